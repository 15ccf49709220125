// How fast can the chip take 16-byte records?  Patterns of the send half's ring writes, timed with
// HIP events.  hipcc --offload-arch=gfx950 -O3 store_bench.hip -o store_bench
//   scatter : lane l of a wave appends to its OWN ring (rings 24 KB apart, start offsets random),
//             one record per lane per iteration -- the lane-per-env rounds
//   dense   : a wave appends 64 consecutive records (1 KB) to ONE ring per iteration -- the wave passes
//   chunk64 : lanes 4e..4e+3 write the 4 consecutive records of ring e (64 B, aligned), 16 rings per
//             instruction -- what an LDS transpose of the rounds' records would issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double gvec2 __attribute__((ext_vector_type(2)));
constexpr size_t kRing = 24576;  // bytes per ring (tier-0 slot of pcc_sim: 3 * 512 * 16)

template <int MODE>
__global__ void k(char *base, const uint32_t *start, int iters, int rings_per_wave, const uint32_t *perm) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    gvec2 v; v.x = (double)lane; v.y = (double)wave;
    if (MODE == 0 || MODE == 3) {
        // MODE 3: the wave's 64 rings are scattered over the whole allocation (sorted work lists)
        const size_t ring = MODE == 3 ? perm[wave * 64 + lane] : wave * 64 + lane;
        char *p = base + ring * kRing;
        uint32_t pos = start[ring] & 1023u;
        for (int i = 0; i < iters; i++) {
            *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)((pos + i) & 1023u) * 16) = v;
            v.x += 1.0;
        }
    } else if (MODE >= 4) {
        // the lane-per-env rounds as they are: ~60 dependent ALU instructions between a lane's records.
        // MODE 4 stores each record when it is made; MODE 5 keeps four and stores them back to back.
        // MODE >= 6: the wave's 64 rings come from one group of 2^(MODE+4) consecutive rings (locality of the sorted work lists)
        size_t ring = perm[wave * 64 + lane];
        if (MODE >= 6) { const size_t gsz = (size_t)1 << (MODE + 4); const size_t groups = (size_t)262144 / gsz; ring = (wave % groups) * gsz + ring % gsz; }
        char *p = base + ring * kRing;
        uint32_t pos = start[ring] & 1023u;
        double acc = (double)lane;
        gvec2 r0 = v, r1 = v, r2 = v, r3 = v;
        for (int i = 0; i < iters; i++) {
            for (int d = 0; d < 40; d++) acc = acc * 1.0000001 + 0.5;   // ~40 dependent fp64 ops
            v.x = acc;
            if (MODE == 4) {
                *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)((pos + i) & 1023u) * 16) = v;
            } else {
                if ((i & 3) == 0) r0 = v; else if ((i & 3) == 1) r1 = v; else if ((i & 3) == 2) r2 = v; else r3 = v;
                if ((i & 3) == 3) {
                    *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)((pos + i - 3) & 1023u) * 16) = r0;
                    *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)((pos + i - 2) & 1023u) * 16) = r1;
                    *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)((pos + i - 1) & 1023u) * 16) = r2;
                    *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)((pos + i) & 1023u) * 16) = r3;
                }
            }
        }
    } else if (MODE == 1) {
        for (int r = 0; r < rings_per_wave; r++) {
            const size_t ring = wave * rings_per_wave + r;
            char *p = base + ring * kRing;
            uint32_t pos = start[ring] & 1023u;
            for (int i = 0; i < iters; i += 64) {
                *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)((pos + i + lane) & 1023u) * 16) = v;
                v.x += 1.0;
            }
        }
    } else if (MODE == 10) {
        for (int i = 0; i < iters; i += 4) {
            for (int c = 0; c < 4; c++) {
                const size_t ring = perm[wave * 64 + 16 * c + (lane >> 2)];
                char *p = base + ring * kRing;
                const uint32_t pos = ((start[ring] & 1020u) + i + (lane & 3u)) & 1023u;
                *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)pos * 16) = v;
                v.x += 1.0;
            }
        }
    } else {
        // 4 iterations of 64 rings -> 4 instructions of 16 rings x 64 B
        for (int i = 0; i < iters; i += 4) {
            for (int c = 0; c < 4; c++) {
                const size_t ring = wave * 64 + 16 * c + (lane >> 2);
                char *p = base + ring * kRing;
                const uint32_t pos = ((start[ring] & 1020u) + i + (lane & 3u)) & 1023u;
                *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)pos * 16) = v;
                v.x += 1.0;
            }
        }
    }
}

int main() {
    const int n_rings = 262144;
    char *base; uint32_t *start;
    CK(hipMalloc(&base, (size_t)n_rings * kRing));
    CK(hipMemset(base, 0, (size_t)n_rings * kRing));
    std::vector<uint32_t> h(n_rings);
    srand(3);
    for (auto &x : h) x = (uint32_t)rand();
    CK(hipMalloc(&start, n_rings * 4));
    CK(hipMemcpy(start, h.data(), n_rings * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 256;
    std::vector<uint32_t> hp(n_rings);
    for (int i = 0; i < n_rings; i++) hp[i] = i;
    for (int i = n_rings - 1; i > 0; i--) { int j = rand() % (i + 1); std::swap(hp[i], hp[j]); }
    uint32_t *perm; CK(hipMalloc(&perm, n_rings * 4)); CK(hipMemcpy(perm, hp.data(), n_rings * 4, hipMemcpyHostToDevice));
    for (int mode = 2; mode < 11; mode++) {
        if (mode == 5 || mode == 2 || mode == 10) continue;
        for (int waves : {1024, 4096}) {
            // `waves` wavefronts, each owning 64 rings; records per launch = waves * 64 * iters
            const int rpw = 64;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, base, start, iters, rpw, perm);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, base, start, iters, rpw, perm);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, base, start, iters, rpw, perm);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(waves), dim3(64), 0, 0, base, start, iters, rpw, perm);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(waves), dim3(64), 0, 0, base, start, iters, rpw, perm);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(waves), dim3(64), 0, 0, base, start, iters, rpw, perm);
                if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(waves), dim3(64), 0, 0, base, start, iters, rpw, perm);
                if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(waves), dim3(64), 0, 0, base, start, iters, rpw, perm);
                if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(waves), dim3(64), 0, 0, base, start, iters, rpw, perm);
                if (mode == 9) hipLaunchKernelGGL(k<9>, dim3(waves), dim3(64), 0, 0, base, start, iters, rpw, perm);
                if (mode == 10) hipLaunchKernelGGL(k<10>, dim3(waves), dim3(64), 0, 0, base, start, iters, rpw, perm);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double recs = (double)waves * 64 * iters;
            printf("%-8s waves %5d  %8.3f ms  %7.2f G records/s  %7.1f GB/s\n", mode == 0 ? "scatter" : mode == 1 ? "dense" : mode == 2 ? "chunk64" : mode == 3 ? "scat-rnd" : mode == 4 ? "alu+1st" : mode == 5 ? "alu+4st" : mode == 6 ? "grp1024" : mode == 7 ? "grp2048" : mode == 8 ? "grp4096" : mode == 9 ? "grp8192" : "chunk-rnd",
                   waves, ms, recs / ms * 1e-6, recs * 16 / ms * 1e-6);
        }
    }
    return 0;
}
