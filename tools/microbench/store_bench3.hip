// What would writing the lane rounds' records LINE-WISE buy?  (profiles/r04_experiments.json "what_bounds_the_light_items")
//   hipcc --offload-arch=gfx950 -O3 store_bench3.hip -o store_bench3
// A wavefront runs 64 rings, a lane each, one 16-byte record per lane per iteration behind ~40 dependent fp64 operations
// (the lane rounds of the send half).  Modes:
//   alu     : no stores at all (the floor)
//   scatter : every lane stores its record to its own ring when it is made (what the send half does)
//   lds8    : the record goes to LDS (row of 8 records per lane, 144-byte stride); every 8 iterations the wavefront writes the
//             rows out in 8 store instructions, each 8 rings x 8 consecutive records (lanes 8g..8g+7 = ring 8j+g) -- a ring's
//             8 records are 128 consecutive bytes, one line when aligned, two pieces otherwise (start offsets are random)
//   lds4    : the same with rows of 4 (4 instructions of 16 rings x 64 bytes every 4 iterations)
// Timed (a) one wavefront per compute unit -- the latency of an iteration, what the longest light item is made of -- and
// (b) 16 per compute unit -- the chip-wide rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double gvec2 __attribute__((ext_vector_type(2)));
constexpr size_t kRing = 24576;
constexpr int kRowStride = 144;   // bytes of a lane's LDS row: 8 records + 16 of padding (the 64 rows spread over the banks)

template <int MODE, int ALU>
__global__ __launch_bounds__(256) void k(char *base, const uint32_t *start, int iters, const uint32_t *perm, int n_rings) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + wv;
    const size_t ring = perm[(wave * 64 + lane) % (size_t)n_rings];
    char *p = base + ring * kRing;
    uint32_t pos = start[ring] & 1023u;
    __shared__ __attribute__((aligned(16))) char s_rows[4][64 * kRowStride];
    __shared__ __attribute__((aligned(16))) uint64_t s_meta[4][64][2];
    char *rows = s_rows[wv];
    double acc = (double)lane;
    uint32_t first = pos;   // ring index of the oldest record still in LDS
    for (int i = 0; i < iters; i++) {
        for (int d = 0; d < ALU; d++) acc = acc * 1.0000001 + 0.5;
        gvec2 v; v.x = acc; v.y = acc;
        if (MODE == 1) {
            *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)(pos & 1023u) * 16) = v;
            pos++;
        } else if (MODE == 2 || MODE == 3) {
            constexpr uint32_t R = MODE == 2 ? 8u : 4u;
            *(gvec2 *)(rows + lane * kRowStride + (pos & (R - 1u)) * 16u) = v;
            pos++;
            if ((i & (R - 1u)) == R - 1u) {
                s_meta[wv][lane][0] = (uint64_t)p;
                s_meta[wv][lane][1] = (uint64_t)first | ((uint64_t)(pos - first) << 32);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                constexpr uint32_t per = 64u / R;   // rings per store instruction
#pragma unroll
                for (uint32_t j = 0; j < R; j++) {
                    const uint32_t e = j * per + lane / R, r = lane & (R - 1u);
                    const uint64_t b = s_meta[wv][e][0], m = s_meta[wv][e][1];
                    const uint32_t f = (uint32_t)m, n = (uint32_t)(m >> 32);
                    const uint32_t idx = f + r;
                    const gvec2 rec = *(const gvec2 *)(rows + e * kRowStride + (idx & (R - 1u)) * 16u);
                    if (r < n) *(__attribute__((address_space(1))) gvec2 *)(void *)((char *)b + (size_t)(idx & 1023u) * 16) = rec;
                }
                __builtin_amdgcn_wave_barrier();
                first = pos;
            }
        }
    }
    if (MODE == 0 && acc == 1.2345) *(double *)p = acc;
}

int main() {
    const int n_rings = 262144;
    char *base; uint32_t *start;
    CK(hipMalloc(&base, (size_t)n_rings * kRing));
    CK(hipMemset(base, 0, (size_t)n_rings * kRing));
    std::vector<uint32_t> h(n_rings), hp(n_rings);
    srand(3);
    for (auto &x : h) x = (uint32_t)rand();
    for (int i = 0; i < n_rings; i++) hp[i] = i;
    for (int i = n_rings - 1; i > 0; i--) { int j = rand() % (i + 1); std::swap(hp[i], hp[j]); }
    CK(hipMalloc(&start, n_rings * 4)); CK(hipMemcpy(start, h.data(), n_rings * 4, hipMemcpyHostToDevice));
    uint32_t *perm; CK(hipMalloc(&perm, n_rings * 4)); CK(hipMemcpy(perm, hp.data(), n_rings * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 512;
    const char *names[4] = {"alu", "scatter", "lds8", "lds4"};
    auto run = [&](int mode, int alu, int blocks, int threads) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
#define L(M, A) hipLaunchKernelGGL((k<M, A>), dim3(blocks), dim3(threads), 0, 0, base, start, iters, perm, n_rings)
            if (alu == 40) { if (mode == 0) L(0, 40); if (mode == 1) L(1, 40); if (mode == 2) L(2, 40); if (mode == 3) L(3, 40); }
            else { if (mode == 0) L(0, 0); if (mode == 1) L(1, 0); if (mode == 2) L(2, 0); if (mode == 3) L(3, 0); }
#undef L
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double recs = (double)blocks * threads * iters;
        printf("%-8s alu %2d  blocks %5d x %4d  %8.3f ms  %7.1f ns per iteration of a wavefront  %7.2f G records/s\n", names[mode], alu, blocks,
               threads, ms, ms * 1e6 / iters, recs / ms * 1e-6);
    };
    for (int alu : {40, 0})
        for (int mode = 0; mode < 4; mode++) {
            run(mode, alu, 256, 64);     // one wavefront per compute unit
            run(mode, alu, 256, 256);    // four (one per SIMD)
            run(mode, alu, 1024, 256);   // sixteen
        }
    return 0;
}
