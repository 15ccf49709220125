// A lane-round wavefront in its habitat: ~160 ns of dependent fp64 arithmetic per iteration + ONE scattered 16-byte store
// of `active` lanes, with and without neighbours on its compute unit.  What does the store add to the iteration?
//   hipcc --offload-arch=gfx950 -O3 store_bench6.hip -o store_bench6
// Workgroups of 256 threads on every compute unit.  Wavefront 0 of a workgroup is the probe (timed by itself: it stamps
// s_memrealtime around its loop); wavefronts 1..3 are neighbours of the chosen kind:
//   0 none   1 the same loop as the probe (lane rounds)   2 dense record stores (64 consecutive records per instruction)
//   3 pure arithmetic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double gvec2 __attribute__((ext_vector_type(2)));
constexpr size_t kRing = 24576;

__device__ __forceinline__ void lane_rounds(char *p, uint32_t pos, int iters, int alu, bool stores, double &acc) {
    for (int i = 0; i < iters; i++) {
        for (int d = 0; d < alu; d++) acc = acc * 1.0000001 + 0.5;
        gvec2 v; v.x = acc; v.y = acc;
        if (stores) *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)((pos + i) & 1023u) * 16) = v;
    }
}

__global__ __launch_bounds__(256) void k(char *base, const uint32_t *start, const uint32_t *perm, int iters, int alu, int active, int neighbours,
                                         int n_neigh, long long *out) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + wv;
    const size_t ring = perm[wave * 64 + lane];
    char *p = base + ring * kRing;
    const uint32_t pos = start[ring] & 1023u;
    double acc = (double)lane;
    if (wv == 0) {
        const long long t0 = (long long)__builtin_readcyclecounter();
        const long long r0 = (long long)wall_clock64();
        if ((int)lane < active) lane_rounds(p, pos, iters, alu, true, acc);
        else lane_rounds(p, pos, iters, alu, false, acc);
        const long long r1 = (long long)wall_clock64();
        if (lane == 0) out[blockIdx.x] = r1 - r0;
        (void)t0;
    } else if ((int)wv <= n_neigh) {
        if (neighbours == 1) lane_rounds(p, pos, iters, alu, true, acc);
        else if (neighbours == 2) {
            gvec2 v; v.x = acc; v.y = acc;
            char *q = base + (size_t)perm[wave] * kRing;
            for (int i = 0; i < iters * 4; i++) {
                *(__attribute__((address_space(1))) gvec2 *)(void *)(q + (size_t)((i * 64 + lane) & 1023u) * 16) = v;
                for (int d = 0; d < alu / 4; d++) acc = acc * 1.0000001 + 0.5;
                v.x = acc;
            }
        } else if (neighbours == 3) lane_rounds(p, pos, iters, alu, false, acc);
    }
    if (acc == 1.2345) *(double *)p = acc;
}

int main() {
    const int n_rings = 65536;   // 1.5 GB: within an XCD's translation reach
    char *base; uint32_t *start, *perm; long long *out;
    CK(hipMalloc(&base, (size_t)n_rings * kRing));
    CK(hipMemset(base, 0, (size_t)n_rings * kRing));
    std::vector<uint32_t> h(n_rings), hp(n_rings);
    srand(3);
    for (auto &x : h) x = (uint32_t)rand();
    for (int i = 0; i < n_rings; i++) hp[i] = i;
    for (int i = n_rings - 1; i > 0; i--) { int j = rand() % (i + 1); std::swap(hp[i], hp[j]); }
    CK(hipMalloc(&start, n_rings * 4)); CK(hipMemcpy(start, h.data(), n_rings * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&perm, n_rings * 4)); CK(hipMemcpy(perm, hp.data(), n_rings * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 256 * 8));
    const int iters = 450;
    const char *nn[4] = {"none", "lane rounds", "dense stores", "arithmetic"};
    for (int alu : {68, 0})
        for (int neighbours : {0, 1, 2, 3})
            for (int n_neigh : {1, 3}) {
                if (neighbours == 0 && n_neigh == 3) continue;
                for (int active : {64, 32, 16, 0}) {
                    std::vector<long long> ho(256);
                    for (int rep = 0; rep < 2; rep++) {
                        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, base, start, perm, iters, alu, active, neighbours, n_neigh, out);
                        CK(hipDeviceSynchronize());
                    }
                    CK(hipMemcpy(ho.data(), out, 256 * 8, hipMemcpyDeviceToHost));
                    std::sort(ho.begin(), ho.end());
                    printf("alu %2d  neighbours: %d x %-12s  probe lanes storing %2d  ->  %6.1f ns per iteration (median of 256 CUs; max %6.1f)\n", alu,
                           neighbours ? n_neigh : 0, nn[neighbours], active, ho[128] * 10.0 / iters, ho[255] * 10.0 / iters);
                }
            }
    return 0;
}
