// Where does the chip-wide rate of scattered 16-byte record stores (the lane-per-env rounds of the send half:
// ~80 G records/s, tools/microbench/store_bench.hip) saturate -- in every CU's own address path, or further down?
//   hipcc --offload-arch=gfx950 -O3 store_bench2.hip -o store_bench2
// The same scattered-ring store loop, launched (a) with fewer and fewer wavefronts of 64 threads (spread over all CUs)
// and (b) as workgroups of 1 024 threads (16 wavefronts on ONE CU), so that the same number of wavefronts sits on
// 1/16 of the CUs; plus (c) the loop with 8-byte and 4-byte records, and (d) 32-byte (two records per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double gvec2 __attribute__((ext_vector_type(2)));
typedef double gvec4 __attribute__((ext_vector_type(4)));
constexpr size_t kRing = 24576;

template <int BYTES>
__global__ void k(char *base, const uint32_t *start, int iters, const uint32_t *perm, int n_rings) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const size_t ring = perm[(wave * 64 + lane) % (size_t)n_rings];
    char *p = base + ring * kRing;
    const uint32_t pos = start[ring] & 1023u;
    double x = (double)lane;
    for (int i = 0; i < iters; i++) {
        char *q = p + (size_t)((pos + i * (BYTES == 32 ? 2 : 1)) & 1023u) * 16;
        if (BYTES == 16) { gvec2 v; v.x = x; v.y = x; *(__attribute__((address_space(1))) gvec2 *)(void *)q = v; }
        if (BYTES == 8) *(__attribute__((address_space(1))) double *)(void *)q = x;
        if (BYTES == 4) *(__attribute__((address_space(1))) float *)(void *)q = (float)x;
        if (BYTES == 32) { gvec4 v; v.x = x; v.y = x; v.z = x; v.w = x; *(__attribute__((address_space(1))) gvec4 *)(void *)(p + (size_t)((pos + 2 * i) & 1022u) * 16) = v; }
        x += 1.0;
    }
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
    auto run = [&](const char *name, int bytes, int blocks, int threads) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (bytes == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(threads), 0, 0, base, start, iters, perm, n_rings);
            if (bytes == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(threads), 0, 0, base, start, iters, perm, n_rings);
            if (bytes == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, base, start, iters, perm, n_rings);
            if (bytes == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(threads), 0, 0, base, start, iters, perm, n_rings);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double stores = (double)blocks * threads * iters;
        printf("%-28s %2d B  blocks %5d x %4d threads (%5d waves)  %8.3f ms  %7.2f G stores/s  %7.1f GB/s\n", name, bytes, blocks, threads,
               blocks * threads / 64, ms, stores / ms * 1e-6, stores * bytes / ms * 1e-6);
    };
    for (int waves : {128, 256, 512, 1024, 2048, 4096}) run("64-thread blocks", 16, waves, 64);
    for (int blocks : {16, 32, 64, 128, 256}) run("1024-thread blocks (1 CU each)", 16, blocks, 1024);
    for (int bytes : {4, 8, 32}) run("record size", bytes, 1024, 64);
    for (int bytes : {4, 8, 32}) run("record size", bytes, 4096, 64);
    return 0;
}
