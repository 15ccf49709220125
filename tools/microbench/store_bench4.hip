// Does the cost of a scattered 16-byte record store depend on how much memory the rings span?  (address translation reach)
//   hipcc --offload-arch=gfx950 -O3 store_bench4.hip -o store_bench4
// 1 024 wavefronts (four per compute unit), every lane appends to a ring of its own picked at random among the first
// n_rings rings of one allocation; rings `spacing` bytes apart.  Swept: n_rings x spacing = the span.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double gvec2 __attribute__((ext_vector_type(2)));

__global__ void k(char *base, const uint32_t *start, int iters, const uint32_t *perm, size_t spacing, uint32_t window) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const size_t ring = perm[wave * 64 + lane];
    char *p = base + ring * spacing;
    const uint32_t pos = start[ring] & 1023u;
    gvec2 v; v.x = (double)lane; v.y = (double)wave;
    for (int i = 0; i < iters; i++) {
        *(__attribute__((address_space(1))) gvec2 *)(void *)(p + (size_t)((pos + (uint32_t)i % window) & 1023u) * 16) = v;
        v.x += 1.0;
    }
}

int main() {
    const size_t max_bytes = (size_t)24 << 30;
    char *base; uint32_t *start, *perm;
    CK(hipMalloc(&base, max_bytes));
    CK(hipMemset(base, 0, max_bytes));
    const int lanes = 1024 * 64;
    std::vector<uint32_t> h(1 << 20), hp(lanes);
    srand(3);
    for (auto &x : h) x = (uint32_t)rand();
    CK(hipMalloc(&start, h.size() * 4)); CK(hipMemcpy(start, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&perm, lanes * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 256;
    for (size_t spacing : {(size_t)24576, (size_t)4096, (size_t)98304})
        for (int n_rings : {65536, 131072, 262144, 524288, 1048576}) {
            if ((size_t)n_rings * spacing > max_bytes) continue;
            // lanes pick distinct rings: a random subset of n_rings (n_rings >= lanes)
            std::vector<uint32_t> all(n_rings);
            for (int i = 0; i < n_rings; i++) all[i] = i;
            for (int i = 0; i < lanes; i++) { int j = i + rand() % (n_rings - i); std::swap(all[i], all[j]); hp[i] = all[i]; }
            CK(hipMemcpy(perm, hp.data(), lanes * 4, hipMemcpyHostToDevice));
            for (uint32_t window : {1024u, 32u}) {
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, base, start, iters, perm, spacing, window);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                printf("spacing %6zu B  rings %8d  span %7.2f GB  window %4u records  %7.3f ms  %7.2f G records/s\n", spacing, n_rings,
                       (double)n_rings * spacing / 1073741824.0, window, ms, (double)lanes * iters / ms * 1e-6);
            }
        }
    return 0;
}
