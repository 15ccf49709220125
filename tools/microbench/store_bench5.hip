// What does ONE wavefront pay per store instruction of 64 x 16 bytes, by the number of distinct 128-byte lines it touches
// and by how many lanes are active?  (the latency of a lane-round iteration; store_bench3/4 measure chip-wide rates)
//   hipcc --offload-arch=gfx950 -O3 store_bench5.hip -o store_bench5
// One wavefront per compute unit (256 blocks of 64 threads), `iters` dependent-free store instructions back to back.
// lanes_per_line = 1: every lane its own ring (64 lines per instruction); 2, 4, 8: groups of that many lanes write
// consecutive 16-byte records of one ring (32, 16, 8 lines per instruction).  active = lanes that store at all.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double gvec2 __attribute__((ext_vector_type(2)));
constexpr size_t kRing = 24576;

template <int BYTES>
__global__ void k(char *base, int iters, const uint32_t *perm, int lanes_per_line, int active, int waves_per_block_used) {
    const uint32_t lane = threadIdx.x & 63u;
    if ((int)(threadIdx.x >> 6) >= waves_per_block_used) return;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const uint32_t grp = lane / (uint32_t)lanes_per_line, sub = lane % (uint32_t)lanes_per_line;
    const size_t ring = perm[wave * 64 + grp];
    char *p = base + ring * kRing;
    gvec2 v; v.x = (double)lane; v.y = (double)wave;
    if ((int)lane >= active) return;
    for (int i = 0; i < iters; i++) {
        // every instruction a fresh line of the ring (8 records per line, rings hold 192 lines)
        char *q = p + (size_t)((uint32_t)i % 190u) * 128 + sub * 16;
        if (BYTES == 16) *(__attribute__((address_space(1))) gvec2 *)(void *)q = v;
        else if (BYTES == 8) *(__attribute__((address_space(1))) double *)(void *)q = v.x;
        else { *(__attribute__((address_space(1))) double *)(void *)q = v.x; *(__attribute__((address_space(1))) double *)(void *)(q + 8) = v.y; }
        v.x += 1.0;
    }
}

int main(int argc, char **argv) {
    const int n_rings = argc > 1 ? atoi(argv[1]) : 32768;   // span = n_rings * 24 KB
    char *base;
    CK(hipMalloc(&base, (size_t)n_rings * kRing));
    CK(hipMemset(base, 0, (size_t)n_rings * kRing));
    std::vector<uint32_t> hp(n_rings);
    srand(3);
    for (int i = 0; i < n_rings; i++) hp[i] = i;
    for (int i = n_rings - 1; i > 0; i--) { int j = rand() % (i + 1); std::swap(hp[i], hp[j]); }
    uint32_t *perm; CK(hipMalloc(&perm, n_rings * 4)); CK(hipMemcpy(perm, hp.data(), n_rings * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2048;
    printf("span %.2f GB\n", (double)n_rings * kRing / 1073741824.0);
    auto run = [&](int bytes, int lpl, int active, int blocks, int threads) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (bytes == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(threads), 0, 0, base, iters, perm, lpl, active, threads / 64);
            if (bytes == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(threads), 0, 0, base, iters, perm, lpl, active, threads / 64);
            if (bytes == 88) hipLaunchKernelGGL(k<88>, dim3(blocks), dim3(threads), 0, 0, base, iters, perm, lpl, active, threads / 64);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("bytes %2d  lanes/line %d  active %2d  waves/CU %2d  %8.3f ms  %7.1f ns per store instruction of a wavefront\n", bytes, lpl, active,
               blocks * (threads / 64) / 256, ms, ms * 1e6 / iters);
    };
    for (int wpc : {1, 4})
        for (int lpl : {1, 2, 4, 8}) run(16, lpl, 64, 256, 64 * wpc);
    for (int active : {32, 16, 8}) run(16, 1, active, 256, 64);
    run(8, 1, 64, 256, 64);
    run(88, 1, 64, 256, 64);
    return 0;
}
