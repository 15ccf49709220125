// Why does a scattered record store add ~55 ns to a lane-round iteration whatever the number of lanes that store?
// (tools/send_timeline.py: 215 ns with the store, 160 without; 32 or 18 live lanes: the same)
//   hipcc --offload-arch=gfx950 -O3 store_bench7.hip -o store_bench7
// One wavefront per compute unit; per "packet" a chain of `alu` dependent fp64 multiply-adds, then a 16-byte store of the
// result to the lane's own ring.  Variants of how the four packets of a block use registers and issue their stores:
//   0  no stores                          1  store right after each packet (one live record)
//   2  four packets, four distinct records kept live, each stored right after it is made
//   3  four packets made, then four stores back to back
//   4  like 1, but the stored value is a copy made one packet earlier (the store never reads a register the next
//      instructions write)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double gvec2 __attribute__((ext_vector_type(2)));
constexpr size_t kRing = 24576;
#define ST(P, POS, V) *(__attribute__((address_space(1))) gvec2 *)(void *)((P) + (size_t)((POS) & 1023u) * 16) = (V)

template <int MODE>
__global__ __launch_bounds__(64) void k(char *base, const uint32_t *start, const uint32_t *perm, int blocks4, int alu, long long *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t ring = perm[(size_t)blockIdx.x * 64 + lane];
    char *p = base + ring * kRing;
    uint32_t pos = start[ring] & 1023u;
    double acc = (double)lane;
    gvec2 prev; prev.x = acc; prev.y = acc;
    const long long r0 = (long long)wall_clock64();
    for (int b = 0; b < blocks4; b++) {
        gvec2 r[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            for (int d = 0; d < alu; d++) acc = acc * 1.0000001 + 0.5;
            r[q].x = acc; r[q].y = acc + 1.0;
            if (MODE == 1 || MODE == 2) ST(p, pos + q, r[q]);
            if (MODE == 4) { ST(p, pos + q, prev); prev = r[q]; }
        }
        if (MODE == 3) { ST(p, pos, r[0]); ST(p, pos + 1, r[1]); ST(p, pos + 2, r[2]); ST(p, pos + 3, r[3]); }
        if (MODE == 2) asm volatile("" :: "v"(r[0].x), "v"(r[1].x), "v"(r[2].x), "v"(r[3].x));   // (keeps the four records in distinct registers)
        pos += 4;
    }
    const long long r1 = (long long)wall_clock64();
    if (lane == 0) out[blockIdx.x] = r1 - r0;
    if (acc == 1.2345) *(double *)p = acc + prev.x;
}

int main() {
    const int n_rings = 65536;
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
    const int blocks4 = 112;
    for (int alu : {0, 4, 8, 12, 16})
        for (int mode = 0; mode < 5; mode++) {
            std::vector<long long> ho(256);
            for (int rep = 0; rep < 2; rep++) {
#define L(M) hipLaunchKernelGGL(k<M>, dim3(256), dim3(64), 0, 0, base, start, perm, blocks4, alu, out)
                if (mode == 0) L(0); if (mode == 1) L(1); if (mode == 2) L(2); if (mode == 3) L(3); if (mode == 4) L(4);
#undef L
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(ho.data(), out, 256 * 8, hipMemcpyDeviceToHost));
            std::sort(ho.begin(), ho.end());
            printf("alu %2d  mode %d  %6.1f ns per packet (median of 256 CUs)\n", alu, mode, ho[128] * 10.0 / (blocks4 * 4));
        }
    return 0;
}
