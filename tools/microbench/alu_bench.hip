// Does plain ALU work scale with wavefronts per compute unit?  Each wavefront runs `iters` rounds of a
// dependent fp64 chain (like the send recurrences); single-wavefront workgroups.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off alu_bench.hip -o alu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int CHAINS>
__global__ void k(double *out, int iters) {
    double a[CHAINS];
    for (int c = 0; c < CHAINS; c++) a[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 64 / CHAINS; r++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                const double t = a[c] * 1.0000001;
                a[c] = t > 3.0 ? t - 1.5 : t + 0.25;   // mul, cmp, 2 adds, 2 cndmask: like link_send's selects
            }
    }
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += a[c];
    if (s == 12345.678) out[0] = s;
}
int main() {
    double *out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int chains : {1, 4}) {
        for (int wg : {64, 256}) {
            for (int wpc : {1, 2, 4, 8, 16, 32}) {
                const int waves = 256 * wpc, blocks = waves * 64 / wg;
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    if (chains == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(wg), 0, 0, out, iters);
                    else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(wg), 0, 0, out, iters);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                // 64 steps x ~6 VALU instructions per round
                printf("chains %d  wg %3d  waves/CU %2d  %8.3f ms  %7.1f cycles@2.4GHz per 6-instr step per wave  chip G steps/s %.1f\n", chains, wg, wpc, ms,
                       ms * 1e-3 * 2.4e9 / (iters * 64.0), (double)waves * iters * 64.0 / ms * 1e-6);
            }
        }
    }
    return 0;
}
