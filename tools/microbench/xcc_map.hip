// Does block b of a launch land on XCD b % 8?  (pcc_dev.h "partitions" assumes it for speed, never for results.)
//   hipcc --offload-arch=gfx950 -O3 xcc_map.hip -o xcc_map
// Launches of the send half's shape (1 096 workgroups of 256 threads, 4 per compute unit resident) and of the retire half's
// (4 616 of 128), back to back on one stream, every workgroup spinning for a while so that the next launch starts into a
// busy machine; each workgroup records the XCC_ID hardware register.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k(uint32_t *out, int spin) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = xcc & 0xF;
    // uneven run times, like work items: block b spins (b % 7 + 1) * spin iterations
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)(blockIdx.x % 7 + 1) * spin) {}
}

int main() {
    uint32_t *d;
    CK(hipMalloc(&d, 3 * 8192 * 4));
    std::vector<uint32_t> h(3 * 8192);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k, dim3(1096), dim3(256), 0, 0, d, 400);            // ~4-28 us per workgroup
        hipLaunchKernelGGL(k, dim3(4616), dim3(128), 0, 0, d + 8192, 200);
        hipLaunchKernelGGL(k, dim3(1096), dim3(256), 0, 0, d + 16384, 400);
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    const int n[3] = {1096, 4616, 1096};
    for (int l = 0; l < 3; l++) {
        int ok = 0, hist[8] = {0};
        for (int b = 0; b < n[l]; b++) { ok += (h[l * 8192 + b] == (uint32_t)(b % 8)); hist[h[l * 8192 + b] & 7]++; }
        printf("launch %d: %d of %d workgroups on XCD (block %% 8); workgroups per XCD:", l, ok, n[l]);
        for (int x = 0; x < 8; x++) printf(" %d", hist[x]);
        printf("\n");
    }
    return 0;
}
