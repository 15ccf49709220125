// cu_key.hip -- census of the compute-unit key the fused step puts into its ready-queue granules (pcc_dev.h: cu_key):
// XCC_ID[3:0] << 8 | HW_ID[15:8] (CU_ID, SH_ID, SE_ID).  It must be the same for every wavefront of a workgroup (a workgroup
// never spans compute units) and take as many distinct values as the device has compute units.
//   hipcc --offload-arch=gfx950 -O2 -o cu_key cu_key.hip && ./cu_key
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void census(unsigned *key, unsigned *hwid) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        key[w] = ((xcc & 0xFu) << 8) | ((hw >> 8) & 0xFFu);
        hwid[w] = hw;
    }
    // keep the workgroups resident for a while so that the launch spreads over every compute unit
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 200000ull) {}
}
int main() {
    const int blocks = 4096, waves = blocks * 4;
    unsigned *key, *hw;
    hipMalloc(&key, waves * 4); hipMalloc(&hw, waves * 4);
    census<<<blocks, 256>>>(key, hw);
    std::vector<unsigned> k(waves), h(waves);
    hipMemcpy(k.data(), key, waves * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h.data(), hw, waves * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> keys(k.begin(), k.end());
    int split = 0;
    for (int b = 0; b < blocks; b++)
        for (int w = 1; w < 4; w++) split += k[b * 4 + w] != k[b * 4];
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("{\"compute_units\": %d, \"distinct_keys\": %zu, \"workgroups_with_more_than_one_key\": %d, \"example_hw_id\": \"0x%08x\"}\n",
           p.multiProcessorCount, keys.size(), split, h[0]);
    return keys.size() == (size_t)p.multiProcessorCount && split == 0 ? 0 : 1;
}
