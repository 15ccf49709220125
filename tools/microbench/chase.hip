// Pointer-chase latency vs footprint and page stride (diagnostics, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o chase chase.hip && ./chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

__global__ void chase(const uint64_t *buf, uint64_t start, int hops, uint64_t *out, uint64_t *ticks) {
    uint64_t p = start;
    const uint64_t t0 = wall_clock64();
    for (int i = 0; i < hops; i++) p = buf[p];
    const uint64_t t1 = wall_clock64();
    out[0] = p;
    ticks[0] = t1 - t0;
}

__global__ void scatter(uint64_t *buf, const uint64_t *idx, size_t slots, size_t step_words) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < slots) buf[i * step_words] = idx[i];
}

int main() {
    const size_t max_bytes = (size_t)64 << 30;
    uint64_t *buf;
    if (hipMalloc(&buf, max_bytes) != hipSuccess) { printf("malloc failed\n"); return 1; }
    uint64_t *out, *ticks;
    uint64_t *staging;
    (void)hipMalloc(&out, 8); (void)hipMalloc(&ticks, 8); (void)hipMalloc(&staging, (size_t)8 << 20);
    std::mt19937_64 rng(1);
    const size_t strides[] = {4096, 65536, 2u << 20};
    for (size_t stride : strides) {
        for (size_t fp = (size_t)1 << 20; fp <= max_bytes; fp <<= 2) {
            size_t slots = fp / stride;
            if (slots < 2) continue;
            if (slots > (1u << 20)) slots = 1u << 20;   // at most 1M slots spread over the footprint
            const size_t step = fp / slots;             // >= stride
            std::vector<uint32_t> perm(slots);
            for (size_t i = 0; i < slots; i++) perm[i] = (uint32_t)i;
            std::shuffle(perm.begin(), perm.end(), rng);
            // cyclic: slot perm[i] -> perm[i+1]; write only the head word of each slot
            std::vector<uint64_t> idx(slots);
            for (size_t i = 0; i < slots; i++) idx[perm[i]] = (uint64_t)perm[(i + 1) % slots] * (step / 8);
            (void)hipMemcpy(staging, idx.data(), slots * 8, hipMemcpyHostToDevice);
            scatter<<<(unsigned)((slots + 255) / 256), 256>>>(buf, staging, slots, step / 8);
            const int hops = 20000;
            chase<<<1, 1>>>(buf, 0, hops, out, ticks);   // warm
            chase<<<1, 1>>>(buf, 0, hops, out, ticks);
            (void)hipDeviceSynchronize();
            uint64_t t;
            (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            printf("stride %8zu footprint %8zu MB slots %8zu : %.0f ns/hop\n", stride, fp >> 20, slots, (double)t * 10.0 / hops);
            fflush(stdout);
        }
    }
    return 0;
}
