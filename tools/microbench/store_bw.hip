// How fast does an MI355X take STORES?  (round 6: the send launch writes 180 MB of 16-byte records in a busy phase of ~55 us = 3.3 TB/s,
// and every coding of the lane rounds' scattered stores ends at the same ~205 G records/s -- is that the memory system's write rate?)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/store_bw.hip -o store_bw && ./store_bw
// Each wavefront writes `n` 16-byte records per lane:
//   dense      lane l of instruction j writes bytes [1024 j + 16 l, +16) of the wavefront's own region (1 KB per instruction)
//   dense nt   the same, nontemporal
//   ring       lane l writes record (start_l + j) mod n of ITS OWN ring (n * 16 bytes apart: 64 lines per instruction, 8 consecutive
//              instructions fill a line; without the per-lane start all 64 lanes hit the same L2 channel: 63 G records/s)
//   ring nt    the same, nontemporal
//   line       8 adjacent lanes write one whole 128-byte line (8 records) of a ring, an instruction 8 lines of 8 different rings
//   line nt    the same, nontemporal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double gvec2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k_store(char *buf, int n, size_t wave_bytes) {
    const size_t wave = (size_t)blockIdx.x * 4 + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    char *base = buf + wave * wave_bytes;
    gvec2 v;
    v.x = (double)wave; v.y = (double)lane;
    for (int j = 0; j < n; j++) {
        // (ring modes: every lane starts at a position of its own, like the rings of 64 envs, and wraps)
        const uint32_t start = ((uint32_t)(wave * 64 + lane) * 2654435761u >> 8) % (uint32_t)n;
        char *p = (MODE & 2) ? base + (size_t)lane * (wave_bytes / 64) + (size_t)((start + (uint32_t)j) % (uint32_t)n) * 16 : base + (size_t)j * 1024 + lane * 16;
        v.x += 1.0;
        if (MODE & 4) {   // line modes: 8 adjacent lanes write ONE whole line (8 records) of a ring; an instruction writes whole lines of 8 rings
            const uint32_t ring = (uint32_t)(lane >> 3) + 8u * ((uint32_t)j & 7u);   // the 64 rings of the wavefront take turns, 8 per instruction
            const uint32_t st8 = (((uint32_t)(wave * 64 + ring) * 2654435761u >> 8) % (uint32_t)n) & ~7u;
            const uint32_t rec = (st8 + 8u * ((uint32_t)j >> 3) + ((uint32_t)lane & 7u)) % ((uint32_t)n & ~7u);
            p = base + (size_t)ring * (wave_bytes / 64) + (size_t)rec * 16;
        }
        if (MODE & 1) __builtin_nontemporal_store(v, (__attribute__((address_space(1))) gvec2 *)(void *)p);
        else *(__attribute__((address_space(1))) gvec2 *)(void *)p = v;
    }
}

int main(int argc, char **argv) {
    const int waves_per_cu = argc > 1 ? atoi(argv[1]) : 8;
    const int n = argc > 2 ? atoi(argv[2]) : 1024;            // records per lane
    const int n_waves = 256 * waves_per_cu;
    const size_t wave_bytes = (size_t)n * 1024;               // (a lane's ring in the ring modes: n * 16 bytes, 64 of them)
    char *buf;
    CK(hipMalloc(&buf, wave_bytes * n_waves));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[6] = {"dense   ", "dense nt", "ring    ", "ring nt ", "line    ", "line nt "};
    for (int mode = 0; mode < 6; mode++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_store<0>, dim3(n_waves / 4), dim3(256), 0, 0, buf, n, wave_bytes);
            if (mode == 1) hipLaunchKernelGGL(k_store<1>, dim3(n_waves / 4), dim3(256), 0, 0, buf, n, wave_bytes);
            if (mode == 2) hipLaunchKernelGGL(k_store<2>, dim3(n_waves / 4), dim3(256), 0, 0, buf, n, wave_bytes);
            if (mode == 3) hipLaunchKernelGGL(k_store<3>, dim3(n_waves / 4), dim3(256), 0, 0, buf, n, wave_bytes);
            if (mode == 4) hipLaunchKernelGGL(k_store<4>, dim3(n_waves / 4), dim3(256), 0, 0, buf, n, wave_bytes);
            if (mode == 5) hipLaunchKernelGGL(k_store<5>, dim3(n_waves / 4), dim3(256), 0, 0, buf, n, wave_bytes);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double bytes = (double)wave_bytes * n_waves;
        printf("%s  %d wavefronts per compute unit x %d records per lane: %8.3f ms  %6.2f TB/s  %6.1f G records/s\n", names[mode], waves_per_cu, n, best,
               bytes / best / 1e9, bytes / 16.0 / best / 1e6);
    }
    return 0;
}
