// Known-traffic kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on the access patterns
// of the simulator (16-byte records at scattered addresses).  Every kernel touches >= 1 GiB, well past
// the 256 MiB Infinity Cache.  The program prints the bytes each kernel REQUESTS as one JSON line;
// tools/pmc_calibrate.sh runs it under `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` and divides.
//   hipcc --offload-arch=gfx950 -O3 pmc_calib.hip -o pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double gvec2 __attribute__((ext_vector_type(2)));
#define G1(T, p) (*(__attribute__((address_space(1))) T *)(void *)(p))

constexpr size_t kRing = 24576;     // bytes per ring: the tier-0 slot of pcc_sim (3 * 512 records)
constexpr int kRingRecs = 1536;

// 16 B per lane, consecutive lanes consecutive addresses
__global__ void calib_dense_read16(const char *base, size_t n_vec, double *sink) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        const gvec2 v = G1(const gvec2, base + i * 16);
        acc += v.x + v.y;
    }
    if (acc == 1.2345) *sink = acc;
}
__global__ void calib_dense_write16(char *base, size_t n_vec) {
    gvec2 v; v.x = threadIdx.x; v.y = blockIdx.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x)
        G1(gvec2, base + i * 16) = v;
}
// one 16-byte record per 128-byte line: 16 of every 128 bytes are touched
__global__ void calib_line_read16(const char *base, size_t n_lines, double *sink) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) {
        const gvec2 v = G1(const gvec2, base + i * 128 + 16 * (i & 7));
        acc += v.x + v.y;
    }
    if (acc == 1.2345) *sink = acc;
}
__global__ void calib_line_write16(char *base, size_t n_lines) {
    gvec2 v; v.x = threadIdx.x; v.y = blockIdx.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x)
        G1(gvec2, base + i * 128 + 16 * (i & 7)) = v;
}
// the lane-per-env rounds of the send half: lane l appends `recs` consecutive 16-byte records to its own ring,
// one record per instruction, rings 24 KB apart
__global__ void calib_ring_append16(char *base, const uint32_t *start, int recs) {
    const size_t ring = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    char *p = base + ring * kRing;
    const uint32_t pos = start[ring] % kRingRecs;
    gvec2 v; v.x = threadIdx.x; v.y = blockIdx.x;
    for (int i = 0; i < recs; i++) { G1(gvec2, p + (size_t)((pos + i) % kRingRecs) * 16) = v; v.x += 1.0; }
}
// the retire half's mean-RTT reads: 16 lanes per ring, each lane the 8-byte second word of every 16th record
__global__ void calib_ring_read8(const char *base, const uint32_t *start, int recs, double *sink) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t ring = t >> 4; const uint32_t l = t & 15u;
    const char *p = base + ring * kRing;
    const uint32_t pos = start[ring] % kRingRecs;
    double acc = 0;
    for (int i = l; i < recs; i += 16) acc += G1(const double, p + (size_t)((pos + i) % kRingRecs) * 16 + 8);
    if (acc == 1.2345) *sink = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;                 // dense kernels: 2 GiB
    const size_t n_lines = ((size_t)4 << 30) / 128;       // line kernels: 32 Mi lines over 4 GiB
    const int n_rings = 131072, recs = 512;               // ring kernels: 3 GiB of rings, 1 GiB of records
    char *base; CK(hipMalloc(&base, (size_t)4 << 30)); CK(hipMemset(base, 0, (size_t)4 << 30));
    double *sink; CK(hipMalloc(&sink, 8));
    uint32_t *start; CK(hipMalloc(&start, n_rings * 4));
    uint32_t *h = (uint32_t *)malloc(n_rings * 4); srand(5); for (int i = 0; i < n_rings; i++) h[i] = (uint32_t)rand();
    CK(hipMemcpy(start, h, n_rings * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(calib_dense_read16, dim3(4096), dim3(256), 0, 0, base, bytes / 16, sink);
        hipLaunchKernelGGL(calib_dense_write16, dim3(4096), dim3(256), 0, 0, base, bytes / 16);
        hipLaunchKernelGGL(calib_line_read16, dim3(4096), dim3(256), 0, 0, base, n_lines, sink);
        hipLaunchKernelGGL(calib_line_write16, dim3(4096), dim3(256), 0, 0, base, n_lines);
        hipLaunchKernelGGL(calib_ring_append16, dim3(n_rings / 64), dim3(64), 0, 0, base, start, recs);
        hipLaunchKernelGGL(calib_ring_read8, dim3(n_rings * 16 / 64), dim3(64), 0, 0, base, start, recs, sink);
        CK(hipDeviceSynchronize());
    }
    printf("{\"calib_dense_read16\": {\"read\": %zu, \"write\": 0}, \"calib_dense_write16\": {\"read\": 0, \"write\": %zu}, "
           "\"calib_line_read16\": {\"read\": %zu, \"write\": 0, \"lines_x128\": %zu}, \"calib_line_write16\": {\"read\": 0, \"write\": %zu, \"lines_x128\": %zu}, "
           "\"calib_ring_append16\": {\"read\": 0, \"write\": %zu}, \"calib_ring_read8\": {\"read\": %zu, \"write\": 0, \"records_x16\": %zu}}\n",
           bytes, bytes, n_lines * 16, n_lines * 128, n_lines * 16, n_lines * 128,
           (size_t)n_rings * recs * 16, (size_t)n_rings * recs * 8, (size_t)n_rings * recs * 16);
    return 0;
}
