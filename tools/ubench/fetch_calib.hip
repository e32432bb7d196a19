// Calibration of rocprofv3's FETCH_SIZE for THIS kernel's access pattern (scattered dword loads), as the MI355X guide
// asks before trusting an absolute: a 2 GiB buffer (far beyond L2 + MALL) read once with one dword per 128-byte line,
// one dword per 64-byte half line, and fully (16 B per lane, coalesced).  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib
// and compare FETCH_SIZE (KiB) per kernel with the bytes printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void gather_stride(const uint32_t *__restrict__ p, uint64_t n_loads, uint32_t stride_words, uint32_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint64_t k = i; k < n_loads; k += (uint64_t)gridDim.x * blockDim.x) acc += p[k * stride_words];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void stream_x4(const uint4 *__restrict__ p, uint64_t n, uint32_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint64_t k = i; k < n; k += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = p[k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint64_t bytes = 2ull << 30;
    uint32_t *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("no device / no memory\n"); return 2; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int blocks = 256 * 8, threads = 256;
    gather_stride<<<blocks, threads>>>(buf, bytes / 128, 32, out);   // one dword per 128-byte line
    hipDeviceSynchronize();
    gather_stride<<<blocks, threads>>>(buf, bytes / 64, 16, out);    // one dword per 64 bytes
    hipDeviceSynchronize();
    stream_x4<<<blocks, threads>>>((const uint4 *)buf, bytes / 16, out); // everything, 16 B per lane
    hipDeviceSynchronize();
    printf("buffer %llu bytes (%llu KiB): kernel 1 touches every 128-byte line once with 4 bytes, kernel 2 every 64 bytes once, kernel 3 reads it all\n",
           (unsigned long long)bytes, (unsigned long long)(bytes >> 10));
    return 0;
}
