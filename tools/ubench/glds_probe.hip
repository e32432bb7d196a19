// Probe for the LDS-staged brick walk of vrt_path_kernel: global_load_lds_dwordx4 under a partial EXEC mask with a per-lane
// global address.  Expected: lane l's 16 bytes land at lds_base + 16*l (l = lane id, not a compacted index); inactive lanes'
// slots are left untouched.  Build: hipcc --offload-arch=gfx950 -O2 -o glds_probe glds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__global__ void probe(const uint32_t *src, const uint32_t *brick_of_lane, unsigned long long mask, uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 2048u; i += blockDim.x) lds[i] = 0xDEAD0000u + i;
    __syncthreads();
    uint32_t *wave_lds = lds + wave * 1024u; // 4 KiB per wave: 4 chunks of 64 lanes x 16 bytes
    if ((mask >> lane) & 1ull) {
        const uint32_t *g = src + (size_t)brick_of_lane[threadIdx.x] * 16u;
#pragma unroll
        for (int c = 0; c < 4; c++)
            __builtin_amdgcn_global_load_lds((glb_void *)(g + 4 * c), (lds_void *)(wave_lds + 256 * c), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // read back through the address the walk loop will form: chunk (w>>2), lane*16, (w&3)*4
    for (uint32_t w = 0; w < 16u; w++) out[(size_t)threadIdx.x * 16u + w] = wave_lds[(w >> 2) * 256u + lane * 4u + (w & 3u)];
}

int main() {
    const int nbricks = 1000;
    static uint32_t hsrc[nbricks * 16], hb[128], hout[128 * 16];
    for (int i = 0; i < nbricks * 16; i++) hsrc[i] = 0xB0000000u + i;
    for (int i = 0; i < 128; i++) hb[i] = (uint32_t)((i * 7919u + 13u) % nbricks);
    uint32_t *dsrc, *db, *dout;
    if (hipMalloc(&dsrc, sizeof hsrc) != hipSuccess) { printf("no device\n"); return 2; }
    hipMalloc(&db, sizeof hb); hipMalloc(&dout, sizeof hout);
    hipMemcpy(dsrc, hsrc, sizeof hsrc, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    const unsigned long long mask = 0xF0F0A5A5FFFF0001ull;
    probe<<<1, 128, 8192>>>(dsrc, db, mask, dout);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 128; t++) {
        const int lane = t & 63, wave = t >> 6;
        for (int w = 0; w < 16; w++) {
            const uint32_t want = ((mask >> lane) & 1ull) ? hsrc[hb[t] * 16 + w] : 0xDEAD0000u + wave * 1024 + (w >> 2) * 256 + lane * 4 + (w & 3);
            if (hout[t * 16 + w] != want) { if (bad < 6) printf("thread %d word %d got %08x want %08x\n", t, w, hout[t * 16 + w], want); bad++; }
        }
    }
    printf(bad ? "GLDS_PROBE_FAIL %d\n" : "GLDS_PROBE_OK\n", bad);
    return bad ? 1 : 0;
}
