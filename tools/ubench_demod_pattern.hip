// The demodulation stage's ACCESS PATTERN without its arithmetic: what do its loads cost alone?  373 K forwarded positions, ascending,
// ~1440 samples apart (hashed), over 1 GiB of resident samples.
//   stage 1   one lane per position: the 15-sample window = two 16-byte loads at byte 2 p (any 2-byte alignment)
//   stage 2   72 % of the positions, 8 lanes each: the 112 first-half samples = 224 bytes from sample p + 16, lane t takes bytes [32 t, 32 t + 32)
// ALIGN: 2 = loads at the true 2-byte-aligned address, 4 = rounded down to 4 bytes (the production kernels: dword-aligned + v_alignbit),
// 16 = rounded down to 16 bytes (+ one more 16-byte load per lane to cover the span).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_demod_pattern.hip -o /tmp/u && /tmp/u
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t pos_of(uint32_t i) { return i * 1440u + hash32(i) % 1440u; }     // sample index, ascending

template <int ALIGN, int STAGE>
__global__ void pattern(const uint8_t *base, uint32_t npos, uint32_t lds_bytes, uint32_t *out) {
    extern __shared__ uint32_t lds[];
    if (lds_bytes && threadIdx.x == 0) lds[0] = 1;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, 0xffffffffu, 0x00020000);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    if (STAGE == 1) {
        for (uint32_t i = gid; i < npos; i += nthreads) {
            uint32_t off = 2u * pos_of(i);
            if (ALIGN > 2) off &= ~(uint32_t)(ALIGN - 1);
            acc ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0) ^ __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
            if (ALIGN == 16) acc ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32, 0, 0);
        }
    } else {
        for (uint32_t g = gid >> 3; g < npos; g += nthreads >> 3) {
            if (hash32(g ^ 0x5bd1e995u) % 100u >= 72u) continue;
            uint32_t off = 2u * (pos_of(g) + 16u) + 32u * (gid & 7);
            if (ALIGN > 2) off &= ~(uint32_t)(ALIGN - 1);
            acc ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0) ^ __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
            if (ALIGN == 16 && (gid & 7) == 7) acc ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32, 0, 0);
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u) out[gid] = 1;
}

template <int ALIGN, int STAGE>
static float run(const uint8_t *d, uint32_t npos, int W, uint32_t lds, int grid, uint32_t *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((pattern<ALIGN, STAGE>), dim3(grid), dim3(W * 64), lds, 0, d, npos, lds, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((pattern<ALIGN, STAGE>), dim3(grid), dim3(W * 64), lds, 0, d, npos, lds, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1000;
}

int main() {
    const uint64_t nbytes = 1ull << 30;
    uint8_t *d; uint32_t *out;
    hipMalloc(&d, nbytes + (1 << 20)); hipMalloc(&out, 1 << 24);
    hipMemset(d, 0x7f, nbytes + (1 << 20));
    const uint32_t npos = 372000;
    struct C { int W; uint32_t lds; int wgs; };
    for (C c : {C{8, 80000, 2}, C{16, 76000, 2}, C{4, 0, 8}}) {
        const int grid = 256 * c.wgs;
        printf("-- %d waves x %d workgroups per CU\n", c.W, c.wgs);
        printf("   stage 1 (2 x 16 B per position, one lane each):   2-byte aligned %.1f us, 4-byte %.1f us, 16-byte (+1 load) %.1f us\n",
               run<2, 1>(d, npos, c.W, c.lds, grid, out), run<4, 1>(d, npos, c.W, c.lds, grid, out), run<16, 1>(d, npos, c.W, c.lds, grid, out));
        printf("   stage 2 (224 B per preamble, 8 lanes x 2 x 16 B): 2-byte aligned %.1f us, 4-byte %.1f us, 16-byte (+1 load) %.1f us\n",
               run<2, 2>(d, npos, c.W, c.lds, grid, out), run<4, 2>(d, npos, c.W, c.lds, grid, out), run<16, 2>(d, npos, c.W, c.lds, grid, out));
    }
    return 0;
}
