// What does rocprofv3's FETCH_SIZE count on gfx950 for access patterns whose HBM line traffic is KNOWN?  (the
// guide calibrates the counter - x 2 - for wide coalesced streaming reads only; the demodulation stage reads scattered 16-byte pieces.)
// Four kernels over 1 GiB of resident bytes, each touching every 128-byte line it touches exactly once per launch:
//   calib_stream   the scan kernel's loads: 16 B per lane, consecutive lanes consecutive bytes            -> 2^30 bytes
//   calib_lines    N scattered whole lines, 8 lanes x 16 B each (tools/ubench_scatter.hip)                -> N x 128 bytes
//   calib_stage1   the stage-1 pattern: one lane, two 16-byte loads at a 2-byte aligned address           -> lines counted on the host
//   calib_stage2   the stage-2 pattern: 8 lanes x two 16-byte loads, dword-aligned, 224+ bytes            -> lines counted on the host
// The program prints the bytes of DISTINCT 128-byte lines each kernel touches per launch; tools/fetch_calib.sh runs it under
// rocprofv3 --pmc FETCH_SIZE and divides.   hipcc -O3 --offload-arch=gfx950 tools/ubench_fetch_calib.hip -o tools/ubench_fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <set>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__host__ __device__ inline uint32_t pos_of(uint32_t i) { return i * 1440u + hash32(i) % 1440u; }     // sample index, ascending
__host__ __device__ inline bool is_preamble(uint32_t g) { return hash32(g ^ 0x5bd1e995u) % 100u < 72u; }
__host__ __device__ inline uint32_t line_of(uint32_t i, uint32_t stride) { return i * (stride / 128) + (i * 2654435761u >> 20) % (stride / 128); }

__global__ void calib_stream(const uint8_t *base, uint64_t nbytes, uint32_t *out) {
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t o = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; o < nbytes; o += (uint64_t)gridDim.x * blockDim.x * 16)
        acc ^= *reinterpret_cast<const u32x4 *>(base + o);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u) out[threadIdx.x] = 1;
}
__global__ void calib_lines(const uint8_t *base, uint32_t nlines, uint32_t stride, uint32_t *out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, ngroups = gridDim.x * blockDim.x / 8;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t i = gid >> 3; i < nlines; i += ngroups)
        acc ^= *reinterpret_cast<const u32x4 *>(base + (uint64_t)line_of(i, stride) * 128 + (gid & 7) * 16);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u) out[threadIdx.x] = 1;
}
__global__ void calib_stage1(const uint8_t *base, uint32_t npos, uint32_t *out) {
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, 0xffffffffu, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npos; i += gridDim.x * blockDim.x) {
        const uint32_t off = 2u * pos_of(i);
        acc ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0) ^ __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u) out[threadIdx.x] = 1;
}
__global__ void calib_stage2(const uint8_t *base, uint32_t npos, uint32_t *out) {
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, 0xffffffffu, 0x00020000);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t g = gid >> 3; g < npos; g += nthreads >> 3) {
        if (!is_preamble(g)) continue;
        const uint32_t off = (2u * (pos_of(g) + 16u) + 28u * (gid & 7)) & ~3u;         // select_kernel: lane t takes pairs 7 t .. 7 t + 6, dword-aligned
        acc ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0) ^ __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u) out[threadIdx.x] = 1;
}

int main() {
    const uint64_t nbytes = 1ull << 30;
    uint8_t *d; uint32_t *out;
    hipMalloc(&d, nbytes + (1 << 20)); hipMalloc(&out, 1 << 16);
    hipMemset(d, 0x7f, nbytes + (1 << 20));
    const uint32_t npos = 372000, nlines = 1210000, stride = (uint32_t)(nbytes / nlines) / 128 * 128;
    // distinct lines per launch, on the host
    std::set<uint32_t> l1, l2;
    for (uint32_t i = 0; i < npos; i++) {
        const uint32_t off = 2u * pos_of(i);
        for (uint32_t b = off; b < off + 32; b += 16) { l1.insert(b >> 7); l1.insert((b + 15) >> 7); }
        if (is_preamble(i))
            for (uint32_t t = 0; t < 8; t++) {
                const uint32_t o = (2u * (pos_of(i) + 16u) + 28u * t) & ~3u;
                for (uint32_t b = o; b < o + 32; b += 16) { l2.insert(b >> 7); l2.insert((b + 15) >> 7); }
            }
    }
    std::set<uint32_t> l0;
    for (uint32_t i = 0; i < nlines; i++) l0.insert(line_of(i, stride));
    printf("{\"calib_stream\": %llu, \"calib_lines\": %llu, \"calib_stage1\": %llu, \"calib_stage2\": %llu, \"launches\": 4}\n",
           (unsigned long long)nbytes, (unsigned long long)l0.size() * 128, (unsigned long long)l1.size() * 128, (unsigned long long)l2.size() * 128);
    for (int rep = 0; rep < 4; rep++) {
        hipLaunchKernelGGL(calib_stream, dim3(256 * 16), dim3(256), 0, 0, d, nbytes, out);
        hipLaunchKernelGGL(calib_lines, dim3(512), dim3(512), 0, 0, d, nlines, stride, out);
        hipLaunchKernelGGL(calib_stage1, dim3(512), dim3(512), 0, 0, d, npos, out);
        hipLaunchKernelGGL(calib_stage2, dim3(512), dim3(1024), 0, 0, d, npos, out);
    }
    hipDeviceSynchronize();
    return 0;
}
