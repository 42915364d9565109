// How fast can this chip fetch SCATTERED 128-byte lines (the demodulation stage's access pattern: ~1.2 M lines per GiB, ascending, one DRAM
// page each)?  N lines spread over 1 GiB at stride 1 GiB / N with a hashed jitter; a group of 8 lanes reads one line (16 B per lane), `F`
// independent loads in flight per lane, W waves per workgroup, LDS padding for the occupancy; default cache policy (what the kernels use).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_scatter.hip -o /tmp/u && /tmp/u
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int F>
__global__ void scatter(const uint8_t *__restrict__ base, uint32_t nlines, uint32_t stride, uint32_t lds_bytes, uint32_t *out) {
    extern __shared__ uint32_t lds[];
    if (lds_bytes && threadIdx.x == 0) lds[0] = 1;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, ngroups = gridDim.x * blockDim.x / 8;
    const uint32_t grp = gid >> 3, sub = gid & 7;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t i = grp; i < nlines; i += ngroups * F) {
        u32x4 v[F];
#pragma unroll
        for (int f = 0; f < F; f++) {
            const uint32_t line = i + f * ngroups;
            const uint32_t jitter = (line * 2654435761u >> 20) % (stride / 128 ? stride / 128 : 1);
            const uint64_t off = (uint64_t)(line < nlines ? line : 0) * stride + (uint64_t)jitter * 128 + sub * 16;
            v[f] = *reinterpret_cast<const u32x4 *>(base + off);
        }
#pragma unroll
        for (int f = 0; f < F; f++) acc ^= v[f];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u) out[gid] = 1;
}

template <int F>
static float run(const uint8_t *d, uint32_t nlines, uint32_t stride, int W, uint32_t lds, int grid, uint32_t *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((scatter<F>), dim3(grid), dim3(W * 64), lds, 0, d, nlines, stride, lds, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((scatter<F>), dim3(grid), dim3(W * 64), lds, 0, d, nlines, stride, lds, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}

int main() {
    const uint64_t nbytes = 1ull << 30;
    uint8_t *d; uint32_t *out;
    hipMalloc(&d, nbytes + (1 << 20)); hipMalloc(&out, 1 << 24);
    hipMemset(d, 0x7f, nbytes + (1 << 20));
    for (uint32_t nlines : {1210000u, 600000u, 2400000u}) {
        const uint32_t stride = (uint32_t)(nbytes / nlines) / 128 * 128;
        printf("-- %u lines of 128 B over 1 GiB (stride %u B): %.1f MB\n", nlines, stride, nlines * 128 / 1e6);
        struct C { int W; uint32_t lds; int wgs_per_cu; int F; };
        for (C c : {C{8, 80000, 2, 1}, C{8, 80000, 2, 2}, C{8, 80000, 2, 4}, C{16, 76000, 2, 1}, C{16, 76000, 2, 2}, C{16, 76000, 2, 4}, C{16, 76000, 2, 8},
                    C{4, 0, 8, 1}, C{4, 0, 8, 2}, C{4, 0, 8, 4}, C{4, 0, 8, 8}, C{4, 0, 4, 4}, C{4, 0, 2, 8}}) {
            const int grid = 256 * c.wgs_per_cu;
            float ms = -1;
            if (c.F == 1) ms = run<1>(d, nlines, stride, c.W, c.lds, grid, out);
            if (c.F == 2) ms = run<2>(d, nlines, stride, c.W, c.lds, grid, out);
            if (c.F == 4) ms = run<4>(d, nlines, stride, c.W, c.lds, grid, out);
            if (c.F == 8) ms = run<8>(d, nlines, stride, c.W, c.lds, grid, out);
            printf("   %2d waves x %d workgroups per CU (%2d waves per CU), %d in flight: %.4f ms = %.2f TB/s of lines, %.1f G lines/s\n", c.W, c.wgs_per_cu,
                   c.W * c.wgs_per_cu, c.F, ms, nlines * 128.0 / (ms * 1e-3) / 1e12, nlines / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
