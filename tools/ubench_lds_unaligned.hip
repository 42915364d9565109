// Does ds_read_b128 at a 2-byte aligned LDS address work on gfx950 (unaligned-access mode), and what does it cost?
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_lds_unaligned tools/ubench_lds_unaligned.hip && tools/ubench_lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void check(uint32_t *out, int off_bytes) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 8 + 64];
    for (int i = threadIdx.x; i < 64 * 8 + 64; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)reinterpret_cast<uintptr_t>(lds) + threadIdx.x * 16 + off_bytes;
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}

template <int OFF>
__global__ void timeit(uint32_t *out, int iters, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4][64 * 8 + 64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 64 * 8 + 64; i += 64) lds[w][i] = (uint16_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)reinterpret_cast<uintptr_t>(lds[w]) + lane * 16 + OFF;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        u32x4 a, b, c, d;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr) : "memory");
        acc += a + b + c + d;
    }
    const unsigned long long t1 = clock64();
    out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    uint32_t *d; unsigned long long *dc;
    hipMalloc(&d, 4096 * 4); hipMalloc(&dc, 8);
    for (int off : {0, 2, 4, 6, 8}) {
        check<<<1, 64>>>(d, off);
        std::vector<uint32_t> h(256);
        hipError_t e = hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
        bool ok = e == hipSuccess;
        for (int t = 0; t < 64 && ok; t++)
            for (int k = 0; k < 4; k++) {
                const uint32_t i0 = t * 8 + off / 2 + 2 * k;
                const uint32_t want = (i0 & 0xffff) | ((i0 + 1) << 16);
                if (h[t * 4 + k] != want) ok = false;
            }
        printf("ds_read_b128 at +%d bytes: %s (%s)\n", off, ok ? "correct" : "WRONG", hipGetErrorString(e));
    }
    unsigned long long c0 = 0, c2 = 0;
    for (int rep = 0; rep < 2; rep++) {
        timeit<0><<<256 * 2, 256>>>(d, 20000, dc); hipMemcpy(&c0, dc, 8, hipMemcpyDeviceToHost);
        timeit<2><<<256 * 2, 256>>>(d, 20000, dc); hipMemcpy(&c2, dc, 8, hipMemcpyDeviceToHost);
    }
    printf("4 waves per SIMD-group, 4 x ds_read_b128 per iteration: aligned %.1f cycles / read, +2 bytes %.1f cycles / read\n",
           c0 / 20000.0 / 4, c2 / 20000.0 / 4);
    return 0;
}
