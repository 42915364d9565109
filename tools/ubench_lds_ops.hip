// What do LDS operations cost per CU when many wavefronts issue them back to back?  (LDS-pipe cycles per wave-instruction.)
// Aligned and 2-byte-misaligned reads of 4 / 8 / 16 bytes per lane, 2-byte writes, 16-byte writes: the scan kernel's candidates for moving
// work from the VALU (95 % busy) to the LDS pipe.   hipcc -O3 --offload-arch=gfx950 tools/ubench_lds_ops.hip -o /tmp/u && /tmp/u
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

enum { R32 = 0, R64, R128, W16x8, W128, R16, W32, R32x2 };

template <int OP, int OFF>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[16][1024 + 64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < (1024 + 64) / 4; i += 64) reinterpret_cast<uint32_t *>(lds[w])[i] = i;
    __syncthreads();
    const uint32_t base = (uint32_t)reinterpret_cast<uintptr_t>(lds[w]);
    const uint32_t a16 = base + lane * 16 + OFF, a4 = base + lane * 4 + OFF, a8 = base + lane * 8 + OFF;
    uint32_t acc = 0;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (OP == R32) {
            uint32_t a, b, c, d;
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(a4) : "memory");
            acc += a + b + c + d;
        } else if (OP == R64) {
            u32x2 a, b, c, d;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:8\n\tds_read_b64 %3, %4 offset:520\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(a8) : "memory");
            acc += a.x + b.y + c.x + d.y;
        } else if (OP == R128) {
            u32x4 a, b, c, d;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(a16) : "memory");
            acc += a.x + b.y + c.z + d.w;
        } else if (OP == R16) {
            uint32_t a, b, c, d;
            asm volatile("ds_read_u16 %0, %4\n\tds_read_u16 %1, %4 offset:2\n\tds_read_u16 %2, %4 offset:4\n\tds_read_u16 %3, %4 offset:6\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(a16) : "memory");
            acc += a + b + c + d;
        } else if (OP == W16x8) {       // one lane's 16 bytes as eight 2-byte writes: counted as 8 instructions
            asm volatile("ds_write_b16 %0, %1\n\tds_write_b16 %0, %2 offset:2\n\tds_write_b16 %0, %3 offset:4\n\tds_write_b16 %0, %4 offset:6\n\t"
                         "ds_write_b16 %0, %1 offset:8\n\tds_write_b16 %0, %2 offset:10\n\tds_write_b16 %0, %3 offset:12\n\tds_write_b16 %0, %4 offset:14\n\ts_waitcnt lgkmcnt(0)"
                         : : "v"(a16), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "memory");
        } else if (OP == W128) {
            asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(a16), "v"(v) : "memory");
        } else if (OP == W32) {
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1 offset:256\n\tds_write_b32 %0, %1 offset:512\n\tds_write_b32 %0, %1 offset:768\n\ts_waitcnt lgkmcnt(0)" : : "v"(a4), "v"(v.x) : "memory");
        } else if (OP == R32x2) {
            u32x2 a, b, c, d;
            asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %4 offset0:64 offset1:65\n\tds_read2_b32 %2, %4 offset0:128 offset1:129\n\tds_read2_b32 %3, %4 offset0:192 offset1:193\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(a4) : "memory");
            acc += a.x + b.y + c.x + d.y;
        }
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP, int OFF>
static void run(const char *name, int per_iter, uint32_t *d, unsigned long long *dc) {
    unsigned long long c = 0;
    const int iters = 4000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k<OP, OFF>), dim3(256), dim3(1024), 0, 0, d, iters, dc);
        hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    }
    // 16 waves per CU, each issuing per_iter instructions per iteration: LDS-pipe cycles per wave-instruction (CU level)
    printf("%-34s +%d bytes: %7.1f cycles per wave-instruction per CU (one wave alone would see %.0f per instruction)\n", name, OFF,
           (double)c / iters / (16.0 * per_iter), (double)c / iters / per_iter);
}

int main() {
    uint32_t *d; unsigned long long *dc;
    hipMalloc(&d, 256 * 1024 * 4); hipMalloc(&dc, 8);
    run<R32, 0>("ds_read_b32", 4, d, dc);   run<R32, 2>("ds_read_b32", 4, d, dc);
    run<R32x2, 0>("ds_read2_b32", 4, d, dc); run<R32x2, 2>("ds_read2_b32", 4, d, dc);
    run<R64, 0>("ds_read_b64", 4, d, dc);   run<R64, 2>("ds_read_b64", 4, d, dc);
    run<R128, 0>("ds_read_b128", 4, d, dc); run<R128, 2>("ds_read_b128", 4, d, dc);
    run<R16, 0>("ds_read_u16 (4 per 16 B lane stride)", 4, d, dc);
    run<W128, 0>("ds_write_b128", 4, d, dc);
    run<W16x8, 0>("ds_write_b16 x 8 (one lane's 16 B)", 8, d, dc);
    run<W32, 0>("ds_write_b32", 4, d, dc);
    return 0;
}
