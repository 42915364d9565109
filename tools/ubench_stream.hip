// Which read-only streaming pattern does this chip like?  1 GiB resident, every byte read once per launch, 16 B per lane
// through a raw buffer descriptor, an XOR keeps the loads alive (stream_read_kernel of the product is pattern "runs").
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o /tmp/ubench_stream && /tmp/ubench_stream
// Patterns (chunk = 1 KiB = one wave-wide 16 B/lane load):
//   runs     wavefront r owns chunks [r R, (r + 1) R)                  - the scan kernel's (R = 32, 2 waves per workgroup)
//   xcd      the same, workgroups of one XCD (blockIdx mod 8) own one contiguous eighth of the stream
//   weave    a workgroup of W waves owns W R consecutive chunks, wave w reads chunks w, w + W, ...
//   stride   persistent: grid = G workgroups, wave g reads chunks g, g + nwaves, ...
// LDS bytes per workgroup set the occupancy (the scan kernel: 12800 per 2 waves -> 24 waves per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kChunk = 1024;
enum { RUNS = 0, XCD = 1, WEAVE = 2, STRIDE = 3 };

template <int AUX, int INFLIGHT>
__global__ void stream_kernel(const uint8_t *__restrict__ iq, uint32_t nchunks, uint32_t R, int pattern, uint32_t lds_bytes, uint32_t *out) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const uint32_t W = blockDim.x >> 6;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (lds_bytes && threadIdx.x == 0) lds[0] = 1;                           // keep the allocation
    uint32_t c0, step, n;                                                    // this wave reads chunks c0, c0 + step, ... (n of them)
    const uint32_t nruns = (nchunks + R - 1) / R;
    if (pattern == RUNS || pattern == XCD) {
        uint32_t b = blockIdx.x;
        if (pattern == XCD) { const uint32_t per = (gridDim.x + 7) / 8; b = (blockIdx.x & 7) * per + (blockIdx.x >> 3); }
        const uint32_t run = b * W + wave;
        if (run >= nruns) return;
        c0 = run * R; step = 1; n = min(R, nchunks - c0);
    } else if (pattern == WEAVE) {
        const uint32_t base = blockIdx.x * W * R;
        if (base + wave >= nchunks) return;
        c0 = base + wave; step = W; n = min(R, (nchunks - c0 + W - 1) / W);
    } else {
        const uint32_t nw = gridDim.x * W, g = blockIdx.x * W + wave;
        if (g >= nchunks) return;
        c0 = g; step = nw; n = (nchunks - g + nw - 1) / nw;
    }
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(iq), 0, 0xffffffffu, 0x00020000);
    const uint32_t lane_off = (uint32_t)lane * 16u;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 buf[INFLIGHT];
    const uint32_t last = c0 + (n - 1) * step;
#pragma unroll
    for (int i = 0; i < INFLIGHT; i++) buf[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, min(c0 + i * step, last) * kChunk, AUX);
    uint32_t c = c0 + INFLIGHT * step;
    for (uint32_t k = 0; k < n; k += INFLIGHT, c += INFLIGHT * step) {
#pragma unroll
        for (int i = 0; i < INFLIGHT; i++) {
            if (k + i < n) acc ^= buf[i];
            buf[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, min(c + i * step, last) * kChunk, AUX);
        }
    }
    const uint32_t v = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (v == 0x9E3779B9u) out[blockIdx.x] = v;
}

struct Cfg { const char *name; int pattern; uint32_t R, W, lds, grid; int aux, inflight; };

template <int AUX, int INFLIGHT>
static float run_one(const Cfg &c, const uint8_t *d, uint32_t nchunks, uint32_t *out, int warm, int timed) {
    uint32_t grid;
    if (c.pattern == STRIDE) grid = c.grid;
    else if (c.pattern == WEAVE) grid = (nchunks + c.W * c.R - 1) / (c.W * c.R);
    else { const uint32_t nruns = (nchunks + c.R - 1) / c.R; grid = (nruns + c.W - 1) / c.W; if (c.pattern == XCD) grid = (grid + 7) / 8 * 8; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < warm; i++) hipLaunchKernelGGL((stream_kernel<AUX, INFLIGHT>), dim3(grid), dim3(c.W * 64), c.lds, 0, d, nchunks, c.R, c.pattern, c.lds, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < timed; i++) hipLaunchKernelGGL((stream_kernel<AUX, INFLIGHT>), dim3(grid), dim3(c.W * 64), c.lds, 0, d, nchunks, c.R, c.pattern, c.lds, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / timed;
}

static float dispatch(const Cfg &c, const uint8_t *d, uint32_t nchunks, uint32_t *out, int warm, int timed) {
#define CASE(A, F) if (c.aux == A && c.inflight == F) return run_one<A, F>(c, d, nchunks, out, warm, timed);
    CASE(18, 1) CASE(18, 2) CASE(18, 3) CASE(18, 4) CASE(0, 2) CASE(2, 2) CASE(3, 2) CASE(19, 2) CASE(2, 4) CASE(0, 4)
#undef CASE
    return -1.f;
}

int main(int argc, char **argv) {
    const uint64_t nbytes = 1ull << 30;
    const uint32_t nchunks = (uint32_t)(nbytes / kChunk);
    uint8_t *d; uint32_t *out;
    hipMalloc(&d, nbytes + 4096); hipMalloc(&out, 1 << 22);
    hipMemset(d, 0x7f, nbytes + 4096);
    std::vector<Cfg> cfgs;
    const uint32_t LDS2 = 12800;                                             // the scan kernel's per 2 waves
    cfgs.push_back({"runs R32 W2 (scan)", RUNS, 32, 2, LDS2, 0, 18, 2});
    for (uint32_t R : {8u, 16u, 24u, 32u, 48u, 64u, 128u, 172u}) cfgs.push_back({"runs", RUNS, R, 2, LDS2, 0, 18, 2});
    for (uint32_t W : {1u, 4u, 8u, 16u}) cfgs.push_back({"runs W", RUNS, 32, W, LDS2 / 2 * W, 0, 18, 2});
    for (int f : {1, 3, 4}) cfgs.push_back({"runs inflight", RUNS, 32, 2, LDS2, 0, 18, f});
    for (int a : {0, 2, 3, 19}) cfgs.push_back({"runs aux", RUNS, 32, 2, LDS2, 0, a, 2});
    cfgs.push_back({"runs no LDS (occupancy 32+)", RUNS, 32, 2, 0, 0, 18, 2});
    cfgs.push_back({"runs LDS 16 waves/CU", RUNS, 32, 2, 20000, 0, 18, 2});
    cfgs.push_back({"runs LDS 16 waves/CU inflight 4", RUNS, 32, 2, 20000, 0, 18, 4});
    for (uint32_t R : {16u, 32u, 64u}) cfgs.push_back({"xcd", XCD, R, 2, LDS2, 0, 18, 2});
    for (uint32_t W : {2u, 4u, 8u, 16u}) for (uint32_t R : {16u, 32u, 64u}) cfgs.push_back({"weave", WEAVE, R, W, LDS2 / 2 * W, 0, 18, 2});
    for (uint32_t G : {1536u, 3072u}) for (int f : {2, 4}) cfgs.push_back({"stride W2", STRIDE, 0, 2, LDS2, G, 18, f});
    cfgs.push_back({"stride W8 G768", STRIDE, 0, 8, LDS2 * 4, 768, 18, 2});
    cfgs.push_back({"stride W2 aux0", STRIDE, 0, 2, LDS2, 3072, 0, 2});
    cfgs.push_back({"runs R32 W2 (scan) again", RUNS, 32, 2, LDS2, 0, 18, 2});
    // settle the clocks
    for (int i = 0; i < 3; i++) dispatch(cfgs[0], d, nchunks, out, 40, 40);
    const int rounds = argc > 1 ? atoi(argv[1]) : 2;
    std::vector<std::vector<float>> ms(cfgs.size());
    for (int r = 0; r < rounds; r++)
        for (size_t i = 0; i < cfgs.size(); i++) ms[i].push_back(dispatch(cfgs[i], d, nchunks, out, 10, 40));
    for (size_t i = 0; i < cfgs.size(); i++) {
        const Cfg &c = cfgs[i];
        const float best = *std::min_element(ms[i].begin(), ms[i].end());
        printf("%-34s R %3u W %2u lds %6u grid %5u aux %2d inflight %d :", c.name, c.R, c.W, c.lds, c.grid, c.aux, c.inflight);
        for (float m : ms[i]) printf(" %.4f", m);
        printf(" ms  -> %.0f GB/s\n", nbytes / (best * 1e-3) / 1e9);
    }
    return 0;
}
