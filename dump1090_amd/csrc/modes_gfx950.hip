// modes_gfx950.hip - gfx950 (MI355X / CDNA4) kernels and the C ABI of include/modes_gfx950.h.
//
// Data path (DESIGN.md has the full picture; line numbers cite /root/reference/dump1090.c):
//
//   u8 I/Q stream in HBM ──scan_kernel──► forwarded positions (per-run slots, u32)
//        (2 B/sample, read once)              │
//                                             ▼
//                                  demod_kernel: exact preamble test (:1602-1650),
//                                  bit slicing + noise gate (:1666-1726), phase-corrected
//                                  retry (:1498-1558), syndrome + repair lookup (:733,:854)
//                                             │
//                                             ▼  64-byte modes_record list ──► host resolve
//
// scan_kernel is the only stage that touches every sample; it never materialises
// magnitudes.  Each lane turns 16 bytes (8 I/Q pairs) into 8 packed-u16 powers
// s = (I-127)^2+(Q-127)^2, shares them with its wavefront through a wave-private
// 1 KiB LDS ring (no workgroup barrier anywhere), reads the two lanes before it and
// evaluates 8 preamble positions (alpha: the ordering relations, modes_order8_swar);
// the ~1.5 % survivors are queued in LDS and a dense second pass (beta) applies the
// level tests as a necessary condition in s.  Because the reference's magnitude LUT is
// strictly monotone in s, the ordering tests are exact on s; everything forwarded
// (~0.07 % of positions) is re-checked exactly (LUT) by demod_kernel.
//
// No MFMA (integer scan: the roofline is HBM, the limiter in practice VALU issue - DESIGN.md 3.1),
// no CUDA compatibility layer, wave64 only.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/modes_gfx950.h"
#include "modes_core.h"

static_assert(sizeof(modes_attempt) == 28, "modes_attempt layout");
static_assert(sizeof(modes_record) == 64, "modes_record layout");

namespace {

constexpr int kWave = 64;
constexpr int kChunkSamples = 512;          // one wavefront iteration: 64 lanes x 8 samples
constexpr int kChunkBytes = 1024;
constexpr int kLookback = 16;               // positions of the previous chunk scanned with this one

// ------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------

struct DeviceTables {
    const uint16_t *lut;     // the reference's magnitudes (dump1090.c:359-364, built on the host in double) by saturated power
    const uint32_t *esyn;    // 112 single-bit syndromes
};

// 16 stream bytes at byte offset `off` (may be negative / past the end: 127 there,
// which is what the reference pads with, dump1090.c:344,506).
// 16 stream bytes at byte offset `off` relative to the 16-byte aligned base `iq`; the valid bytes
// are [lo, hi), everything else reads as 127 - what the reference pads with (dump1090.c:344,506).
// The byte-wise path is a rolled loop (small code) taken only by lanes that straddle an end of
// the span.
__device__ __forceinline__ uint4 load_iq16(const uint8_t *iq, int64_t off, int64_t lo, int64_t hi) {
    if (off >= lo && off + 16 <= hi) return *reinterpret_cast<const uint4 *>(iq + off);
    uint64_t a = 0x7f7f7f7f7f7f7f7full, b = 0x7f7f7f7f7f7f7f7full;
#pragma nounroll
    for (int t = 0; t < 16; t++) {
        const int64_t o = off + t;
        if (o >= lo && o < hi) {
            const int sh = (t & 7) * 8;
            const uint64_t v = (uint64_t)iq[o] << sh, keep = ~(0xffull << sh);
            if (t < 8) a = (a & keep) | v; else b = (b & keep) | v;
        }
    }
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}

// min(s, 32767): what the production scan kernel works on (modes_core.h)
__device__ __forceinline__ uint4 power16_sat(uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
    modes_power8_sat(w, o);
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// The scan kernel's powers by the byte dot product: w ^ 0x7f7f7f7f turns every byte b into the signed byte 127 - b
// (0 -> 127, 255 -> -128: it always fits), v_dot4_i32_i8 against the same dword with one sample's two bytes kept
// gives (127-I)^2 + (127-Q)^2 in one instruction, and its clamp against the accumulator 0x7fff8000 saturates the one
// power that needs 16 bits (32768: I = Q = 255) to 32767: the low half of the result is 0x8000 | min(s, 32767).  The
// constant bit 15 changes no ordering and no difference; the level bound of the beta pass accounts for it (kPowBias).
constexpr uint32_t kPowBias = 0x8000u;
__device__ __forceinline__ uint4 power16_scan(uint4 v) {
    // One block, hand-ordered: hipcc fuses the xor into the two ANDs (v_bitop3, a half-rate VOP3 - 12 of them instead of
    // 12 full-rate VOP2) and leaves s_nops between a dot product and the v_perm that packs it; here the eight dot
    // products sit between their producers and consumers (a dot result is read no sooner than 3 instructions later).
    uint32_t o0, o1, o2, o3, t0, t1, t2, t3, a0, a1, a2, a3, b0, b1, b2, b3;
    asm("v_xor_b32 %4, %20, %16\n\t"
        "v_xor_b32 %5, %20, %17\n\t"
        "v_xor_b32 %6, %20, %18\n\t"
        "v_xor_b32 %7, %20, %19\n\t"
        "v_and_b32 %8, %21, %4\n\t"
        "v_and_b32 %12, %22, %4\n\t"
        "v_and_b32 %9, %21, %5\n\t"
        "v_and_b32 %13, %22, %5\n\t"
        "v_and_b32 %10, %21, %6\n\t"
        "v_and_b32 %14, %22, %6\n\t"
        "v_and_b32 %11, %21, %7\n\t"
        "v_and_b32 %15, %22, %7\n\t"
        "v_dot4_i32_i8 %8, %4, %8, %23 clamp\n\t"
        "v_dot4_i32_i8 %12, %4, %12, %23 clamp\n\t"
        "v_dot4_i32_i8 %9, %5, %9, %23 clamp\n\t"
        "v_dot4_i32_i8 %13, %5, %13, %23 clamp\n\t"
        "v_dot4_i32_i8 %10, %6, %10, %23 clamp\n\t"
        "v_dot4_i32_i8 %14, %6, %14, %23 clamp\n\t"
        "v_dot4_i32_i8 %11, %7, %11, %23 clamp\n\t"
        "v_dot4_i32_i8 %15, %7, %15, %23 clamp\n\t"
        "v_perm_b32 %0, %12, %8, %24\n\t"
        "v_perm_b32 %1, %13, %9, %24\n\t"
        "v_perm_b32 %2, %14, %10, %24\n\t"
        "v_perm_b32 %3, %15, %11, %24\n\t"
        "s_nop 0"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),
          "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
        : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "s"(0x7F7F7F7Fu), "s"(0x0000FFFFu), "s"(0xFFFF0000u), "s"(0x7FFF8000u),
          "s"(0x05040100u));
    return make_uint4(o0, o1, o2, o3);
}

// Reference magnitude of buffer sample q (0 outside the stream).
struct MagAt {
    const uint8_t *iq;
    int64_t lo, hi;
    const uint16_t *lut;
    int64_t p;
    __device__ __forceinline__ int operator()(int t) const {
        int64_t o = 2 * (p + t);
        if (o < lo || o >= hi) return 0;
        uint32_t ib = iq[o];
        uint32_t qb = (o + 1 < hi) ? iq[o + 1] : 127u;
        return lut[modes_lut_index(ib, qb)];
    }
};

// The magnitude table (MODES_LUT_ENTRIES u16 = 64 KiB, modes_core.h) into LDS: 16 bytes per lane,
// four loads of a thread in flight at a time.
constexpr int kLutVec = MODES_LUT_ENTRIES * 2 / 16;       // 4096
template <int THREADS>
__device__ __forceinline__ void stage_lut(uint16_t *s_lut, const uint16_t *lut) {
    const uint4 *src = reinterpret_cast<const uint4 *>(lut);
    uint4 *dst = reinterpret_cast<uint4 *>(s_lut);
    constexpr bool kWhole = kLutVec % (4 * THREADS) == 0;                     // no bounds test needed
#pragma unroll 1
    for (int i = (int)threadIdx.x; i < kLutVec; i += 4 * THREADS) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (kWhole || i + k * THREADS < kLutVec) v[k] = src[i + k * THREADS];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (kWhole || i + k * THREADS < kLutVec) dst[i + k * THREADS] = v[k];
    }
}

// ------------------------------------------------------------------------------------
// K1 (parity of computeMagnitudeVector, dump1090.c:1454-1469): u8 I/Q -> u16 magnitude.
// LUT staged in LDS; 8 samples per lane per iteration (16 B in, 16 B out).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void magnitude_kernel(const uint8_t *__restrict__ iq, uint64_t nsamples,
                                                        const uint16_t *__restrict__ lut, uint16_t *__restrict__ mag) {
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[kLutVec * 8];
    stage_lut<256>(s_lut, lut);
    __syncthreads();
    const uint64_t ngroups = (nsamples + 7) / 8;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = load_iq16(iq, (int64_t)(g * 16), 0, (int64_t)(nsamples * 2));
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint16_t m[8];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            m[2 * d] = s_lut[modes_lut_index(w[d] & 0xff, (w[d] >> 8) & 0xff)];
            m[2 * d + 1] = s_lut[modes_lut_index((w[d] >> 16) & 0xff, w[d] >> 24)];
        }
        if (g * 8 + 8 <= nsamples) {
            uint4 o = make_uint4(m[0] | (uint32_t)m[1] << 16, m[2] | (uint32_t)m[3] << 16, m[4] | (uint32_t)m[5] << 16,
                                 m[6] | (uint32_t)m[7] << 16);
            *reinterpret_cast<uint4 *>(mag + g * 8) = o;
        } else {
            for (int t = 0; t < 8 && g * 8 + t < nsamples; t++) mag[g * 8 + t] = m[t];
        }
    }
}

// Debug tap: the saturated powers, computed both ways the kernels compute them - the packed multiply / multiply-add the
// demod kernel indexes its table with (power16_sat) and the byte dot product of the scan kernel (power16_scan, which
// carries a constant bit 15).  A sample on which the two disagree reads 0xffff.
__global__ __launch_bounds__(256) void power_kernel(const uint8_t *__restrict__ iq, uint64_t nsamples,
                                                    uint16_t *__restrict__ out) {
    const uint64_t ngroups = (nsamples + 7) / 8;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = load_iq16(iq, (int64_t)(g * 16), 0, (int64_t)(nsamples * 2));
        const uint4 s = power16_sat(v), d = power16_scan(v);
        const uint32_t w[4] = {s.x, s.y, s.z, s.w}, x[4] = {d.x, d.y, d.z, d.w};
        for (int t = 0; t < 8 && g * 8 + t < nsamples; t++) {
            const uint32_t a = (w[t >> 1] >> (16 * (t & 1))) & 0xffffu, b = (x[t >> 1] >> (16 * (t & 1))) & 0xffffu;
            out[g * 8 + t] = (uint16_t)(b == (a | kPowBias) ? a : 0xffffu);
        }
    }
}

// ------------------------------------------------------------------------------------
// scan_kernel - the HBM-bound stage.
//
// Work decomposition: chunk c = buffer samples [512c, 512c+512) = 1 KiB = one coalesced
// 16 B/lane load per wavefront.  A *run* is `run_chunks` consecutive chunks owned by ONE
// wavefront; iteration c scans positions [512c-16, 512c+496) (their 15-sample windows
// end inside chunk c), so a run scans [512*c0-16, 512*c1-16) and needs, besides its own
// chunks, only the last 32 bytes of chunk c0-1.  Runs are independent: no inter-wave
// synchronisation, no atomics; forwarded positions go to the run's private slot list in
// ascending order, so the concatenation over runs is already sorted.
// ------------------------------------------------------------------------------------
// Device-side state of one detect call, zeroed by the scan kernel.
struct ResultHeader {
    uint32_t n_records;               // staging slots reserved so far = records (exact reservation: no holes)
    uint32_t ordered;                 // finalize_kernel has put the (short) list in order itself: order_kernel has nothing to do
    uint32_t pad[2];
};

// What one demod workgroup found: plain stores to device memory when it retires, summed by finalize_kernel.  (A ticket
// that lets the LAST demod workgroup do the summing needs an agent-scope release per workgroup = an L2 write-back on its
// XCD; the workgroups that finish early then slow their neighbours' loads down - measured: the kernel went from 38 to
// 57 us, profiles/r02p.  A kernel boundary does that cache maintenance once.)
struct WgTotals {
    unsigned long long n_forwarded, n_preambles;
    uint32_t flags;          // bit 0: a run overflowed its slot list; bit 1: internal inconsistency (pre-test vs full gate)
    uint32_t pad;
};

// What finalize_kernel tells the host, in pinned host memory: the totals of the call, whether the ordered list is
// complete already, and - written LAST, behind a system-scope release - the sequence number of the call.
// modes_gpu_fetch polls that word: the completion of a call costs no packet in the stream (an event record between the
// last kernel and the next call's scan kernel is 5.6 us of idle GPU; timing events around the kernels 9 us each:
// profiles/r02f), so consecutive calls run back to back.
struct HostHeader {
    unsigned long long n_forwarded, n_preambles;
    uint32_t n_records;
    uint32_t flags;
    uint32_t ordered;
    uint32_t seq;
};

struct ScanParams {
    const uint8_t *iq;    // 16-byte aligned base
    int64_t lo, hi;       // valid bytes [lo, hi) relative to iq (lo < 16: alignment slack)
    uint64_t g0;          // framed coordinate of buffer sample 0
    int64_t p_begin;      // positions [p_begin, p_end) are tested
    int64_t p_end;
    uint32_t nchunks;
    uint32_t run_chunks;
    uint32_t nruns;
    uint32_t slot_cap;
    uint32_t *slots;      // [nruns][slot_cap]
    uint32_t *counts;     // [nruns]  (true count, may exceed slot_cap -> overflow flag)
    ResultHeader *hdr;
};

// ------------------------------------------------------------------------------------
// demod_kernel parameters
// ------------------------------------------------------------------------------------
#ifdef MODES_TRACE
__device__ unsigned long long g_trace[8192 * 8];      // per demod wavefront: start, table staged, end, preambles of its workgroup,
                                                      // then summed over its batches: set-up, stage 1, stage 2a, stages 2b + 3
#define TRACE_T(var) const unsigned long long var = wall_clock64()
#else
#define TRACE_T(var)
#endif
// Records: a workgroup reserves the staging slots of one block of positions with ONE global atomic (it knows
// the number of records of the block before it writes any: the noise-gate pre-test is exact) and writes them
// ranked by position; every slot also gets a key (batch, rank inside the batch).  order_kernel then moves the
// records to (exclusive prefix of the batch counts) + rank: the final list is in stream order on the DEVICE -
// no holes, no host sort - ready for a device-to-host copy or for a gather over RCCL.
constexpr uint32_t kNoPos = 0xFFFFFFFFu;          // a survivor whose first gate failed after all (edge of the span)
constexpr uint64_t kNoKey = ~0ull;

constexpr int kDemodGroup = 64;       // runs whose slot lists one demod workgroup walks together (one per lane of a wavefront)

struct DemodParams {
    const uint8_t *iq;
    int64_t lo, hi;
    uint64_t g0;
    uint32_t nruns;
    uint32_t run_chunks;
    uint32_t slot_cap;
    const uint32_t *slots;
    const uint32_t *counts;
    DeviceTables tab;
    int maxfix;                // 0 = --no-fix, 1 = default, 2 = --aggressive   (dump1090.c:1115)
    int aggressive;            // Modes.aggressive by itself (dump1090.c:1731: it also admits attempts with 1-2 slicing errors)
    uint32_t *cand_slots;      // [nruns][slot_cap] or nullptr
    uint32_t *cand_counts;     // [nbatches]
    modes_record *staging;     // records in completion order
    uint64_t *keys;            // per staging slot: batch << 32 | rank of the record inside its batch
    uint32_t *batch_count;     // [nbatches] records of each batch
    uint32_t *batch_off;       // [nbatches] exclusive prefix of batch_count (written by the last workgroup to retire)
    uint32_t nbatches;
    ResultHeader *hdr;         // device: slot reservation counter (zeroed by the scan kernel)
    WgTotals *totals;          // [gridDim.x] what each workgroup found (device memory)
    uint32_t max_records;
};

// ------------------------------------------------------------------------------------
// scan_kernel - two passes over every chunk, because the stage is VALU-issue-bound (DESIGN.md 3.1 has the measured
// issue costs: every packed / VOP3 op is half rate on gfx950):
//
//   alpha  every position: the ten ORDERING relations (exact on s for even positions, superset
//          for odd ones: modes_order8_swar).  ~1.5 % of positions survive; lanes owning a survivor
//          push {their 24-sample window, the four result words, the window's position} into a
//          wave-private LDS queue.
//   beta   whenever the queue (59 entries) cannot take the next chunk's: one lane per entry, plain 32-bit integers: the level
//          bound 9*max(quiet) < s0+s2+s7+s9 for the survivors of that entry.  Dense lanes, so the
//          level test costs ~10x less per chunk than evaluating it packed at every position.
//
// The chunk loop is unrolled by two so that the ring's two address sets are static: all LDS addresses are
// loop-invariant VGPRs, the prefetched chunk lives in the registers it was loaded into (no
// rotation moves), and global addresses are SGPR base + constant lane offset.
//
// Forwarded positions of a run ARE in ascending order (scan_beta), and so is the concatenation over runs.
// ------------------------------------------------------------------------------------
// Cache policy of the stream loads (gfx940+ bits: 1 = sc0, 2 = nt, 16 = sc1).  Every byte is read once: nt | sc1
// takes the kernel from 0.206 to 0.194 ms per GiB (nt alone 0.196, sc0 / sc1 alone: nothing; tools/ab_scan.py).
#ifndef SCAN_AUX
#define SCAN_AUX 18
#endif
constexpr int kScanAux = SCAN_AUX;
#ifndef DEMOD_AUX
#define DEMOD_AUX 0
#endif
constexpr int kDemodAux = DEMOD_AUX;   // the demod kernel's sample loads (candidates only: ~150 MB per GiB): default policy.
                                       // nt or nt | sc1: 0.0408 ms instead of 0.0354 (they re-read what the scan just streamed), sc0: same
// LDS per wavefront: the ring (66 x 16 B) + the queue.  59 entries: 2 x (1056 + 59 x 68) = 10,136 B per workgroup = 16 workgroups =
// 32 wavefronts per CU, the most a CU holds (8 per SIMD; 55 VGPRs).  With the two-slot ring and 64 entries of rounds 1-3 it was 24:
// profiles/r07/ab_scan_ring66.txt (-3 %; the kernel is VALU-bound and wants every wavefront it can get: 20 per CU +4 %, 16 +9 %).
#ifndef SCAN_QCAP
#define SCAN_QCAP 59
#endif
constexpr int kQCap = SCAN_QCAP;      // queue entries per wavefront: one beta pass (a lane per entry: at most 64; the push handles < 64)
static_assert(kQCap >= 32 && kQCap <= 64, "queue capacity");
constexpr int kRingEntries = 66;
constexpr int kQStride = 17;          // dwords per entry: 11 window + 4 results + 1 position + 1 pad (odd: no bank conflicts)
constexpr int kScan2Waves = 2;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wavefront execute in issue order: a read sees an earlier write of the
    // same wavefront without a barrier; only the compiler must not reorder them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One pass over the queue: a lane per entry.  Everything that is the same for (almost) every entry is kept out of the
// per-survivor loop - the pass costs ~12 % of the kernel, and all of it is instruction issue:
//   * the eight ordering flags of an entry (bits 15 / 31 of its four result words) are gathered with two v_perm and
//     walked in whatever bit order that leaves them in (the level bound does not care);
//   * the span and framing rules (p_begin <= p < p_end, j < 131070) hold for all eight positions of an entry except
//     at the two ends of a call and for one window per buffer: one test per entry, a per-position loop only there;
//   * a lane forwards at most one position in all but ~1 pass in 200: one ballot ranks the lanes; the general
//     prefix (a ballot per bit of the per-lane count) runs only when some lane has two.
__device__ __forceinline__ void scan_beta(const ScanParams &P, const uint32_t *queue, uint32_t nb, int lane,
                                          uint32_t *my_slots, uint32_t &count) {
    const bool act = (uint32_t)lane < nb;
    const uint32_t *e = queue + lane * kQStride;
    uint32_t f = 0, p0 = 0;
    if (act) {
        // byte k of `lo` = the flag byte of position k (its bit 7), byte k of `hi` = position 4 + k
        const uint32_t lo = __builtin_amdgcn_perm(e[12], e[11], 0x07050301u), hi = __builtin_amdgcn_perm(e[14], e[13], 0x07050301u);
        f = ((lo & 0x80808080u) >> 7) | ((hi & 0x80808080u) >> 3);          // bit 8k <-> position k, bit 8k + 4 <-> position 4 + k
        p0 = e[15];
    }
    const uint16_t *w = reinterpret_cast<const uint16_t *>(e);
    uint32_t fwd8 = 0;                                                       // bit i: position p0 + i is forwarded
    while (f) {                                                              // lanes with alpha survivors only
        const uint32_t b = (uint32_t)__builtin_ctz(f);
        f &= f - 1;
        const uint32_t i = (b >> 3) | (b & 4u);
        const uint16_t *x = w + i;
        const uint32_t s0 = x[0], s2 = x[2], s4 = x[4], s5 = x[5], s7 = x[7], s9 = x[9], s11 = x[11], s12 = x[12],
                       s13 = x[13], s14 = x[14];
        const uint32_t quiet = max(max(max(s4, s5), max(s11, s12)), max(s13, s14));
        // (with kPowBias b on every value: 9 (q + b) < sum + 4 b + 4 + 5 b)
        fwd8 |= (modes_level_bound(s0, s2, s7, s9 + 5u * kPowBias, quiet) ? 1u : 0u) << i;
    }
    // 32-bit forms of p_begin <= p < p_end and of the framing rule j < 131070 (:1593): positions of a call are below
    // 2^32 - 2^15, so a wrapped look-back position (the 16 positions in front of chunk 0) fails the first test, and j
    // only needs p + g0 mod 2^17.  `plain`: they hold for p0 .. p0 + 7 alike.
    const uint32_t p_begin32 = (uint32_t)P.p_begin, p_span32 = (uint32_t)(P.p_end - P.p_begin), g0_lo = (uint32_t)P.g0;
    const uint32_t span_m7 = p_span32 >= 8u ? p_span32 - 7u : 0u;            // wave-uniform
    const bool plain = (p0 - p_begin32) < span_m7 && ((p0 + g0_lo) & (MODES_BLOCK_STRIDE - 1)) < MODES_BLOCK_POSITIONS - 7;
    if (fwd8 != 0 && !plain) {                                               // ends of the call, one window per buffer
        uint32_t keep = 0;
        for (uint32_t m = fwd8; m; m &= m - 1) {
            const uint32_t i = (uint32_t)__builtin_ctz(m), p = p0 + i;
            if ((p - p_begin32) < p_span32 && ((p + g0_lo) & (MODES_BLOCK_STRIDE - 1)) < MODES_BLOCK_POSITIONS) keep |= 1u << i;
        }
        fwd8 = keep;
    }
    // Entries were queued in ascending p0 and own disjoint 8-position windows, so (lane, bit) order is position
    // order: the slot list of a run is ASCENDING, and so is the concatenation over runs - the demod kernel ranks
    // its records by position without a sort.
    const uint64_t any = __ballot(fwd8 != 0);
    if (any == 0) return;                                                    // wave-uniform
    const uint64_t multi = __ballot((fwd8 & (fwd8 - 1)) != 0);
    if (multi == 0) {                                                        // one position per forwarding lane
        const uint32_t idx = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
        if (fwd8 != 0 && idx < P.slot_cap) my_slots[idx] = p0 + (uint32_t)__builtin_ctz(fwd8);
        count += (uint32_t)__builtin_popcountll(any);
        return;
    }
    // exclusive prefix over the lanes of a count <= 8: one ballot per bit
    const uint32_t cnt = (uint32_t)__builtin_popcount(fwd8);
    uint32_t excl = 0, total = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const uint64_t bb = __ballot(((cnt >> b) & 1u) != 0);
        if (bb) {                                                            // wave-uniform
            excl += __builtin_amdgcn_mbcnt_hi((uint32_t)(bb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bb, 0u)) << b;
            total += (uint32_t)__builtin_popcountll(bb) << b;
        }
    }
    uint32_t idx = count + excl;
    while (fwd8) {
        const int i = __builtin_ctz(fwd8);
        fwd8 &= fwd8 - 1;
        if (idx < P.slot_cap) my_slots[idx] = p0 + (uint32_t)i;
        idx++;
    }
    count += total;
}

// GUARD: the run touches an end of the span (byte-wise bounds checks on every load).  Otherwise
// chunks come through a raw buffer descriptor: SGPR base and offset + constant lane offset, no
// address arithmetic in the vector unit.
template <bool GUARD>
__device__ __forceinline__ void scan_run(const ScanParams &P, uint32_t run, int lane, uint4 *ring4, uint32_t *queue) {
    const int64_t c0 = (int64_t)run * P.run_chunks;                          // even (run_chunks is even)
    const int64_t c1 = (c0 + (int64_t)P.run_chunks < (int64_t)P.nchunks) ? c0 + (int64_t)P.run_chunks : (int64_t)P.nchunks;
    const uint8_t *iq = P.iq;
    const int64_t lo = P.lo, hi = P.hi;
    const int64_t base_off = c0 * kChunkBytes;
    const uint32_t lane_off = (uint32_t)lane * 16u;

    // A lane's 24-sample window = the 16 samples of the two lanes before it (previous chunk's lanes 62, 63 for lanes 0, 1) + its own
    // 8 samples, which never leave its registers.  The ring: loop-invariant LDS pointers, the chunk loop unrolled by two.
    // 66 entries of 16 bytes instead of two slots of 64: lanes 0 .. 61 of EVERY chunk write entries 2 .. 63 (a wavefront's LDS
    // operations execute in order: the previous chunk's reads are done), only lanes 62, 63 - the two that the NEXT chunk's lanes
    // 0, 1 look back to - alternate between entries 64, 65 (even chunks) and 0, 1 (odd chunks).  An even chunk then reads entries
    // lane and lane + 1; an odd chunk the same except lanes 0, 1 (entries 64, 65) and lane 63's second read (entry 0).  Entries 0, 1
    // lie 64 entries below 64, 65: the same banks as the two-slot ring, every 16-lane group of a b128 operation conflict-free.
    // Every address is still a loop-invariant VGPR; 992 bytes less LDS per wavefront for no instruction.
    uint4 *const wr0 = ring4 + 2 + lane, *const wr1 = ring4 + (lane < 62 ? 2 + lane : lane - 62);
    const uint32_t ring_lds = (uint32_t)reinterpret_cast<uintptr_t>(ring4);  // LDS byte address (low half of the flat address)
    const uint32_t rd0a = ring_lds + 16u * lane, rd0b = rd0a + 16u;
    const uint32_t rd1a = ring_lds + 16u * (lane < 2 ? 64 + lane : lane), rd1b = ring_lds + 16u * (lane == 0 ? 65 : lane == 63 ? 0 : lane + 1);

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(iq + (GUARD ? 0 : base_off)), 0, 0x7fffffff, 0x00020000);
    // Chunk c0 + k is at byte offset k * 1024 of the run.  The prefetch runs two chunks ahead; past the end of the
    // run it re-reads the run's last chunk (a cache hit, one s_min) instead of pulling the next run's first two
    // chunks from HBM a second time (that cost 6 % extra traffic).  The offset is a scalar that just counts.
    const uint32_t nk = (uint32_t)(c1 - c0), last_off = (nk - 1) * kChunkBytes;
    auto load_at = [&](uint32_t off) -> uint4 {
        const uint32_t o = off < last_off ? off : last_off;
        if (!GUARD)
            return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, o, kScanAux));
        return load_iq16(iq, base_off + (int64_t)o + lane_off, lo, hi);
    };

    uint32_t count = 0, qn = 0;                                              // wave-uniform
    uint32_t *my_slots = P.slots + (uint64_t)run * P.slot_cap;

    // one chunk: powers `s` to slot `wr`, neighbours' through `rda/rdb`, alpha, push, maybe beta
    auto step = [&](const uint4 &s, uint4 *wr, uint32_t rda, uint32_t rdb, int64_t c) {
        *wr = s;
        wave_lds_fence();
        // two ds_read_b128, spelled out: hipcc splits the same loads written in C++ into six narrower
        // LDS reads here.  LDS operations of a wavefront complete in order, so these see the write above.
        u32x4 a, b;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a), "=&v"(b) : "v"(rda), "v"(rdb) : "memory");
        wave_lds_fence();
        const uint32_t E[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, s.x, s.y, s.z, s.w};
        uint32_t r[4];
        modes_order8_swar(E, r);
#if defined(SCAN_ABL_EXTRA)
        // calibration build (timing only; tools/ab_scan.py): SCAN_ABL_EXTRA more full-rate VALU instructions per chunk, on a
        // chain that feeds the survivor test so that they cannot be dropped - what one more instruction per chunk costs a
        // kernel that is VALU-bound (the price list of everything one might still fuse into it)
        {
            uint32_t extra = r[0];
#pragma unroll
            for (int t = 0; t < SCAN_ABL_EXTRA; t++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(extra) : "v"(E[t % 12]));
            r[3] |= extra & 1u;                                              // bit 0 is not a flag bit (MODES_ORDER_FLAGS)
        }
#endif

        // ---- alpha survivors -> queue; beta (level bound) whenever the next push might not fit ----
        const bool any = (((r[0] | r[1]) | (r[2] | r[3])) & MODES_ORDER_FLAGS) != 0;
        const uint64_t hb = __ballot(any);
#if defined(SCAN_ABL)
        count += (uint32_t)__builtin_popcountll(hb);        // ablation build (timing only): alpha kept alive, no hand-off
        if (false) {
#else
        if (hb) {
#endif
            const uint32_t npush = (uint32_t)__builtin_popcountll(hb);
            const uint32_t pos = (uint32_t)(c * kChunkSamples - kLookback + 8 * lane);
            auto push_entry = [&](uint32_t slot) {
                uint32_t *e = queue + __umul24(slot, (uint32_t)kQStride);                // v_mad_u32_u24, not a 32-bit multiply
#pragma unroll
                for (int t = 0; t < 11; t++) {
#if defined(SCAN_ABL_PUSH)
                    if (t >= 11 - SCAN_ABL_PUSH) continue;                   // ablation build (timing only): fewer LDS writes per push
#endif
                    e[t] = E[t];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) e[11 + q] = r[q];
                e[15] = pos;
            };
            // The level pass empties the queue whenever this chunk's entries do not fit.  `base`: pushing lanes (by rank) that are
            // queued already - not 0 only when MORE lanes push than the whole queue holds (kQCap < 64; every lane of a wavefront can
            // own an ordering survivor, tests/test_gpu_parity.py::test_every_lane_pushes): kQCap of them go in and through the pass
            // first.  Entries stay in lane order = position order either way.
            uint32_t base = 0;
            bool mine = any;                                                 // this lane still has to push
            while (qn + (npush - base) > kQCap) {
                if (kQCap < kWave && qn == 0) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0u)) - base;
                    if (mine && rank < (uint32_t)kQCap) push_entry(rank);
                    mine = mine && rank >= (uint32_t)kQCap;
                    qn = kQCap;
                    base += kQCap;
                    wave_lds_fence();
                }
#if !defined(SCAN_ABL_NOBETA)
                scan_beta(P, queue, qn, lane, my_slots, count);
#endif
                qn = 0;
                wave_lds_fence();
            }
            if (mine) {
                // rank among the pushing lanes: v_mbcnt counts the mask bits below this lane
                // (v_mbcnt adds the count to its last operand: the queue fill comes in for free)
                push_entry(__builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, qn - base)));
            }
            qn += npush - base;
            wave_lds_fence();
        }
    };

    // prologue: the last 16 samples of chunk c0-1 go where an odd chunk's lanes 62, 63 write (entries 0, 1)
    {
        const uint4 s = power16_scan(load_iq16(iq, base_off - kChunkBytes + lane_off, lo, hi));
        if (lane >= 62) *wr1 = s;
    }
    uint4 x = load_at(0), y = load_at(kChunkBytes);                          // two chunks in flight
    uint32_t k = 0, off = 2 * kChunkBytes;
    for (; k + 2 <= nk; k += 2, off += 2 * kChunkBytes) {
        const uint4 sx = power16_scan(x);
        x = load_at(off);
        step(sx, wr0, rd0a, rd0b, c0 + k);
        const uint4 sy = power16_scan(y);
        y = load_at(off + kChunkBytes);
        step(sy, wr1, rd1a, rd1b, c0 + k + 1);
    }
    if (k < nk) step(power16_scan(x), wr0, rd0a, rd0b, c0 + k);             // odd tail (last run only)
#if !defined(SCAN_ABL_NOBETA)
    if (qn) scan_beta(P, queue, qn, lane, my_slots, count);
#endif
    if (lane == 0) P.counts[run] = count;                 // true count; demod_kernel flags count > slot_cap
}

__global__ __launch_bounds__(kScan2Waves * kWave) void scan_kernel(ScanParams P) {
    __shared__ uint4 ring_all[kScan2Waves][kRingEntries];
    __shared__ uint32_t queue_all[kScan2Waves][kQCap * kQStride];
#if defined(SCAN_LDS_PAD)
    // occupancy experiment (tools/ab_scan.py): SCAN_LDS_PAD more bytes of LDS per workgroup -> fewer wavefronts per CU
    __shared__ uint32_t lds_pad[SCAN_LDS_PAD / 4];
    if (P.nruns == 0xFFFFFFFFu) { lds_pad[threadIdx.x] = 1; __syncthreads(); P.counts[threadIdx.x] = lds_pad[threadIdx.x ^ 1]; }   // never: keeps the array
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform: addresses stay in SGPRs
    const uint32_t run = blockIdx.x * kScan2Waves + wave;
    if (blockIdx.x == 0 && threadIdx.x == 0) *P.hdr = ResultHeader{0, 0, {0, 0}};   // demod_kernel reserves record slots in it
    if (run >= P.nruns) return;
    // Runs whose loads (chunks c0-1 .. c1+1, the prefetch runs two chunks ahead) all lie inside the
    // span take the unguarded instantiation; that is every run except the first and the last few.
    const int64_t c0 = (int64_t)run * P.run_chunks;
    const int64_t first = (c0 - 1) * kChunkBytes, last = (c0 + (int64_t)P.run_chunks + 2) * kChunkBytes;
    if (first >= P.lo && last <= P.hi) scan_run<false>(P, run, lane, ring_all[wave], queue_all[wave]);
    else                               scan_run<true>(P, run, lane, ring_all[wave], queue_all[wave]);
}

// Sample / dword loads for the demod kernel.  GUARD = false: plain loads (the caller has checked,
// wave-uniformly, that every byte it will touch lies inside the span) - no branches, so the loads
// of one iteration pipeline.  GUARD = true: bytes outside [lo, hi) read as 127 (no signal); used
// only by wavefronts that work next to an end of the span.
template <bool GUARD>
__device__ __forceinline__ uint32_t load_dword(const uint8_t *iq, int64_t o, int64_t lo, int64_t hi) {
    if (!GUARD || (o >= lo && o + 4 <= hi)) return *reinterpret_cast<const uint32_t *>(iq + o);
    uint32_t v = 0x7f7f7f7fu;
#pragma nounroll
    for (int b = 0; b < 4; b++)
        if (o + b >= lo && o + b < hi) v = (v & ~(0xffu << (8 * b))) | ((uint32_t)iq[o + b] << (8 * b));
    return v;
}
template <bool GUARD>
__device__ __forceinline__ uint32_t load_sample(const uint8_t *iq, int64_t sample, int64_t lo, int64_t hi) {
    const int64_t o = 2 * sample;
    if (!GUARD || (o >= lo && o + 2 <= hi)) return *reinterpret_cast<const uint16_t *>(iq + o);
    uint32_t v = 0x7f7fu;
    if (o >= lo && o < hi) v = (v & 0xff00u) | iq[o];
    if (o + 1 >= lo && o + 1 < hi) v = (v & 0x00ffu) | ((uint32_t)iq[o + 1] << 8);
    return v;
}
// true when samples [first, last] are entirely inside the span
__device__ __forceinline__ bool samples_inside(int64_t first, int64_t last, int64_t lo, int64_t hi) {
    return 2 * first >= lo && 2 * last + 2 <= hi;
}
// The magnitude table as the demodulation kernels see it: the whole table - the reference's LUT by saturated power - in
// 64 KiB of LDS, shared by the wavefronts of a workgroup.  (Until round 3 a second form lived here - the first 512 entries
// in 1 KiB of LDS, modes_mag_exact beyond, for 4-wave workgroups small enough to sit next to the scan kernel's: 0.055-0.079 ms
// alone against 0.036, and a kernel that runs UNDER the scan slows the scan down by more than it hides - DESIGN.md 3.2.  The
// table left in global memory and read through L1 / L2: 2.5x slower alone, the scan +28 % - profiles/r02a.)
struct LutFull {
    const uint16_t *p;
    __device__ __forceinline__ uint32_t operator[](uint32_t idx) const { return p[idx]; }
};
// magnitude of a sample packed as I | Q << 8 (low 16 bits)
template <class Lut>
__device__ __forceinline__ int mag_of(const Lut lut, uint32_t iq16) {
    return lut[modes_lut_index(iq16 & 0xff, (iq16 >> 8) & 0xff)];
}

// Sum / XOR over the wavefront (every lane active), the result in every lane - in fact in a scalar register.  Four DPP steps leave
// every lane of a row of 16 with its row's value (quad butterflies, then the mirrored half row and the mirrored row), four
// v_readlane collect the rows: ~80 cycles.  (__shfl_xor compiles to ds_bpermute_b32, a round trip through the LDS crossbar per
// step: six of them per reduction, ~24 reductions per record - a quarter of record_kernel's time, which is one dependent chain per
// wavefront; profiles/r08/ab_record_kernel.txt.)
__device__ __forceinline__ int wave_rows_sum(int v) {
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);                  // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);                  // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);                 // row_half_mirror
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true);                 // row_mirror
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
    v = wave_rows_sum(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// bits of `mask` below this lane (v_mbcnt: no per-lane mask register to keep alive)
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Ballot of a per-pair flag: lane L contributes pair L (f1) and pair L+64 (f2, lanes < 48).
__device__ __forceinline__ modes_m128 pair_ballot(bool f1, bool f2) {
    return m128_make(__ballot(f1), __ballot(f2) & 0x0000FFFFFFFFFFFFull);
}
__device__ __forceinline__ bool m128_bit(modes_m128 m, int k) {      // k may be per-lane
    return (((k < 64) ? (m.lo >> (k & 63)) : (m.hi >> (k & 63))) & 1ull) != 0;
}

// XOR over the wavefront.
__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
    int x = (int)v;
    x ^= __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true);
    x ^= __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true);
    x ^= __builtin_amdgcn_mov_dpp(x, 0x141, 0xf, 0xf, true);
    x ^= __builtin_amdgcn_mov_dpp(x, 0x140, 0xf, 0xf, true);
    return (uint32_t)(__builtin_amdgcn_readlane(x, 0) ^ __builtin_amdgcn_readlane(x, 16) ^ __builtin_amdgcn_readlane(x, 32) ^ __builtin_amdgcn_readlane(x, 48));
}

// Syndrome and repair lookup of one attempt by the whole wavefront (dump1090.c:1104, 1112-1117, 854-880).
// `bits`: mask of the demodulated message (bit k = message bit k, MSB first), wave-uniform.
// The syndrome is linear: the XOR of x^(111-p) mod G over the set bits, p = frame position = k + 112 - nbits;
// lane L contributes frame positions L and L + 64.  The repair search of modes_find_fix (same result:
// all 1- and 2-bit syndromes are distinct) runs one candidate first position per lane.  s_esyn: the
// 112 single-bit syndromes in LDS.  Results are wave-uniform.
struct AttemptFix {
    uint32_t syndrome;
    uint8_t nfix, pos0, pos1;
};
// The 112 single-bit syndromes by value: a 512-entry open-addressing table (key << 8 | position, 0xFFFFFFFF = empty) the host
// builds next to them (modes_gpu_create: multiplicative hash, linear probing; the longest run of occupied cells is 3, so four
// probes decide).  -> the position whose single-bit syndrome is `key`, 255 if there is none.
constexpr int kSynHashBits = 9, kSynProbes = 4;
constexpr uint32_t kSynHashMul = 0x85EBCA6Bu, kSynEmpty = 0xFFFFFFFFu;
constexpr int kSynWords = 112 + (1 << kSynHashBits);                         // esyn[112] | table[512]
__device__ __forceinline__ uint32_t syn_lookup(const uint32_t *table, uint32_t key) {
    const uint32_t h = (key * kSynHashMul) >> (32 - kSynHashBits);
    uint32_t q = 255u;
#pragma unroll
    for (int t = 0; t < kSynProbes; t++) {
        const uint32_t v = table[(h + (uint32_t)t) & ((1u << kSynHashBits) - 1)];
        if ((v >> 8) == key && v != kSynEmpty) q = v & 0xffu;
    }
    return q;
}
__device__ __forceinline__ int df_of_bits(modes_m128 bits) {                 // msg[0] >> 3
    const uint32_t v = (uint32_t)bits.lo;
    return (int)(((v & 1u) << 4) | ((v & 2u) << 2) | (v & 4u) | ((v & 8u) >> 2) | ((v & 16u) >> 4));
}
__device__ __forceinline__ AttemptFix wave_finish_attempt(modes_m128 bits, bool gate_ok, int maxfix, int lane,
                                                          const uint32_t *s_esyn) {
    AttemptFix r{0, 0, 0xff, 0xff};
    if (!gate_ok) return r;
    const int df = df_of_bits(bits);
    const int nbits = modes_len_by_df(df);
    const int shift = 112 - nbits;
    const int pA = lane, pB = lane + 64;                                     // this lane's two frame positions
    const uint32_t eA = s_esyn[pA], eB = pB < 112 ? s_esyn[pB] : 0u;
    uint32_t c = 0;                                                          // message bit k sits at frame position k + shift
    {
        const int kA = pA - shift, kB = pB - shift;
        if (kA >= 0 && ((kA < 64 ? bits.lo >> kA : bits.hi >> (kA - 64)) & 1ull)) c ^= eA;
        if (pB < 112 && kB >= 0 && ((kB < 64 ? bits.lo >> kB : bits.hi >> (kB - 64)) & 1ull)) c ^= eB;
    }
    const uint32_t syn = wave_xor(c);
    r.syndrome = syn;
    if (syn == 0 || maxfix < 1 || !(df == 11 || df == 17 || df == 18)) return r;
    const int first = (nbits == 112) ? 5 : 56;                               // frame bits this length may repair
    const bool okA = pA >= first, okB = pB >= first && pB < 112;
    {   // one flipped bit
        const uint64_t bA = __ballot(okA && eA == syn), bB = __ballot(okB && eB == syn);
        if (bA | bB) {
            const int p = bA ? __builtin_ctzll(bA) : 64 + __builtin_ctzll(bB);
            r.nfix = 1;
            r.pos0 = (uint8_t)(p - shift);
            return r;
        }
    }
    if (maxfix < 2) return r;
    // Two flipped bits p < q: syn == esyn[p] ^ esyn[q].  Lanes hold p; the q that goes with it - if any - is the position whose
    // single-bit syndrome is syn ^ esyn[p]: one hash lookup per candidate p instead of a walk over all q (107 wave-wide
    // rounds of two ballots each - it made the record stage of the low-SNR --aggressive workload 0.08 ms per GiB).  All one- and
    // two-bit syndromes over the repairable positions are distinct (dump1090.c:795-841 relies on it), so at most one pair
    // answers and the order of the search cannot matter.  s_esyn[112 ..]: the table (syn_lookup).
    {
        const uint32_t qA = okA ? syn_lookup(s_esyn + 112, syn ^ eA) : 255u, qB = okB ? syn_lookup(s_esyn + 112, syn ^ eB) : 255u;
        const uint64_t bA = __ballot(qA < 112u && (int)qA > pA), bB = __ballot(qB < 112u && (int)qB > pB);
        if (bA | bB) {
            const int src = bA ? __builtin_ctzll(bA) : __builtin_ctzll(bB);                 // the lane that holds p
            const int p = bA ? src : 64 + src;
            const int q = (int)(bA ? __builtin_amdgcn_readlane((int)qA, src) : __builtin_amdgcn_readlane((int)qB, src));
            r.nfix = 2;
            r.pos0 = (uint8_t)(p - shift);
            r.pos1 = (uint8_t)(q - shift);
        }
    }
    return r;
}

// The record, written by the whole wavefront: lane i computes BYTE i of the 64-byte modes_record from the wave-uniform results (lanes
// 0-7 block and j; 8-35 attempt 0; 36-63 attempt 1 - msg by one shift and v_bfrev per byte, the class byte and whitelist slot of
// include/modes_gfx950.h MODES_CLS_* / modes_classify once for BOTH attempts, each on the lanes of its own half), three quad
// permutes pack the four bytes of a dword into every lane of their quad, and lanes 4k store dword k: one coalesced 64-byte store
// (lanes 4k + 1: the same dword to the host's pinned copy).  Round 6: one lane used to write the record field by field - 14 byte
// reversals, a classification and 30 stores per attempt, twice -, and the class byte had made record_kernel 14 % slower on the
// frames stream (43.9 -> 50.1 us per 4 GiB call).  `maxfix` / `aggressive`: the context's repair policy.
__device__ __forceinline__ void store_record(modes_record *dev, modes_record *host, int lane, uint32_t block, uint32_t j,
                                             modes_m128 bits0, uint32_t err0, AttemptFix f0, modes_m128 bits1, uint32_t err1, bool gate1,
                                             AttemptFix f1, int maxfix, int aggressive) {
    const bool second = lane >= 36;                                          // this lane's attempt
    const int o = lane - (second ? 36 : 8);                                  // its byte's offset in the attempt (negative: the header)
    const uint64_t lo = second ? bits1.lo : bits0.lo, hi = second ? bits1.hi : bits0.hi;
    const uint32_t err = second ? err1 : err0, gate = second ? (gate1 ? 1u : 0u) : 1u;
    const uint32_t syn = second ? f1.syndrome : f0.syndrome, nfix = second ? f1.nfix : f0.nfix;
    const uint32_t pos0 = second ? f1.pos0 : f0.pos0, pos1 = second ? f1.pos1 : f0.pos1;
    // message byte b: pairs 8b .. 8b + 7 are bits 8b .. of the mask, pair 8b + t is bit 7 - t of the byte (modes_bits_to_msg)
    auto msg_byte = [&](int b) -> uint32_t {
        const uint32_t v = (uint32_t)((b < 8 ? lo >> (8 * b) : hi >> (8 * (b - 8))) & 0xFF);
        return __builtin_bitreverse32(v) >> 24;
    };
    const uint32_t m0 = msg_byte(0), m1 = msg_byte(1), m2 = msg_byte(2), m3 = msg_byte(3);
    const uint32_t cls = modes_classify((int)(m0 >> 3), err, gate, syn, nfix, maxfix > 0 ? 1u : 0u, aggressive ? 1u : 0u);
    const uint32_t wslot = modes_class_slot(cls, m1, m2, m3, syn);
    uint32_t byte;
    if (lane < 4) byte = block >> (8 * lane);
    else if (lane < 8) byte = j >> (8 * (lane - 4));
    else if (o < 14) byte = msg_byte(o);
    else if (o < 24) {
        const uint32_t low = err | (gate << 8) | (nfix << 16) | (pos0 << 24);                // bytes 14 .. 17
        const uint64_t mid = (uint64_t)low | ((uint64_t)(pos1 | ((cls & 0xFF) << 8) | ((wslot & 0xFFFF) << 16)) << 32);   // .. 21; 22, 23: 0
        byte = (uint32_t)(mid >> (8 * (o - 14)));
    } else byte = syn >> (8 * (o - 24));
    byte &= 0xFF;
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)byte, 0x00, 0xf, 0xf, true);   // quad_perm [0,0,0,0]
    const uint32_t b1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)byte, 0x55, 0xf, 0xf, true);   // [1,1,1,1]
    const uint32_t b2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)byte, 0xAA, 0xf, 0xf, true);   // [2,2,2,2]
    const uint32_t b3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)byte, 0xFF, 0xf, 0xf, true);   // [3,3,3,3]
    const uint32_t dword = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    const int q = lane & 3;
    uint32_t *dst = q == 0 ? reinterpret_cast<uint32_t *>(dev) : reinterpret_cast<uint32_t *>(host);
    if (q < (host ? 2 : 1)) dst[lane >> 2] = dword;
}

// The delta sums of dump1090.c:1713-1717 over the first 56 and over all 112 pairs (lane L: pairs L, L + 64).
__device__ __forceinline__ void delta_sums(int lane, int lo1, int hi1, int lo2, int hi2, int *sum56, int *sum112) {
    const int d1 = lo1 > hi1 ? lo1 - hi1 : hi1 - lo1, d2 = lo2 > hi2 ? lo2 - hi2 : hi2 - lo2;
    *sum112 = wave_sum(d1 + (lane < 48 ? d2 : 0));                           // < 2^23
    *sum56 = wave_sum(lane < 56 ? d1 : 0);
}

// One slicing pass for the whole wavefront: lane L holds pairs k1 = L and k2 = L + 64.
// Returns the packed message (wave-uniform) and, on request, the two delta sums.
__device__ __forceinline__ modes_m128 slice_pass(int lane, int lo1, int hi1, int lo2, int hi2, uint8_t *errors,
                                                 int *sum56, int *sum112) {
    bool w1, s1, w2, s2;
    int d1, d2;
    modes_pair_flags(lane, lo1, hi1, &w1, &s1, &d1);
    modes_pair_flags(lane + 64, lo2, hi2, &w2, &s2, &d2);
    const bool two = lane < 48;
    const modes_m128 weak = pair_ballot(w1, two && w2);
    const modes_m128 strong = pair_ballot(s1, two && s2);
    const bool first_equal = (__ballot(lo1 == hi1) & 1ull) != 0;             // pair 0 lives in lane 0
    const modes_m128 bits = modes_pack_bits(weak, strong, first_equal, errors);
    if (sum56) {
        // both sums in one reduction: sum112 < 2^23, sum56 < 2^22
        const int all = d1 + (two ? d2 : 0);
        const int first = (lane < 56) ? d1 : 0;
        *sum112 = wave_sum(all);
        *sum56 = wave_sum(first);
    }
    return bits;
}

// Both LUT indices of a dword I0 Q0 I1 Q1 (two samples) in one packed value: the saturated powers.
__device__ __forceinline__ uint32_t pk_lut_index(uint32_t w) { return modes_power_pair_sat(w); }

// Exact preamble predicate (dump1090.c:1602-1650) at buffer sample p, magnitudes from the LDS LUT.
// Guarded form: 2-byte loads, bytes outside the span read as 127.
template <class Lut>
__device__ __forceinline__ bool preamble_at_guarded(const uint8_t *iq, int64_t lo, int64_t hi, const Lut lut, uint32_t p) {
    int m[15];
#pragma unroll
    for (int t = 0; t < 15; t++) m[t] = mag_of(lut, load_sample<true>(iq, (int64_t)p + t, lo, hi));
    struct Win { const int *m; __device__ int operator()(int t) const { return m[t]; } };
    return modes_preamble_exact(Win{m});
}
// Fast form: the 15 samples are two 16-byte loads at a 2-byte aligned address (unaligned-access mode, raw buffer
// descriptor); voff = byte offset of sample p from the descriptor's base.  Indexed and looked up dword by dword.
template <class Lut>
__device__ __forceinline__ bool preamble_at_fast(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, const Lut lut) {
    const u32x4 wa = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, kDemodAux), wb = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16u, 0, kDemodAux);
    const uint32_t w[8] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
    int m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t ix = pk_lut_index(w[i]);
        m[2 * i] = (int)lut.p[ix & 0xffffu];
        m[2 * i + 1] = (int)lut.p[ix >> 16];
    }
    struct Win { const int *m; __device__ int operator()(int t) const { return m[t]; } };
    return modes_preamble_exact(Win{m});
}

// Noise-gate pre-test (dump1090.c:1713-1723): the sum of |lo - hi| over 56 consecutive bit pairs
// (a short message, or the second half of a long one) by a group of kGateLanes = 4 lanes.
// The 56 pairs are 224 consecutive bytes at a 2-byte aligned address; lane t of the group fetches the
// 16-byte pieces t, t + 4, t + 8 and (t < 2) t + 12 through a raw buffer descriptor (the hardware runs
// in unaligned-access mode; no alignment is promised to the compiler), so the four lanes consume 64
// consecutive bytes per instruction and every line is pulled into the L1 once.  A bit pair is one
// dword (I_lo Q_lo I_hi Q_hi); both LUT indices (saturated powers) come out of one packed
// multiply + multiply-add.  voff = byte offset of the first pair from the descriptor's base.
constexpr uint32_t kUnknown = 0xFFFFFFFFu;      // a delta sum the pre-test did not need
constexpr int kGateLanes = 4;                   // lanes per preamble in the gate pre-test
__device__ __forceinline__ void half_load(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, int t, u32x4 (&w)[4]) {
    const uint32_t o = voff + 16u * (uint32_t)t;
#pragma unroll
    for (int i = 0; i < 3; i++) w[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + 64u * i, 0, kDemodAux);
    w[3] = u32x4{0, 0, 0, 0};                                                // four equal bytes: a pair with |lo - hi| = 0
    if (t < 2) w[3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + 192u, 0, kDemodAux);
}
// Returns this lane's part of the sum.  *first: flags of the lane's first four pairs (pairs 4t .. 4t+3 of
// the 56), bit k = |lo - hi| < 256, bit 4 + k = lo > hi, bit 8 = (lo == hi) of its first pair.  Indexed and looked up
// dword by dword: the gathers spread over the VALU work.
template <class Lut>
__device__ __forceinline__ uint32_t half_eval(const u32x4 (&w)[4], const Lut lut, uint32_t *first) {
    uint32_t acc = 0, f = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t ix = pk_lut_index(w[i][k]);                       // both LUT indices
            const uint32_t a = lut.p[ix & 0xffffu], b = lut.p[ix >> 16];
            if (i == 0 && first) {
                const uint32_t d = __builtin_amdgcn_sad_u16(a, b, 0u);
                f |= (d < 256u ? 1u : 0u) << k | (a > b ? 1u : 0u) << (4 + k);
                if (k == 0) f |= (a == b ? 1u : 0u) << 8;
                acc += d;
            } else {
                acc = __builtin_amdgcn_sad_u16(a, b, acc);                   // += |a - b| (high halves are zero)
            }
        }
    }
    if (first) *first = f;
    return acc;
}
__device__ __forceinline__ void surv_push(uint32_t *list, uint32_t k, uint32_t p, uint32_t sum56, uint32_t sum112) {
    list[3 * k] = p;
    list[3 * k + 1] = sum56;
    list[3 * k + 2] = sum112;
}

// First half of the demodulation of one preamble by the whole wavefront: magnitudes, the delta sums of
// dump1090.c:1713-1717 as far as they are needed, the first slicing pass and its noise gate.
// known56 / known112: the sums the pre-test already has (kUnknown otherwise).
struct Front {
    int lo1, hi1, lo2, hi2, pre;
    int sum56, sum112;
    bool have112, gate0;
    modes_m128 bits0;
    uint8_t err0;
};
// The samples of one preamble's message window as the wavefront loads them (raw I | Q << 8): lane L holds bit pairs k1 = L and
// k2 = L + 64 (samples 16 + 2 k, 17 + 2 k) and, lanes 0 .. 11, sample L - 1 of the preamble.  Loading and evaluating are separate so
// that record_kernel can have the NEXT record's loads in flight while it demodulates the current one.
struct FrontRaw {
    uint32_t lo1, hi1, lo2, hi2, pre;
};
template <bool GUARD>
__device__ __forceinline__ FrontRaw front_load(const DemodParams &P, int lane, int64_t pc) {
    const uint8_t *iq = P.iq;
    const int64_t lo = P.lo, hi = P.hi;
    const bool two = lane < 48;
    FrontRaw r;
    r.lo1 = load_sample<GUARD>(iq, pc + 16 + 2 * lane, lo, hi);
    r.hi1 = load_sample<GUARD>(iq, pc + 17 + 2 * lane, lo, hi);
    r.lo2 = two ? load_sample<GUARD>(iq, pc + 144 + 2 * lane, lo, hi) : 0x7f7fu;
    r.hi2 = two ? load_sample<GUARD>(iq, pc + 145 + 2 * lane, lo, hi) : 0x7f7fu;
    r.pre = (lane < 12) ? load_sample<GUARD>(iq, pc - 1 + lane, lo, hi) : 0x7f7fu;
    return r;
}
template <class Lut>
__device__ __forceinline__ Front front_eval(const Lut lut, int lane, const FrontRaw &r, uint32_t known56, uint32_t known112) {
    const bool two = lane < 48;
    Front f;
    f.lo1 = mag_of(lut, r.lo1);
    f.hi1 = mag_of(lut, r.hi1);
    f.lo2 = two ? mag_of(lut, r.lo2) : 0;
    f.hi2 = two ? mag_of(lut, r.hi2) : 0;
    f.pre = (lane < 12) ? mag_of(lut, r.pre) : 0;
    f.sum56 = (int)known56;
    f.sum112 = (int)known112;
    f.have112 = known112 != kUnknown;
    if (known56 == kUnknown) { delta_sums(lane, f.lo1, f.hi1, f.lo2, f.hi2, &f.sum56, &f.sum112); f.have112 = true; }
    f.bits0 = slice_pass(lane, f.lo1, f.hi1, f.lo2, f.hi2, &f.err0, nullptr, nullptr);
    const bool long0 = modes_len_by_df(df_of_bits(f.bits0)) == 112;
    if (long0 && !f.have112) { delta_sums(lane, f.lo1, f.hi1, f.lo2, f.hi2, &f.sum56, &f.sum112); f.have112 = true; }
    f.gate0 = long0 ? (f.sum112 / 56 >= 2550) : (f.sum56 / 28 >= 2550);     // dump1090.c:1717-1723
    return f;
}
template <bool GUARD, class Lut>
__device__ __forceinline__ Front demod_front(const DemodParams &P, const Lut lut, int lane, int64_t pc, uint32_t known56,
                                             uint32_t known112) {
    return front_eval(lut, lane, front_load<GUARD>(P, lane, pc), known56, known112);
}

// Full demodulation (both attempts) -> the record in staging slot `slot`, keyed `key`.  Returns false when the first
// noise gate fails after all (then nothing is written; the caller has already ruled that out for its entries).
template <bool KEYED = true, class Lut>
__device__ __forceinline__ bool demod_rest(const DemodParams &P, const Lut lut, const uint32_t *s_esyn, int lane, int64_t pc,
                                           const FrontRaw &raw, uint32_t known56, uint32_t known112, uint32_t slot, uint64_t key,
                                           modes_record *host_rec = nullptr) {
    const uint64_t g = (uint64_t)pc + P.g0;
    const uint32_t j = (uint32_t)(g & (MODES_BLOCK_STRIDE - 1));
    const bool two = lane < 48;
    Front f = front_eval(lut, lane, raw, known56, known112);
    if (!f.gate0) return false;                                              // dump1090.c:1723-1726: position ends
    const int lo1 = f.lo1, hi1 = f.hi1, lo2 = f.lo2, hi2 = f.hi2, pre = f.pre;
    const modes_m128 bits0 = f.bits0;
    const uint8_t err0 = f.err0;

    uint8_t err1 = err0;
    modes_m128 bits1 = bits0;
    bool gate1 = true;
    if (j != 0) {                                                            // dump1090.c:1660
        uint32_t up, dn;
        const bool backward = modes_phase_factors(
            (uint32_t)__builtin_amdgcn_readlane(pre, 0), (uint32_t)__builtin_amdgcn_readlane(pre, 1),
            (uint32_t)__builtin_amdgcn_readlane(pre, 3), (uint32_t)__builtin_amdgcn_readlane(pre, 4),
            (uint32_t)__builtin_amdgcn_readlane(pre, 7), (uint32_t)__builtin_amdgcn_readlane(pre, 8),
            (uint32_t)__builtin_amdgcn_readlane(pre, 10), (uint32_t)__builtin_amdgcn_readlane(pre, 11), &up, &dn);
        int nlo1 = lo1, nhi1 = hi1, nlo2 = lo2, nhi2 = hi2;
        if (backward) {
            // hi of every pair is rescaled, walking from pair 111 down (dump1090.c:1519-1534)
            const int hu1 = (int)modes_scale((uint32_t)hi1, up), hd1 = (int)modes_scale((uint32_t)hi1, dn);
            const int hu2 = (int)modes_scale((uint32_t)hi2, up), hd2 = (int)modes_scale((uint32_t)hi2, dn);
            const modes_m128 Up = pair_ballot(lo1 > hu1, two && lo2 > hu2);
            const modes_m128 Dn = pair_ballot(lo1 > hd1, two && lo2 > hd2);
            modes_m128 Pm = m128_andn(Dn, Up);
            Pm.hi &= ~(1ull << 47);                                          // pair 111 starts the chain: c_111 = Up_111
            const modes_m128 cm = modes_chain_down(Up, Pm);
            nhi1 = m128_bit(cm, lane + 1) ? hd1 : hu1;                       // pair k uses c_(k+1)
            nhi2 = (lane == 47) ? hu2 : (m128_bit(cm, lane + 65) ? hd2 : hu2);
        } else {
            // lo of every pair is rescaled, walking from pair 0 up (dump1090.c:1542-1556)
            const int lu1 = (int)modes_scale((uint32_t)lo1, up), ld1 = (int)modes_scale((uint32_t)lo1, dn);
            const int lu2 = (int)modes_scale((uint32_t)lo2, up), ld2 = (int)modes_scale((uint32_t)lo2, dn);
            const modes_m128 Up = pair_ballot(lu1 > hi1, two && lu2 > hi2);
            const modes_m128 Dn = pair_ballot(ld1 > hi1, two && ld2 > hi2);
            modes_m128 Gm = Dn, Pm = m128_andn(Up, Dn);
            Gm.lo = (Gm.lo & ~1ull) | (Up.lo & 1ull);                        // pair 0 starts the chain: c_0 = Up_0
            Pm.lo &= ~1ull;
            const modes_m128 cm = modes_chain(Gm, Pm);
            nlo1 = (lane == 0) ? lu1 : (m128_bit(cm, lane - 1) ? lu1 : ld1);   // pair k uses c_(k-1)
            nlo2 = m128_bit(cm, lane + 63) ? lu2 : ld2;
        }
        bits1 = slice_pass(lane, nlo1, nhi1, nlo2, nhi2, &err1, nullptr, nullptr);
        // the gate of the retry: uncorrected deltas, the retry's own length (dump1090.c:1708-1723)
        const bool long1 = modes_len_by_df(df_of_bits(bits1)) == 112;
        if (long1 && !f.have112) delta_sums(lane, lo1, hi1, lo2, hi2, &f.sum56, &f.sum112);
        gate1 = long1 ? (f.sum112 / 56 >= 2550) : (f.sum56 / 28 >= 2550);
    }
    // syndromes and repair positions: the whole wavefront, both attempts (wave-uniform results)
    const AttemptFix f0 = wave_finish_attempt(bits0, true, P.maxfix, lane, s_esyn);
    const AttemptFix f1 = wave_finish_attempt(bits1, gate1, P.maxfix, lane, s_esyn);
    // the whole wavefront writes the record (store_record); with a host copy to fill (record_kernel, short lists) the same dwords there
    if (slot < P.max_records) {                                              // (wave-uniform)
        store_record(&P.staging[slot], host_rec, lane, (uint32_t)(g / MODES_BLOCK_STRIDE), j, bits0, err0, f0, bits1, err1, gate1, f1,
                     P.maxfix, P.aggressive);
        if (KEYED && lane == 0) P.keys[slot] = key;
    }
    return true;
}
template <bool GUARD, bool KEYED = true, class Lut>
__device__ __forceinline__ bool demod_full(const DemodParams &P, const Lut lut, const uint32_t *s_esyn, int lane, int64_t pc,
                                           uint32_t known56, uint32_t known112, uint32_t slot, uint64_t key,
                                           modes_record *host_rec = nullptr) {
    return demod_rest<KEYED>(P, lut, s_esyn, lane, pc, front_load<GUARD>(P, lane, pc), known56, known112, slot, key, host_rec);
}

// ------------------------------------------------------------------------------------
// demod_kernel - persistent workgroups; a workgroup takes a batch of kDemodGroup consecutive runs at a
// time and walks the concatenation of their (ascending) slot lists kDemodThreads positions per block,
// all its wavefronts together (the work per run varies several-fold; the sum over a batch does not):
//   stage 1: one thread per forwarded position: exact preamble predicate (dump1090.c:1602-1650)
//            on table magnitudes; survivors compacted into a workgroup list (the "valid preambles"
//            of --stats).
//   stage 2: noise-gate pre-test (dump1090.c:1713-1723), kGateLanes lanes per preamble:
//            a) the first 56 bit pairs: their sum of |lo-hi|, and from the first six pairs the DF of
//               the first slicing pass, i.e. the message length the gate is evaluated on - a short
//               message is decided here, a long one queues for b) its other 56 pairs.
//            A position that fails ends here (nearly every preamble found in noise).
//   stage 3: the survivors ARE the records of the block (the few next to an end of the span get their
//            gate decided first, with guarded loads).  They are ranked by position, their staging slots
//            reserved with one global atomic, then one wavefront per survivor, all 64 lanes cooperating
//            (coalesced sample loads; the sequential parts of the reference become carry chains, see
//            modes_core.h): both attempts, syndrome and repair search -> record + (batch, rank) key.
// Lists live in LDS; appends from stages 2a/2b reserve their slots with one LDS atomic per wavefront.
// finalize_kernel (one workgroup, next in the stream) turns the per-batch record counts into offsets.
// ------------------------------------------------------------------------------------
template <int kDemodWaves, class Lut>
__global__ __launch_bounds__(kDemodWaves * 64) __attribute__((amdgpu_num_sgpr(80))) void demod_kernel(DemodParams P) {
    constexpr int kDemodThreads = kDemodWaves * 64;
    constexpr int kGatePerRound = kDemodThreads / kGateLanes;   // preambles a workgroup tests per round
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[MODES_LUT_ENTRIES];
    __shared__ uint32_t s_pre[kDemodGroup + 1];        // exclusive prefix of the batch's (clamped) run counts
    __shared__ uint32_t s_list[kDemodThreads];         // preambles awaiting the gate pre-test
    __shared__ uint32_t s_long[2 * kDemodThreads];     // (position, sum over the first 56 pairs) of those that decode as long
    __shared__ uint32_t s_surv[3 * kDemodThreads];     // (position, the two delta sums or kUnknown) of those that go to stage 3
    __shared__ uint32_t s_n[2][4];                     // list counters of the current / the next block (see stage 1)
    __shared__ uint32_t s_blk[4];                      // [0] survivors dropped by the edge gate, [1] staging base, [2] records of the block
    __shared__ uint32_t s_esyn[kSynWords];             // the 112 single-bit syndromes, then their hash table (syn_lookup)
    __shared__ unsigned long long s_tot[2];
    __shared__ uint32_t s_flags[2];                    // [1]: WgTotals.flags bits this workgroup raises
#ifdef MODES_TRACE
    const unsigned long long t_start = wall_clock64();
#endif
    stage_lut<kDemodThreads>(s_lut, P.tab.lut);
    for (uint32_t i = threadIdx.x; i < (uint32_t)kSynWords; i += kDemodThreads) s_esyn[i] = P.tab.esyn[i];
    if (threadIdx.x < 2) s_tot[threadIdx.x] = 0;
    if (threadIdx.x < 2) s_flags[threadIdx.x] = 0;
    if (threadIdx.x < 4) s_blk[threadIdx.x] = 0;
    if (threadIdx.x < 8) s_n[threadIdx.x >> 2][threadIdx.x & 3] = 0;
    __syncthreads();
#ifdef MODES_TRACE
    const unsigned long long t_lut = wall_clock64();
#endif

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint8_t *iq = P.iq;
    const Lut lut{s_lut};
    const int64_t lo = P.lo, hi = P.hi;
    const uint64_t below = (1ull << lane) - 1;
    unsigned long long tot_fwd = 0, tot_cand = 0;       // tot_fwd: per lane of wavefront 0; tot_cand: workgroup-uniform
    uint32_t blk = 0;                                   // blocks processed so far: parity selects the counter set
#ifdef MODES_TRACE
    unsigned long long tr_t[4] = {0, 0, 0, 0};          // batch set-up, stage 1, stage 2a, stages 2b + 3
#define TRACE_ADD(k, since) tr_t[k] += wall_clock64() - (since)
#else
#define TRACE_ADD(k, since)
#endif

    for (uint32_t batch = blockIdx.x; batch < P.nbatches; batch += gridDim.x) {
        const uint32_t run0 = batch * kDemodGroup;
        TRACE_T(tb0);
        // every position of the batch is >= gbase: 32-bit byte offsets from there through a raw buffer descriptor
        const int64_t gbase = (int64_t)run0 * P.run_chunks * kChunkSamples - 64;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(iq) + 2 * gbase, 0,
                                                                              0x7fffffff, 0x00020000);
        if (wave == 0) {                                                     // one run per lane
            uint32_t cnt = 0;
            if (run0 + lane < P.nruns) {
                const uint32_t true_count = P.counts[run0 + lane];
                if (true_count > P.slot_cap) atomicOr(&s_flags[1], 1u);        // the scan dropped positions: the call fails
                cnt = min(true_count, P.slot_cap);
                tot_fwd += true_count;
            }
            uint32_t incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            s_pre[lane] = incl - cnt;
            if (lane == 63) s_pre[64] = incl;
        }
        __syncthreads();
        const uint32_t n = s_pre[kDemodGroup];                               // forwarded positions of the batch
        TRACE_ADD(0, tb0);
        uint32_t ncand = 0;
        uint32_t prior = 0;                                                  // records of the batch's earlier blocks
        const uint64_t cand_base = (uint64_t)batch * kDemodGroup * P.slot_cap;
        // position number e of the batch: entry e - s_pre[rr] of run rr, the last run that starts at or before e
        auto slot_of = [&](uint32_t e) -> uint32_t {
            uint32_t rr = 0;
#pragma unroll
            for (int step = kDemodGroup / 2; step >= 1; step >>= 1)
                if (s_pre[rr + step] <= e) rr += step;
            return P.slots[(uint64_t)(run0 + rr) * P.slot_cap + (e - s_pre[rr])];
        };
        uint32_t p_next = (uint32_t)tid < n ? slot_of((uint32_t)tid) : 0u;
        for (uint32_t base = 0; base < n; base += kDemodThreads) {
            // ---------------- stage 1 ----------------
            TRACE_T(ts1);
            const uint32_t e = base + (uint32_t)tid;
            const bool active = e < n;
            const uint32_t p = p_next;
            // the next block's position travels through this block's stages: one dependent gather less per block
            p_next = e + kDemodThreads < n ? slot_of(e + kDemodThreads) : 0u;
            // fast path when every lane's 16-sample window is inside the span (wave-uniform test)
            const bool in1 = !active || samples_inside((int64_t)p, (int64_t)p + 15, lo, hi);
            bool ok;
            if (__all(in1)) ok = active && preamble_at_fast(rsrc, (uint32_t)(2 * ((int64_t)p - gbase)), lut);
            else            ok = active && preamble_at_guarded(iq, lo, hi, lut, p);
            // Preambles whose whole message window (samples p-1 .. p+239) lies inside the span go to the
            // pre-test list; the few next to an end of the span go straight to stage 3 (guarded loads).
            const bool whole = samples_inside((int64_t)p - 1, (int64_t)p + 239, lo, hi);
            const uint64_t okb = __ballot(ok), listb = __ballot(ok && whole), edgeb = okb & ~listb;
            // every wavefront claims its places in the three lists with one LDS atomic each (order within a list is free)
            uint32_t *cn = s_n[blk & 1];                                     // [0] long, [1] stage-3 queue, [2] pre-test list, [3] candidates
            uint32_t lbase = 0, ebase = 0, cbase = 0;
            if (lane == 0) {
                if (listb) lbase = atomicAdd(&cn[2], (uint32_t)__builtin_popcountll(listb));
                if (edgeb) ebase = atomicAdd(&cn[1], (uint32_t)__builtin_popcountll(edgeb));
                if (okb) cbase = atomicAdd(&cn[3], (uint32_t)__builtin_popcountll(okb));
            }
            lbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)lbase);
            ebase = (uint32_t)__builtin_amdgcn_readfirstlane((int)ebase);
            cbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)cbase);
            if (ok) {
                if (whole) s_list[lbase + (uint32_t)__builtin_popcountll(listb & below)] = p;
                else       surv_push(s_surv, ebase + (uint32_t)__builtin_popcountll(edgeb & below), p, kUnknown, kUnknown);
                if (P.cand_slots) P.cand_slots[cand_base + ncand + cbase + (uint32_t)__builtin_popcountll(okb & below)] = p;
            }
            __syncthreads();
            if (tid < 4) s_n[(blk + 1) & 1][tid] = 0;                        // the next block's counters (nobody reads them before two more barriers)
            const uint32_t nlist = cn[2];
            ncand += cn[3];
            blk++;

            // ---------------- stage 2 ----------------
            TRACE_ADD(1, ts1);
            TRACE_T(ts2);
            const int grp = tid / kGateLanes, t = tid % kGateLanes;
            const int head = lane & ~(kGateLanes - 1);                       // first lane of this lane's group
            {   // a: the next round's loads are issued before this one's sums
                u32x4 wn[4] = {};
                uint32_t npc = 0;
                bool nact = false;
                auto issue = [&](uint32_t c0) {
                    const uint32_t c = c0 + (uint32_t)grp;
                    nact = c < nlist;
                    npc = nact ? s_list[c] : 0u;
                    if (nact) half_load(rsrc, (uint32_t)(2 * ((int64_t)npc - gbase)) + 32u, t, wn);
                };
                if (nlist) issue(0);
                for (uint32_t c0 = 0; c0 < nlist; c0 += kGatePerRound) {
                    u32x4 w[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) w[i] = wn[i];
                    const uint32_t pc = npc;
                    const bool gact = nact;
                    if (c0 + kGatePerRound < nlist) issue(c0 + kGatePerRound);
                    uint32_t first;
                    uint32_t d56 = half_eval(w, lut, &first);
#pragma unroll
                    for (int o = kGateLanes / 2; o >= 1; o >>= 1) d56 += (uint32_t)__shfl_xor((int)d56, o, 64);
                    // lane 1 of the group holds pairs 4, 5; pairs 0..3 come from lane 0
                    const uint32_t f0 = (uint32_t)__shfl((int)first, head, 64);
                    const int df = modes_df_first6((f0 & 0xfu) | ((first & 0x3u) << 4), ((f0 >> 4) & 0xfu) | (((first >> 4) & 0x3u) << 4),
                                                   ((f0 >> 8) & 1u) != 0);
                    const bool is_long = modes_len_by_df(df) == 112;
                    const bool mine = gact && t == 1;
                    const bool pass = mine && !is_long && d56 / 28 >= 2550;  // dump1090.c:1717-1723, short message
                    const bool more = mine && is_long;
                    const uint64_t pb = __ballot(pass), mb = __ballot(more);
                    uint32_t sbase = 0, qbase = 0;
                    if (lane == 0) {
                        if (pb) sbase = atomicAdd(&cn[1], (uint32_t)__builtin_popcountll(pb));
                        if (mb) qbase = atomicAdd(&cn[0], (uint32_t)__builtin_popcountll(mb));
                    }
                    sbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)sbase);
                    qbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)qbase);
                    if (pass) surv_push(s_surv, sbase + (uint32_t)__builtin_popcountll(pb & below), pc, d56, kUnknown);
                    if (more) {
                        const uint32_t k = qbase + (uint32_t)__builtin_popcountll(mb & below);
                        s_long[2 * k] = pc;
                        s_long[2 * k + 1] = d56;
                    }
                }
            }
            __syncthreads();
            TRACE_ADD(2, ts2);
            TRACE_T(ts3);
            {   // b: the other 56 pairs of the long ones
                const uint32_t nlong = cn[0];
                for (uint32_t c0 = 0; c0 < nlong; c0 += kGatePerRound) {
                    const uint32_t c = c0 + (uint32_t)grp;
                    const bool gact = c < nlong;
                    const uint32_t pc = gact ? s_long[2 * c] : 0u;
                    const uint32_t d56 = gact ? s_long[2 * c + 1] : 0u;
                    uint32_t d112 = 0;
                    if (gact) {
                        u32x4 w[4];
                        half_load(rsrc, (uint32_t)(2 * ((int64_t)pc - gbase)) + 32u + 224u, t, w);
                        d112 = half_eval(w, lut, nullptr);
                    }
#pragma unroll
                    for (int o = kGateLanes / 2; o >= 1; o >>= 1) d112 += (uint32_t)__shfl_xor((int)d112, o, 64);
                    d112 += d56;
                    const bool pass = gact && t == 1 && d112 / 56 >= 2550;   // dump1090.c:1717-1723, long message
                    const uint64_t pb = __ballot(pass);
                    uint32_t sbase = 0;
                    if (lane == 0 && pb) sbase = atomicAdd(&cn[1], (uint32_t)__builtin_popcountll(pb));
                    sbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)sbase);
                    if (pass) surv_push(s_surv, sbase + (uint32_t)__builtin_popcountll(pb & below), pc, d56, d112);
                }
            }
            __syncthreads();
            // ---------------- stage 3 ----------------
            const uint32_t nsurv = cn[1];
            if (nsurv) {                                                     // workgroup-uniform
                // a) survivors that skipped the pre-test: first gate now (guarded loads); a failure drops the entry
                for (uint32_t k = (uint32_t)wave; k < nsurv; k += kDemodWaves) {
                    if (s_surv[3 * k + 1] != kUnknown) continue;             // wave-uniform
                    const Front f = demod_front<true>(P, lut, lane, (int64_t)s_surv[3 * k], kUnknown, kUnknown);
                    if (lane == 0) {
                        if (f.gate0) {
                            s_surv[3 * k + 1] = (uint32_t)f.sum56;
                            s_surv[3 * k + 2] = f.have112 ? (uint32_t)f.sum112 : kUnknown;
                        } else {
                            s_surv[3 * k] = kNoPos;
                            atomicAdd(&s_blk[0], 1u);
                        }
                    }
                }
                __syncthreads();
                // b) rank by position (the block's records in stream order; the ranks go where the long list was: stage 2b is
                //    done with it), one global atomic for their staging slots
                if ((uint32_t)tid < nsurv) {
                    const uint32_t mine = s_surv[3 * tid];
                    uint32_t r = 0;
                    for (uint32_t k = 0; k < nsurv; k++) r += s_surv[3 * k] < mine ? 1u : 0u;   // kNoPos is never below a live one
                    s_long[tid] = r;
                }
                if (tid == 0) {
                    const uint32_t live = nsurv - s_blk[0];
                    s_blk[0] = 0;
                    s_blk[2] = live;
                    s_blk[1] = live ? atomicAdd(&P.hdr->n_records, live) : 0u;
                }
                __syncthreads();
                // c) one wavefront per record
                const uint32_t sbase = s_blk[1], live = s_blk[2];
                for (uint32_t k = (uint32_t)wave; k < nsurv; k += kDemodWaves) {
                    const uint32_t pk = s_surv[3 * k];
                    if (pk == kNoPos) continue;
                    const int64_t pcs = (int64_t)pk;
                    const uint32_t k56 = s_surv[3 * k + 1], k112 = s_surv[3 * k + 2], rank = s_long[k];
                    const uint32_t slot = sbase + rank;
                    const uint64_t key = ((uint64_t)batch << 32) | (uint64_t)(prior + rank);
                    bool done;
                    if (samples_inside(pcs - 1, pcs + 239, lo, hi)) done = demod_full<false>(P, lut, s_esyn, lane, pcs, k56, k112, slot, key);
                    else                                            done = demod_full<true>(P, lut, s_esyn, lane, pcs, k56, k112, slot, key);
                    if (!done && lane == 0) {                                // cannot happen: the pre-test IS the first gate
                        if (slot < P.max_records) P.keys[slot] = kNoKey;
                        atomicOr(&s_flags[1], 2u);
                    }
                }
                prior += live;
                __syncthreads();
            }
            TRACE_ADD(3, ts3);
        }
        tot_cand += ncand;
        if (tid == 0) {
            P.cand_counts[batch] = ncand;
            P.batch_count[batch] = prior;
        }
        __syncthreads();                                                     // s_pre is rewritten by the next batch
    }
#ifdef MODES_TRACE
    if (lane == 0) {
        const uint32_t w = blockIdx.x * kDemodWaves + (uint32_t)wave;
        if (w < 8192) {
            unsigned long long *tr = &g_trace[8 * w];
            tr[0] = t_start; tr[1] = t_lut; tr[2] = wall_clock64(); tr[3] = tot_cand;
            tr[4] = tr_t[0]; tr[5] = tr_t[1]; tr[6] = tr_t[2]; tr[7] = tr_t[3];
        }
    }
#endif
    // totals (tot_fwd: lanes of wavefront 0; tot_cand: the same number in every thread)
    if (wave == 0) {
        tot_fwd = (unsigned long long)wave_sum((int)tot_fwd);                // < 2^31 per workgroup and call (slot lists are u32-indexed)
        if (lane == 0) { s_tot[0] = tot_fwd; s_tot[1] = tot_cand; }
    }
    __syncthreads();
    if (tid == 0) P.totals[blockIdx.x] = WgTotals{s_tot[0], s_tot[1], s_flags[1], 0};
}

// ------------------------------------------------------------------------------------
// The demodulation in TWO kernels (demod_variant 2; demod_variant 0 takes it for calls that follow a record-rich call):
//
//   select_kernel   stages 1 and 2 for EVERY forwarded position - the exact preamble predicate and the noise-gate
//                   pre-test - and nothing else: on noise that is all there is to do (~270,000 preambles per GiB, none of
//                   which survives).  16-wave workgroups held to 64 VGPRs: two of them (32 wavefronts) per CU share the
//                   two 64 KiB tables, twice the wavefronts per CU of demod_kernel, whose register count the record
//                   stage sets - the stage is bound by how many wavefronts wait on its gathers at once.  The survivors
//                   of a batch leave as a position list in ascending order.
//   record_kernel   stage 3 for the survivors only: one wavefront per record, written straight to its FINAL place in the
//                   ordered list (every workgroup derives the batch offsets from the batch counts itself: no staging
//                   list, no keys, no order kernel).  A workgroup whose batches have no survivor retires before it
//                   stages a table: on noise the kernel is a few microseconds.
//
// demod_kernel (one kernel, stage 3 inline, staging + keys + order) is what record-free input runs on (demod_variant 3: always).
// ------------------------------------------------------------------------------------
constexpr double kSplitAbove = 4096.0;                     // records per GiB beyond which demod_variant 0 takes the two-kernel path
constexpr int kSelWaves = 16;
constexpr int kSelThreads = kSelWaves * 64;
constexpr int kSelLanes = 8;                               // lanes per preamble in the gate pre-test
constexpr int kSelPerRound = kSelThreads / kSelLanes;      // 128 preambles per round
#ifndef SEL_AHEAD
#define SEL_AHEAD 1
#endif
constexpr int kSelAhead = SEL_AHEAD;                       // rounds whose loads are in flight together.  1: the kernel fits its 64 VGPRs (two 16-wave
                                                           // workgroups per CU) without a private segment; 2 spills 24 bytes per lane for no measured gain
                                                           // (1, 2, 3, 5 or 9 rounds in flight: the stage waits for HBM lines either way - DESIGN.md 3.2)
// s_flag values (per preamble of the block); LONG carries the first half's delta sum in its low 22 bits (<= 56 * 65167)
constexpr uint32_t kSelFail = 0u, kSelPass = 1u << 30, kSelEdge = 2u << 30, kSelLong = 3u << 30, kSelKind = 3u << 30;

struct SelectParams {
    const uint8_t *iq;
    int64_t lo, hi;
    uint32_t nruns, run_chunks, slot_cap;
    const uint32_t *slots;      // [nruns][slot_cap] forwarded positions, ascending per run (scan kernel)
    const uint32_t *counts;     // [nruns]
    const uint16_t *lut;
    uint32_t *cand_slots;       // [nbatches][kDemodGroup * slot_cap] preamble positions (keep_candidates) or nullptr
    uint32_t *cand_counts;      // [nbatches]
    uint32_t *surv;             // [nbatches][kDemodGroup * slot_cap] survivor positions of each batch, ascending
    uint32_t *batch_count;      // [nbatches] survivors = records of each batch
    uint32_t nbatches;
    WgTotals *totals;           // [gridDim.x]
};

// 56 bit pairs = 224 bytes from byte offset voff (2-byte aligned: a sample is two bytes) by a group of 8 lanes: lane t takes
// pairs 7t .. 7t + 6.  The loads are DWORD-ALIGNED - eight dwords from (voff + 28 t) rounded down to a dword, as two x4 loads -
// and an odd sample offset is taken out afterwards with one v_alignbit per pair (2-byte-misaligned loads keep the texture
// addresser busy ~47 cycles per wave instruction: TA_BUSY 63 % of the kernel, profiles/r05; aligned, the stage is ~1 us
// shorter - it is bound by the lines it pulls from HBM either way).
__device__ __forceinline__ void sel_load(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, int t, uint32_t (&d)[8]) {
    const uint32_t o = (voff + 28u * (uint32_t)t) & ~3u;
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, kDemodAux), b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + 16u, 0, kDemodAux);
    d[0] = a[0]; d[1] = a[1]; d[2] = a[2]; d[3] = a[3]; d[4] = b[0]; d[5] = b[1]; d[6] = b[2]; d[7] = b[3];
}
// pair k of the lane: the dword at 2-byte phase `sh16` (0 or 16 bits) inside d[k], d[k + 1]
__device__ __forceinline__ uint32_t sel_dword(const uint32_t (&d)[8], int k, uint32_t sh16) {
    return __builtin_amdgcn_alignbit(d[k + 1], d[k], sh16);
}
// Table address of both samples of a pair (dword I0 Q0 I1 Q1) by the scan kernel's byte dot product (power16_scan):
// x = w ^ 0x7f7f7f7f holds the signed bytes 127 - b; v_dot4_i32_i8 of x with x masked to one sample, clamped against the
// accumulator 0x7fff8000, is 0x7fff0000 | 0x8000 | min(s, 32767); shifted left by one that is 0xffff0000 + 2 s (mod 2^32), so
// adding (table base + 0x10000) gives the LDS byte address of entry s - five instructions per pair before the two
// shift-adds, three of them full rate (the packed multiply / multiply-add form: six, one full rate, plus wait states).
__device__ __forceinline__ void sel_pair(uint32_t w, uint32_t lut_adj, uint32_t &a, uint32_t &b) {
    const uint32_t x = w ^ 0x7f7f7f7fu;
    const uint32_t pa = (uint32_t)__builtin_amdgcn_sdot4((int)x, (int)(x & 0x0000ffffu), 0x7fff8000, true);
    const uint32_t pb = (uint32_t)__builtin_amdgcn_sdot4((int)x, (int)(x & 0xffff0000u), 0x7fff8000, true);
    // (an LDS pointer is its 32-bit byte address)
    a = *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>((uintptr_t)((pa << 1) + lut_adj));
    b = *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>((uintptr_t)((pb << 1) + lut_adj));
}
// sum over the 8 lanes of a group, in every lane of it: quad butterflies, then the mirrored half row (lane i <-> 7 - i
// swaps the two quads, whose lanes all hold their quad's sum by then) - three DPP adds, no LDS round trip
__device__ __forceinline__ uint32_t sel_reduce8(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);    // row_half_mirror
    return v;
}
// This lane's part of the delta sum of the group's 56 pairs (its seven).  *first (FLAGS: meaningful in the group's lane 0, which
// holds pairs 0 .. 6): bit k = |lo - hi| < 256 of pair k, bit 8 + k = lo > hi, k = 0 .. 5; bit 16 = (lo == hi) of pair 0.
template <bool FLAGS>
__device__ __forceinline__ uint32_t sel_sum(const uint32_t (&d)[8], uint32_t sh16, uint32_t lut_adj, uint32_t *first) {
    uint32_t acc = 0, f = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        uint32_t a, b;
        sel_pair(sel_dword(d, k, sh16), lut_adj, a, b);
        if (FLAGS && k < 6) {
            const uint32_t dd = __builtin_amdgcn_sad_u16(a, b, 0u);
            f |= (dd < 256u ? 1u : 0u) << k | (a > b ? 1u : 0u) << (8 + k);
            if (k == 0) f |= (a == b ? 1u : 0u) << 16;
            acc += dd;
        } else {
            acc = __builtin_amdgcn_sad_u16(a, b, acc);
        }
    }
    if (FLAGS) *first = f;
    return acc;
}

__device__ __forceinline__ int sel_tid() {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
__global__ __launch_bounds__(kSelThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void select_kernel(SelectParams P) {
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[MODES_LUT_ENTRIES];
    __shared__ uint32_t s_pre[kDemodGroup + 1];        // exclusive prefix of the batch's (clamped) run counts
    __shared__ uint32_t s_list[kSelThreads];           // preamble positions of the block, ASCENDING
    __shared__ uint32_t s_flag[kSelThreads];           // their state (kSel*)
    __shared__ uint16_t s_work[kSelThreads];           // indices into s_list: the long ones, then (from the top) the edge ones
    __shared__ uint32_t s_wcnt[kSelWaves + 1];         // per-wavefront counts -> exclusive prefix
    __shared__ uint32_t s_n[2];                        // [0] long ones, [1] edge ones of the block
    __shared__ uint32_t s_flags;                       // WgTotals.flags bits
    __shared__ uint32_t s_fwd;                         // forwarded positions of the workgroup's batches (a call has < 2^32 positions)
#ifdef MODES_TRACE
    const unsigned long long t_start = wall_clock64();
    unsigned long long tr_t[4] = {0, 0, 0, 0};          // batch set-up, stage 1 (+ compaction), stage 2a, stage 2b + edge + write-out
#endif
    stage_lut<kSelThreads>(s_lut, P.lut);              // (staged UNDER the first batch's position loads instead: 7 us longer - the
                                                       //  table's loads queue behind them; profiles/r05)
    const int tid = (int)threadIdx.x;
    [[maybe_unused]] const int lane = tid & 63;             // (the trace build reads it)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) { s_flags = 0; s_fwd = 0; s_n[0] = 0; s_n[1] = 0; }
    __syncthreads();
#ifdef MODES_TRACE
    const unsigned long long t_lut = wall_clock64();
#endif
    const uint8_t *iq = P.iq;
    const int64_t lo = P.lo, hi = P.hi;
    const uint32_t lut_adj = (uint32_t)reinterpret_cast<uintptr_t>(s_lut) + 0x10000u;     // sel_pair: LDS byte address of the table, adjusted
    uint32_t tot_cand = 0;                             // workgroup-uniform

    for (uint32_t batch = blockIdx.x; batch < P.nbatches; batch += gridDim.x) {
        // The thread's number is taken afresh per batch and per block of positions (sel_tid): everything derived from it - a
        // dozen LDS addresses - is then recomputed where it is used instead of living in a register for the whole kernel
        // (hoisted out of the loops they end up in the private segment: this kernel has exactly its 64 registers).
        const int tid = sel_tid(), lane = tid & 63;
        const uint32_t run0 = batch * kDemodGroup;
        TRACE_T(tb0);
        const int64_t gbase = (int64_t)run0 * P.run_chunks * kChunkSamples - 64;     // every position of the batch is >= gbase
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(iq) + 2 * gbase, 0, 0x7fffffff, 0x00020000);
        if (wave == 0) {                                                     // one run per lane
            uint32_t cnt = 0;
            if (run0 + lane < P.nruns) {
                const uint32_t true_count = P.counts[run0 + lane];
                if (true_count > P.slot_cap) atomicOr(&s_flags, 1u);           // the scan dropped positions: the call fails
                cnt = min(true_count, P.slot_cap);
                atomicAdd(&s_fwd, true_count);                                 // (one LDS instruction per batch: a running per-lane sum would
                                                                               //  be one more register alive across everything below)
            }
            uint32_t incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            s_pre[lane] = incl - cnt;
            if (lane == 63) s_pre[64] = incl;
        }
        __syncthreads();
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pre[kDemodGroup]);     // workgroup-uniform values live in SGPRs:
        const uint64_t list_base = (uint64_t)batch * kDemodGroup * P.slot_cap;   // of the batch's candidate and survivor lists
        uint32_t ncand = 0, prior = 0;
        auto slot_of = [&](uint32_t e) -> uint32_t {
            uint32_t rr = 0;
#pragma unroll
            for (int step = kDemodGroup / 2; step >= 1; step >>= 1)
                if (s_pre[rr + step] <= e) rr += step;
            return P.slots[(uint64_t)(run0 + rr) * P.slot_cap + (e - s_pre[rr])];
        };
        // exclusive prefix of one count per wavefront over the workgroup: -> (this wavefront's base, total); two barriers
        // (lane w takes wavefront w's count and the prefix runs over the lanes: a loop over w with "w < wave" is sixteen wave-uniform
        //  comparisons that the compiler hoists out of the batch loop - 32 scalar registers this kernel then spills into vector lanes)
        auto wave_prefix = [&](uint32_t mine, uint32_t *total) -> uint32_t {
            if (lane == 0) s_wcnt[wave] = mine;
            __syncthreads();
            const uint32_t c = lane < kSelWaves ? s_wcnt[lane] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int off = 1; off < kSelWaves; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            __syncthreads();
            *total = (uint32_t)__builtin_amdgcn_readlane((int)incl, kSelWaves - 1);
            return (uint32_t)__builtin_amdgcn_readlane((int)(incl - c), wave);
        };
        uint32_t p_next = (uint32_t)tid < n ? slot_of((uint32_t)tid) : 0u;
        TRACE_ADD(0, tb0);
        for (uint32_t base = 0; base < n; base += kSelThreads) {
            const int tid = sel_tid(), lane = tid & 63;
            TRACE_T(ts1);
            // ---------------- stage 1: the exact predicate, survivors compacted IN ORDER ----------------
            const uint32_t e = base + (uint32_t)tid;
            const bool active = e < n;
            const uint32_t p = p_next;
            p_next = e + kSelThreads < n ? slot_of(e + kSelThreads) : 0u;
            const bool in1 = !active || samples_inside((int64_t)p, (int64_t)p + 15, lo, hi);
            bool ok;
            const LutFull lut{s_lut};
            if (__all(in1)) ok = active && preamble_at_fast(rsrc, (uint32_t)(2 * ((int64_t)p - gbase)), lut);
            else            ok = active && preamble_at_guarded(iq, lo, hi, lut, p);
            // (two samples of slack behind the message window: the pre-test's dword-aligned loads read up to four bytes past it)
            const bool whole = samples_inside((int64_t)p - 1, (int64_t)p + 241, lo, hi);
            const uint64_t okb = __ballot(ok);
            uint32_t nlist;
            const uint32_t wbase = wave_prefix((uint32_t)__builtin_popcountll(okb), &nlist);
            if (ok) {
                const uint32_t c = wbase + lanes_below(okb);
                s_list[c] = p;
                s_flag[c] = whole ? kSelFail : kSelEdge;
                if (P.cand_slots) P.cand_slots[list_base + ncand + c] = p;
            }
            // the few next to an end of the span (their message window is not all inside): a list of their own
            const uint64_t edgeb = __ballot(ok && !whole);
            if (edgeb) {                                                      // wave-uniform, rare
                uint32_t eb = 0;
                if (lane == 0) eb = atomicAdd(&s_n[1], (uint32_t)__builtin_popcountll(edgeb));
                eb = (uint32_t)__builtin_amdgcn_readfirstlane((int)eb);
                if (ok && !whole) s_work[kSelThreads - 1 - (eb + lanes_below(edgeb))] =
                    (uint16_t)(wbase + lanes_below(okb));
            }
            __syncthreads();
            ncand += nlist;
            TRACE_ADD(1, ts1);
            TRACE_T(ts2);
            // ---------------- stage 2a: the first 56 pairs and the DF of the first slicing pass, 8 lanes per preamble ----------------
            const int grp = tid / kSelLanes, t = tid % kSelLanes;
            const int head = lane & ~(kSelLanes - 1);
            // The loads of kSelAhead rounds are issued together.  (It hardly matters - 1, 2, 3, 5 or 9 rounds in flight, 4 or 8 lanes
            // per preamble, 8, 12 or 16 wavefronts, half the instructions: the stage takes 11-17 us for a 1 GiB call and 2 us
            // when only 64 workgroups run.  What it waits for is HBM: ~2.75 lines of 128 B per preamble, 95 MB per GiB of noise
            // at the ~6.5 TB/s the chip streams: DESIGN.md 3.2, profiles/r05.)
            for (uint32_t s0 = 0; s0 < nlist; s0 += kSelAhead * kSelPerRound) {
                uint32_t w[kSelAhead][8];
                uint32_t sh[kSelAhead];
                bool act[kSelAhead];
#pragma unroll
                for (int r = 0; r < kSelAhead; r++) {
                    const uint32_t c = s0 + (uint32_t)r * kSelPerRound + (uint32_t)grp;
                    act[r] = c < nlist && s_flag[c < nlist ? c : 0] != kSelEdge;
                    const uint32_t pc = act[r] ? s_list[c] : 0u;
                    const uint32_t voff = (uint32_t)(2 * ((int64_t)pc - gbase)) + 32u;
                    sh[r] = ((voff + 28u * (uint32_t)t) & 2u) << 3;          // 16 when the lane's first pair starts in the middle of a dword
#pragma unroll
                    for (int k = 0; k < 8; k++) w[r][k] = 0;
                    if (act[r]) sel_load(rsrc, voff, t, w[r]);
                }
#pragma unroll
                for (int r = 0; r < kSelAhead; r++) {
                    if (s0 + (uint32_t)r * kSelPerRound >= nlist) break;      // workgroup-uniform
                    const uint32_t c = s0 + (uint32_t)r * kSelPerRound + (uint32_t)grp;
                    const bool gact = act[r];
                    uint32_t first = 0, d56 = 0;
                    if (gact) d56 = sel_sum<true>(w[r], sh[r], lut_adj, &first);
                    d56 = sel_reduce8(d56);
                    const uint32_t f0 = (uint32_t)__shfl((int)first, head, 64);   // pairs 0 .. 5 live in the group's first lane
                    const bool is_long = modes_len_by_df(modes_df_first6(f0 & 0x3fu, (f0 >> 8) & 0x3fu, ((f0 >> 16) & 1u) != 0)) == 112;
                    const bool mine = gact && t == 0;
                    if (mine && !is_long) s_flag[c] = d56 / 28 >= 2550 ? kSelPass : kSelFail;     // dump1090.c:1717-1723, short message
                    const bool more = mine && is_long;
                    const uint64_t mb = __ballot(more);
                    if (mb) {                                                 // wave-uniform
                        uint32_t qb = 0;
                        if (lane == 0) qb = atomicAdd(&s_n[0], (uint32_t)__builtin_popcountll(mb));
                        qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
                        if (more) {
                            s_work[qb + lanes_below(mb)] = (uint16_t)c;
                            s_flag[c] = kSelLong | d56;
                        }
                    }
                }
            }
            __syncthreads();
            TRACE_ADD(2, ts2);
            TRACE_T(ts3);
            // ---------------- stage 2b: the other 56 pairs of the long ones ----------------
            const uint32_t nlong = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_n[0]), nedge = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_n[1]);
            for (uint32_t c0 = 0; c0 < nlong; c0 += kSelPerRound) {
                const uint32_t k = c0 + (uint32_t)grp;
                const bool gact = k < nlong;
                const uint32_t c = gact ? s_work[k] : 0u;
                const uint32_t pc = s_list[c];
                uint32_t d = 0;
                if (gact) {
                    uint32_t w[8];
                    const uint32_t voff = (uint32_t)(2 * ((int64_t)pc - gbase)) + 32u + 224u;
                    sel_load(rsrc, voff, t, w);
                    d = sel_sum<false>(w, ((voff + 28u * (uint32_t)t) & 2u) << 3, lut_adj, nullptr);
                }
                d = sel_reduce8(d);
                if (gact && t == 0) s_flag[c] = (d + (s_flag[c] & ~kSelKind)) / 56 >= 2550 ? kSelPass : kSelFail;   // :1717-1723, long message
            }
            // ---------------- the edge ones: first slicing pass and gate with guarded loads, a wavefront each ----------------
            for (uint32_t k = (uint32_t)wave; k < nedge; k += kSelWaves) {
                const uint32_t c = s_work[kSelThreads - 1 - k];
                const int64_t pc = (int64_t)s_list[c];
                const bool two = lane < 48;
                const int lo1 = mag_of(lut, load_sample<true>(iq, pc + 16 + 2 * lane, lo, hi));
                const int hi1 = mag_of(lut, load_sample<true>(iq, pc + 17 + 2 * lane, lo, hi));
                const int lo2 = two ? mag_of(lut, load_sample<true>(iq, pc + 144 + 2 * lane, lo, hi)) : 0;
                const int hi2 = two ? mag_of(lut, load_sample<true>(iq, pc + 145 + 2 * lane, lo, hi)) : 0;
                int sum56, sum112;
                delta_sums(lane, lo1, hi1, lo2, hi2, &sum56, &sum112);
                const int d1 = lo1 > hi1 ? lo1 - hi1 : hi1 - lo1;
                const uint32_t weak = (uint32_t)__ballot(d1 < 256) & 0x3fu, gt = (uint32_t)__ballot(lo1 > hi1) & 0x3fu;
                const bool eq0 = (__ballot(lo1 == hi1) & 1ull) != 0;
                const bool is_long = modes_len_by_df(modes_df_first6(weak, gt, eq0)) == 112;
                if (lane == 0) s_flag[c] = (is_long ? sum112 / 56 >= 2550 : sum56 / 28 >= 2550) ? kSelPass : kSelFail;
            }
            __syncthreads();
            // ---------------- the survivors of the block, in order, behind those of the batch's earlier blocks ----------------
            {
                const bool pass = (uint32_t)tid < nlist && s_flag[tid] == kSelPass;
                const uint64_t pb = __ballot(pass);
                uint32_t nsurv;
                const uint32_t sb = wave_prefix((uint32_t)__builtin_popcountll(pb), &nsurv);
                if (pass) P.surv[list_base + prior + sb + lanes_below(pb)] = s_list[tid];
                prior += nsurv;
                if (tid == 0) { s_n[0] = 0; s_n[1] = 0; }                      // (wave_prefix ended with a barrier: nobody reads them now)
            }
            __syncthreads();
            TRACE_ADD(3, ts3);
        }
        tot_cand += ncand;
        if (tid == 0) {
            P.cand_counts[batch] = ncand;
            P.batch_count[batch] = prior;
        }
        __syncthreads();                                                     // s_pre is rewritten by the next batch
    }
#ifdef MODES_TRACE
    if (lane == 0) {
        const uint32_t w = blockIdx.x * kSelWaves + (uint32_t)wave;
        if (w < 8192) {
            unsigned long long *tr = &g_trace[8 * w];
            tr[0] = t_start; tr[1] = t_lut; tr[2] = wall_clock64(); tr[3] = tot_cand;
            tr[4] = tr_t[0]; tr[5] = tr_t[1]; tr[6] = tr_t[2]; tr[7] = tr_t[3];
        }
    }
#endif
    __syncthreads();
    if (tid == 0) P.totals[blockIdx.x] = WgTotals{s_fwd, tot_cand, s_flags, 0};
}

struct RecordParams {
    DemodParams d;                 // iq, lo, hi, g0, tab, maxfix, max_records, staging (= the ORDERED device list here)
    const uint32_t *surv;          // [nbatches][kDemodGroup * slot_cap]
    const uint32_t *batch_count;   // [nbatches]
    modes_record *host_out;        // pinned host copy (device view) or nullptr: the first direct_cap records go there too
    uint32_t direct_cap;
    uint32_t *wg_flags;            // [gridDim.x] bit 1: the full demodulation disagreed with the pre-test (cannot happen)
};
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_sgpr(80))) void record_kernel(RecordParams R) {
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[MODES_LUT_ENTRIES];
    __shared__ uint32_t s_esyn[kSynWords];
    __shared__ uint32_t s_red[8];
    __shared__ uint32_t s_bad;
    const DemodParams &P = R.d;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MODES_TRACE
    const unsigned long long t_start = wall_clock64();
    unsigned long long tr_counts = 0, tr_table = 0, tr_records = 0, tr_n = 0, tr_nb = 0;     // (second half of g_trace: select_kernel has the first)
#endif
    // sum over the workgroup of one value per thread (batch_count[from + tid], from <= .. < to: to - from <= 512, the grid never exceeds
    // 512 workgroups)
    auto count_reduce = [&](uint32_t v) -> uint32_t {
        v = (uint32_t)wave_sum((int)v);
        __syncthreads();                                                     // s_red of the previous call has been read
        if (lane == 0) s_red[wave] = v;
        __syncthreads();
        uint32_t all = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) all += s_red[w];
        return all;
    };
    // the samples of the record at buffer sample pc (wave-uniform): plain loads when its window is inside the span
    auto load_front = [&](int64_t pc) -> FrontRaw {
        return samples_inside(pc - 1, pc + 239, P.lo, P.hi) ? front_load<false>(P, lane, pc) : front_load<true>(P, lane, pc);
    };
    if (tid == 0) s_bad = 0;
    bool staged = false;
    uint32_t off = 0, prev = 0;                                              // records in front of batch `prev`
    for (uint32_t batch = blockIdx.x; batch < P.nbatches; batch += gridDim.x) {
        // Everything a wavefront needs to start is requested TOGETHER: the batch's count, the counts in front of it, and the positions
        // of the wavefront's first two records (their slots exist whatever the count is) - one round trip to memory, then the first
        // record's samples together with the table: two in front of the first demodulation instead of five (round 5).
        const uint64_t list_base = (uint64_t)batch * kDemodGroup * P.slot_cap;
        const uint32_t nb = R.batch_count[batch];
        uint32_t r = (uint32_t)wave;
        uint32_t p0 = R.surv[list_base + r], p1 = R.surv[list_base + r + 8];
        TRACE_T(tc0);
        off += count_reduce(prev + (uint32_t)tid < batch ? R.batch_count[prev + (uint32_t)tid] : 0u);
        prev = batch;
#ifdef MODES_TRACE
        tr_counts += wall_clock64() - tc0; tr_nb += nb;
#endif
        if (nb == 0) continue;                                               // workgroup-uniform: noise ends here, before any table is staged
        FrontRaw raw0 = FrontRaw{0x7f7fu, 0x7f7fu, 0x7f7fu, 0x7f7fu, 0x7f7fu};
        if (r < nb) raw0 = load_front((int64_t)p0);
        TRACE_T(tt0);
        if (!staged) {
            stage_lut<512>(s_lut, P.tab.lut);
            for (int i = tid; i < kSynWords; i += 512) s_esyn[i] = P.tab.esyn[i];
            __syncthreads();
            staged = true;
        }
#ifdef MODES_TRACE
        tr_table += wall_clock64() - tt0;
        const unsigned long long tr0 = wall_clock64();
#endif
        const LutFull lut{s_lut};
        // one wavefront per record, software-pipelined: while record r is demodulated the samples of the wavefront's next record
        // (r + 8) and the position of the one after (r + 16) are already on their way
        for (; r < nb; r += 8) {
            const uint32_t p2 = r + 16 < nb ? R.surv[list_base + r + 16] : 0u;
            FrontRaw raw1 = raw0;
            if (r + 8 < nb) raw1 = load_front((int64_t)p1);
            const uint32_t slot = off + r;                                   // its place in the ordered list
            // demod_rest writes P.staging[slot] (here: the ordered list itself); keys are not used on this path.
            // the first direct_cap records also go to the host's pinned copy (a short list needs no copy operation then)
            modes_record *host_rec = (R.host_out && slot < R.direct_cap) ? &R.host_out[slot] : nullptr;
            const bool done = demod_rest<false>(P, lut, s_esyn, lane, (int64_t)p0, raw0, kUnknown, kUnknown, slot, 0, host_rec);
            if (!done && lane == 0) atomicOr(&s_bad, 2u);
            p0 = p1; p1 = p2; raw0 = raw1;
#ifdef MODES_TRACE
            tr_n++;
#endif
        }
#ifdef MODES_TRACE
        tr_records += wall_clock64() - tr0;
#endif
    }
#ifdef MODES_TRACE
    if (lane == 0) {
        const uint32_t w = blockIdx.x * 8 + (uint32_t)wave;
        if (w < 4096) {
            unsigned long long *tr = &g_trace[8 * (4096 + w)];
            tr[0] = t_start; tr[1] = tr_counts; tr[2] = wall_clock64(); tr[3] = tr_n; tr[4] = tr_table; tr[5] = tr_records; tr[6] = tr_nb; tr[7] = 0;
        }
    }
#endif
    __syncthreads();
    if (tid == 0) R.wg_flags[blockIdx.x] = s_bad;
}

// finalize of the two-kernel path: totals, the number of records, the verdict for the host.  Nothing to put in order.
struct Finalize2Params {
    ResultHeader *hdr;
    const WgTotals *totals;
    uint32_t ntotals;
    const uint32_t *wg_flags;
    uint32_t nwg;
    const uint32_t *batch_count;
    uint32_t nbatches;
    unsigned long long *d_count;
    HostHeader *host_hdr;
    uint32_t seq;
};
__global__ __launch_bounds__(512) void finalize2_kernel(Finalize2Params P) {
    __shared__ unsigned long long s_sum[3][8];
    __shared__ uint32_t s_fl[8];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long f = 0, c = 0, n = 0;
    uint32_t fl = 0;
    for (uint32_t i = (uint32_t)tid; i < P.ntotals; i += 512) { f += P.totals[i].n_forwarded; c += P.totals[i].n_preambles; fl |= P.totals[i].flags; }
    for (uint32_t i = (uint32_t)tid; i < P.nwg; i += 512) fl |= P.wg_flags[i];
    for (uint32_t i = (uint32_t)tid; i < P.nbatches; i += 512) n += P.batch_count[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        f += (unsigned long long)__shfl_xor((long long)f, off, 64);
        c += (unsigned long long)__shfl_xor((long long)c, off, 64);
        n += (unsigned long long)__shfl_xor((long long)n, off, 64);
        fl |= (uint32_t)__shfl_xor((int)fl, off, 64);
    }
    if (lane == 0) { s_sum[0][wave] = f; s_sum[1][wave] = c; s_sum[2][wave] = n; s_fl[wave] = fl; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long nf = 0, nc = 0, nr = 0;
        uint32_t flags = 0;
        for (int w = 0; w < 8; w++) { nf += s_sum[0][w]; nc += s_sum[1][w]; nr += s_sum[2][w]; flags |= s_fl[w]; }
        if (P.d_count) *P.d_count = nr;
        P.hdr->n_records = (uint32_t)nr;
        P.hdr->ordered = 1u;
        HostHeader *h = P.host_hdr;
        h->n_forwarded = nf;
        h->n_preambles = nc;
        h->n_records = (uint32_t)nr;
        h->flags = flags;
        h->ordered = 1u;
        __hip_atomic_store(&h->seq, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------
// finalize_kernel - ONE workgroup behind the demod kernel: sums the workgroups' totals, turns the per-batch record
// counts into offsets (exclusive prefix), puts a SHORT list (<= inline_cap records) in stream order right away - on the
// device and in the host's pinned copy - and publishes the call's verdict to the host (HostHeader; the sequence number
// last, behind a system-scope release).  Long lists are left to order_kernel.
// ------------------------------------------------------------------------------------
struct FinalizeParams {
    ResultHeader *hdr;
    const WgTotals *totals;
    uint32_t ntotals;
    const uint32_t *batch_count;
    uint32_t *batch_off;
    uint32_t nbatches;
    const modes_record *staging;
    const uint64_t *keys;
    modes_record *out;             // the ordered list on the device
    modes_record *host_out;        // its pinned host copy (device view), or nullptr
    unsigned long long *d_count;   // the caller's device count word, or nullptr
    HostHeader *host_hdr;          // device view of the pinned header
    uint32_t max_records;
    uint32_t inline_cap;
    uint32_t seq;
};
__global__ __launch_bounds__(512) void finalize_kernel(FinalizeParams P) {
    __shared__ unsigned long long s_sum[2][8];
    __shared__ uint32_t s_w[8], s_fl[8];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // totals
    unsigned long long f = 0, c = 0;
    uint32_t fl = 0;
    for (uint32_t i = (uint32_t)tid; i < P.ntotals; i += 512) { f += P.totals[i].n_forwarded; c += P.totals[i].n_preambles; fl |= P.totals[i].flags; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        f += (unsigned long long)__shfl_xor((long long)f, off, 64);
        c += (unsigned long long)__shfl_xor((long long)c, off, 64);
        fl |= (uint32_t)__shfl_xor((int)fl, off, 64);
    }
    if (lane == 0) { s_sum[0][wave] = f; s_sum[1][wave] = c; s_fl[wave] = fl; }
    // exclusive prefix of the batch counts, 512 batches per pass
    uint32_t running = 0;
    for (uint32_t b0 = 0; b0 < P.nbatches; b0 += 512) {
        const uint32_t b = b0 + (uint32_t)tid;
        const uint32_t cnt = b < P.nbatches ? P.batch_count[b] : 0u;
        uint32_t incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        __syncthreads();                                                     // s_w of the previous pass has been read
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint32_t before = running;
        for (int w = 0; w < 8; w++) {
            const uint32_t t = s_w[w];
            if (w < wave) before += t;
            running += t;
        }
        if (b < P.nbatches) P.batch_off[b] = before + incl - cnt;
    }
    __syncthreads();                                                         // batch_off is written (this workgroup reads it back)
    const uint32_t total = running;                                          // records of the call
    const bool inl = total <= P.inline_cap && total <= P.max_records;
    if (inl) {                                                               // order_kernel's loop, one workgroup
        const uint4 *src = reinterpret_cast<const uint4 *>(P.staging);
        uint4 *dst = reinterpret_cast<uint4 *>(P.out), *hdst = reinterpret_cast<uint4 *>(P.host_out);
        for (uint32_t piece = (uint32_t)tid; piece < 4u * total; piece += 512) {
            const uint64_t key = P.keys[piece >> 2];
            if (key == kNoKey) continue;
            const uint32_t dest = P.batch_off[key >> 32] + (uint32_t)key;
            if (dest >= P.max_records) continue;
            const uint4 v = src[piece];
            dst[4 * dest + (piece & 3)] = v;
            if (hdst) hdst[4 * dest + (piece & 3)] = v;
        }
    }
    __syncthreads();                                                         // the ordered list is written
    if (tid == 0) {
        unsigned long long nf = 0, nc = 0;
        uint32_t flags = 0;
        for (int w = 0; w < 8; w++) { nf += s_sum[0][w]; nc += s_sum[1][w]; flags |= s_fl[w]; }
        if (P.d_count) *P.d_count = total;
        P.hdr->ordered = inl ? 1u : 0u;
        HostHeader *h = P.host_hdr;
        h->n_forwarded = nf;
        h->n_preambles = nc;
        h->n_records = total;
        h->flags = flags;
        h->ordered = inl ? 1u : 0u;
        __hip_atomic_store(&h->seq, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------
// order_kernel - staging (completion order) -> the final list in stream order: record of key (batch, rank)
// goes to batch_off[batch] + rank.  Four lanes per record (16 bytes each).  Lists of at most direct_cap
// records are ALSO written straight to the host's pinned copy, so that a sparse call needs no copy
// operation (and no second synchronisation) at all; d_count receives the number of records on the device
// (what a gather over RCCL exchanges first).
// ------------------------------------------------------------------------------------
struct OrderParams {
    const ResultHeader *hdr;
    const modes_record *staging;
    const uint64_t *keys;
    const uint32_t *batch_off;
    modes_record *out;             // device, max_records
    modes_record *host_out;        // device view of the pinned host list, or nullptr
    unsigned long long *d_count;   // optional
    uint32_t max_records;
    uint32_t direct_cap;
};
__global__ __launch_bounds__(256) void order_kernel(OrderParams P) {
    if (P.hdr->ordered) return;                                              // the demod kernel's last workgroup did it (short list)
    const uint32_t total = P.hdr->n_records;
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.d_count) *P.d_count = total;
    const uint32_t n = min(total, P.max_records);
    const bool direct = P.host_out != nullptr && total <= P.direct_cap;
    const uint4 *src = reinterpret_cast<const uint4 *>(P.staging);
    uint4 *dst = reinterpret_cast<uint4 *>(P.out), *hdst = reinterpret_cast<uint4 *>(P.host_out);
    const uint64_t npieces = 4ull * n;
    for (uint64_t piece = (uint64_t)blockIdx.x * 256 + threadIdx.x; piece < npieces; piece += (uint64_t)gridDim.x * 256) {
        const uint64_t key = P.keys[piece >> 2];
        if (key == kNoKey) continue;
        const uint64_t dest = (uint64_t)P.batch_off[key >> 32] + (uint32_t)key;
        if (dest >= P.max_records) continue;
        const uint4 v = src[piece];
        dst[4 * dest + (piece & 3)] = v;
        if (direct) hdst[4 * dest + (piece & 3)] = v;
    }
}

// Debug tap: modes_mag_exact for every saturated power (the demod kernel's table-free magnitude, device square root).
__global__ __launch_bounds__(256) void mag_exact_kernel(uint16_t *out) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s < MODES_LUT_ENTRIES) out[s] = (uint16_t)modes_mag_exact(s);
}

// ------------------------------------------------------------------------------------
// prefix_kernel - one workgroup: exclusive prefix of the per-run preamble counts, only needed
// when the host asked for the dense candidate list (keep_candidates, i.e. --stats).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void prefix_kernel(const uint32_t *cand_counts, uint32_t nruns, uint64_t *cand_offsets) {
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nruns + 1023) / 1024;
    const uint32_t lo = min(t * per, nruns), hi = min(lo + per, nruns);
    uint64_t c = 0;
    for (uint32_t r = lo; r < hi; r++) c += cand_counts[r];
    part[t] = c;
    __syncthreads();
    for (uint32_t step = 1; step < 1024; step <<= 1) {       // Hillis-Steele inclusive scan
        uint64_t add = 0;
        if (t >= step) add = part[t - step];
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint64_t off = part[t] - c;                              // exclusive prefix of this thread's first run
    for (uint32_t r = lo; r < hi; r++) { cand_offsets[r] = off; off += cand_counts[r]; }
}

__global__ __launch_bounds__(256) void compact_candidates_kernel(const uint32_t *cand_slots, const uint32_t *cand_counts,
                                                                 const uint64_t *cand_offsets, uint32_t nruns,
                                                                 uint32_t slot_cap, uint64_t g0, uint64_t *out) {
    const int lane = threadIdx.x & 63;
    const uint32_t run = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (run >= nruns) return;
    const uint32_t n = cand_counts[run];
    const uint64_t off = cand_offsets[run];
    for (uint32_t e = lane; e < n; e += 64) out[off + e] = (uint64_t)cand_slots[(uint64_t)run * slot_cap + e] + g0;
}

// ------------------------------------------------------------------------------------
// synthetic input (tests/synth.py:noise_bytes, integer for integer)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_noise_kernel(uint8_t *out, uint64_t first_byte, uint64_t nbytes, uint64_t seed,
                                                          uint32_t sigma_q16) {
    const uint64_t ngroups = (nbytes + 15) / 16;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const uint64_t idx = first_byte + g * 16 + b;
            const uint64_t h = mix64(seed + idx * 0x9E3779B97F4A7C15ull);
            // sum of the 8 bytes of h
            uint64_t t = (h & 0x00FF00FF00FF00FFull) + ((h >> 8) & 0x00FF00FF00FF00FFull);
            t = (t & 0x0000FFFF0000FFFFull) + ((t >> 16) & 0x0000FFFF0000FFFFull);
            const int32_t gsum = (int32_t)((t & 0xFFFFFFFFull) + (t >> 32));
            int32_t v = 127 + (((gsum - 1020) * (int32_t)sigma_q16 + 58982) >> 16);
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            w[b >> 2] |= (uint32_t)v << (8 * (b & 3));
        }
        if (g * 16 + 16 <= nbytes) {
            *reinterpret_cast<uint4 *>(out + g * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int b = 0; b < 16 && g * 16 + b < nbytes; b++) out[g * 16 + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
        }
    }
}

__global__ __launch_bounds__(256) void fill_kernel(uint8_t *out, uint64_t nbytes, uint8_t value) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = value;
}

// ------------------------------------------------------------------------------------
// stream_read_kernel - the chip's read-only streaming rate with the scan kernel's own access pattern: runs of
// `run_chunks` 1 KiB chunks per wavefront, 16 B per lane through a raw buffer descriptor with the same cache policy
// (nt | sc1), two chunks in flight, two wavefronts per workgroup - and nothing else (an XOR keeps the loads alive).
// bench.py reports it next to the 8 TB/s specification as roofline.measured_ceiling.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kScan2Waves * kWave) void stream_read_kernel(const uint8_t *__restrict__ iq, uint32_t nchunks,
                                                                          uint32_t run_chunks, uint32_t nruns, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t run = blockIdx.x * kScan2Waves + wave;
    if (run >= nruns) return;
    const uint32_t c0 = run * run_chunks;
    const uint32_t nk = min(run_chunks, nchunks - c0), last_off = (nk - 1) * kChunkBytes;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(iq) + (uint64_t)c0 * kChunkBytes, 0,
                                                                    0x7fffffff, 0x00020000);
    const uint32_t lane_off = (uint32_t)lane * 16u;
    auto load_at = [&](uint32_t off) -> u32x4 {
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, off < last_off ? off : last_off, kScanAux);
    };
    u32x4 acc = {0, 0, 0, 0};
    u32x4 x = load_at(0), y = load_at(kChunkBytes);
    uint32_t k = 0, off = 2 * kChunkBytes;
    for (; k + 2 <= nk; k += 2, off += 2 * kChunkBytes) {
        acc ^= x;
        x = load_at(off);
        acc ^= y;
        y = load_at(off + kChunkBytes);
    }
    if (k < nk) acc ^= x;
    const uint32_t v = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (v == 0x9E3779B9u) out[run] = v;                                      // (practically) never: the loads stay, nothing is written
}

}  // namespace

// ======================================================================================
// host side of the ABI
// ======================================================================================

struct modes_gpu {
    modes_gpu_config cfg{};
    int maxfix = 1;
    hipStream_t own_stream = nullptr;
    hipStream_t last_stream = nullptr;
    // Kernel times: start / stop events ATTACHED to the three dispatches (hipExtLaunchKernelGGL: the timestamps of the
    // dispatch's own completion signal - no marker packets between the kernels).  ev_done: the results are complete.
    hipEvent_t ev_k[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // scan, demod, order: start, stop
    hipEvent_t ev_done = nullptr;
    std::string err;

    uint16_t *d_lut = nullptr;
    uint32_t *d_esyn = nullptr;

    // per-detect scratch (grown on demand)
    uint32_t *d_slots = nullptr;      size_t slots_bytes = 0;
    uint32_t *d_surv = nullptr;       size_t surv_bytes = 0;         // select_kernel's survivor lists (same geometry as the candidate lists)
    uint32_t *d_wg_flags = nullptr;                                  // record_kernel: one word per workgroup (<= 512)
    bool split_path = false;          // the detect in flight ran select + record + finalize2 (the list is complete and in order)
    uint32_t *d_cand_slots = nullptr; size_t cand_slots_bytes = 0;
    uint32_t *d_counts = nullptr;     size_t counts_elems = 0;       // counts | cand_counts | batch_count | batch_off
    uint64_t *d_cand_offsets = nullptr;
    uint64_t *d_cand_dense = nullptr; size_t cand_dense_elems = 0;
    modes_record *d_staging = nullptr;  // records in completion order + their keys
    uint64_t *d_keys = nullptr;
    modes_record *d_records = nullptr;  // the ordered list (own allocation; unused while the caller supplies one)
    modes_record *d_user_records = nullptr;   // modes_gpu_set_output
    unsigned long long *d_user_count = nullptr;
    uint32_t list_cap = 0;              // capacity of d_staging / d_keys / d_records / h_records
    ResultHeader *d_hdr = nullptr;

    WgTotals *d_totals = nullptr;     // device: one per resident demod workgroup
    HostHeader *h_hdr = nullptr;      // pinned + mapped: the verdict of the call in flight (HostHeader)
    HostHeader *h_hdr_dev = nullptr;
    uint32_t seq = 0;                 // sequence number of the last detect
    bool timing = true;               // attach timing events to the kernels of the next detects (modes_gpu_set_timing)
    bool timed = false;               // ... of the detect in flight
    bool done_recorded = false;       // ev_done was recorded behind the detect in flight (kernels follow the demod kernel)
    hipStream_t tail_stream = nullptr;  // the stream the last kernel of the detect in flight runs on
    bool order_launched = false;      // order_kernel is part of the detect in flight (device-output mode)
    OrderParams order_params{};       // ... or is launched by fetch when the list turns out to be long
    uint32_t demod_grid = 0;          // workgroups of the demod launch in flight
    modes_record *h_records = nullptr;  // pinned + mapped, list_cap
    modes_record *h_records_dev = nullptr;
    std::vector<uint64_t> h_cands;

    uint8_t *d_stage = nullptr;       size_t stage_bytes = 0;
    uint32_t *d_ceiling = nullptr;    size_t ceiling_bytes = 0;     // modes_gpu_stream_ceiling's (unused) output words

    // host-side cost of modes_gpu_detect, seconds, accumulated (modes_gpu_host_profile): 0 hipSetDevice, 1 geometry + list
    // growth + parameter blocks, 2 scan launch, 3 demod launch, 4 finalize launch, 5 the rest (order / prefix kernels,
    // hipGetLastError, event records), 6 calls
    double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    uint32_t demod_wgs = 1024;        // workgroups of demod_kernel that are resident at once (occupancy x CUs)
    uint32_t select_wgs = 512;        // ... of select_kernel
    // demod_variant 0 chooses per call: the one-kernel path on (nearly) record-free input, select + record when the
    // previous call of this context left more than kSplitAbove records per GiB (+4 % there, -4 % on noise)
    double records_per_gib = 0.0;
    bool auto_records = false;        // max_records was 0: the record list grows when a call needs more
    bool full_slots = false;          // a run once overflowed the automatic slot_cap: size the lists for the worst case
    modes_gpu_span last_span{};       // what the detect in flight was asked to do (for the overflow retry)

    // geometry of the detect in flight
    bool in_flight = false;
    uint32_t nruns = 0, slot_cap = 0, nbatches = 0;
    uint64_t g0 = 0;
};

// text of the last failed create: per thread, so that hosts which create one context per thread (one per GPU) do not race
static thread_local char g_create_error[512] = "";

static int fail(modes_gpu *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else { strncpy(g_create_error, buf, sizeof g_create_error - 1); g_create_error[sizeof g_create_error - 1] = 0; }
    return code;
}

#define HIP_TRY(ctx, call)                                                                               \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return fail(ctx, MODES_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));  \
    } while (0)

template <class T>
static int grow(modes_gpu *ctx, T **ptr, size_t *have, size_t want_bytes) {
    if (*have >= want_bytes && *ptr) return MODES_OK;
    if (*ptr) { (void)hipFree(*ptr); *ptr = nullptr; *have = 0; }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(ptr), want_bytes);
    if (e != hipSuccess) return fail(ctx, MODES_ERR_NOMEM, "hipMalloc(%zu): %s", want_bytes, hipGetErrorString(e));
    *have = want_bytes;
    return MODES_OK;
}

static int wait_results(modes_gpu *ctx);

static int grid_for(uint64_t items, int per_block) {
    uint64_t b = (items + per_block - 1) / per_block;
    return (int)std::max<uint64_t>(1, std::min<uint64_t>(b, 256 * 8));
}

// (re)allocate the record lists for `cap` records: staging + keys + the ordered list on the device, the pinned host copy
static int alloc_lists(modes_gpu *ctx, uint32_t cap) {
    if (ctx->d_staging) (void)hipFree(ctx->d_staging);
    if (ctx->d_keys) (void)hipFree(ctx->d_keys);
    if (ctx->d_records) (void)hipFree(ctx->d_records);
    if (ctx->h_records) (void)hipHostFree(ctx->h_records);
    ctx->d_staging = nullptr; ctx->d_keys = nullptr; ctx->d_records = nullptr; ctx->h_records = nullptr; ctx->h_records_dev = nullptr;
    ctx->list_cap = 0;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_staging), (size_t)cap * sizeof(modes_record)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_keys), (size_t)cap * sizeof(uint64_t)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_records), (size_t)cap * sizeof(modes_record)));
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_records), (size_t)cap * sizeof(modes_record), hipHostMallocMapped));
    HIP_TRY(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->h_records_dev), ctx->h_records, 0));
    ctx->list_cap = cap;
    return MODES_OK;
}

extern "C" {

int modes_gpu_abi_version(void) { return MODES_GFX950_ABI; }

const char *modes_gpu_last_error(const modes_gpu *ctx) { return ctx ? ctx->err.c_str() : g_create_error; }

int modes_gpu_create(const modes_gpu_config *cfg, modes_gpu **out) {
    if (!cfg || !out) return fail(nullptr, MODES_ERR_ARG, "modes_gpu_create: null argument");
    *out = nullptr;
    // MODES_GPU_CREATE_TRACE=1: where the time of a create goes, to stderr (the C host's start-up is most of a short file's wall clock)
    using clk = std::chrono::steady_clock;
    const bool trace = getenv("MODES_GPU_CREATE_TRACE") != nullptr;
    clk::time_point t_prev = clk::now();
    double t_phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto phase = [&](int k) { const clk::time_point now = clk::now(); t_phase[k] += std::chrono::duration<double>(now - t_prev).count(); t_prev = now; };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, MODES_ERR_HIP, "no HIP device: libmodes_gfx950 has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, MODES_ERR_ARG, "device %d of %d", cfg->device, ndev);
    modes_gpu *ctx = new (std::nothrow) modes_gpu;
    if (!ctx) return fail(nullptr, MODES_ERR_NOMEM, "out of memory");
    ctx->cfg = *cfg;
    // Measurement / test knob: run a whole suite on one demodulation path.  It only fills in the AUTOMATIC choice - a caller that
    // asks for a path by number gets that path (the parity tests that cross-check the two paths against each other must not
    // collapse into one) - and it must be a number the library knows.
    if (const char *v = getenv("MODES_GPU_DEMOD_VARIANT")) {
        char *end = nullptr;
        const unsigned long n = strtoul(v, &end, 10);
        if (end == v || *end != 0 || !(n == 0 || n == 2 || n == 3)) {
            delete ctx;
            return fail(nullptr, MODES_ERR_ARG, "MODES_GPU_DEMOD_VARIANT='%s': 0 (automatic), 2 (two kernels) or 3 (one kernel)", v);
        }
        if (ctx->cfg.demod_variant == 0) ctx->cfg.demod_variant = (uint32_t)n;
    }
    if (ctx->cfg.scan_variant != 0) {
        const uint32_t v = ctx->cfg.scan_variant;
        delete ctx;
        return fail(nullptr, MODES_ERR_ARG, "scan_variant %u: there is one scan kernel (the single-pass first version was removed in round 4)", v);
    }
    if (ctx->cfg.demod_variant == 1 || ctx->cfg.demod_variant > 3) {
        const uint32_t v = ctx->cfg.demod_variant;
        delete ctx;
        return fail(nullptr, MODES_ERR_ARG, "demod_variant %u: 0 (automatic), 2 (two kernels) or 3 (one kernel)", v);
    }
    ctx->auto_records = ctx->cfg.max_records == 0;
    if (ctx->auto_records) ctx->cfg.max_records = 1u << 18;          // 16 MiB of records; grows on demand
    if (ctx->cfg.direct_records == 0) ctx->cfg.direct_records = 4096;
    ctx->maxfix = cfg->fix_errors ? (cfg->aggressive ? 2 : 1) : 0;
    auto bail = [&](int rc) { fail(nullptr, rc, "%s", ctx->err.c_str()); modes_gpu_destroy(ctx); return rc; };
#define CREATE_TRY(call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) { fail(ctx, MODES_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); return bail(MODES_ERR_HIP); } \
    } while (0)
    phase(0);                                                               // runtime start-up (first call of the process)
    CREATE_TRY(hipSetDevice(cfg->device));
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    phase(1);                                                               // device context, stream
    {   // demod_kernel is persistent: launch exactly as many workgroups as fit on the chip at once
        hipDeviceProp_t prop;
        int per_cu = 0;
        CREATE_TRY(hipGetDeviceProperties(&prop, cfg->device));
        CREATE_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, demod_kernel<8, LutFull>, 512, 0));
        ctx->demod_wgs = (uint32_t)std::max(1, per_cu) * (uint32_t)std::max(1, prop.multiProcessorCount);
        CREATE_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, select_kernel, kSelThreads, 0));
        ctx->select_wgs = (uint32_t)std::max(1, per_cu) * (uint32_t)std::max(1, prop.multiProcessorCount);
    }
    phase(2);                                                               // code object load (first occupancy query)
    for (auto &e : ctx->ev_k) CREATE_TRY(hipEventCreate(&e));
    CREATE_TRY(hipEventCreate(&ctx->ev_done));
    if (const char *m = getenv("MODES_GPU_TIMING"))              // measurement knob (tools/gpu_round.sh): 0 = no kernel events
        ctx->timing = atoi(m) != 0;
    // tables: the magnitude LUT exactly as the reference builds it (dump1090.c:359-364, double
    // arithmetic on the host) and the 112 single-bit syndromes.
    std::vector<uint16_t> lut(MODES_LUT_ENTRIES, 0);
    for (int i = 0; i <= 128; i++)
        for (int q = 0; q <= 128; q++)
            lut[std::min(i * i + q * q, 32767)] = (uint16_t)std::round(std::sqrt((double)(i * i + q * q)) * 360.0);
    uint32_t esyn[kSynWords];
    for (int p = 0; p < 112; p++) esyn[p] = modes_bit_syndrome(p);
    {   // ... and the same by value (syn_lookup): open addressing, linear probing
        const uint32_t mask = (1u << kSynHashBits) - 1;
        uint32_t *table = esyn + 112;
        for (uint32_t i = 0; i <= mask; i++) table[i] = kSynEmpty;
        for (uint32_t p = 0; p < 112; p++) {
            uint32_t i = (esyn[p] * kSynHashMul) >> (32 - kSynHashBits);
            while (table[i] != kSynEmpty) i = (i + 1) & mask;
            table[i] = esyn[p] << 8 | p;
        }
        uint32_t run = 0, longest = 0;
        for (uint32_t i = 0; i < 2 * (mask + 1); i++) {
            run = table[i & mask] != kSynEmpty ? run + 1 : 0;
            longest = std::max(longest, run);
        }
        if (longest >= (uint32_t)kSynProbes) { fail(ctx, MODES_ERR_ARG, "syndrome table: a run of %u cells, %d probes", longest, kSynProbes); return bail(MODES_ERR_ARG); }
    }
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_lut), lut.size() * 2));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_esyn), sizeof esyn));
    CREATE_TRY(hipMemcpy(ctx->d_lut, lut.data(), lut.size() * 2, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(ctx->d_esyn, esyn, sizeof esyn, hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_hdr), sizeof(ResultHeader)));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_totals), sizeof(WgTotals) * std::max(ctx->demod_wgs, ctx->select_wgs)));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_wg_flags), sizeof(uint32_t) * 512));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_hdr), sizeof(HostHeader), hipHostMallocMapped));
    CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->h_hdr_dev), ctx->h_hdr, 0));
    memset(ctx->h_hdr, 0, sizeof(HostHeader));
#undef CREATE_TRY
    phase(3);                                                               // tables, small buffers
    if (alloc_lists(ctx, ctx->cfg.max_records) != MODES_OK) return bail(MODES_ERR_NOMEM);
    phase(4);                                                               // record lists (device + pinned host)
    if (trace)
        fprintf(stderr, "modes_gpu_create: runtime %.4f s, context %.4f, code object %.4f, tables %.4f, lists %.4f\n", t_phase[0], t_phase[1],
                t_phase[2], t_phase[3], t_phase[4]);
    *out = ctx;
    return MODES_OK;
}

void modes_gpu_destroy(modes_gpu *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->in_flight) (void)wait_results(ctx);                            // kernels of a detect nobody fetched
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    void *dev[] = {ctx->d_lut, ctx->d_esyn, ctx->d_slots, ctx->d_cand_slots, ctx->d_counts, ctx->d_cand_offsets,
                   ctx->d_cand_dense, ctx->d_staging, ctx->d_keys, ctx->d_records, ctx->d_hdr, ctx->d_stage, ctx->d_totals, ctx->d_ceiling, ctx->d_surv, ctx->d_wg_flags};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    if (ctx->h_hdr) (void)hipHostFree(ctx->h_hdr);
    if (ctx->h_records) (void)hipHostFree(ctx->h_records);
    for (auto &e : ctx->ev_k)
        if (e) (void)hipEventDestroy(e);
    if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int modes_gpu_set_output(modes_gpu *ctx, void *d_records, uint64_t capacity, void *d_count) {
    if (!ctx) return MODES_ERR_ARG;
    if (ctx->in_flight) return fail(ctx, MODES_ERR_STATE, "set_output: a detect is in flight on this context");
    if (d_records && (capacity == 0 || capacity > 0xFFFFFFFFull || (reinterpret_cast<uintptr_t>(d_records) & 15)))
        return fail(ctx, MODES_ERR_ARG, "set_output: the list must be 16-byte aligned and hold 1 .. 2^32-1 records");
    if (d_count && (reinterpret_cast<uintptr_t>(d_count) & 7)) return fail(ctx, MODES_ERR_ARG, "set_output: d_count must be 8-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    ctx->d_user_records = static_cast<modes_record *>(d_records);
    ctx->d_user_count = static_cast<unsigned long long *>(d_count);
    if (d_records) {
        // the caller's list cannot grow: its capacity is the capacity of the call
        ctx->auto_records = false;
        ctx->cfg.max_records = (uint32_t)capacity;
        if (ctx->list_cap < capacity) { int rc = alloc_lists(ctx, (uint32_t)capacity); if (rc != MODES_OK) return rc; }
    }
    return MODES_OK;
}

// `stream` is a hipStream_t; NULL is HIP's default stream (which is also what torch's default
// stream is), NOT the context's private stream - work must stay ordered with the caller's.
static hipStream_t pick_stream(modes_gpu *, void *stream) { return static_cast<hipStream_t>(stream); }

int modes_gpu_compute_magnitude(modes_gpu *ctx, const void *d_iq, uint64_t nsamples, void *d_mag, void *stream) {
    if (!ctx) return MODES_ERR_ARG;
    if (!d_iq || !d_mag) return fail(ctx, MODES_ERR_ARG, "compute_magnitude: null pointer");
    if (nsamples == 0) return MODES_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    hipLaunchKernelGGL(magnitude_kernel, dim3(grid_for((nsamples + 7) / 8, 256)), dim3(256), 0, pick_stream(ctx, stream),
                       static_cast<const uint8_t *>(d_iq), nsamples, ctx->d_lut, static_cast<uint16_t *>(d_mag));
    HIP_TRY(ctx, hipGetLastError());
    return MODES_OK;
}

int modes_gpu_compute_power(modes_gpu *ctx, const void *d_iq, uint64_t nsamples, void *d_s, void *stream) {
    if (!ctx) return MODES_ERR_ARG;
    if (!d_iq || !d_s) return fail(ctx, MODES_ERR_ARG, "compute_power: null pointer");
    if (nsamples == 0) return MODES_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    hipLaunchKernelGGL(power_kernel, dim3(grid_for((nsamples + 7) / 8, 256)), dim3(256), 0, pick_stream(ctx, stream),
                       static_cast<const uint8_t *>(d_iq), nsamples, static_cast<uint16_t *>(d_s));
    HIP_TRY(ctx, hipGetLastError());
    return MODES_OK;
}

int modes_gpu_debug_tables(modes_gpu *ctx, void *d_lut, void *d_exact, void *stream) {
    if (!ctx) return MODES_ERR_ARG;
    if (!d_lut || !d_exact) return fail(ctx, MODES_ERR_ARG, "debug_tables: null pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipMemcpyAsync(d_lut, ctx->d_lut, MODES_LUT_ENTRIES * 2, hipMemcpyDeviceToDevice, pick_stream(ctx, stream)));
    hipLaunchKernelGGL(mag_exact_kernel, dim3(MODES_LUT_ENTRIES / 256), dim3(256), 0, pick_stream(ctx, stream),
                       static_cast<uint16_t *>(d_exact));
    HIP_TRY(ctx, hipGetLastError());
    return MODES_OK;
}

int modes_gpu_synth_noise(modes_gpu *ctx, void *d_out, uint64_t first_byte, uint64_t nbytes, uint64_t seed,
                          uint32_t sigma_q16, void *stream) {
    if (!ctx) return MODES_ERR_ARG;
    if (!d_out) return fail(ctx, MODES_ERR_ARG, "synth_noise: null pointer");
    if (nbytes == 0) return MODES_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    hipLaunchKernelGGL(synth_noise_kernel, dim3(grid_for((nbytes + 15) / 16, 256)), dim3(256), 0, pick_stream(ctx, stream),
                       static_cast<uint8_t *>(d_out), first_byte, nbytes, seed, sigma_q16);
    HIP_TRY(ctx, hipGetLastError());
    return MODES_OK;
}

int modes_gpu_fill(modes_gpu *ctx, void *d_out, uint64_t nbytes, uint8_t value, void *stream) {
    if (!ctx) return MODES_ERR_ARG;
    if (!d_out) return fail(ctx, MODES_ERR_ARG, "fill: null pointer");
    if (nbytes == 0) return MODES_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(nbytes, 256)), dim3(256), 0, pick_stream(ctx, stream),
                       static_cast<uint8_t *>(d_out), nbytes, value);
    HIP_TRY(ctx, hipGetLastError());
    return MODES_OK;
}

int modes_gpu_detect(modes_gpu *ctx, const modes_gpu_span *span, void *stream) {
    if (!ctx) return MODES_ERR_ARG;
    if (!span || !span->iq) return fail(ctx, MODES_ERR_ARG, "detect: null span");
    if (ctx->in_flight) return fail(ctx, MODES_ERR_STATE, "detect: a detect is already in flight on this context (fetch it first)");
    if (span->nblocks == 0) return fail(ctx, MODES_ERR_ARG, "detect: nblocks == 0");
    if ((reinterpret_cast<uintptr_t>(span->iq) & 1) != 0) return fail(ctx, MODES_ERR_ARG, "detect: iq must be 2-byte aligned");
    if (span->stream_byte0 & 1) return fail(ctx, MODES_ERR_ARG, "detect: stream_byte0 must be even");
    if (span->nbytes > (1ull << 33) - 65536) return fail(ctx, MODES_ERR_ARG, "detect: at most 8 GiB - 64 KiB per call");
    // buffer `first_block` starts 476 bytes before stream byte 262144*first_block: the span must
    // reach back that far (or start at the beginning of the stream, where the carry is 127s).
    const uint64_t need0 = span->first_block * (uint64_t)MODES_DATA_LEN;
    const uint64_t carry0 = need0 >= MODES_CARRY_BYTES ? need0 - MODES_CARRY_BYTES : 0;
    if (span->stream_byte0 > carry0)
        return fail(ctx, MODES_ERR_ARG, "detect: span starts at stream byte %llu, buffer %llu needs bytes from %llu",
                    (unsigned long long)span->stream_byte0, (unsigned long long)span->first_block, (unsigned long long)carry0);
    using clk = std::chrono::steady_clock;
    clk::time_point t_mark = clk::now();
    auto mark = [&](int k) { const clk::time_point now = clk::now(); ctx->prof[k] += std::chrono::duration<double>(now - t_mark).count(); t_mark = now; };
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    mark(0);
    hipStream_t st = pick_stream(ctx, stream);

    // The kernels address the stream from a 16-byte aligned base; a span that starts inside a
    // 16-byte line (e.g. a 476-byte carry cut out of a bigger device buffer) gets `mis` bytes of
    // slack in front, which read as "outside the span".
    const uint64_t mis = reinterpret_cast<uintptr_t>(span->iq) & 15;
    const uint8_t *base = static_cast<const uint8_t *>(span->iq) - mis;
    const int64_t byte_lo = (int64_t)mis, byte_hi = (int64_t)(mis + span->nbytes);

    // framed coordinates: g = 131072*k + j;  base sample p  <->  g = p + g0
    const uint64_t g0 = span->stream_byte0 / 2 + MODES_CARRY_SAMPLES - mis / 2;
    const uint64_t g_begin = span->first_block * (uint64_t)MODES_BLOCK_STRIDE;
    const uint64_t g_end = (span->first_block + span->nblocks) * (uint64_t)MODES_BLOCK_STRIDE;
    // positions before the span or past its last sample have magnitude 0 at m[0] and can never
    // satisfy m[0] > m[1] (dump1090.c:1602), so only samples inside the span need scanning.
    int64_t p_begin = (int64_t)g_begin - (int64_t)g0;
    int64_t p_end = (int64_t)g_end - (int64_t)g0;
    if (p_begin < byte_lo / 2) p_begin = byte_lo / 2;
    if (p_end > (byte_hi + 1) / 2) p_end = (byte_hi + 1) / 2;
    if (p_end < p_begin) p_end = p_begin;

    const uint64_t nchunks = (uint64_t)(p_end + kLookback + kChunkSamples - 1) / kChunkSamples;
    uint32_t R = ctx->cfg.run_chunks;
    // automatic run length: long runs amortise a run's set-up in the scan (16 chunks is enough), but the
    // demod kernel wants at least one batch of kDemodGroup runs per resident workgroup (two per workgroup, dealt
    // dynamically, measured worse: the last batch handed out ends one batch time after the average - profiles/r02k)
    if (R == 0) R = (uint32_t)std::max<uint64_t>(4, std::min<uint64_t>(32, nchunks / ((uint64_t)kDemodGroup * ctx->demod_wgs)));
    if (R > 8192) return fail(ctx, MODES_ERR_ARG, "run_chunks=%u: at most 8192", R);
    R += R & 1;                                                   // the scan loop is unrolled by two chunks
    const uint32_t nruns = (uint32_t)std::max<uint64_t>(1, (nchunks + R - 1) / R);
    uint32_t cap = ctx->cfg.slot_cap;
    if (cap == 0) cap = ctx->full_slots ? R * kChunkSamples : std::max<uint32_t>(64, R * 32);   // 1/16 of the run's positions
    if (cap > R * (uint32_t)kChunkSamples) cap = R * kChunkSamples;

    int rc;
    const uint32_t nbatches = (nruns + kDemodGroup - 1) / kDemodGroup;      // demod_kernel's unit of work, of candidate lists and of record order
    size_t want = (size_t)nbatches * kDemodGroup * cap * sizeof(uint32_t);
    if ((rc = grow(ctx, &ctx->d_slots, &ctx->slots_bytes, want)) != MODES_OK) return rc;
    const bool split = ctx->cfg.demod_variant == 2 || (ctx->cfg.demod_variant == 0 && ctx->records_per_gib > kSplitAbove);
    if (split) {
        // record_kernel requests a batch's first sixteen survivor slots TOGETHER with its count (one round trip instead of two) - slots
        // select_kernel has not written when the batch has fewer survivors; the values are never used, but they are read: the list is
        // zeroed once when it is (re)allocated, and a batch always has sixteen slots
        static_assert(kDemodGroup >= 16, "record_kernel reads slots wave and wave + 8 of a batch before it knows the batch's count");
        uint32_t *const before = ctx->d_surv;
        const size_t had = ctx->surv_bytes;
        if ((rc = grow(ctx, &ctx->d_surv, &ctx->surv_bytes, want)) != MODES_OK) return rc;
        if (ctx->d_surv != before || ctx->surv_bytes != had) HIP_TRY(ctx, hipMemsetAsync(ctx->d_surv, 0, ctx->surv_bytes, st));
    }
    if (ctx->cfg.keep_candidates && (rc = grow(ctx, &ctx->d_cand_slots, &ctx->cand_slots_bytes, want)) != MODES_OK) return rc;
    if (ctx->counts_elems < nruns) {
        if (ctx->d_counts) (void)hipFree(ctx->d_counts);
        if (ctx->d_cand_offsets) (void)hipFree(ctx->d_cand_offsets);
        ctx->d_counts = nullptr; ctx->d_cand_offsets = nullptr; ctx->counts_elems = 0;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_counts), (size_t)nruns * 4 * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_cand_offsets), (size_t)nruns * sizeof(uint64_t)));
        ctx->counts_elems = nruns;
    }
    // counts [nruns] | cand_counts | batch_count | batch_off (nbatches <= nruns each)
    uint32_t *d_counts = ctx->d_counts, *d_cand_counts = ctx->d_counts + nruns, *d_batch_count = ctx->d_counts + 2 * (size_t)nruns,
             *d_batch_off = ctx->d_counts + 3 * (size_t)nruns;

    ScanParams sp{};
    sp.iq = base;
    sp.lo = byte_lo;
    sp.hi = byte_hi;
    sp.g0 = g0;
    sp.p_begin = p_begin;
    sp.p_end = p_end;
    sp.nchunks = (uint32_t)nchunks;
    sp.run_chunks = R;
    sp.nruns = nruns;
    sp.slot_cap = cap;
    sp.slots = ctx->d_slots;
    sp.counts = d_counts;
    sp.hdr = ctx->d_hdr;

    DemodParams dp{};
    dp.iq = sp.iq;
    dp.lo = sp.lo;
    dp.hi = sp.hi;
    dp.g0 = g0;
    dp.nruns = nruns;
    dp.run_chunks = R;
    dp.slot_cap = cap;
    dp.slots = ctx->d_slots;
    dp.counts = d_counts;
    dp.tab = DeviceTables{ctx->d_lut, ctx->d_esyn};
    dp.maxfix = ctx->maxfix;
    dp.aggressive = ctx->cfg.aggressive ? 1 : 0;
    dp.cand_slots = ctx->cfg.keep_candidates ? ctx->d_cand_slots : nullptr;
    dp.cand_counts = d_cand_counts;
    dp.staging = ctx->d_staging;
    dp.keys = ctx->d_keys;
    dp.batch_count = d_batch_count;
    dp.batch_off = d_batch_off;
    dp.nbatches = nbatches;
    dp.hdr = ctx->d_hdr;
    dp.totals = ctx->d_totals;
    dp.max_records = ctx->cfg.max_records;

    FinalizeParams fp{};
    fp.hdr = ctx->d_hdr;
    fp.totals = ctx->d_totals;
    fp.batch_count = d_batch_count;
    fp.batch_off = d_batch_off;
    fp.nbatches = nbatches;
    fp.staging = ctx->d_staging;
    fp.keys = ctx->d_keys;
    fp.out = ctx->d_user_records ? ctx->d_user_records : ctx->d_records;
    fp.host_out = ctx->h_records_dev;
    fp.d_count = ctx->d_user_count;
    fp.host_hdr = ctx->h_hdr_dev;
    fp.max_records = ctx->cfg.max_records;
    fp.inline_cap = std::min(ctx->cfg.direct_records, ctx->cfg.max_records);
    fp.seq = ++ctx->seq;

    OrderParams op{};
    op.hdr = ctx->d_hdr;
    op.staging = ctx->d_staging;
    op.keys = ctx->d_keys;
    op.batch_off = d_batch_off;
    op.out = ctx->d_user_records ? ctx->d_user_records : ctx->d_records;
    op.host_out = ctx->h_records_dev;
    op.d_count = ctx->d_user_count;
    op.max_records = ctx->cfg.max_records;
    op.direct_cap = std::min(ctx->cfg.direct_records, ctx->cfg.max_records);

    // The header is zeroed by the scan kernel itself, so a detect is exactly: scan, demod, finalize (, order) (, prefix) - and,
    // unless the caller asked for kernel times, NOTHING else in the stream: the results' completion is a word finalize_kernel
    // writes to pinned host memory (HostHeader.seq), so the scan kernel of the next call starts when this call's kernels end.
    // Kernel times (modes_gpu_set_timing; on by default): start / stop events attached to the dispatches - they cost ~9 us
    // of idle GPU per kernel boundary, so a pipelined host samples them (bench.py: one call in eight).
    const bool timed = ctx->timing;
    ctx->timed = timed;
    hipEvent_t *ek = ctx->ev_k;
    auto ev = [&](int k) -> hipEvent_t { return timed ? ek[k] : nullptr; };
    const dim3 scan_grid((nruns + kScan2Waves - 1) / kScan2Waves);
    mark(1);
    hipExtLaunchKernelGGL(scan_kernel, scan_grid, dim3(kScan2Waves * kWave), 0, st, ev(0), ev(1), 0, sp);
    mark(2);
    // cfg.overlap == 1: everything after the scan moves to the context's own stream (ordered behind the
    // scan), so the next kernel on the caller's stream - typically another context's scan -
    // may run concurrently with this latency-bound tail (in practice a demod workgroup's 77 KiB of LDS only
    // find room when the scan drains: DESIGN.md 3.2).
    hipStream_t st2 = st;
    if (ctx->cfg.overlap == 1 && st != ctx->own_stream) {
        st2 = ctx->own_stream;
        if (!timed) HIP_TRY(ctx, hipEventRecord(ek[1], st));
        HIP_TRY(ctx, hipStreamWaitEvent(st2, ek[1], 0));
    }
    ctx->split_path = split;
    ctx->demod_grid = std::min<uint32_t>(nbatches, split ? ctx->select_wgs : ctx->demod_wgs);   // persistent workgroups: what is resident at once
    if (ctx->split_path) {
        SelectParams sel{};
        sel.iq = sp.iq; sel.lo = sp.lo; sel.hi = sp.hi;
        sel.nruns = nruns; sel.run_chunks = R; sel.slot_cap = cap;
        sel.slots = ctx->d_slots; sel.counts = d_counts; sel.lut = ctx->d_lut;
        sel.cand_slots = dp.cand_slots; sel.cand_counts = d_cand_counts;
        sel.surv = ctx->d_surv; sel.batch_count = d_batch_count; sel.nbatches = nbatches; sel.totals = ctx->d_totals;
        hipExtLaunchKernelGGL(select_kernel, dim3(ctx->demod_grid), dim3(kSelThreads), 0, st2, ev(2), ev(3), 0, sel);
        mark(3);
        RecordParams rp{};
        rp.d = dp;
        rp.d.staging = fp.out;                                              // records go straight to their place in the ordered list
        rp.surv = ctx->d_surv; rp.batch_count = d_batch_count;
        rp.host_out = ctx->h_records_dev; rp.direct_cap = fp.inline_cap; rp.wg_flags = ctx->d_wg_flags;
        const uint32_t rec_grid = std::min<uint32_t>(nbatches, 512);
        hipExtLaunchKernelGGL(record_kernel, dim3(rec_grid), dim3(512), 0, st2, ev(4), ev(5), 0, rp);
        Finalize2Params f2{};
        f2.hdr = ctx->d_hdr; f2.totals = ctx->d_totals; f2.ntotals = ctx->demod_grid; f2.wg_flags = ctx->d_wg_flags; f2.nwg = rec_grid;
        f2.batch_count = d_batch_count; f2.nbatches = nbatches; f2.d_count = ctx->d_user_count; f2.host_hdr = ctx->h_hdr_dev; f2.seq = fp.seq;
        hipExtLaunchKernelGGL(finalize2_kernel, dim3(1), dim3(512), 0, st2, nullptr, nullptr, 0, f2);
        mark(4);
    } else {
    hipExtLaunchKernelGGL((demod_kernel<8, LutFull>), dim3(ctx->demod_grid), dim3(512), 0, st2, ev(2), ev(3), 0, dp);
    mark(3);
    fp.ntotals = ctx->demod_grid;
    hipExtLaunchKernelGGL(finalize_kernel, dim3(1), dim3(512), 0, st2, nullptr, nullptr, 0, fp);
    mark(4);
    }
    // Lists of up to direct_records records are put in order by finalize_kernel.  order_kernel only follows in the
    // stream when the caller consumes the list on the device in stream order (MODES_GPU_ORDER_IN_STREAM); otherwise
    // modes_gpu_fetch / modes_gpu_fetch_device launch it when a list turns out to be long.
    const bool tail = ctx->cfg.keep_candidates != 0;                       // prefix_kernel follows
    ctx->order_launched = !ctx->split_path && ctx->d_user_records != nullptr && (ctx->cfg.flags & MODES_GPU_ORDER_IN_STREAM) != 0;
    ctx->order_params = op;
    if (ctx->order_launched) {
        if (ctx->cfg.overlap == 2 && st != ctx->own_stream) {              // only the order kernel leaves the caller's stream
            st2 = ctx->own_stream;
            if (!timed) HIP_TRY(ctx, hipEventRecord(ek[3], st));
            HIP_TRY(ctx, hipStreamWaitEvent(st2, ek[3], 0));
        }
        hipExtLaunchKernelGGL(order_kernel, dim3(512), dim3(256), 0, st2, ev(4), ev(5), 0, op);
    }
    if (tail)
        hipLaunchKernelGGL(prefix_kernel, dim3(1), dim3(1024), 0, st2, d_cand_counts, nbatches, ctx->d_cand_offsets);
    HIP_TRY(ctx, hipGetLastError());
    // kernels behind finalize_kernel: their completion is an event (the host word covers the kernels up to finalize)
    ctx->done_recorded = ctx->order_launched || tail;
    if (ctx->done_recorded) HIP_TRY(ctx, hipEventRecord(ctx->ev_done, st2));
    ctx->tail_stream = st2;
    mark(5);
    ctx->prof[6] += 1.0;

    ctx->last_stream = st;
    ctx->last_span = *span;
    ctx->in_flight = true;
    ctx->nruns = nruns;
    ctx->nbatches = nbatches;
    ctx->slot_cap = cap;
    ctx->g0 = g0;
    return MODES_OK;
}

// Blocks until the results of the detect in flight are complete.  The demod kernel's last workgroup publishes the call's
// sequence number in pinned host memory: poll it (no packet in the stream); while polling, look at the stream from time
// to time so that a faulted kernel is an error, not a hang.
static int wait_results(modes_gpu *ctx) {
    volatile uint32_t *seq = &ctx->h_hdr->seq;
    // The word is polled hot for as long as a call of the throughput workloads can take (2 ms: a nap oversleeps by 50 us and more,
    // and with naps from 0.25 ms on the 1 GiB low-SNR step went from 0.236 to 0.269 ms), then politely: a host has one such thread
    // per context in flight and per rank, next to its resolver's pool, and a container's CPU quota counts a spinning thread like
    // a working one (the leases hold a 256-thread host to 16 CPUs) - a host that waits for a live stream's next
    // buffer, or for a call of many GiB, does so in 20 us naps.
    using clk = std::chrono::steady_clock;
    clk::time_point t0{};
    uint32_t naps = 0;
    for (uint64_t spins = 1;; spins++) {
        if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == ctx->seq) break;
        if ((spins & 0x3FF) == 0) {
            if (spins == 0x400) t0 = clk::now();
            const double waited_us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            if (waited_us > 2000.0) {
                if ((++naps & 15) == 0) {                                   // every ~1 ms: has the stream failed, or ended without a word?
                    const hipError_t q = hipStreamQuery(ctx->tail_stream);
                    if (q == hipSuccess) {                                  // everything on the stream has run
                        if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == ctx->seq) break;
                        return fail(ctx, MODES_ERR_HIP, "the kernels of the call completed without publishing results");
                    }
                    if (q != hipErrorNotReady) return fail(ctx, MODES_ERR_HIP, "while waiting for results: %s", hipGetErrorString(q));
                }
                struct timespec nap = {0, 20000};
                nanosleep(&nap, nullptr);
            }
        }
    }
    if (ctx->done_recorded) HIP_TRY(ctx, hipEventSynchronize(ctx->ev_done));
    return MODES_OK;
}

// Common part of fetch / fetch_device: waits for the detect, handles the two overflow cases (which repeat the
// call unless MODES_GPU_NO_RETRY), fills the counters of *res.  On return *n_out = records of the call.
static int finish_detect(modes_gpu *ctx, modes_gpu_result *res, bool to_host) {
    if (!res) return fail(ctx, MODES_ERR_ARG, "fetch: null result");
    if (!ctx->in_flight) return fail(ctx, MODES_ERR_STATE, "fetch: no detect in flight");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    // Wait for THIS detect only (other work may already be queued behind it on the caller's stream): the word the demod
    // kernel's last workgroup publishes, then the event behind any kernel that followed it.  What fetch itself launches
    // runs on the context's own stream.
    int rcw = wait_results(ctx);
    if (rcw != MODES_OK) { ctx->in_flight = false; return rcw; }             // the call is lost; the context stays usable
    hipStream_t st = ctx->own_stream;
    ctx->in_flight = false;
    memset(res, 0, sizeof *res);
    const HostHeader hh = *ctx->h_hdr;
    const uint64_t n_forwarded = hh.n_forwarded, n_preambles = hh.n_preambles, n_records = hh.n_records;
    const uint32_t flags = hh.flags;
    const bool no_retry = (ctx->cfg.flags & MODES_GPU_NO_RETRY) != 0;
    if (flags & 2u) return fail(ctx, MODES_ERR_HIP, "internal: the noise-gate pre-test and the full demodulation disagree");
    if (flags & 1u) {
        // More than slot_cap positions of one run look like preambles (the automatic cap is 1/16 of
        // the positions; only a periodic, preamble-like signal gets there).  Nothing may be dropped:
        // with the automatic cap, repeat the call with worst-case lists and keep them from now on.
        if (ctx->cfg.slot_cap != 0 || ctx->full_slots || no_retry) {
            const bool can_grow = ctx->cfg.slot_cap == 0 && !ctx->full_slots;
            ctx->full_slots = ctx->full_slots || can_grow;              // a resubmitted call gets the big lists
            return fail(ctx, MODES_ERR_OVERFLOW, "scan forwarded more than slot_cap=%u positions in one run%s", ctx->slot_cap,
                        can_grow ? "; resubmit the call (the lists are worst-case sized now)" : "; raise slot_cap");
        }
        ctx->full_slots = true;
        const modes_gpu_span again = ctx->last_span;
        int rc = modes_gpu_detect(ctx, &again, ctx->last_stream);
        if (rc != MODES_OK) return rc;
        return finish_detect(ctx, res, to_host);
    }
    if (n_records > ctx->cfg.max_records) {
        // With the automatic capacity, grow the lists with some slack and repeat the call; nothing may be dropped.
        const uint64_t want = n_records + (n_records >> 2) + 65536;
        res->n_records = n_records;                                     // what the call needs
        if (!ctx->auto_records || want > 0xFFFFFFFFull)
            return fail(ctx, MODES_ERR_OVERFLOW, "%llu records exceed max_records=%u", (unsigned long long)n_records, ctx->cfg.max_records);
        ctx->cfg.max_records = (uint32_t)want;
        int rc = alloc_lists(ctx, (uint32_t)want);
        if (rc != MODES_OK) return rc;
        if (no_retry)
            return fail(ctx, MODES_ERR_OVERFLOW, "%llu records exceeded the list; it holds %llu now: resubmit the call",
                        (unsigned long long)n_records, (unsigned long long)want);
        const modes_gpu_span again = ctx->last_span;
        rc = modes_gpu_detect(ctx, &again, ctx->last_stream);
        if (rc != MODES_OK) return rc;
        return finish_detect(ctx, res, to_host);
    }
    const modes_record *d_list = ctx->d_user_records ? ctx->d_user_records : ctx->d_records;
    bool ordered_late = false;
    if (ctx->split_path) {                                              // in order already; the first direct_records are on the host too
        if (to_host && n_records > std::min(ctx->cfg.direct_records, ctx->cfg.max_records))
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_records, d_list, (size_t)n_records * sizeof(modes_record), hipMemcpyDeviceToHost, st));
    } else if (!hh.ordered) {                                           // a long list: short ones are complete, on the host too
        if (!ctx->order_launched) {
            hipExtLaunchKernelGGL(order_kernel, dim3(512), dim3(256), 0, st, ctx->timed ? ctx->ev_k[4] : nullptr,
                                  ctx->timed ? ctx->ev_k[5] : nullptr, 0, ctx->order_params);
            HIP_TRY(ctx, hipGetLastError());
            ordered_late = true;
        }
        if (to_host)
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_records, d_list, (size_t)n_records * sizeof(modes_record), hipMemcpyDeviceToHost, st));
    }
    if (ctx->cfg.keep_candidates && n_preambles) {
        if (ctx->cand_dense_elems < n_preambles) {
            if (ctx->d_cand_dense) (void)hipFree(ctx->d_cand_dense);
            ctx->d_cand_dense = nullptr; ctx->cand_dense_elems = 0;
            const size_t want = (size_t)n_preambles + (n_preambles >> 2) + 1024;
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_cand_dense), want * sizeof(uint64_t)));
            ctx->cand_dense_elems = want;
        }
        hipLaunchKernelGGL(compact_candidates_kernel, dim3((ctx->nbatches + 3) / 4), dim3(256), 0, st, ctx->d_cand_slots,
                           ctx->d_counts + ctx->nruns, ctx->d_cand_offsets, ctx->nbatches, ctx->slot_cap * kDemodGroup, ctx->g0,
                           ctx->d_cand_dense);
        HIP_TRY(ctx, hipGetLastError());
        ctx->h_cands.resize(n_preambles);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_cands.data(), ctx->d_cand_dense, (size_t)n_preambles * sizeof(uint64_t),
                                    hipMemcpyDeviceToHost, st));
    } else {
        ctx->h_cands.clear();
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    // candidates: the workgroup lists keep no order inside a block of positions
    std::sort(ctx->h_cands.begin(), ctx->h_cands.end());
    ctx->records_per_gib = (double)n_records * 1073741824.0 / (double)std::max<uint64_t>(ctx->last_span.nbytes, 1);
    res->records = to_host ? ctx->h_records : d_list;
    res->n_records = n_records;
    res->candidates = ctx->h_cands.empty() ? nullptr : ctx->h_cands.data();
    res->n_candidates = ctx->h_cands.size();
    res->n_forwarded = n_forwarded;
    res->n_preambles = n_preambles;
    if (ctx->timed) {
        // the host word is published before the demod kernel has retired: its stop event may still be pending
        const bool third = ctx->order_launched || ordered_late || ctx->split_path;     // split path: the record kernel's events
        (void)hipEventSynchronize(ctx->ev_k[third ? 5 : 3]);
        (void)hipEventElapsedTime(&res->scan_ms, ctx->ev_k[0], ctx->ev_k[1]);
        (void)hipEventElapsedTime(&res->demod_ms, ctx->ev_k[2], ctx->ev_k[3]);
        if (third) (void)hipEventElapsedTime(&res->order_ms, ctx->ev_k[4], ctx->ev_k[5]);
    }
    return MODES_OK;
}

int modes_gpu_host_profile(modes_gpu *ctx, double out[8], int reset) {
    if (!ctx || !out) return MODES_ERR_ARG;
    for (int k = 0; k < 8; k++) out[k] = ctx->prof[k];
    if (reset)
        for (double &v : ctx->prof) v = 0.0;
    return MODES_OK;
}

int modes_gpu_stream_ceiling(modes_gpu *ctx, const void *d_iq, uint64_t nbytes, uint32_t launches, uint32_t time_every,
                             float *avg_ms, float *min_ms, void *stream) {
    if (!ctx) return MODES_ERR_ARG;
    if (!d_iq || !avg_ms || launches == 0 || time_every == 0) return fail(ctx, MODES_ERR_ARG, "stream_ceiling: bad argument");
    if (reinterpret_cast<uintptr_t>(d_iq) & 15) return fail(ctx, MODES_ERR_ARG, "stream_ceiling: d_iq must be 16-byte aligned");
    const uint64_t nchunks64 = nbytes / kChunkBytes;
    if (nchunks64 == 0 || nchunks64 > 0x7fffffffull / 2) return fail(ctx, MODES_ERR_ARG, "stream_ceiling: 1 KiB .. 1 TiB");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    hipStream_t st = pick_stream(ctx, stream);
    const uint32_t nchunks = (uint32_t)nchunks64, R = 32, nruns = (nchunks + R - 1) / R;
    int rc;
    if ((rc = grow(ctx, &ctx->d_ceiling, &ctx->ceiling_bytes, (size_t)nruns * sizeof(uint32_t))) != MODES_OK) return rc;
    // the launches are queued back to back without a host round trip (an idle chip boosts its clocks: DESIGN.md 3.1); one
    // in time_every carries start / stop events attached to its dispatch - the scan kernel's own timing method
    const uint32_t ntimed = (launches + time_every - 1) / time_every;
    struct Events {                                                          // destroyed on every way out of this function
        std::vector<hipEvent_t> v;
        ~Events() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); }
    } guard;
    std::vector<hipEvent_t> &evs = guard.v;
    evs.assign(2 * (size_t)ntimed, nullptr);
    for (auto &e : evs) HIP_TRY(ctx, hipEventCreate(&e));
    uint32_t t = 0;
    for (uint32_t i = 0; i < launches; i++) {
        const bool timed = (i % time_every) == time_every - 1 || (i == launches - 1 && t < ntimed);
        hipExtLaunchKernelGGL(stream_read_kernel, dim3((nruns + kScan2Waves - 1) / kScan2Waves), dim3(kScan2Waves * kWave), 0, st,
                              timed ? evs[2 * t] : nullptr, timed ? evs[2 * t + 1] : nullptr, 0,
                              static_cast<const uint8_t *>(d_iq), nchunks, R, nruns, ctx->d_ceiling);
        if (timed) t++;
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    double sum = 0.0;
    float best = 1e30f;
    for (uint32_t k = 0; k < t; k++) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, evs[2 * k], evs[2 * k + 1]));
        sum += ms;
        best = std::min(best, ms);
    }
    *avg_ms = t ? (float)(sum / t) : 0.f;
    if (min_ms) *min_ms = t ? best : 0.f;
    return MODES_OK;
}

int modes_gpu_stream_wait(modes_gpu *ctx, void *stream) {
    if (!ctx) return MODES_ERR_ARG;
    if (!ctx->in_flight) return fail(ctx, MODES_ERR_STATE, "stream_wait: no detect in flight");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    if (!ctx->done_recorded) {                                              // the detect left no event behind: record one now
        HIP_TRY(ctx, hipEventRecord(ctx->ev_done, ctx->tail_stream));
        ctx->done_recorded = true;
    }
    HIP_TRY(ctx, hipStreamWaitEvent(static_cast<hipStream_t>(stream), ctx->ev_done, 0));
    return MODES_OK;
}

int modes_gpu_set_timing(modes_gpu *ctx, int on) {
    if (!ctx) return MODES_ERR_ARG;
    ctx->timing = on != 0;
    return MODES_OK;
}

int modes_gpu_fetch(modes_gpu *ctx, modes_gpu_result *res) {
    if (!ctx) return MODES_ERR_ARG;
    return finish_detect(ctx, res, true);
}

int modes_gpu_fetch_device(modes_gpu *ctx, modes_gpu_result *res) {
    if (!ctx) return MODES_ERR_ARG;
    return finish_detect(ctx, res, false);
}

int modes_gpu_submit_host(modes_gpu *ctx, const uint8_t *iq, uint64_t nbytes, uint64_t stream_byte0, uint64_t first_block,
                          uint64_t nblocks) {
    if (!ctx) return MODES_ERR_ARG;
    if (!iq && nbytes) return fail(ctx, MODES_ERR_ARG, "submit_host: null iq");
    if (ctx->in_flight) return fail(ctx, MODES_ERR_STATE, "submit_host: a detect is already in flight on this context");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    int rc;
    const size_t want = (size_t)((nbytes + 15) & ~15ull) + 16;
    if ((rc = grow(ctx, &ctx->d_stage, &ctx->stage_bytes, want)) != MODES_OK) return rc;
    // host -> HBM on the context's own stream, then the kernels behind it: asynchronous when `iq` is
    // pinned (modes_gpu_host_alloc), so the caller can fill its other buffer meanwhile
    if (nbytes) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage, iq, nbytes, hipMemcpyHostToDevice, ctx->own_stream));
    modes_gpu_span span{ctx->d_stage, nbytes, stream_byte0, first_block, nblocks};
    return modes_gpu_detect(ctx, &span, ctx->own_stream);
}

int modes_gpu_demod_host(modes_gpu *ctx, const uint8_t *iq, uint64_t nbytes, uint64_t stream_byte0, uint64_t first_block,
                         uint64_t nblocks, modes_gpu_result *res) {
    const int rc = modes_gpu_submit_host(ctx, iq, nbytes, stream_byte0, first_block, nblocks);
    return rc != MODES_OK ? rc : modes_gpu_fetch(ctx, res);
}

int modes_gpu_host_alloc(modes_gpu *ctx, size_t nbytes, void **out) {
    if (!ctx || !out) return MODES_ERR_ARG;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    hipError_t e = hipHostMalloc(out, nbytes ? nbytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, MODES_ERR_NOMEM, "hipHostMalloc(%zu): %s", nbytes, hipGetErrorString(e));
    return MODES_OK;
}

void modes_gpu_host_free(modes_gpu *ctx, void *p) {
    if (ctx && p) { (void)hipSetDevice(ctx->cfg.device); (void)hipHostFree(p); }
}

#ifdef MODES_TRACE
int modes_gpu_trace(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 8192 * 8) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
