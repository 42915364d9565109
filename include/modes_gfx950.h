/* modes_gfx950.h - C ABI of libmodes_gfx950.so, the MI355X (gfx950) replacement for
 * dump1090's IQ -> candidate-record hot path.
 *
 * The reference has no plugin interface: the path is entered by two calls the
 * main thread makes once per 256 KiB buffer,
 *
 *     computeMagnitudeVector();                          dump1090.c:2974 -> :1454
 *     detectModeS(Modes.magnitude, Modes.data_len/2);    dump1090.c:2986 -> :1563
 *
 * and left through useModesMessage(&mm) (dump1090.c:1777).  This header is what a
 * C host binds instead (INTEGRATION.md shows the patch): the two stateless stages
 * run on the GPU over MANY buffers at once and come back as fixed-size records,
 * one per preamble position that survives the first noise gate; the stateful,
 * in-order remainder of detectModeS()/decodeModesMessage() (skip window, retry,
 * ICAO whitelist; dump1090.c:1731-1791, 1181-1210) stays on the host in
 * modes_host.h and consumes those records.
 *
 * Plain C: pointers, sizes, POD structs.  No C++ / torch types.  Every function
 * returns 0 or a negative MODES_ERR_*; modes_gpu_last_error() has the text.  A
 * context is used from one host thread at a time (like the reference's main
 * thread).  There is NO CPU fallback: without a HIP device create() fails.
 */
#ifndef MODES_GFX950_H
#define MODES_GFX950_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MODES_OK            0
#define MODES_ERR_ARG      (-1)   /* bad argument / span geometry                     */
#define MODES_ERR_HIP      (-2)   /* HIP runtime error (text has hipGetErrorString)   */
#define MODES_ERR_NOMEM    (-3)
#define MODES_ERR_OVERFLOW (-4)   /* more records than max_records                    */
#define MODES_ERR_STATE    (-5)   /* fetch without a detect in flight, ...            */

/* Buffer geometry of the reference (dump1090.c:54,61,331): buffer k holds stream
 * bytes [262144k - 476, 262144(k+1)), 127 outside the stream; 131310 magnitudes;
 * detectModeS tests local offsets j in [0,131070).  "Framed coordinate"
 * g = 131072*k + j  <->  file sample g - 238. */
#ifndef MODES_DATA_LEN                    /* dump1090.c:54 has its own spelling of the same number */
#define MODES_DATA_LEN        262144u
#endif
#define MODES_CARRY_BYTES     476u
#define MODES_BLOCK_STRIDE    131072u     /* samples between buffer starts            */
#define MODES_BLOCK_POSITIONS 131070u     /* tested j per buffer (dump1090.c:1593)    */
#define MODES_CARRY_SAMPLES   238u

typedef struct modes_gpu modes_gpu;       /* opaque context                           */

typedef struct {
    int32_t  device;           /* HIP device ordinal                                  */
    int32_t  fix_errors;       /* Modes.fix_errors (dump1090.c:167; default 1)        */
    int32_t  aggressive;       /* Modes.aggressive (dump1090.c:179; default 0)        */
    int32_t  keep_candidates;  /* also return every preamble position (for --stats)   */
    uint32_t run_chunks;       /* tuning: 512-sample chunks per wavefront run, 0=auto */
    uint32_t slot_cap;         /* tuning: forwarded positions per run, 0=auto         */
    uint32_t max_records;      /* record-list capacity; 0 = automatic (starts at 1<<18 and grows)  */
    uint32_t scan_variant;     /* must be 0 (1 was the single-pass first version of the scan kernel, removed in round 4: MODES_ERR_ARG) */
    uint32_t overlap;          /* 0: all kernels in order on the caller's stream.  1: only the scan kernel runs
                                  there; the demod and order kernels follow on the context's own stream, so work
                                  the caller queues next (another context's scan) may overlap them.  2: only the
                                  order kernel leaves the caller's stream (it fits next to anything).          */
    uint32_t flags;            /* MODES_GPU_* below                                                       */
    uint32_t direct_records;   /* lists of at most this many records reach the host with the kernels
                                  (zero-copy stores, no copy operation); 0 = 4096                          */
    uint32_t demod_variant;    /* 0 = automatic (production): one kernel - 8-wave workgroups with the whole 64 KiB magnitude
                                  table in LDS, every stage inline, records put in order behind it - unless the context's
                                  previous call left more than 4096 records per GiB: then two kernels, select (preamble test +
                                  noise-gate pre-test, 16 wavefronts at 64 VGPRs) and record (one wavefront per survivor,
                                  written straight to its final place: no staging list, no order kernel): 4 % faster where
                                  there are records, 4 % slower on pure noise.  3 = always the one kernel,
                                  2 = always the two (each is the other's cross-check in the parity tests); 1 was a
                                  small-table form of the one kernel, removed in round 3: MODES_ERR_ARG                    */
} modes_gpu_config;

/* modes_gpu_config.flags */
#define MODES_GPU_NO_RETRY 1u  /* never repeat a call inside modes_gpu_fetch(): a list overflow returns
                                  MODES_ERR_OVERFLOW (the lists are enlarged; the caller resubmits)      */
#define MODES_GPU_ORDER_IN_STREAM 2u  /* with modes_gpu_set_output: the kernel that puts a LONG list (more than
                                  direct_records) in order follows the detect in its stream, and so does an event -
                                  ~11 us of the stream per call - so that a device consumer queued behind the detect
                                  sees the complete list without a host round trip.  Default: that kernel runs
                                  inside modes_gpu_fetch_device() when a list turns out to be long */

/* One demodulation attempt at a preamble position: dump1090.c:1666-1726 (bit
 * slicing, packing, noise gate) plus the syndrome / repair lookup of
 * dump1090.c:1104,1112-1117,854-880.  Pure function of the magnitudes. */
typedef struct {
    uint8_t  msg[14];      /* packed bits as demodulated (before any repair)          */
    uint8_t  errors;       /* dump1090.c:1682 counter                                 */
    uint8_t  gate_ok;      /* 1 if mean |lo-hi| >= 2550 (dump1090.c:1723)             */
    uint8_t  nfix;         /* bits fixBitErrors would flip: 0, 1 or 2                 */
    uint8_t  fixpos[2];    /* their message-relative positions, 0xff if unused        */
    uint8_t  cls;          /* MODES_CLS_*: what decodeModesMessage() will make of this attempt (below); 0 = not classified */
    uint16_t slot;         /* ICAO-whitelist slot (dump1090.c:898-905) of the address a CLEAN / IID / AP attempt writes or asks for */
    uint8_t  pad[2];
    uint32_t syndrome;     /* modesChecksum() of msg (dump1090.c:733); 0 if !gate_ok  */
} modes_attempt;

/* modes_attempt.cls (ABI 5) - the per-message decisions of decodeModesMessage() that do not depend on the ICAO whitelist, made on
 * the GPU by the wavefront that demodulated the attempt (it holds the DF, the syndrome and the repair): which branch of
 * dump1090.c:1099-1128 / :1183-1210 / :1731 the attempt takes.  The host's --raw resolve (modes_host_resolve_raw*) is then: skip
 * window, this byte, at most one whitelist access through `slot`, and the hex line.  A host configured differently from the context
 * that wrote the byte (MODES_CLS_FIX / MODES_CLS_AGGRESSIVE disagree), or handed records without it (cls == 0: another producer),
 * classifies by itself - modes_classify() in dump1090_amd/csrc/modes_core.h is the one definition both sides compile. */
#define MODES_CLS_KIND       0x07u
#define MODES_CLS_GATE       1u   /* the noise gate failed: the position ends here (dump1090.c:1723-1726)                      */
#define MODES_CLS_SKIP       2u   /* too many slicing errors for this mode (dump1090.c:1731): not decoded, the retry follows   */
#define MODES_CLS_CLEAN      3u   /* DF11/17/18, syndrome 0: crcok, the address goes on the whitelist (dump1090.c:1198)        */
#define MODES_CLS_FIXED      4u   /* DF11/17/18 repaired by nfix bits (dump1090.c:1112-1128): crcok, no whitelist access       */
#define MODES_CLS_IID        5u   /* DF11, syndrome 1..79, no repair: crcok iff the whitelist knows msg[1..3] (dump1090.c:1204) */
#define MODES_CLS_AP         6u   /* DF0/4/5/16/20/21/24: crcok iff the whitelist knows the address = the syndrome (dump1090.c:942-983) */
#define MODES_CLS_BAD        7u   /* decoded, never crcok                                                                      */
#define MODES_CLS_LONG       0x08u  /* 112-bit message (the DF as demodulated, dump1090.c:1100)                                */
#define MODES_CLS_NOERR      0x10u  /* errors == 0 (the "demodulated with zero errors" counter, dump1090.c:1739)              */
#define MODES_CLS_FIX        0x20u  /* classified with fix_errors on ...                                                       */
#define MODES_CLS_AGGRESSIVE 0x40u  /* ... with aggressive on                                                                  */
#define MODES_CLS_VALID      0x80u  /* the byte has been written                                                               */

/* att[0]: samples as received.  att[1]: after applyPhaseCorrection
 * (dump1090.c:1498-1558; identical to att[0] when j == 0, dump1090.c:1660).
 * Emitted only for positions whose att[0].gate_ok == 1 (a failed first gate ends
 * the position in the reference: dump1090.c:1723-1726). */
typedef struct {
    uint32_t      block;   /* buffer index k                                          */
    uint32_t      j;       /* buffer-local offset                                     */
    modes_attempt att[2];
} modes_record;            /* 64 bytes                                                */

/* A device-resident piece of the sample stream and the buffers to demodulate. */
typedef struct {
    const void *iq;            /* DEVICE pointer (2-byte aligned): interleaved u8 I,Q  */
    uint64_t    nbytes;        /* valid bytes at iq; bytes outside read as 127        */
    uint64_t    stream_byte0;  /* offset of iq[0] in the whole stream (even)          */
    uint64_t    first_block;   /* first buffer index to demodulate                    */
    uint64_t    nblocks;       /* number of buffers                                   */
} modes_gpu_span;

typedef struct {
    const modes_record *records;       /* ascending (block, j); host memory owned by the context (fetch) or   */
    uint64_t            n_records;     /*   DEVICE memory (fetch_device); valid until the next detect         */
    const uint64_t     *candidates;    /* framed g of every preamble position,        */
    uint64_t            n_candidates;  /*   ascending (keep_candidates only)          */
    uint64_t            n_forwarded;   /* positions the s-domain scan forwarded       */
    uint64_t            n_preambles;   /* positions where dump1090.c:1602-1650 holds  */
    float               scan_ms;       /* HIP-event time of the scan kernel           */
    float               demod_ms;      /* of the demod kernel                         */
    float               order_ms;      /* of the third kernel of the call: the order kernel (records into stream order) on the
                                          one-kernel path, the RECORD kernel on the two-kernel path (demod_ms is then the select
                                          kernel's) - "the kernels behind the scan" add up to demod_ms + order_ms on either path */
    float               reserved;
} modes_gpu_result;

int  modes_gpu_create(const modes_gpu_config *cfg, modes_gpu **out);
void modes_gpu_destroy(modes_gpu *ctx);
/* Text of the last error on ctx (or of the last failed create when ctx == NULL). */
const char *modes_gpu_last_error(const modes_gpu *ctx);

/* Replaces computeMagnitudeVector() (dump1090.c:1454-1469): nsamples interleaved
 * I/Q byte pairs at d_iq -> nsamples u16 magnitudes at d_mag (both DEVICE).
 * stream: the hipStream_t to launch on (NULL = HIP's default stream). */
int modes_gpu_compute_magnitude(modes_gpu *ctx, const void *d_iq, uint64_t nsamples,
                                void *d_mag, void *stream);

/* Replaces computeMagnitudeVector()+detectModeS() up to (not including) the
 * stateful decode, for span->nblocks buffers: launches the scan, demod and order
 * kernels asynchronously on `stream` (a hipStream_t; NULL = HIP's default stream).
 * One detect per context at a time (MODES_ERR_STATE otherwise; use several contexts to pipeline).
 *
 * LIFETIME: span->iq and `stream` must stay valid, and the bytes unmodified, until the matching
 * modes_gpu_fetch()/modes_gpu_fetch_device() has RETURNED - in every mode.  When a call needs
 * longer lists than the context has (a run with more than 1/16 preamble-like positions; more records than
 * the automatic capacity), fetch enlarges them and REPEATS the call on the same span and stream; nothing
 * is ever truncated.  A host that cannot keep the input alive sets MODES_GPU_NO_RETRY: fetch then returns
 * MODES_ERR_OVERFLOW (result->n_records = records the call needs) with the lists already enlarged, and
 * the host resubmits the span itself. */
int modes_gpu_detect(modes_gpu *ctx, const modes_gpu_span *span, void *stream);

/* Waits for the last modes_gpu_detect() and returns its records in stream order (host memory).  The list
 * is put in order on the device; up to direct_records records arrive with the kernels, longer lists
 * take one device-to-host copy here. */
int modes_gpu_fetch(modes_gpu *ctx, modes_gpu_result *res);

/* The same without the copy: res->records is the DEVICE list (the context's, or the one given to
 * modes_gpu_set_output) - for hosts that hand the records to another device consumer, e.g. the gather of
 * the per-GPU lists over RCCL. */
int modes_gpu_fetch_device(modes_gpu *ctx, modes_gpu_result *res);

/* Makes `stream` (a hipStream_t) wait for the results of the detect in flight - the device list and count of
 * modes_gpu_set_output are complete for everything queued on `stream` afterwards.  Needed with overlap = 1,
 * where the caller's own stream does not wait for the demod and order kernels. */
int modes_gpu_stream_wait(modes_gpu *ctx, void *stream);

/* Kernel times in modes_gpu_result (scan_ms, demod_ms, order_ms): HIP events attached to the kernels of every detect
 * that follows (on: the default) or none (off: the times read 0).  The events cost ~9 us of idle GPU per kernel
 * boundary; a pipelined host switches them off, or on for a sample of its calls. */
int modes_gpu_set_timing(modes_gpu *ctx, int on);

/* Caller-owned device output: the ordered list is written to d_records (16-byte aligned, room for
 * `capacity` records; more records than that are MODES_ERR_OVERFLOW at fetch) and, if d_count is not
 * NULL, the number of records of the call to the 8-byte device word d_count.  The count and a list of up to
 * direct_records records are written by the kernels of modes_gpu_detect, in stream order; a longer list is complete
 * when modes_gpu_fetch_device() has returned (or in stream order too, with MODES_GPU_ORDER_IN_STREAM) - a host that
 * gathers the lists over RCCL queues its collectives after that call, with no event in the detect's stream.
 * (NULL, 0, NULL) returns to the context's own list (which keeps the capacity it has). */
int modes_gpu_set_output(modes_gpu *ctx, void *d_records, uint64_t capacity, void *d_count);

/* Host-buffer convenience used by the C host: stages `nbytes` stream bytes that
 * start at stream offset `stream_byte0` into the context's device buffer, then
 * detect + fetch for buffers [first_block, first_block+nblocks). */
int modes_gpu_demod_host(modes_gpu *ctx, const uint8_t *iq, uint64_t nbytes,
                         uint64_t stream_byte0, uint64_t first_block, uint64_t nblocks,
                         modes_gpu_result *res);

/* The two halves of modes_gpu_demod_host, for hosts that overlap reading with the GPU the way the
 * reference overlaps its reader thread with detectModeS (dump1090.c:2965-2990): submit copies the
 * bytes to the context's device buffer and queues the kernels, all asynchronously when `iq` is pinned
 * memory from modes_gpu_host_alloc; modes_gpu_fetch() later returns the records.  `iq` must stay
 * untouched until that fetch.  One submit per context at a time (use two contexts to double-buffer). */
int modes_gpu_submit_host(modes_gpu *ctx, const uint8_t *iq, uint64_t nbytes,
                          uint64_t stream_byte0, uint64_t first_block, uint64_t nblocks);
/* Pinned (page-locked) host memory for modes_gpu_submit_host. */
int  modes_gpu_host_alloc(modes_gpu *ctx, size_t nbytes, void **out);
void modes_gpu_host_free(modes_gpu *ctx, void *p);

/* Debug / test tap of the kernels' powers: min((I-127)^2 + (Q-127)^2, 32767) per sample (u16) for nsamples at d_iq,
 * computed on the device both ways the kernels do (packed multiply-add: the demod kernel's table index; byte dot
 * product: the scan kernel) - a sample on which the two differ reads 0xffff.  Ordering compares on s equal ordering
 * compares on the magnitude (the LUT is strictly monotone in s). */
int modes_gpu_compute_power(modes_gpu *ctx, const void *d_iq, uint64_t nsamples,
                            void *d_s, void *stream);

/* Debug / test tap of the demod kernel's magnitudes: the 32768-entry table (the reference's LUT by saturated power)
 * to d_lut and modes_mag_exact() of every index, computed on the device, to d_exact (32768 u16 each). */
int modes_gpu_debug_tables(modes_gpu *ctx, void *d_lut, void *d_exact, void *stream);

/* Synthetic-input generator (bench / tests): the integer noise of
 * tests/synth.py:noise_bytes, bytes [first_byte, first_byte+nbytes) -> d_out. */
int modes_gpu_synth_noise(modes_gpu *ctx, void *d_out, uint64_t first_byte, uint64_t nbytes,
                          uint64_t seed, uint32_t sigma_q16, void *stream);

/* Fills d_out[0..nbytes) with `value` (127 = no signal) - tail padding helper. */
int modes_gpu_fill(modes_gpu *ctx, void *d_out, uint64_t nbytes, uint8_t value, void *stream);

/* Measurement taps (bench.py; no counterpart in the reference).
 * modes_gpu_host_profile: host seconds spent inside modes_gpu_detect since creation / the last reset, by section -
 *   out[0] hipSetDevice, [1] geometry, list growth and parameter blocks, [2] scan launch, [3] demod launch,
 *   [4] finalize launch, [5] everything after it (order / prefix kernels, hipGetLastError, event records), [6] calls.
 * modes_gpu_stream_ceiling: the chip's read-only streaming rate over the nbytes at d_iq (16-byte aligned) with the scan
 *   kernel's own access pattern and cache policy and no arithmetic: `launches` kernels queued back to back on `stream`,
 *   one in `time_every` timed by events attached to its dispatch (the scan kernel's timing method); *avg_ms / *min_ms =
 *   average / shortest timed launch.  SURVEY.md 8d: the measured ceiling next to the 8 TB/s specification. */
int modes_gpu_host_profile(modes_gpu *ctx, double out[8], int reset);
int modes_gpu_stream_ceiling(modes_gpu *ctx, const void *d_iq, uint64_t nbytes, uint32_t launches, uint32_t time_every,
                             float *avg_ms, float *min_ms, void *stream);

/* ABI version of this header. */
#define MODES_GFX950_ABI 5
int modes_gpu_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
