/* modes_dropin.c - the gfx950 path bound into the reference's own main().
 *
 * Not a translation unit of its own: integration/dump1090_gfx950.patch #includes it into dump1090.c right
 * before main(), where `Modes`, struct modesMessage and useModesMessage() are in scope, and swaps the two
 * hot-path calls of the main loop (dump1090.c:2974, :2986) for the two functions below.  Everything else -
 * argument parsing, the reader thread, the sink (displayModesMessage, the network outputs, the interactive
 * list), --stats printing - is the reference's own code, untouched.
 *
 *     computeMagnitudeVector();                        ->  modesGpuDemod();      (still under data_mutex:
 *                                                          Modes.data belongs to the reader again afterwards)
 *     detectModeS(Modes.magnitude, Modes.data_len/2);  ->  modesGpuResolve();
 *
 * One 256 KiB buffer per GPU call keeps the reference's structure (and its latency); it is the wrong
 * granularity for throughput - dump1090_amd/csrc/main.cpp is the host for that.
 */
#define MODES_HOST_NO_MESSAGE_STRUCT        /* dump1090.c:211-260 is the definition in this translation unit */
#include "modes_gfx950.h"
#include "modes_host.h"
#include <stddef.h>

_Static_assert(sizeof(struct modesMessage) == MODES_MESSAGE_SIZE, "struct modesMessage: size differs from libmodes_host's");
_Static_assert(offsetof(struct modesMessage, flight) == MODES_MESSAGE_OFFSET_FLIGHT, "struct modesMessage: layout");
_Static_assert(offsetof(struct modesMessage, unit) == MODES_MESSAGE_OFFSET_UNIT, "struct modesMessage: layout");

static modes_gpu  *dropin_gpu;
static modes_host *dropin_host;
static modes_gpu_result dropin_res;
static uint64_t dropin_nbuf;                 /* buffers handed over so far */

/* once, after modesInit() (dump1090.c:2943) */
static void modesInitGpu(void) {
    modes_gpu_config gc;
    modes_host_config hc;
    memset(&gc, 0, sizeof(gc));
    gc.device = 0;
    gc.fix_errors = Modes.fix_errors;
    gc.aggressive = Modes.aggressive;
    gc.keep_candidates = Modes.stats;    /* --stats counts preambles whose first gate fails too (dump1090.c:1651) */
    hc.fix_errors = Modes.fix_errors;
    hc.aggressive = Modes.aggressive;
    hc.check_crc = Modes.check_crc;
    hc.reserved = 0;
    if (modes_gpu_create(&gc, &dropin_gpu) != MODES_OK) {
        fprintf(stderr, "GPU path: %s\n", modes_gpu_last_error(NULL));   /* no CPU fallback */
        exit(1);                                                          /* like dump1090.c:339-343 */
    }
    if ((dropin_host = modes_host_create(&hc)) == NULL) {
        fprintf(stderr, "GPU path: out of memory\n");
        exit(1);
    }
}

/* In place of computeMagnitudeVector(): Modes.data holds [476-byte carry | 262144 new bytes]
 * (dump1090.c:449-451, 481-483).  The carry of the very first buffer is the 127-fill of modesInit() -
 * "before the stream", which the library supplies itself. */
static void modesGpuDemod(void) {
    const unsigned char *p = dropin_nbuf ? Modes.data : Modes.data + (MODES_FULL_LEN - 1) * 4;
    const uint64_t carry = dropin_nbuf ? MODES_CARRY_BYTES : 0;
    const uint64_t stream0 = dropin_nbuf * (uint64_t)MODES_DATA_LEN - carry;
    if (modes_gpu_demod_host(dropin_gpu, p, carry + MODES_DATA_LEN, stream0, dropin_nbuf, 1, &dropin_res) != MODES_OK) {
        fprintf(stderr, "GPU path: %s\n", modes_gpu_last_error(dropin_gpu));
        exit(1);
    }
    dropin_nbuf++;
}

/* the reference's own sink; the callback's struct IS the reference's struct (asserted above) */
static void dropin_sink(const struct modesMessage *mm, uint32_t block, uint32_t j, void *user) {
    (void)block; (void)j; (void)user;
    useModesMessage((struct modesMessage *)mm);                          /* dump1090.c:1802 */
}

/* In place of detectModeS(): the stateful, in-order half, feeding useModesMessage() at the points and in the
 * order the reference does; the --stats counters land where the reference prints them from. */
static void modesGpuResolve(void) {
    modes_host_stats st;
    /* the ICAO whitelist's 60 s TTL (dump1090.c:913,924) runs on the caller's clock in libmodes_host: this host also
     * serves live input (RTL-SDR, --loop, a pipe), so it advances with the wall clock like the reference's does.
     * (Frames that arrive over the raw TCP input port go through the reference's own decodeModesMessage and its own
     * Modes.icao_cache - a second whitelist; INTEGRATION.md says what that means.) */
    modes_host_set_time(dropin_host, (int64_t)time(NULL));
    modes_host_resolve(dropin_host, dropin_res.records, dropin_res.n_records, dropin_res.candidates,
                       dropin_res.n_candidates, dropin_sink, NULL);
    modes_host_get_stats(dropin_host, &st);
    Modes.stat_valid_preamble = st.valid_preamble;
    Modes.stat_out_of_phase = st.out_of_phase;
    Modes.stat_demodulated = st.demodulated;
    Modes.stat_goodcrc = st.goodcrc;
    Modes.stat_badcrc = st.badcrc;
    Modes.stat_fixed = st.fixed;
    Modes.stat_single_bit_fix = st.single_bit_fix;
    Modes.stat_two_bits_fix = st.two_bits_fix;
}
