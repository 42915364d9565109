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
 * That is integration/dump1090_gfx950.patch: one 256 KiB buffer per GPU call keeps the reference's structure (and its
 * latency); it is the wrong granularity for throughput - 4,096 synchronous GPU calls per GiB.
 *
 * integration/dump1090_gfx950_batched.patch adds ONE more edit: the file reader (dump1090.c:460-512, called at :524) becomes
 * modesGpuReadFile() below, which hands over K x 262,144 bytes per round trip through data_mutex / data_cond instead of one
 * buffer (K = $MODES_DROPIN_BLOCKS, default 512 = 128 MiB; 1 with --interactive, which replays at the radio's pace).  The
 * library does the framing of a multi-buffer call itself (modes_gpu_submit_host(..., nblocks)), so modesGpuDemod() only passes
 * the count on (input that cannot seek - stdin, a FIFO - is handed over at the pace it delivers: whole buffers, when the batch is
 * full or $MODES_DROPIN_FLUSH_MS = 66 ms after it began, so that a radio's stream through a pipe is printed within two buffers, as the
 * reference does); two pinned buffers and two GPU contexts alternate, so that the reader fills one buffer while the GPU works on
 * the other and the main thread resolves the batch before; the resolve runs once per hand-off.  Live RTL-SDR input
 * (rtlsdrCallback, dump1090.c:442-456) is untouched and keeps one buffer per call.  The main loop (:2969-2990) is the same
 * four edits; the reference's EOF race (its loop usually drops the last buffer) does not exist on this path -
 * the reader raises Modes.exit only after the last hand-off has been consumed.
 */
#define MODES_HOST_NO_MESSAGE_STRUCT        /* dump1090.c:211-260 is the definition in this translation unit */
#include "modes_gfx950.h"
#include "modes_host.h"
#include <poll.h>
#include <stddef.h>
#include <time.h>

_Static_assert(sizeof(struct modesMessage) == MODES_MESSAGE_SIZE, "struct modesMessage: size differs from libmodes_host's");
_Static_assert(offsetof(struct modesMessage, flight) == MODES_MESSAGE_OFFSET_FLIGHT, "struct modesMessage: layout");
_Static_assert(offsetof(struct modesMessage, unit) == MODES_MESSAGE_OFFSET_UNIT, "struct modesMessage: layout");

static modes_gpu  *dropin_gpu;
static modes_host *dropin_host;
static modes_gpu_result dropin_res;
static uint64_t dropin_nbuf;                 /* buffers handed over so far */

/* ---- batched file input (integration/dump1090_gfx950_batched.patch) ---- */
static int            dropin_batched;        /* modesGpuReadFile() is the reader of this run                              */
static uint64_t       dropin_k = 512;        /* buffers per hand-off                                                      */
static modes_gpu     *dropin_ctx[2];         /* [0] = dropin_gpu                                                          */
static unsigned char *dropin_buf[2];         /* pinned: 476-byte carry + K x 262144 bytes                                 */
static struct {                              /* the hand-off, written by the reader under data_mutex before data_ready = 1 */
    int      which;                          /* buffer / context of this batch                                            */
    size_t   carry, fresh;                   /* valid carry bytes in front (0 for the first batch), new bytes behind them  */
    uint64_t first_block;
    int      last;                           /* the stream ends with this batch (it carries the EOF buffer)               */
    int      empty;                          /* nothing to do: the wake-up call that follows the last batch               */
} dropin_hand;
static int dropin_inflight = -1;             /* context whose batch has been submitted but not fetched                    */
static modes_gpu_result dropin_res_last;     /* the last batch's records, fetched together with its predecessor's         */
static int dropin_have[2];                   /* dropin_res / dropin_res_last hold a batch to resolve                      */
/* $MODES_DROPIN_TIMING: one JSON line on stderr when the last batch has been resolved - how long the GPU start-up took and how
 * long the stream itself (first read .. last message), which is what a long-running host sees (tools/dropin_rate.py) */
static double dropin_t0, dropin_t_init, dropin_t_first_read;
static uint64_t dropin_bytes;
static double dropin_now(void) {                                          /* (monotonic: test runs pin time() and gettimeofday()) */
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* once, after modesInit() (dump1090.c:2943) */
static void modesInitGpu(void) {
    modes_gpu_config gc;
    modes_host_config hc;
    dropin_t0 = dropin_now();
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
    dropin_ctx[0] = dropin_gpu;
    dropin_t_init = dropin_now();
}

static void dropin_die(const char *what, modes_gpu *g) {
    fprintf(stderr, "GPU path: %s: %s\n", what, modes_gpu_last_error(g));
    exit(1);                                                              /* like dump1090.c:339-343 */
}

/* OPT-IN ($MODES_DROPIN_FAST_EXIT=1): a file run lives half a second, and what main()'s `return 0` would still
 * do - the HIP runtime's exit handlers, unpinning and freeing the buffers one by one (~0.1 s) - the kernel does for a dead
 * process anyway.  Registered AFTER the runtime's own handlers (so it runs before them), keeps the exit status, flushes
 * stdio first.  It is a process-wide side effect - every atexit / on_exit handler registered earlier, by the hosting program or
 * its libraries, is skipped, on error exits too - so a file that is #included into someone else's main() does not do it
 * unasked: the default is the orderly exit. */
static void dropin_fast_exit(int status, void *arg) {
    (void)arg;
    fflush(NULL);
    _exit(status);
}

/* K, the second context, the two pinned buffers - by the reader thread, before its first read */
static void dropin_setup_batched(void) {
    modes_gpu_config gc;
    const char *k = getenv("MODES_DROPIN_BLOCKS");
    int i;
    if (k && atoll(k) > 0) dropin_k = (uint64_t)atoll(k);
    if (Modes.interactive) dropin_k = 1;                                  /* replayed at the radio's pace, dump1090.c:470-476 */
    if (dropin_k > 16384) dropin_k = 16384;                               /* 4 GiB per call */
    memset(&gc, 0, sizeof(gc));
    gc.device = 0;
    gc.fix_errors = Modes.fix_errors;
    gc.aggressive = Modes.aggressive;
    gc.keep_candidates = Modes.stats;
    if (modes_gpu_create(&gc, &dropin_ctx[1]) != MODES_OK) dropin_die("second context", NULL);
    for (i = 0; i < 2; i++) {
        void *p = NULL;
        if (modes_gpu_host_alloc(dropin_ctx[i], MODES_CARRY_BYTES + dropin_k * (size_t)MODES_DATA_LEN, &p) != MODES_OK)
            dropin_die("pinned buffer", dropin_ctx[i]);
        dropin_buf[i] = p;
        modes_gpu_set_timing(dropin_ctx[i], 0);                           /* no timing events between the kernels */
    }
    if (getenv("MODES_DROPIN_FAST_EXIT")) on_exit(dropin_fast_exit, NULL);
    dropin_batched = 1;
}

/* A regular file is read by several threads at once (pread on disjoint slices): one thread copying out of the page
 * cache is an order of magnitude slower than the PCIe link behind it. */
#define DROPIN_READERS 16
struct dropin_slice { int fd; unsigned char *dst; size_t want, got; off_t at; };
static void *dropin_read_slice(void *arg) {
    struct dropin_slice *s = arg;
    while (s->got < s->want) {
        ssize_t n = pread(s->fd, s->dst + s->got, s->want - s->got, s->at + (off_t)s->got);
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) break;
        s->got += (size_t)n;
    }
    return NULL;
}
/* -> bytes read (short only at end of file) */
static size_t dropin_read_parallel(int fd, off_t at, unsigned char *dst, size_t want) {
    struct dropin_slice sl[DROPIN_READERS];
    pthread_t th[DROPIN_READERS];
    size_t per = (want / DROPIN_READERS + 4095) & ~(size_t)4095, got = 0;
    int i, n = 0;
    for (i = 0; i < DROPIN_READERS && (size_t)i * per < want; i++, n++) {
        sl[i].fd = fd; sl[i].dst = dst + (size_t)i * per; sl[i].at = at + (off_t)((size_t)i * per); sl[i].got = 0;
        sl[i].want = want - (size_t)i * per < per ? want - (size_t)i * per : per;
        if (i > 0 && pthread_create(&th[i], NULL, dropin_read_slice, &sl[i]) != 0) { perror("pthread_create"); exit(1); }
    }
    dropin_read_slice(&sl[0]);
    for (i = 1; i < n; i++) pthread_join(th[i], NULL);
    for (i = 0; i < n; i++) {
        got += sl[i].got;
        if (sl[i].got < sl[i].want) break;                                /* end of file inside this slice */
    }
    return got;
}

/* In place of readDataFromFile() (dump1090.c:460-512; batched patch): the same protocol - fill under data_mutex, publish with
 * data_ready = 1, wait until the main thread has taken the hand-off - with K buffers per round trip.  At the end of the
 * file the stream gets its EOF buffer (dump1090.c:499-504 pads the short read with 127: the library reads bytes outside the
 * span as 127) and Modes.exit is raised only once that hand-off has been consumed. */
#if defined(__GNUC__)
__attribute__((unused))                                                   /* only the batched patch calls it */
#endif
static void modesGpuReadFile(void) {
    unsigned char tail[MODES_CARRY_BYTES];
    size_t carry = 0;
    uint64_t first_block = 0;
    off_t pos = 0;
    int w = 0, last = 0, seekable, paced;
    static unsigned char pendbuf[MODES_DATA_LEN];                         /* read beyond the last whole buffer of a paced hand-off */
    size_t pend = 0;
    double dropin_flush_s = 0.066;
    dropin_setup_batched();
    dropin_t_first_read = dropin_now();
    seekable = Modes.fd != STDIN_FILENO && !Modes.loop && lseek(Modes.fd, 0, SEEK_CUR) != (off_t)-1;
    paced = lseek(Modes.fd, 0, SEEK_CUR) == (off_t)-1;                    /* a pipe, a FIFO, a socket */
    if (getenv("MODES_DROPIN_FLUSH_MS")) dropin_flush_s = atof(getenv("MODES_DROPIN_FLUSH_MS")) * 1e-3;
    pthread_mutex_lock(&Modes.data_mutex);
    while (!last) {
        const size_t batch = dropin_k * (size_t)MODES_DATA_LEN;
        unsigned char *p;
        size_t got = 0;
        if (Modes.data_ready) {
            pthread_cond_wait(&Modes.data_cond, &Modes.data_mutex);
            continue;
        }
        if (Modes.interactive) {                                          /* dump1090.c:470-476 */
            pthread_mutex_unlock(&Modes.data_mutex);
            usleep(5000);
            pthread_mutex_lock(&Modes.data_mutex);
        }
        p = dropin_buf[w];
        if (carry) memcpy(p, tail, MODES_CARRY_BYTES);                    /* dump1090.c:481 */
        if (seekable) {
            got = dropin_read_parallel(Modes.fd, pos, p + carry, batch);
            pos += (off_t)got;
            last = got < batch;
        } else if (paced) {
            /* Input that cannot seek (stdin, a FIFO) is handed over at the pace it delivers - the reference's reader publishes every
             * buffer as it fills (dump1090.c:460-512) and a radio at 2 Msps fills one in 65.5 ms: waiting for K = 512 of them would be
             * 33 s of silence.  So the hand-off is what HAS arrived - whole buffers - when the batch is full or $MODES_DROPIN_FLUSH_MS
             * (default 66) after it began; until one whole buffer is there it waits.  The bytes read beyond the last whole buffer open
             * the next hand-off.  A fast pipe (cat file |) still fills its batches. */
            const double t0 = dropin_now();
            int ended = 0;
            if (pend) memcpy(p + carry, pendbuf, pend);
            got = pend;
            pend = 0;
            while (got < batch) {                                             /* (under data_mutex, like the reference's read, dump1090.c:484) */
                struct pollfd pf;
                int timeout = -1, pr;
                ssize_t n;
                if (got >= MODES_DATA_LEN) {
                    const double left = t0 + dropin_flush_s - dropin_now();
                    if (left <= 0) break;
                    timeout = (int)(left * 1e3) + 1;
                }
                pf.fd = Modes.fd; pf.events = POLLIN; pf.revents = 0;
                pr = poll(&pf, 1, timeout);
                if (pr < 0 && errno == EINTR) continue;
                if (pr <= 0) break;
                n = read(Modes.fd, p + carry + got, batch - got);
                if (n < 0 && (errno == EINTR || errno == EAGAIN)) continue;
                if (n <= 0) { ended = 1; break; }
                got += (size_t)n;
            }
            if (!ended && got < batch) {                                      /* the deadline: whole buffers go, the rest waits */
                pend = got % MODES_DATA_LEN;
                got -= pend;
                memcpy(pendbuf, p + carry + got, pend);
            }
            last = ended;
        } else {
            while (got < batch) {
                ssize_t n = read(Modes.fd, p + carry + got, batch - got);
                if (n == 0 && Modes.filename != NULL && Modes.fd != STDIN_FILENO && Modes.loop) {   /* dump1090.c:488-494 */
                    if (lseek(Modes.fd, 0, SEEK_SET) != -1) continue;
                }
                if (n < 0 && errno == EINTR) continue;
                if (n <= 0) break;
                got += (size_t)n;
            }
            last = got < batch;
        }
        dropin_bytes += got;
        dropin_hand.which = w;
        dropin_hand.carry = carry;
        dropin_hand.fresh = got;
        dropin_hand.first_block = first_block;
        dropin_hand.last = last;
        dropin_hand.empty = 0;
        if (!last) {
            memcpy(tail, p + carry + got - MODES_CARRY_BYTES, MODES_CARRY_BYTES);
            carry = MODES_CARRY_BYTES;
            first_block += got / MODES_DATA_LEN;
            w ^= 1;
        }
        Modes.data_ready = 1;
        pthread_cond_signal(&Modes.data_cond);
    }
    /* the last hand-off is out; when the main thread has taken it, end the main loop (dump1090.c:2989) - with one more,
     * empty hand-off in case it already waits for data again */
    while (Modes.data_ready) pthread_cond_wait(&Modes.data_cond, &Modes.data_mutex);
    Modes.exit = 1;
    dropin_hand.empty = 1;
    Modes.data_ready = 1;
    pthread_cond_signal(&Modes.data_cond);
    pthread_mutex_unlock(&Modes.data_mutex);
}

/* In place of computeMagnitudeVector(): Modes.data holds [476-byte carry | 262144 new bytes]
 * (dump1090.c:449-451, 481-483).  The carry of the very first buffer is the 127-fill of modesInit() -
 * "before the stream", which the library supplies itself. */
static void modesGpuDemod(void) {
    if (dropin_batched) {
        /* The hand-off names a pinned buffer with K buffers' worth of stream: queue it (H2D + kernels, asynchronous) on its
         * own context, then wait for the batch BEFORE it - its buffer is the one the reader fills next, and its records are
         * what modesGpuResolve() works on while the GPU runs this one.  The last batch is fetched right away too. */
        const int w = dropin_hand.which;
        dropin_have[0] = dropin_have[1] = 0;
        if (dropin_hand.empty) return;
        {
            const uint64_t nblocks = dropin_hand.fresh / MODES_DATA_LEN + (dropin_hand.last ? 1 : 0);
            const uint64_t stream0 = dropin_hand.first_block * (uint64_t)MODES_DATA_LEN - dropin_hand.carry;
            if (modes_gpu_submit_host(dropin_ctx[w], dropin_buf[w], dropin_hand.carry + dropin_hand.fresh, stream0,
                                      dropin_hand.first_block, nblocks) != MODES_OK)
                dropin_die("submit", dropin_ctx[w]);
        }
        if (dropin_inflight >= 0) {
            if (modes_gpu_fetch(dropin_ctx[dropin_inflight], &dropin_res) != MODES_OK) dropin_die("fetch", dropin_ctx[dropin_inflight]);
            dropin_have[0] = 1;
        }
        dropin_inflight = w;
        if (dropin_hand.last) {
            if (modes_gpu_fetch(dropin_ctx[w], &dropin_res_last) != MODES_OK) dropin_die("fetch", dropin_ctx[w]);
            dropin_have[1] = 1;
            dropin_inflight = -1;
        }
        return;
    }
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
    if (dropin_batched) {                                                /* the batch before the one the GPU works on, then - at the end - that one */
        if (dropin_have[0])
            modes_host_resolve(dropin_host, dropin_res.records, dropin_res.n_records, dropin_res.candidates, dropin_res.n_candidates,
                               dropin_sink, NULL);
        if (dropin_have[1]) {
            modes_host_resolve(dropin_host, dropin_res_last.records, dropin_res_last.n_records, dropin_res_last.candidates,
                               dropin_res_last.n_candidates, dropin_sink, NULL);
            if (getenv("MODES_DROPIN_TIMING")) {
                const double t = dropin_now(), stream = t - dropin_t_first_read;
                fprintf(stderr, "{\"bytes\": %llu, \"blocks_per_handoff\": %llu, \"gpu_init_s\": %.4f, \"reader_setup_s\": %.4f, \"stream_s\": %.4f, "
                                "\"stream_GBps\": %.2f, \"stream_Msamples_per_s\": %.1f}\n",
                        (unsigned long long)dropin_bytes, (unsigned long long)dropin_k, dropin_t_init - dropin_t0, dropin_t_first_read - dropin_t_init,
                        stream, stream > 0 ? (double)dropin_bytes / stream / 1e9 : 0.0, stream > 0 ? (double)dropin_bytes / 2.0 / stream / 1e6 : 0.0);
            }
        }
    } else
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
