/* modes_host.h - C ABI of libmodes_host.so: the host half of the hot path.
 *
 * What stays sequential in the reference stays on the CPU here: the skip-after-
 * good-message rule, the retry-with-phase-correction state machine
 * (dump1090.c:1568,1597,1723-1726,1731-1792), decodeModesMessage()'s CRC / repair /
 * ICAO-whitelist decisions (dump1090.c:1091-1310, 896-983) and the sink
 * useModesMessage() (dump1090.c:1802-1820).  Input: the modes_record list produced
 * by libmodes_gfx950.so (modes_gfx950.h).  Output: `struct modesMessage` - same
 * field names and types as dump1090.c:211-260 - handed to a callback in stream
 * order, exactly where the reference calls useModesMessage(&mm).
 *
 * Pure host code: loads and runs without a GPU.
 */
#ifndef MODES_HOST_H
#define MODES_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "modes_gfx950.h"

#ifdef __cplusplus
extern "C" {
#endif

/* (guarded: a translation unit that is dump1090.c itself - integration/modes_dropin.c - already has them) */
#ifndef MODES_LONG_MSG_BITS
#define MODES_LONG_MSG_BITS   112
#define MODES_SHORT_MSG_BITS  56
#define MODES_LONG_MSG_BYTES  (112 / 8)
#define MODES_SHORT_MSG_BYTES (56 / 8)
#endif
#ifndef MODES_UNIT_FEET
#define MODES_UNIT_FEET   0
#define MODES_UNIT_METERS 1
#endif

/* Layout of struct modesMessage below (LP64): a host that brings its own definition - the reference's,
 * MODES_HOST_NO_MESSAGE_STRUCT - checks these with _Static_assert (integration/modes_dropin.c). */
#define MODES_MESSAGE_SIZE          180
#define MODES_MESSAGE_OFFSET_FLIGHT 96
#define MODES_MESSAGE_OFFSET_UNIT   176

#ifndef MODES_HOST_NO_MESSAGE_STRUCT
/* dump1090.c:211-260, field for field. */
struct modesMessage {
    /* Generic fields */
    unsigned char msg[MODES_LONG_MSG_BYTES]; /* Binary message. */
    int msgbits;                /* Number of bits in message */
    int msgtype;                /* Downlink format # */
    int crcok;                  /* True if CRC was valid */
    uint32_t crc;               /* Message CRC */
    int errorbit;               /* Bit corrected. -1 if no bit corrected. */
    int aa1, aa2, aa3;          /* ICAO Address bytes 1 2 and 3 */
    int phase_corrected;        /* True if phase correction was applied. */

    /* DF 11 */
    int ca;                     /* Responder capabilities. */
    int iid;                    /* Interrogator Identifier (IID). */

    /* DF 17, 18 */
    int metype;                 /* Extended squitter message type. */
    int mesub;                  /* Extended squitter message subtype. */
    int heading_is_valid;
    int heading;
    int aircraft_type;
    int fflag;                  /* 1 = Odd, 0 = Even CPR message. */
    int tflag;                  /* UTC synchronized? */
    int raw_latitude;           /* Non decoded latitude */
    int raw_longitude;          /* Non decoded longitude */
    char flight[9];             /* 8 chars flight number. */
    int ew_dir;                 /* 0 = East, 1 = West. */
    int ew_velocity;            /* E/W velocity. */
    int ns_dir;                 /* 0 = North, 1 = South. */
    int ns_velocity;            /* N/S velocity. */
    int vert_rate_source;       /* Vertical rate source. */
    int vert_rate_sign;         /* Vertical rate sign. */
    int vert_rate;              /* Vertical rate. */
    int velocity;               /* Computed from EW and NS velocity. */

    /* DF 17, 18: Surface position (metype 5-8). */
    int movement;
    int movement_valid;
    int ground_track;
    int ground_track_valid;

    /* DF4, DF5, DF20, DF21 */
    int fs;                     /* Flight status for DF4,5,20,21 */
    int dr;                     /* Request extraction of downlink request. */
    int um;                     /* Request extraction of downlink request. */
    int identity;               /* 13 bits identity (Squawk). */

    /* Fields used by multiple message types. */
    int altitude, unit;
};
#endif /* MODES_HOST_NO_MESSAGE_STRUCT */

/* The flags of the reference's global `Modes` the path reads
 * (dump1090.c:167,168,179; defaults dump1090.c:305,306,315). */
typedef struct {
    int32_t fix_errors;
    int32_t aggressive;
    int32_t check_crc;
    int32_t reserved;
} modes_host_config;

/* dump1090.c:186-195 (the counters --stats prints, dump1090.c:2993-3006).
 * valid_preamble needs the full candidate list (keep_candidates); it is -1 when
 * the resolve ran on records only. */
typedef struct {
    int64_t valid_preamble;
    int64_t out_of_phase;
    int64_t demodulated;
    int64_t goodcrc;
    int64_t badcrc;
    int64_t fixed;
    int64_t single_bit_fix;
    int64_t two_bits_fix;
} modes_host_stats;

/* Where the reference calls useModesMessage(&mm) (dump1090.c:1777).  Called for
 * every attempt that reaches the decoder; `mm` lives on the caller's stack, do
 * not retain it (same contract as dump1090.c:1732).  block/j locate the frame. */
typedef void (*modes_sink_fn)(const struct modesMessage *mm, uint32_t block, uint32_t j, void *user);

typedef struct modes_host modes_host;   /* resolve state: config, ICAO cache, stats */

modes_host *modes_host_create(const modes_host_config *cfg);
void        modes_host_destroy(modes_host *h);
/* The clock behind the ICAO whitelist's 60 s TTL (the reference reads time(NULL), dump1090.c:913,924).
 * Never called: the clock stands still and nothing expires inside a run (file input; what the
 * parity tests pin).  A host on a live stream calls it with its wall clock before every resolve. */
void        modes_host_set_time(modes_host *h, int64_t now_seconds);

/* Sequential in-order resolve of one batch of records (ascending (block, j), as
 * modes_gpu_fetch returns them).  `candidates` (framed g, ascending) is optional:
 * when given, positions without a record (first noise gate failed) still count
 * as valid preambles outside a skip window, which --stats needs.  The sink is
 * called regardless of crcok, like dump1090.c:1777; use modes_host_wants() for
 * the reference's display filter.  Returns number of sink calls. */
uint64_t modes_host_resolve(modes_host *h, const modes_record *recs, uint64_t nrecs,
                            const uint64_t *candidates, uint64_t ncand,
                            modes_sink_fn sink, void *user);

/* Same resolve, but instead of a callback the messages that pass the display
 * filter (modes_host_wants) are stored, in order, in out[0..cap).  Returns the
 * number of such messages (may exceed cap; only cap are stored). */
typedef struct {
    struct modesMessage mm;
    uint32_t block;
    uint32_t j;
} modes_emitted;
uint64_t modes_host_resolve_to_array(modes_host *h, const modes_record *recs, uint64_t nrecs,
                                     const uint64_t *candidates, uint64_t ncand,
                                     modes_emitted *out, uint64_t cap);

/* Same resolve with the CLI's --raw sink built in: the listing `dump1090 --raw` prints for these buffers
 * (one "*<hex>;\n" line per message that passes the display filter, dump1090.c:1324-1326, :1803) is written
 * to out[0..cap); *nbytes = length of the whole listing (may exceed cap; then only whole lines that fit are
 * stored).  Returns the number of lines. */
uint64_t modes_host_resolve_raw(modes_host *h, const modes_record *recs, uint64_t nrecs,
                                const uint64_t *candidates, uint64_t ncand,
                                char *out, uint64_t cap, uint64_t *nbytes);

/* How the --raw resolvers above and below work since ABI 5 of modes_gfx950.h (the LEAN resolve): the per-message decisions of
 * decodeModesMessage() that do not need the whitelist arrive in modes_attempt.cls / .slot, made by the GPU wavefront that
 * demodulated the attempt; the host keeps the skip window (dump1090.c:1770), answers at most one whitelist question per attempt
 * (dump1090.c:1198, :1204, :942-983) and writes the line.  Records without the byte (cls == 0) or classified for another
 * configuration are classified here, by the same function.  modes_host_resolve() with a sink is the general form. */

/* CPUs this process may use at once: the smallest of online CPUs, the affinity mask and the cgroup CPU quota (cpu.max /
 * cfs_quota_us, rounded up) - what a pool of threads has to be sized by (a container on a 256-thread host may be held to 16).
 * The multi-threaded resolvers below never run more pieces than this at a time. */
int modes_host_cpu_budget(void);

/* modes_attempt.cls / .slot (include/modes_gfx950.h MODES_CLS_*) for records that do not carry them - another producer's, a
 * capture replayed from disk - written in place, for the configuration `cfg`: what the kernels write. */
void modes_host_classify(const modes_host_config *cfg, modes_record *recs, uint64_t nrecs);

/* The same listing (and the same whitelist / counter updates) computed by up to `threads` threads: the batch is cut at
 * buffer boundaries, the pieces are resolved speculatively and confirmed in order (modes_host.cpp) - byte-identical to
 * modes_host_resolve_raw without candidates.  For hosts whose one resolve thread would be the bottleneck: a
 * message-dense stream, or rank 0 of an N-GPU run that resolves N GPUs' records.  (threads < 0: exactly -threads
 * pieces however short the list is - for tests; otherwise a thread gets at least 16384 records and there are never more threads than
 * modes_host_cpu_budget().  The worker threads are created once per process.) */
uint64_t modes_host_resolve_raw_mt(modes_host *h, const modes_record *recs, uint64_t nrecs,
                                   char *out, uint64_t cap, uint64_t *nbytes, int threads);

/* The same for a batch that lies in several arrays - segs[0], segs[1], ... in stream order, each a list of WHOLE
 * buffers (e.g. the lists of N ranks and several calls as they sit in the gather buffers): one parallel resolve over
 * all of them, no concatenation copy.  With more segments than 64 pieces allow it falls back to one call per segment. */
uint64_t modes_host_resolve_raw_mtv(modes_host *h, const modes_record *const *segs, const uint64_t *seg_nrecs, uint32_t nsegs,
                                    char *out, uint64_t cap, uint64_t *nbytes, int threads);

/* The same multi-threaded resolve WITHOUT the gathering copy: the listing stays where its pieces were written, pieces[0 .. *npieces)
 * name them in stream order (at most 65; memory of the library, valid until the CALLING THREAD's next modes_host_resolve_raw_mt* /
 * _pieces call) - for fwrite / writev in order.  *nbytes = length of the whole listing.  Returns the number of lines. */
typedef struct {
    const char *base;
    uint64_t    len;
} modes_text_piece;
uint64_t modes_host_resolve_raw_pieces(modes_host *h, const modes_record *const *segs, const uint64_t *seg_nrecs, uint32_t nsegs,
                                       modes_text_piece *pieces, uint32_t piece_cap, uint32_t *npieces, uint64_t *nbytes, int threads);

/* ---- resolve on the ranks that demodulated ------------------------------------------------------
 * An N-GPU host whose rank 0 would otherwise resolve every rank's records (dump1090.c:896-925, :1183-1210: the
 * whitelist is the one piece of state that crosses buffers) can leave each list where it is: a rank resolves its
 * own records from a GUESSED whitelist - the state the batch started from plus what the ranks before it would
 * write (modes_host_whitelist_guess, exchanged) -, reports what it wrote and which answers it took from the guess
 * (modes_host_resolve_raw_spec), and the ranks confirm each other in stream order (modes_host_whitelist_check
 * against the state the ranks before really left); a rank with a wrong answer resolves again from that state.
 * The listing is the concatenation of the ranks' texts - byte-identical to the sequential resolve, by the argument
 * of modes_host_resolve_raw_mt.  dump1090_amd/distributed.py (RankResolve) is the protocol around these calls. */
#define MODES_ICAO_SLOTS 1024u          /* MODES_ICAO_CACHE_LEN, dump1090.c:65 */
#define MODES_ICAO_NONE 0xFFFFFFFFu     /* "no address" in a guess table (0 is an address a frame can carry) */
typedef struct {
    uint32_t addr;                      /* the address asked for (never 0: dump1090.c:920 answers that without the table) */
    uint32_t known;                     /* 1: the whitelist said yes */
} modes_icao_lookup;

/* The whitelist itself: addr[s] / seen[s] for the MODES_ICAO_SLOTS slots (dump1090.c:896-925). */
void modes_host_get_whitelist(const modes_host *h, uint32_t *addr, int64_t *seen);
void modes_host_set_whitelist(modes_host *h, const uint32_t *addr, const int64_t *seen);

/* guess[s] = the address the records' clean DF11/17/18 frames would leave in slot s (the last one in stream order),
 * MODES_ICAO_NONE where none would write.  A guess: whether such a frame is decoded at all depends on skip windows.
 * Uses up to `threads` threads (the lists are only read). */
void modes_host_whitelist_guess(const modes_host *h, const modes_record *const *segs, const uint64_t *seg_nrecs, uint32_t nsegs,
                                uint32_t *guess, int threads);

/* modes_host_resolve_raw_mtv, and besides: written[s] = 1 for every slot the resolve wrote, and lookups[0 .. *nlookups)
 * = every whitelist question that was answered from the state `h` had on entry (not from a slot written since), in
 * order.  *nlookups may exceed lookup_cap (then only lookup_cap are stored; 2 per record + 16 always suffice). */
uint64_t modes_host_resolve_raw_spec(modes_host *h, const modes_record *const *segs, const uint64_t *seg_nrecs, uint32_t nsegs,
                                     char *out, uint64_t cap, uint64_t *nbytes, int threads,
                                     uint8_t *written, modes_icao_lookup *lookups, uint64_t lookup_cap, uint64_t *nlookups);

/* modes_host_resolve (any sink; candidates for --stats) with the same log: for the sinks the lean --raw resolve does not serve
 * (--stats: the counters of dump1090.c:2993-3006 are sums over the ranks; --onlyaddr, --raw-net: the sink formats the line).  One
 * thread.  A resolve that has to be repeated must not count twice: modes_host_get_stats before, modes_host_set_stats to go back. */
uint64_t modes_host_resolve_spec(modes_host *h, const modes_record *recs, uint64_t nrecs, const uint64_t *candidates, uint64_t ncand,
                                 modes_sink_fn sink, void *user, uint8_t *written, modes_icao_lookup *lookups, uint64_t lookup_cap,
                                 uint64_t *nlookups);
void modes_host_set_stats(modes_host *h, const modes_host_stats *st);

/* 1 when h's whitelist (at h's clock) gives every one of the logged answers, else 0. */
int modes_host_whitelist_check(const modes_host *h, const modes_icao_lookup *lookups, uint64_t n);

/* dump1090.c:1803: would useModesMessage() display/forward this message? */
int modes_host_wants(const modes_host *h, const struct modesMessage *mm);

void modes_host_get_stats(const modes_host *h, modes_host_stats *out);

/* decodeModesMessage() (dump1090.c:1091-1310) on a demodulated frame, with the
 * repair decision already made on the GPU (att->nfix / fixpos).  Updates the
 * ICAO whitelist and the repair counters exactly like the reference. */
void modes_host_decode(modes_host *h, const modes_attempt *att, struct modesMessage *mm);

/* The same for a frame that did not come through the GPU - 14 bytes as the reference's raw TCP
 * input hands them to decodeModesMessage (decodeHexMessage, dump1090.c:2472-2502): syndrome and
 * repair lookup happen here, with the context's fix_errors / aggressive. */
void modes_host_decode_frame(modes_host *h, const unsigned char *frame, struct modesMessage *mm);

/* Formatting of the sink's two machine-readable modes.  buf >= 40 bytes.
 *   raw:      "*<hex>;\n"   (dump1090.c:1324-1326)
 *   onlyaddr: "%02x%02x%02x\n" (dump1090.c:1319)
 * Return the number of characters written. */
int modes_format_raw(const struct modesMessage *mm, char *buf);
/* The line the reference writes to its raw-output TCP clients (port 30002; modesSendRawOutput,
 * dump1090.c:2381-2393): the same frame in UPPER-case hex.  The sockets themselves are out of scope. */
int modes_format_raw_net(const struct modesMessage *mm, char *buf);
int modes_format_onlyaddr(const struct modesMessage *mm, char *buf);
/* The verbose dump of one message - what the reference prints when neither --raw nor --onlyaddr is
 * given: displayModesMessage() (dump1090.c:1314-1450) plus the blank line of useModesMessage()
 * (dump1090.c:1814).  check_crc = Modes.check_crc (only the "DF %d with good CRC" line reads it).
 * At most cap-1 characters are written; 1024 bytes always suffice. */
int modes_format_verbose(const struct modesMessage *mm, int check_crc, char *buf, size_t cap);
/* The 9-line --stats summary (dump1090.c:2994-3005); buf >= 512 bytes. */
int modes_format_stats(const modes_host_stats *st, char *buf);

/* ---- aircraft table, CPR positions and the BaseStation (SBS) sink ------------------------------
 * What the reference keeps behind useModesMessage() while an SBS (port 30003) or HTTP client is
 * connected (dump1090.c:1806-1808).  Sequential host state like the ICAO whitelist; the sockets
 * themselves, the interactive screen and the web map stay out of scope. */

/* struct aircraft (dump1090.c:107-130), the list link replaced by the tracker's own index. */
typedef struct {
    uint32_t addr;              /* ICAO address */
    char hexaddr[7];            /* printable ICAO address */
    char flight[9];             /* flight number */
    int altitude;               /* altitude */
    int speed;                  /* velocity computed from EW and NS components */
    int track;                  /* angle of flight */
    int odd_cprlat, odd_cprlon; /* encoded latitude / longitude of the last odd and even CPR frame */
    int even_cprlat, even_cprlon;
    double lat, lon;            /* coordinates obtained from CPR encoded data */
    int64_t odd_cprtime, even_cprtime;   /* when (ms) those frames arrived */
    int64_t seen_ms;            /* time (ms) of the last message */
    long messages;              /* number of Mode S messages received */
} modes_aircraft;

typedef struct modes_tracker modes_tracker;
modes_tracker *modes_tracker_create(void);
void           modes_tracker_destroy(modes_tracker *tr);
/* interactiveReceiveData() (dump1090.c:2069-2167): account `mm` to its aircraft (created on first
 * sight), update altitude / flight / speed / track and decode the CPR position (airborne pairs at
 * most 10 s apart, dump1090.c:2120; surface frames against the running mean of the airborne
 * positions, dump1090.c:2141-2155).  now_ms is the caller's clock (the reference reads
 * gettimeofday).  Returns the aircraft, or NULL when check_crc is set and mm->crcok is 0
 * (dump1090.c:2073).  The pointer stays valid until that aircraft expires. */
const modes_aircraft *modes_tracker_receive(modes_tracker *tr, const struct modesMessage *mm, int check_crc, int64_t now_ms);
/* interactiveRemoveStaleAircrafts() (dump1090.c:2203-2224): forget aircraft not heard for more than
 * ttl_ms (the reference: 60 s).  Returns how many were removed. */
uint64_t modes_tracker_expire(modes_tracker *tr, int64_t now_ms, int64_t ttl_ms);
uint64_t modes_tracker_count(const modes_tracker *tr);
const modes_aircraft *modes_tracker_get(const modes_tracker *tr, uint64_t i);   /* newest aircraft first */
/* Modes.ref_lat / ref_lon / ref_count (dump1090.c:205-206). */
void modes_tracker_reference(const modes_tracker *tr, double *lat, double *lon, int *count);
/* aircraftsToJson() (dump1090.c:2505-2552), the body the reference serves as /data.json: every
 * aircraft with a decoded position, newest first; metric = Modes.metric.  Returns the length of
 * the text; at most cap - 1 characters (and a terminating 0) are stored. */
size_t modes_tracker_json(const modes_tracker *tr, int metric, char *buf, size_t cap);
/* modesSendSBSOutput() (dump1090.c:2397-2448): the line for `mm`, '\n'-terminated, or 0 when this
 * message type has no SBS line.  `a` = what modes_tracker_receive returned for this message (needed
 * for MSG,3 positions and MSG,4 speed / track).  256 bytes always suffice. */
int modes_format_sbs(const struct modesMessage *mm, const modes_aircraft *a, char *buf, size_t cap);

/* CRC helpers shared with the device code (dump1090.c:703-753). */
uint32_t modes_checksum(const unsigned char *msg, int bits);      /* modesChecksum      */
uint32_t modes_compute_crc(const unsigned char *msg, int bits);   /* modesComputeCRC    */
int      modes_message_len_by_type(int type);                     /* modesMessageLenByType */

/* Number of buffers the reference's reader publishes for a stream of nbytes
 * (dump1090.c:484-510): nbytes/262144 + 1. */
uint64_t modes_block_count(uint64_t nbytes);

#ifdef __cplusplus
}
#endif
#endif
