// modes_host.cpp - libmodes_host.so: the sequential host half of the hot path
// (include/modes_host.h).  Consumes the GPU's modes_record list and reproduces, in
// stream order, what detectModeS()/decodeModesMessage()/useModesMessage() do after the
// demodulator has produced bits.  Pure C++17, no HIP.
//
// Line numbers cite /root/reference/dump1090.c.

#include "../../include/modes_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <memory>
#include <vector>

#include <sched.h>
#include <unistd.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "modes_core.h"

namespace {

constexpr uint32_t kIcaoSlots = 1024;                  // MODES_ICAO_CACHE_LEN, :65
constexpr int64_t kIcaoTtl = 60;                       // MODES_ICAO_CACHE_TTL, :66 (seconds)

// ICAOCacheHashAddress, :898-905.
inline uint32_t icao_slot(uint32_t a) {
    a = ((a >> 16) ^ a) * 0x45d9f3bu;
    a = ((a >> 16) ^ a) * 0x45d9f3bu;
    a = ((a >> 16) ^ a);
    return a & (kIcaoSlots - 1);
}

// x^(111-p) mod G for the 112 frame positions, and the remainders of the 256 byte values (byte-wise long division)
struct Tables {
    uint32_t esyn[112];
    uint32_t byte_rem[256];
    Tables() {
        for (int p = 0; p < 112; p++) esyn[p] = modes_bit_syndrome(p);
        for (uint32_t b = 0; b < 256; b++) {
            uint32_t r = b << 16;                                   // b * x^16, then 8 more shifts: b * x^24 mod G
            for (int t = 0; t < 8; t++) {
                r <<= 1;
                if (r & 0x1000000u) r ^= MODES_CRC_POLY;
            }
            byte_rem[b] = r & 0xFFFFFFu;
        }
    }
};
const Tables kTables;

// modes_syndrome (modes_core.h) a byte at a time: r <- (r * x^8 + byte) mod G
inline uint32_t syndrome_bytes(const unsigned char *msg, int nbytes) {
    uint32_t r = 0;
    for (int b = 0; b < nbytes; b++) r = ((r << 8) & 0xFFFFFFu) ^ kTables.byte_rem[r >> 16] ^ msg[b];
    return r;
}

}  // namespace

struct modes_host {
    modes_host_config cfg{};
    modes_host_stats st{};
    // Recently-seen ICAO addresses (:896-925).  The reference stamps entries with time(NULL) and
    // expires them after 60 s of WALL CLOCK.  The clock here is the caller's (modes_host_set_time): a
    // file run on the GPU finishes in milliseconds and never advances it, so nothing expires - the
    // semantics the parity oracle gets from the constant-clock interposer; a host on a live pipe
    // (--ifile -, --loop) advances it once per batch and gets the reference's 60 s.
    uint32_t icao[kIcaoSlots] = {};
    int64_t icao_seen[kIcaoSlots] = {};   // caller's clock (modes_host_set_time) when the address was last validated
    int64_t now_s = 0;                    // never advanced by a file run: nothing expires (the default)
    bool have_candidates = false;
    struct IcaoLog *log = nullptr;        // set while a piece of a batch is resolved speculatively (modes_host_resolve_raw_mt)
    bool mt_guess = false;                // the multi-threaded resolve runs its guess pass (set by the first piece that had to be resolved again)
    bool lean = false;                    // set by the --raw resolvers: the sink reads msg / msgbits / crcok only, so the decode stops
                                          // when those (and the whitelist) are settled - no altitude, squawk, position, velocity fields
};

// What a speculative piece did to / asked of the whitelist: which slots it wrote, and every lookup that was answered from
// a slot it had NOT written yet - i.e. from the state it was started with, which is a guess until the pieces before it are
// confirmed.
struct IcaoLog {
    struct Lookup { uint32_t addr; bool known; };
    bool written[kIcaoSlots] = {};
    std::vector<Lookup> lookups;
};

extern "C" {

uint32_t modes_checksum(const unsigned char *msg, int bits) { return syndrome_bytes(msg, bits / 8); }

// modesComputeCRC (:703-719) = parity of the data bits only = syndrome XOR received parity.
uint32_t modes_compute_crc(const unsigned char *msg, int bits) {
    const int n = bits / 8;
    const uint32_t rx = ((uint32_t)msg[n - 3] << 16) | ((uint32_t)msg[n - 2] << 8) | msg[n - 1];
    return syndrome_bytes(msg, n) ^ rx;
}

int modes_message_len_by_type(int type) { return modes_len_by_df(type); }

uint64_t modes_block_count(uint64_t nbytes) { return nbytes / MODES_DATA_LEN + 1; }

modes_host *modes_host_create(const modes_host_config *cfg) {
    if (!cfg) return nullptr;
    modes_host *h = new (std::nothrow) modes_host;
    if (h) h->cfg = *cfg;
    return h;
}

void modes_host_destroy(modes_host *h) { delete h; }

void modes_host_set_time(modes_host *h, int64_t now_seconds) { if (h) h->now_s = now_seconds; }

void modes_host_get_stats(const modes_host *h, modes_host_stats *out) {
    *out = h->st;
    if (!h->have_candidates) out->valid_preamble = -1;
}

int modes_host_wants(const modes_host *h, const struct modesMessage *mm) {
    return h->cfg.check_crc == 0 || mm->crcok;                                    // :1803
}

static void icao_remember(modes_host *h, uint32_t addr) {                                             // :910-914
    const uint32_t s = icao_slot(addr);
    h->icao[s] = addr;
    h->icao_seen[s] = h->now_s;
    if (h->log) h->log->written[s] = true;
}
static bool icao_known(const modes_host *h, uint32_t addr) {                                            // :919-925
    const uint32_t s = icao_slot(addr);
    const bool known = addr != 0 && h->icao[s] == addr && h->now_s - h->icao_seen[s] <= kIcaoTtl;
    if (h->log && addr != 0 && !h->log->written[s]) h->log->lookups.push_back({addr, known});
    return known;
}

// decodeAC13Field, :988-1012.
static int decode_ac13(const unsigned char *msg, int *unit) {
    const int m_bit = msg[3] & (1 << 6), q_bit = msg[3] & (1 << 4);
    if (!m_bit) {
        *unit = MODES_UNIT_FEET;
        if (q_bit) {
            const int n = ((msg[2] & 31) << 6) | ((msg[3] & 0x80) >> 2) | ((msg[3] & 0x20) >> 1) | (msg[3] & 15);
            return n * 25 - 1000;
        }
    } else {
        *unit = MODES_UNIT_METERS;
    }
    return 0;
}

// decodeAC12Field, :1016-1030.
static int decode_ac12(const unsigned char *msg, int *unit) {
    if (msg[5] & 1) {
        *unit = MODES_UNIT_FEET;
        const int n = ((msg[5] >> 1) << 4) | ((msg[6] & 0xF0) >> 4);
        return n * 25 - 1000;
    }
    return 0;
}

// A frame that did not come through the GPU (the reference's raw TCP input, decodeHexMessage
// dump1090.c:2472-2502, hands bytes straight to decodeModesMessage): syndrome and repair lookup on
// the host with the same helpers the device uses, then the common decode.
void modes_host_decode_frame(modes_host *h, const unsigned char *frame, struct modesMessage *mm) {
    const uint32_t *esyn = kTables.esyn;
    modes_attempt att;
    memset(&att, 0, sizeof att);
    memcpy(att.msg, frame, MODES_LONG_MSG_BYTES);
    const int bits = modes_len_by_df(frame[0] >> 3);
    att.gate_ok = 1;
    att.syndrome = syndrome_bytes(att.msg, bits / 8);
    att.nfix = (uint8_t)modes_find_fix(att.syndrome, bits, h->cfg.fix_errors ? (h->cfg.aggressive ? 2 : 1) : 0, esyn, att.fixpos);
    modes_host_decode(h, &att, mm);
}

void modes_host_decode(modes_host *h, const modes_attempt *att, struct modesMessage *mm) {
    static const char ais[] = "?ABCDEFGHIJKLMNOPQRSTUVWXYZ????? ???????????????0123456789??????";
    // The reference leaves the fields a message type does not use uninitialised (mm is a stack
    // variable, :1732); zero them so the struct is a function of the input.
    // Lean mode (the --raw sinks of this file, which read msg / msgbits / crcok / the address bytes only) skips the memset AND the
    // per-type fields: modes_host_resolve zeroes its modesMessage ONCE per call instead, so a sink that does read another field in
    // lean mode sees zeros or an earlier message's value of that field, never the stack's bytes.
    if (!h->lean) memset(mm, 0, sizeof *mm);
    memcpy(mm->msg, att->msg, MODES_LONG_MSG_BYTES);
    unsigned char *msg = mm->msg;

    mm->msgtype = msg[0] >> 3;                                                    // :1099
    mm->msgbits = modes_len_by_df(mm->msgtype);                                   // :1100
    mm->crc = att->syndrome;                                                      // :1104 (computed on the GPU)
    mm->errorbit = -1;
    mm->iid = 0;
    mm->crcok = (mm->crc == 0);

    // :1112-1128.  The lookup ran on the GPU with the context's maxfix; apply it here.
    if (!mm->crcok && h->cfg.fix_errors && (mm->msgtype == 11 || mm->msgtype == 17 || mm->msgtype == 18)) {
        const int maxfix = h->cfg.aggressive ? 2 : 1;
        const int nfixed = att->nfix <= maxfix ? att->nfix : 0;
        if (nfixed > 0) {
            // the syndrome is linear in the message bits: flipping message bit k (frame position k + 112 - msgbits)
            // XORs that position's single-bit syndrome in - same value as modesChecksum() of the repaired frame (:1119)
            for (int i = 0; i < nfixed; i++) {
                msg[att->fixpos[i] >> 3] ^= (unsigned char)(0x80u >> (att->fixpos[i] & 7));
                mm->crc ^= kTables.esyn[att->fixpos[i] + 112 - mm->msgbits];
            }
            mm->crcok = (mm->crc == 0);
            mm->errorbit = att->fixpos[0];
            if (nfixed == 1) h->st.single_bit_fix++; else h->st.two_bits_fix++;  // :1122-1126
        }
    }

    mm->ca = msg[0] & 7;                                                          // :1133
    mm->aa1 = msg[1]; mm->aa2 = msg[2]; mm->aa3 = msg[3];                         // :1136
    mm->metype = msg[4] >> 3;                                                     // :1141
    mm->mesub = msg[4] & 7;
    mm->fs = msg[0] & 7;                                                          // :1145
    mm->dr = msg[1] >> 3 & 31;
    mm->um = ((msg[1] & 7) << 3) | msg[2] >> 5;
    if (!h->lean) {   // squawk, :1163-1179 (Gillham interleave C1 A1 C2 A2 C4 A4 0 B1 D1 B2 D2 B4 D4)
        const int a = ((msg[3] & 0x80) >> 5) | ((msg[2] & 0x02) >> 0) | ((msg[2] & 0x08) >> 3);
        const int b = ((msg[3] & 0x02) << 1) | ((msg[3] & 0x08) >> 2) | ((msg[3] & 0x20) >> 5);
        const int c = ((msg[2] & 0x01) << 2) | ((msg[2] & 0x04) >> 1) | ((msg[2] & 0x10) >> 4);
        const int d = ((msg[3] & 0x01) << 2) | ((msg[3] & 0x04) >> 1) | ((msg[3] & 0x10) >> 4);
        mm->identity = a * 1000 + b * 100 + c * 10 + d;
    }

    if (mm->msgtype != 11 && mm->msgtype != 17 && mm->msgtype != 18) {            // :1183
        // bruteForceAP, :942-983: address = AP field XOR parity of the data bits
        mm->crcok = 0;
        const int t = mm->msgtype;
        if (t == 0 || t == 4 || t == 5 || t == 16 || t == 20 || t == 21 || t == 24) {
            const int last = mm->msgbits / 8 - 1;
            // modesComputeCRC (:703-719) = syndrome XOR received parity; these types are never repaired, so the
            // syndrome is the one the GPU computed
            const uint32_t crc = att->syndrome ^ (((uint32_t)msg[last - 2] << 16) | ((uint32_t)msg[last - 1] << 8) | msg[last]);
            const uint32_t b0 = msg[last] ^ (crc & 0xff), b1 = msg[last - 1] ^ ((crc >> 8) & 0xff),
                           b2 = msg[last - 2] ^ ((crc >> 16) & 0xff);
            if (icao_known(h, b0 | (b1 << 8) | (b2 << 16))) {
                mm->aa1 = (int)b2; mm->aa2 = (int)b1; mm->aa3 = (int)b0;
                mm->crcok = 1;
            }
        }
    } else {
        const uint32_t addr = ((uint32_t)mm->aa1 << 16) | ((uint32_t)mm->aa2 << 8) | (uint32_t)mm->aa3;
        if (mm->crcok && mm->errorbit == -1) icao_remember(h, addr);               // :1198
        if (mm->msgtype == 11 && !mm->crcok && mm->crc < 80 && icao_known(h, addr)) {   // :1204
            mm->iid = (int)mm->crc;
            mm->crcok = 1;
        }
    }

    mm->phase_corrected = 0;                                                      // :1309
    if (h->lean) return;                                                          // crcok, the address and the whitelist are settled
    if (mm->msgtype == 0 || mm->msgtype == 4 || mm->msgtype == 16 || mm->msgtype == 20)   // :1213
        mm->altitude = decode_ac13(msg, &mm->unit);

    if (mm->msgtype == 17 || mm->msgtype == 18) {                                 // :1221
        if (mm->metype >= 1 && mm->metype <= 4) {                                 // identification
            mm->aircraft_type = mm->metype - 1;
            mm->flight[0] = ais[msg[5] >> 2];
            mm->flight[1] = ais[((msg[5] & 3) << 4) | (msg[6] >> 4)];
            mm->flight[2] = ais[((msg[6] & 15) << 2) | (msg[7] >> 6)];
            mm->flight[3] = ais[msg[7] & 63];
            mm->flight[4] = ais[msg[8] >> 2];
            mm->flight[5] = ais[((msg[8] & 3) << 4) | (msg[9] >> 4)];
            mm->flight[6] = ais[((msg[9] & 15) << 2) | (msg[10] >> 6)];
            mm->flight[7] = ais[msg[10] & 63];
            mm->flight[8] = '\0';
        } else if (mm->metype >= 5 && mm->metype <= 8) {                          // surface position, :1236
            mm->movement = ((msg[4] & 0x07) << 4) | (msg[5] >> 4);
            mm->movement_valid = (mm->movement != 0);
            mm->ground_track_valid = (msg[5] >> 3) & 1;
            mm->ground_track = (((msg[5] & 0x07) << 4) | (msg[6] >> 4)) * 360 / 128;
            mm->fflag = (msg[6] >> 2) & 1;
            mm->tflag = (msg[6] >> 3) & 1;
            mm->raw_latitude = ((msg[6] & 3) << 15) | (msg[7] << 7) | (msg[8] >> 1);
            mm->raw_longitude = ((msg[8] & 1) << 16) | (msg[9] << 8) | msg[10];
        } else if (mm->metype >= 9 && mm->metype <= 18) {                         // airborne position, :1261
            mm->fflag = msg[6] & (1 << 2);
            mm->tflag = msg[6] & (1 << 3);
            mm->altitude = decode_ac12(msg, &mm->unit);
            mm->raw_latitude = ((msg[6] & 3) << 15) | (msg[7] << 7) | (msg[8] >> 1);
            mm->raw_longitude = ((msg[8] & 1) << 16) | (msg[9] << 8) | msg[10];
        } else if (mm->metype == 19 && mm->mesub >= 1 && mm->mesub <= 4) {        // velocity, :1272
            if (mm->mesub == 1 || mm->mesub == 2) {
                mm->ew_dir = (msg[5] & 4) >> 2;
                mm->ew_velocity = ((msg[5] & 3) << 8) | msg[6];
                mm->ns_dir = (msg[7] & 0x80) >> 7;
                mm->ns_velocity = ((msg[7] & 0x7f) << 3) | ((msg[8] & 0xe0) >> 5);
                mm->vert_rate_source = (msg[8] & 0x10) >> 4;
                mm->vert_rate_sign = (msg[8] & 0x8) >> 3;
                mm->vert_rate = ((msg[8] & 7) << 6) | ((msg[9] & 0xfc) >> 2);
                mm->velocity = (int)std::sqrt((double)(mm->ns_velocity * mm->ns_velocity + mm->ew_velocity * mm->ew_velocity));
                if (mm->velocity) {
                    int ewv = mm->ew_velocity, nsv = mm->ns_velocity;
                    if (mm->ew_dir) ewv = -ewv;
                    if (mm->ns_dir) nsv = -nsv;
                    const double heading = std::atan2((double)ewv, (double)nsv);
                    mm->heading = (int)(heading * 360 / (M_PI * 2));
                    if (mm->heading < 0) mm->heading += 360;
                } else {
                    mm->heading = 0;
                }
            } else {
                mm->heading_is_valid = msg[5] & (1 << 2);
                mm->heading = (int)((360.0 / 128) * (((msg[5] & 3) << 5) | (msg[6] >> 3)));
            }
        }
    }
    mm->phase_corrected = 0;                                                      // :1309
}

// The loop body of detectModeS after the bits exist (:1708-1791), driven by records instead of
// by sample offsets.  State that the reference keeps in locals of one detectModeS() call - the
// skip distance `j += ...` (:1770) and use_correction (:1568) - resets at every buffer, so it
// resets here whenever `block` changes.
uint64_t modes_host_resolve(modes_host *h, const modes_record *recs, uint64_t nrecs, const uint64_t *cands, uint64_t ncand,
                            modes_sink_fn sink, void *user) {
    uint64_t calls = 0;
    uint64_t ri = 0, ci = 0;
    uint32_t cur_block = 0xffffffffu;
    uint32_t skip_to = 0;                 // first block-local offset the scan visits again
    if (cands) h->have_candidates = true;
    struct modesMessage mm;               // one for the whole call, zeroed here: lean decodes (modes_host_decode) leave the fields
    memset(&mm, 0, sizeof mm);            // they do not compute alone, and no sink may ever see the stack's bytes in them

    auto next_is_candidate_only = [&]() -> bool {
        if (!cands || ci >= ncand) return false;
        if (ri >= nrecs) return true;
        const uint64_t grec = (uint64_t)recs[ri].block * MODES_BLOCK_STRIDE + recs[ri].j;
        return cands[ci] < grec;
    };

    while (ri < nrecs || (cands && ci < ncand)) {
        if (next_is_candidate_only()) {
            // a preamble whose first noise gate failed: counted (:1651), then `continue` (:1723-1726)
            const uint32_t blk = (uint32_t)(cands[ci] / MODES_BLOCK_STRIDE), j = (uint32_t)(cands[ci] % MODES_BLOCK_STRIDE);
            ci++;
            if (blk != cur_block) { cur_block = blk; skip_to = 0; }
            if (j >= skip_to) h->st.valid_preamble++;
            continue;
        }
        const modes_record &r = recs[ri++];
        if (cands && ci < ncand && cands[ci] == (uint64_t)r.block * MODES_BLOCK_STRIDE + r.j) ci++;
        if (r.block != cur_block) { cur_block = r.block; skip_to = 0; }
        if (r.j < skip_to) continue;                                             // inside a decoded frame, :1770
        h->st.valid_preamble++;                                                   // :1651

        for (int pass = 0; pass < 2; pass++) {
            const modes_attempt &a = r.att[pass];
            if (pass == 1 && r.j != 0) h->st.out_of_phase++;                      // :1660-1663
            if (!a.gate_ok) break;                                                // :1723-1726 (no retry)
            bool good = false;
            if (a.errors == 0 || (h->cfg.aggressive && a.errors < 3)) {           // :1731
                modes_host_decode(h, &a, &mm);
                mm.phase_corrected = 0;                                           // (set below for this attempt only; the full decode zeroes it anyway)
                if (mm.crcok || pass == 1) {                                      // :1738-1753
                    if (a.errors == 0) h->st.demodulated++;
                    if (mm.errorbit == -1) {
                        if (mm.crcok) h->st.goodcrc++; else h->st.badcrc++;
                    } else {
                        h->st.badcrc++;
                        h->st.fixed++;
                        if (mm.errorbit < MODES_LONG_MSG_BITS) h->st.single_bit_fix++; else h->st.two_bits_fix++;
                    }
                }
                if (mm.crcok) {                                                   // :1769-1774
                    // msglen comes from the demodulated (pre-repair) DF, like the local `msglen` of :1709
                    skip_to = r.j + (8 + (uint32_t)modes_len_by_df(a.msg[0] >> 3)) * 2 + 1;
                    good = true;
                    if (pass == 1) mm.phase_corrected = 1;
                }
                if (sink) sink(&mm, r.block, r.j, user);                          // :1777
                calls++;
            }
            if (good) break;                                                      // :1786-1791
        }
    }
    return calls;
}

namespace {
struct ArraySink {
    modes_host *h;
    modes_emitted *out;
    uint64_t cap, n;
};
void array_sink(const struct modesMessage *mm, uint32_t block, uint32_t j, void *user) {
    ArraySink *s = static_cast<ArraySink *>(user);
    if (!modes_host_wants(s->h, mm)) return;
    if (s->n < s->cap) {
        s->out[s->n].mm = *mm;
        s->out[s->n].block = block;
        s->out[s->n].j = j;
    }
    s->n++;
}
}  // namespace

uint64_t modes_host_resolve_to_array(modes_host *h, const modes_record *recs, uint64_t nrecs, const uint64_t *cands,
                                     uint64_t ncand, modes_emitted *out, uint64_t cap) {
    ArraySink s{h, out, out ? cap : 0, 0};
    modes_host_resolve(h, recs, nrecs, cands, ncand, array_sink, &s);
    return s.n;
}

// ---------------------------------------------------------------------------------------------
// The --raw listing of a batch (what `dump1090 --raw` prints for these buffers, dump1090.c:1324-1326 behind the filter of
// :1803) without a modesMessage: the LEAN resolve.  Everything decodeModesMessage() decides about an attempt that the
// whitelist does not decide is in the attempt's class byte (include/modes_gfx950.h MODES_CLS_*; written by the GPU wavefront
// that demodulated it, or - records of another producer, or of a context configured differently - by the same
// modes_classify() here).  What is left per record: skip window (:1770), the class, at most one whitelist access through
// the precomputed slot (:1198, :1204, :942-983), the counters of :1738-1753, and the line - two 16-byte stores of hex digits.
// modes_host_resolve() with a raw sink is the definition it is tested against (tests/test_host.py, every stream x flag set).
// ---------------------------------------------------------------------------------------------
namespace {
inline bool wl_known(modes_host *h, uint32_t addr, uint32_t slot) {                                // :919-925
    const bool known = addr != 0 && h->icao[slot] == addr && h->now_s - h->icao_seen[slot] <= kIcaoTtl;
    if (h->log && addr != 0 && !h->log->written[slot]) h->log->lookups.push_back({addr, known});
    return known;
}
inline void wl_remember(modes_host *h, uint32_t addr, uint32_t slot) {                            // :910-914
    h->icao[slot] = addr;
    h->icao_seen[slot] = h->now_s;
    if (h->log) h->log->written[slot] = true;
}

// 16 bytes -> 32 lower-case hex digits (the caller uses the first 14 or 28)
inline void hex32(const unsigned char *src, char *dst) {
#if defined(__SSE2__)
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src));
    const __m128i nib = _mm_set1_epi8(0x0f), nine = _mm_set1_epi8(9), zero = _mm_set1_epi8('0'), alpha = _mm_set1_epi8('a' - '0' - 10);
    const __m128i hi = _mm_and_si128(_mm_srli_epi16(v, 4), nib), lo = _mm_and_si128(v, nib);
    __m128i a = _mm_unpacklo_epi8(hi, lo), b = _mm_unpackhi_epi8(hi, lo);
    a = _mm_add_epi8(_mm_add_epi8(a, zero), _mm_and_si128(_mm_cmpgt_epi8(a, nine), alpha));
    b = _mm_add_epi8(_mm_add_epi8(b, zero), _mm_and_si128(_mm_cmpgt_epi8(b, nine), alpha));
    _mm_storeu_si128(reinterpret_cast<__m128i *>(dst), a);
    _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + 16), b);
#else
    static const char digits[] = "0123456789abcdef";
    for (int i = 0; i < 16; i++) { dst[2 * i] = digits[src[i] >> 4]; dst[2 * i + 1] = digits[src[i] & 15]; }
#endif
}

// where the listing goes: out[0 .. cap), whole lines as long as they fit and nothing behind the first that does not
struct LeanOut {
    char *out;
    uint64_t cap;
    uint64_t n = 0;         // length of the whole listing
    uint64_t stored = 0;    // bytes of it in out
    uint64_t msgs = 0;
    bool full = false;
    LeanOut(char *o, uint64_t c) : out(o), cap(o ? c : 0) {}
    // msg: 16 readable bytes (a modes_attempt's msg is followed by two more fields of the same struct)
    inline void line(const unsigned char *msg, bool is_long) {
        const uint64_t len = is_long ? 31 : 17;
        if (!full) {
            if (n + 40 <= cap) {                                                  // room for the two 16-byte stores and the NUL
                char *p = out + n;
                p[0] = '*';
                hex32(msg, p + 1);
                p[len - 2] = ';';
                p[len - 1] = '\n';
                stored = n + len;
            } else if (n + len + 1 <= cap) {
                char tmp[40];
                tmp[0] = '*';
                hex32(msg, tmp + 1);
                tmp[len - 2] = ';';
                tmp[len - 1] = '\n';
                memcpy(out + n, tmp, (size_t)len);
                stored = n + len;
            } else {
                full = true;
            }
        }
        n += len;
        msgs++;
    }
    void finish() { if (out && stored < cap) out[stored] = 0; }
};

void lean_resolve(modes_host *h, const modes_record *recs, uint64_t nrecs, LeanOut &o) {
    const uint32_t fix = h->cfg.fix_errors ? 1u : 0u, aggressive = h->cfg.aggressive ? 1u : 0u;
    const uint32_t cfg_mask = MODES_CLS_VALID | MODES_CLS_FIX | MODES_CLS_AGGRESSIVE;
    const uint32_t cfg_want = MODES_CLS_VALID | (fix ? MODES_CLS_FIX : 0u) | (aggressive ? MODES_CLS_AGGRESSIVE : 0u);
    const bool print_all = h->cfg.check_crc == 0;                                 // :1803
    modes_host_stats st{};                                                        // this call's counts (added to h->st at the end)
    uint32_t cur_block = 0xffffffffu, skip_to = 0;
    for (uint64_t i = 0; i < nrecs; i++) {
        const modes_record &r = recs[i];
        if (r.block != cur_block) { cur_block = r.block; skip_to = 0; }
        if (r.j < skip_to) continue;                                              // inside a decoded frame, :1770
        st.valid_preamble++;                                                      // :1651
        for (int pass = 0; pass < 2; pass++) {
            const modes_attempt &a = r.att[pass];
            if (pass == 1 && r.j != 0) st.out_of_phase++;                         // :1660-1663
            uint32_t cls = a.cls, slot = a.slot;
            if ((cls & cfg_mask) != cfg_want) {                                   // not classified for this configuration: here, the same way
                cls = modes_classify(a.msg[0] >> 3, a.errors, a.gate_ok, a.syndrome, a.nfix, fix, aggressive);
                slot = modes_class_slot(cls, a.msg[1], a.msg[2], a.msg[3], a.syndrome);
            }
            const uint32_t kind = cls & MODES_CLS_KIND;
            if (kind == MODES_CLS_GATE) break;                                    // :1723-1726 (no retry)
            if (kind == MODES_CLS_SKIP) continue;                                 // :1731
            bool crcok = false, repaired = false;
            switch (kind) {
            case MODES_CLS_CLEAN:
                crcok = true;
                wl_remember(h, ((uint32_t)a.msg[1] << 16) | ((uint32_t)a.msg[2] << 8) | a.msg[3], slot);   // :1198
                break;
            case MODES_CLS_FIXED:
                crcok = repaired = true;
                if (a.nfix == 1) st.single_bit_fix++; else st.two_bits_fix++;     // :1122-1126
                break;
            case MODES_CLS_IID:                                                   // :1204
                crcok = wl_known(h, ((uint32_t)a.msg[1] << 16) | ((uint32_t)a.msg[2] << 8) | a.msg[3], slot);
                break;
            case MODES_CLS_AP:                                                    // :942-983: the recovered address is the syndrome
                crcok = wl_known(h, a.syndrome, slot);
                break;
            default: break;                                                       // BAD
            }
            if (crcok || pass == 1) {                                             // :1738-1753
                if (cls & MODES_CLS_NOERR) st.demodulated++;
                if (!repaired) {
                    if (crcok) st.goodcrc++; else st.badcrc++;
                } else {
                    st.badcrc++;
                    st.fixed++;
                    st.single_bit_fix++;                                          // (errorbit < 112 always: the quirk of :1749)
                }
            }
            const bool is_long = (cls & MODES_CLS_LONG) != 0;
            if (print_all || crcok) {                                             // :1777 behind :1803
                if (!repaired) {
                    o.line(a.msg, is_long);
                } else {
                    unsigned char m[16];
                    memcpy(m, a.msg, 16);
                    for (int k = 0; k < a.nfix; k++) m[a.fixpos[k] >> 3] ^= (unsigned char)(0x80u >> (a.fixpos[k] & 7));
                    o.line(m, is_long);
                }
            }
            if (crcok) {                                                          // :1769-1774, :1786-1791
                skip_to = r.j + (8 + (is_long ? 112u : 56u)) * 2 + 1;
                break;
            }
        }
    }
    h->st.valid_preamble += st.valid_preamble;
    h->st.out_of_phase += st.out_of_phase;
    h->st.demodulated += st.demodulated;
    h->st.goodcrc += st.goodcrc;
    h->st.badcrc += st.badcrc;
    h->st.fixed += st.fixed;
    h->st.single_bit_fix += st.single_bit_fix;
    h->st.two_bits_fix += st.two_bits_fix;
}

struct RawSink {
    modes_host *h;
    char *out;
    uint64_t cap, n, msgs;
};
void raw_sink(const struct modesMessage *mm, uint32_t, uint32_t, void *user) {
    RawSink *s = static_cast<RawSink *>(user);
    if (!modes_host_wants(s->h, mm)) return;
    s->msgs++;
    const int len = 3 + mm->msgbits / 4;                                          // '*' + hex + ';' + '\n'
    if (s->out && s->n + (uint64_t)len + 1 <= s->cap) modes_format_raw(mm, s->out + s->n);
    s->n += (uint64_t)len;
}
}  // namespace

// With candidates (--stats: every preamble position counts) the general resolve with a raw sink; without, the lean one.
uint64_t modes_host_resolve_raw(modes_host *h, const modes_record *recs, uint64_t nrecs, const uint64_t *cands, uint64_t ncand,
                                char *out, uint64_t cap, uint64_t *nbytes) {
    if (out && cap) out[0] = 0;                                                   // an empty (or wholly cut) listing is ""
    if (!cands && !getenv("MODES_HOST_NO_LEAN")) {
        LeanOut o(out, cap);
        lean_resolve(h, recs, nrecs, o);
        o.finish();
        if (nbytes) *nbytes = o.n;
        return o.msgs;
    }
    RawSink s{h, out, cap, 0, 0};
    struct Lean { modes_host *h; bool was; ~Lean() { h->lean = was; } } lean{h, h->lean};
    h->lean = true;
    modes_host_resolve(h, recs, nrecs, cands, ncand, raw_sink, &s);
    if (nbytes) *nbytes = s.n;
    return s.msgs;
}

// ---------------------------------------------------------------------------------------------
// The same listing from several threads.  The resolve is sequential only through the ICAO whitelist (and the skip
// window, which never crosses a buffer): the batch is cut at buffer boundaries into one piece per thread, every piece is
// resolved SPECULATIVELY from a guessed whitelist - the state the batch started with, overlaid with the addresses the
// pieces before it will certainly or almost certainly add (frames with a clean CRC) - and logs every lookup that was
// answered from that guess.  Then, in order: a piece whose logged answers all hold under the TRUE state it should have
// started from executed exactly like the sequential resolve (same answers -> same control flow -> same output) and is
// kept; a piece with a wrong answer is resolved again from the true state (rare: it takes a frame inside another frame's
// skip window, or one only a not-yet-seen address validates).  Exact by construction; tests/test_host.py compares it with
// the one-thread listing on every stream.
//
// Round 6: ONE parallel region holds the guess and the speculative resolve (a piece waits for the guesses of the pieces
// before it, not for a second wake-up of the pool), piece i always runs on worker i (its records are in that core's cache
// from the guess), the first piece writes straight into the caller's buffer, and every piece's text buffer, log and state
// live as long as the process (a fresh 12 MB of text per call was 3,000 page faults under one address-space lock: the reason
// four threads were 2.3 x one).  The pool never runs more pieces than the process has CPUs to run them on
// (modes_host_cpu_budget: affinity and cgroup quota, not the machine's core count).
// ---------------------------------------------------------------------------------------------
extern "C" int modes_host_cpu_budget(void) {
    static const int budget = [] {
        long n = sysconf(_SC_NPROCESSORS_ONLN);
        if (n < 1) n = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) {
            const int a = CPU_COUNT(&set);
            if (a >= 1 && a < n) n = a;
        }
        // cgroup v2: "max 100000" or "<quota> <period>"; v1: cpu.cfs_quota_us / cpu.cfs_period_us (-1 = none)
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = "";
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
            fclose(f1);
            if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(f2, "%lld", &period) != 1) period = 0;
                fclose(f2);
            }
        }
        if (quota > 0 && period > 0) {
            const long q = (long)((quota + period - 1) / period);
            if (q >= 1 && q < n) n = q;
        }
        if (const char *e = getenv("MODES_HOST_CPUS")) {                          // (tests, experiments)
            const long v = atol(e);
            if (v >= 1) n = v;
        }
        return (int)n;
    }();
    return budget;
}

namespace {
// Worker threads of the multi-threaded resolve: created once per process, on first use (starting 8-32 threads costs more
// than resolving a batch of 35,000 records).  One parallel loop at a time; callers from different hosts take turns.  Index i
// of a loop ALWAYS runs on worker i - 1 (index 0 on the caller).
class WorkerPool {
public:
    static WorkerPool &instance() {
        static WorkerPool pool;
        return pool;
    }
    // fn(0 .. n-1), the caller runs index 0 itself; returns when all are done
    void run(size_t n, const std::function<void(size_t)> &fn) {
        if (n <= 1) { if (n) fn(0); return; }
        {
            std::unique_lock<std::mutex> g(m_);
            while (workers_.size() < n - 1) {
                const size_t id = workers_.size();
                workers_.emplace_back([this, id] { work(id); });
            }
            fn_ = &fn; total_ = n; left_ = n - 1; gen_++;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return left_ == 0; });
        fn_ = nullptr;
    }
    std::mutex &turn() { return turn_; }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
private:
    void work(size_t id) {
        std::unique_lock<std::mutex> g(m_);
        uint64_t seen = 0;
        for (;;) {
            cv_.wait(g, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            if (id + 1 >= total_ || !fn_) continue;                               // this loop is shorter than the pool
            const std::function<void(size_t)> *fn = fn_;
            g.unlock();
            (*fn)(id + 1);
            g.lock();
            if (--left_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, turn_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t total_ = 0, left_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

struct Piece {
    const modes_record *recs = nullptr;   // the segment the piece lies in
    uint64_t lo = 0, hi = 0;              // records [lo, hi) of it: whole buffers
    modes_host host;                      // private copy: config, clock, whitelist (guess, then the piece's own writes), stats of the piece
    IcaoLog log;
    char *text = nullptr;                 // the piece's listing: its own buffer (kept from call to call) or, the first piece's, the caller's
    char *own = nullptr;
    uint64_t own_cap = 0;
    uint64_t cap = 0, nbytes = 0, stored = 0, msgs = 0;
    // guess: what the piece's clean DF11/17/18 frames would write to the whitelist (last write per slot)
    uint32_t guess_addr[kIcaoSlots];
    bool guess_set[kIcaoSlots];
    std::atomic<int> guessed{0};
    double t_begin = 0, t_guessed = 0, t_started = 0, t_done = 0;   // MODES_HOST_MT_DEBUG: when the piece's worker got going, ...
    ~Piece() { free(own); }
    bool own_text(uint64_t need) {
        if (own_cap < need) {
            free(own);
            own_cap = need + need / 4;
            own = static_cast<char *>(malloc((size_t)own_cap));
            if (!own) { own_cap = 0; return false; }
        }
        text = own;
        cap = own_cap;
        return true;
    }
};
// The pieces of the CALLING THREAD (grow-only): a resolver thread keeps its pieces' text buffers, logs and states from call to call,
// and the listing modes_host_resolve_raw_pieces leaves in them stays put until that thread's next multi-threaded resolve.
std::vector<std::unique_ptr<Piece>> &piece_store() {
    static thread_local std::vector<std::unique_ptr<Piece>> store;
    return store;
}
struct PieceList {                        // modes_host_resolve_raw_pieces: where the caller wants the pieces named
    modes_text_piece *out;
    uint32_t cap, n;
};

inline uint32_t attempt_class(const modes_attempt &a, uint32_t cfg_mask, uint32_t cfg_want, uint32_t fix, uint32_t aggressive) {
    if ((a.cls & cfg_mask) == cfg_want) return a.cls;
    return modes_classify(a.msg[0] >> 3, a.errors, a.gate_ok, a.syndrome, a.nfix, fix, aggressive);
}
// The addresses a piece's clean DF11/17/18 frames put on the whitelist (dump1090.c:1198: crcok without repair).  Which of
// them really get decoded depends on skip windows: a guess, checked when the pieces are confirmed.
void guess_piece(Piece &p, const modes_host_config &cfg) {
    memset(p.guess_set, 0, sizeof p.guess_set);
    const uint32_t fix = cfg.fix_errors ? 1u : 0u, aggressive = cfg.aggressive ? 1u : 0u;
    const uint32_t cfg_mask = MODES_CLS_VALID | MODES_CLS_FIX | MODES_CLS_AGGRESSIVE;
    const uint32_t cfg_want = MODES_CLS_VALID | (fix ? MODES_CLS_FIX : 0u) | (aggressive ? MODES_CLS_AGGRESSIVE : 0u);
    for (uint64_t i = p.lo; i < p.hi; i++) {
        for (int a = 0; a < 2; a++) {
            const modes_attempt &at = p.recs[i].att[a];
            const uint32_t kind = attempt_class(at, cfg_mask, cfg_want, fix, aggressive) & MODES_CLS_KIND;
            if (kind == MODES_CLS_CLEAN) {
                const uint32_t addr = ((uint32_t)at.msg[1] << 16) | ((uint32_t)at.msg[2] << 8) | at.msg[3];
                const uint32_t sl = modes_icao_slot(addr);
                p.guess_addr[sl] = addr;
                p.guess_set[sl] = true;
            }
            if (kind == MODES_CLS_CLEAN || kind == MODES_CLS_FIXED || kind == MODES_CLS_GATE) break;   // the position ends here
        }
    }
}
void run_piece(Piece &p) {
    p.log.lookups.clear();
    memset(p.log.written, 0, sizeof p.log.written);
    p.host.st = modes_host_stats{};
    p.host.have_candidates = false;
    p.host.log = &p.log;
    LeanOut o(p.text, p.cap);
    lean_resolve(&p.host, p.recs + p.lo, p.hi - p.lo, o);
    p.host.log = nullptr;
    p.nbytes = o.n;
    p.stored = o.stored;
    p.msgs = o.msgs;
}
}  // namespace

// `outer` (modes_host_resolve_raw_spec; else null): the log of the CALL as a whole - the slots it wrote, and the lookups
// that were answered from the state `h` had on entry.
// `pl` (modes_host_resolve_raw_pieces; then out == nullptr): the listing stays in the pieces' own buffers, *pl names them.
static uint64_t resolve_raw_mtv_impl(modes_host *h, const modes_record *const *segs, const uint64_t *seg_nrecs, uint32_t nsegs,
                                     char *out, uint64_t cap, uint64_t *nbytes, int threads, IcaoLog *outer, PieceList *pl = nullptr) {
    uint64_t nrecs = 0;
    for (uint32_t g = 0; g < nsegs; g++) nrecs += seg_nrecs[g];
    // threads < 0: exactly -threads pieces however short the list and however few CPUs (tests); otherwise at least 16384 records
    // per thread - 70 us of work at 4.3 ns a record, against the ~50 us it takes to get a sleeping worker going: with 2048 a
    // 35,000-record list woke 15 threads for 10 us each, and the 8 GiB leg's step was 3-5 % slower with 8-15 resolver threads
    // than with 2 (profiles/r09/resolve_threads_ab.txt) - and no more threads than the process may run at once
    const uint64_t kMinPiece = threads < 0 ? 1 : 16384;
    int T = threads < 0 ? -threads : threads;
    if (threads >= 0 && T > modes_host_cpu_budget()) T = modes_host_cpu_budget();
    T = T < 1 ? 1 : (T > 64 ? 64 : T);
    if ((uint64_t)T > nrecs / kMinPiece) T = (int)(nrecs / kMinPiece);
    auto one_thread = [&]() -> uint64_t {                                         // segment after segment
        uint64_t msgs = 0, total = 0;
        struct Log { modes_host *h; IcaoLog *was; ~Log() { h->log = was; } } keep{h, h->log};
        if (outer) h->log = outer;                                                // (the log's `written` carries over from segment to segment)
        if (pl) {                                                                 // the whole listing is one piece, in this thread's first buffer
            std::vector<std::unique_ptr<Piece>> &store = piece_store();
            if (store.empty()) store.emplace_back(new Piece);
            Piece &p = *store[0];
            pl->n = 0;
            if (!p.own_text(nrecs * (h->cfg.check_crc ? 31 : 62) + 64)) { if (nbytes) *nbytes = 0; return 0; }
            for (uint32_t g = 0; g < nsegs; g++) {
                uint64_t nb = 0;
                msgs += modes_host_resolve_raw(h, segs[g], seg_nrecs[g], nullptr, 0, p.text + total, p.cap - total, &nb);
                total += nb;
            }
            if (total && pl->cap) { pl->out[0] = modes_text_piece{p.text, total}; pl->n = 1; }
            if (nbytes) *nbytes = total;
            return msgs;
        }
        for (uint32_t g = 0; g < nsegs; g++) {
            uint64_t nb = 0;
            msgs += modes_host_resolve_raw(h, segs[g], seg_nrecs[g], nullptr, 0, out ? out + total : nullptr, cap > total ? cap - total : 0, &nb);
            total += nb;
        }
        if (nbytes) *nbytes = total;
        return msgs;
    };
    if (T <= 1) return one_thread();
    // pieces: about nrecs / T records each, inside one segment, cut where the buffer changes (the skip window resets
    // there; a segment ends with a whole buffer)
    struct Cut { const modes_record *recs; uint64_t lo, hi; };
    std::vector<Cut> cuts;
    cuts.reserve((size_t)T + nsegs);
    const uint64_t share = (nrecs + (uint64_t)T - 1) / (uint64_t)T;
    for (uint32_t g = 0; g < nsegs; g++) {
        const modes_record *recs = segs[g];
        const uint64_t n = seg_nrecs[g];
        uint64_t lo = 0;
        while (lo < n) {
            uint64_t hi = lo + share < n ? lo + share : n;
            while (hi < n && recs[hi].block == recs[hi - 1].block) hi++;
            if (n - hi < share / 4) hi = n;                                       // no sliver at the end of a segment
            cuts.push_back(Cut{recs, lo, hi});
            lo = hi;
        }
    }
    const size_t P = cuts.size();
    if (P > 64 && pl) return one_thread();                                        // (pieces of several calls would share this thread's buffers)
    if (P > 64) {                                                                 // many short segments: fewer, longer pieces are not worth the code
        uint64_t msgs = 0, total = 0;
        for (uint32_t g = 0; g < nsegs; g++) {
            uint64_t nb = 0;
            msgs += resolve_raw_mtv_impl(h, &segs[g], &seg_nrecs[g], 1, out ? out + total : nullptr, cap > total ? cap - total : 0, &nb, threads, outer);
            total += nb;
        }
        if (nbytes) *nbytes = total;
        return msgs;
    }
    WorkerPool &pool = WorkerPool::instance();
    std::unique_lock<std::mutex> turn(pool.turn());                               // one parallel loop at a time in the process
    std::vector<std::unique_ptr<Piece>> &pieces = piece_store();                  // (this thread's)
    while (pieces.size() < P) pieces.emplace_back(new Piece);
    const bool dbg = getenv("MODES_HOST_MT_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    const uint64_t per_record = h->cfg.check_crc ? 31 : 62;                       // lines a record can print: one behind the filter of :1803, else two
    for (size_t t = 0; t < P; t++) {
        Piece &p = *pieces[t];
        p.recs = cuts[t].recs;
        p.lo = cuts[t].lo;
        p.hi = cuts[t].hi;
        p.guessed.store(0, std::memory_order_relaxed);
        if (t == 0 && out) { p.text = out; p.cap = cap; }                         // the first piece's text is where it belongs
        else if (!p.own_text((p.hi - p.lo) * per_record + 64)) {                  // no memory for a piece's text: the sequential resolve needs none
            turn.unlock();
            return one_thread();
        }
    }
    // ONE parallel region: every piece lists what its clean frames would write; the state a piece starts from is the
    // batch's initial state with the lists of the pieces before it applied in order; then the speculative resolve
    // The guess costs a pass over the records (a third of the call's memory traffic) and only pays where later pieces ASK the
    // whitelist about addresses earlier pieces write: a host whose pieces have never been caught with a wrong answer starts every
    // piece from the batch's own start state (streams of DF11 / DF17 squitters ask nothing); the first wrong answer - resolved
    // again below, like any other - switches the guesses on for the rest of the host's life (traffic with DF0/4/5/20/21 replies).
    const bool guessing = h->mt_guess || getenv("MODES_HOST_MT_GUESS") != nullptr;
    modes_host start = *h;
    start.log = nullptr;
    start.lean = false;
    // (more pieces than threads - many segments: a worker takes pieces w, w + W, ...: never more than T threads at work)
    const size_t W = std::min(P, (size_t)T);
    pool.run(W, [&](size_t w) {
        for (size_t t = w; t < P; t += W) {                                       // every guess of this worker first: nobody waits for a piece
            Piece &p = *pieces[t];                                                // whose worker is still resolving an earlier one
            if (dbg) p.t_begin = now();
            if (guessing) guess_piece(p, start.cfg);
            p.guessed.store(1, std::memory_order_release);
            if (dbg) p.t_guessed = now();
        }
        for (size_t t = w; t < P; t += W) {
            Piece &p = *pieces[t];
            p.host = start;
            for (size_t k = 0; guessing && k < t; k++) {
                const Piece &q = *pieces[k];
                for (int spin = 0; !q.guessed.load(std::memory_order_acquire); spin++)
                    if (spin > 64) std::this_thread::yield();
                for (uint32_t sl = 0; sl < kIcaoSlots; sl++)
                    if (q.guess_set[sl]) { p.host.icao[sl] = q.guess_addr[sl]; p.host.icao_seen[sl] = start.now_s; }
            }
            if (dbg) p.t_started = now();
            run_piece(p);
            if (dbg) p.t_done = now();
        }
    });
    const double t2 = now();
    int reruns = 0;
    // confirm in order: the true state at the start of piece t is the confirmed state at the end of piece t - 1
    modes_host truth = start;                                                     // whitelist + clock of the true sequential run
    for (size_t t = 0; t < P; t++) {
        Piece &p = *pieces[t];
        bool ok = true;
        for (const IcaoLog::Lookup &q : p.log.lookups) {
            if (icao_known(&truth, q.addr) != q.known) { ok = false; break; }
        }
        if (!ok) {                                                                // rare: again, from the true state
            const modes_host_config cfg = p.host.cfg;
            p.host = truth;
            p.host.cfg = cfg;
            run_piece(p);
            reruns++;
            h->mt_guess = true;
        }
        // true state after the piece: its writes over the true state before it
        for (uint32_t s2 = 0; s2 < kIcaoSlots; s2++)
            if (p.log.written[s2]) { truth.icao[s2] = p.host.icao[s2]; truth.icao_seen[s2] = p.host.icao_seen[s2]; }
        if (outer) {
            // the piece's answers now ARE the sequential run's: those it took from a slot no piece of this call had written
            // before it came from the state the call started with
            for (const IcaoLog::Lookup &q : p.log.lookups)
                if (!outer->written[icao_slot(q.addr)]) outer->lookups.push_back(q);
            for (uint32_t s2 = 0; s2 < kIcaoSlots; s2++)
                if (p.log.written[s2]) outer->written[s2] = true;
        }
    }
    const double t3 = now();
    // merge: whitelist, counters, text
    memcpy(h->icao, truth.icao, sizeof h->icao);
    memcpy(h->icao_seen, truth.icao_seen, sizeof h->icao_seen);
    uint64_t total = 0, msgs = 0, stored = 0;
    bool full = false;
    std::vector<uint64_t> at(P, 0), take(P, 0);                                  // where a piece's text goes, how much of it
    for (size_t t = 0; t < P; t++) {
        Piece &p = *pieces[t];
        h->st.valid_preamble += p.host.st.valid_preamble;
        h->st.out_of_phase += p.host.st.out_of_phase;
        h->st.demodulated += p.host.st.demodulated;
        h->st.goodcrc += p.host.st.goodcrc;
        h->st.badcrc += p.host.st.badcrc;
        h->st.fixed += p.host.st.fixed;
        h->st.single_bit_fix += p.host.st.single_bit_fix;
        h->st.two_bits_fix += p.host.st.two_bits_fix;
        // what the header promises when the listing outgrows `cap`: whole lines, as many as fit, nothing behind them -
        // the first piece that does not fit contributes the whole lines of its beginning, every later piece nothing
        if (out && !full) {
            uint64_t n = p.stored;                                                // (== p.nbytes unless the piece's own buffer was too small: never, by its size)
            if (p.stored < p.nbytes || total + n + 1 > cap) {
                full = true;
                if (total + n + 1 > cap) n = cap > total + 1 ? cap - total - 1 : 0;
                while (n > 0 && p.text[n - 1] != '\n') n--;
            }
            at[t] = total;
            take[t] = n;
            stored = total + n;
        }
        total += p.nbytes;
        msgs += p.msgs;
    }
    // the copies themselves on the workers (the first piece's text is in place): 12 MB for the 524,000 lines of a --gpus 8
    // step is 0.4 ms on one thread
    if (out) pool.run(W, [&](size_t w) {
        for (size_t t = w; t < P; t += W)
            if (take[t] && pieces[t]->text != out + at[t]) memcpy(out + at[t], pieces[t]->text, (size_t)take[t]);
    });
    if (out && stored < cap) out[stored] = 0;
    if (pl) {                                                                     // no copy at all: the pieces are the listing
        pl->n = 0;
        for (size_t t = 0; t < P; t++)
            if (pieces[t]->nbytes && pl->n < pl->cap) pl->out[pl->n++] = modes_text_piece{pieces[t]->text, pieces[t]->nbytes};
    }
    if (nbytes) *nbytes = total;
    if (dbg) {
        double wake = 0, guess = 0, wait = 0, run = 0;
        for (size_t t = 0; t < P; t++) {
            const Piece &p = *pieces[t];
            wake = std::max(wake, p.t_begin - t0); guess = std::max(guess, p.t_guessed - p.t_begin);
            wait = std::max(wait, p.t_started - p.t_guessed); run = std::max(run, p.t_done - p.t_started);
        }
        fprintf(stderr, "resolve_raw_mt: %zu pieces in %u segment(s), %llu records: guess + speculative %.3f ms (slowest piece: woke after %.3f, guess %.3f, "
                        "start state %.3f, resolve %.3f), confirm %.3f ms (%d re-run), merge %.3f ms\n",
                P, nsegs, (unsigned long long)nrecs, t2 - t0, wake, guess, wait, run, t3 - t2, reruns, now() - t3);
    }
    return msgs;
}

uint64_t modes_host_resolve_raw_mt(modes_host *h, const modes_record *recs, uint64_t nrecs, char *out, uint64_t cap,
                                   uint64_t *nbytes, int threads) {
    return resolve_raw_mtv_impl(h, &recs, &nrecs, 1, out, cap, nbytes, threads, nullptr);
}

uint64_t modes_host_resolve_raw_mtv(modes_host *h, const modes_record *const *segs, const uint64_t *seg_nrecs, uint32_t nsegs,
                                    char *out, uint64_t cap, uint64_t *nbytes, int threads) {
    return resolve_raw_mtv_impl(h, segs, seg_nrecs, nsegs, out, cap, nbytes, threads, nullptr);
}

// The listing where its pieces were written: nothing is gathered into one buffer (a third of the multi-threaded call's memory
// traffic) - a host that prints hands the pieces to fwrite / writev in order.
uint64_t modes_host_resolve_raw_pieces(modes_host *h, const modes_record *const *segs, const uint64_t *seg_nrecs, uint32_t nsegs,
                                       modes_text_piece *pieces, uint32_t piece_cap, uint32_t *npieces, uint64_t *nbytes, int threads) {
    PieceList pl{pieces, pieces ? piece_cap : 0, 0};
    const uint64_t msgs = resolve_raw_mtv_impl(h, segs, seg_nrecs, nsegs, nullptr, 0, nbytes, threads, nullptr, &pl);
    if (npieces) *npieces = pl.n;
    return msgs;
}

// ---------------------------------------------------------------------------------------------
// Resolve on the ranks that demodulated (include/modes_host.h; dump1090_amd/distributed.py RankResolve is the
// protocol): the speculation of modes_host_resolve_raw_mt one level up - a rank is a piece, its guess and its log
// travel instead of its records.
// ---------------------------------------------------------------------------------------------
static_assert(MODES_ICAO_SLOTS == kIcaoSlots, "the header's table size is the whitelist's");

void modes_host_get_whitelist(const modes_host *h, uint32_t *addr, int64_t *seen) {
    memcpy(addr, h->icao, sizeof h->icao);
    memcpy(seen, h->icao_seen, sizeof h->icao_seen);
}

void modes_host_set_whitelist(modes_host *h, const uint32_t *addr, const int64_t *seen) {
    memcpy(h->icao, addr, sizeof h->icao);
    memcpy(h->icao_seen, seen, sizeof h->icao_seen);
}

void modes_host_whitelist_guess(const modes_host *h, const modes_record *const *segs, const uint64_t *seg_nrecs, uint32_t nsegs,
                                uint32_t *guess, int threads) {
    for (uint32_t s = 0; s < kIcaoSlots; s++) guess[s] = MODES_ICAO_NONE;
    uint64_t nrecs = 0;
    for (uint32_t g = 0; g < nsegs; g++) nrecs += seg_nrecs[g];
    if (nrecs == 0) return;
    // pieces in stream order (any cut will do: the lists are only read); a later piece's address wins its slot
    int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    if (T > modes_host_cpu_budget()) T = modes_host_cpu_budget();
    if ((uint64_t)T > nrecs / 4096 + 1) T = (int)(nrecs / 4096 + 1);
    const uint64_t share = (nrecs + (uint64_t)T - 1) / (uint64_t)T;
    WorkerPool &pool = WorkerPool::instance();
    std::lock_guard<std::mutex> turn(pool.turn());
    std::vector<std::unique_ptr<Piece>> &pieces = piece_store();
    size_t P = 0;
    for (uint32_t g = 0; g < nsegs; g++)
        for (uint64_t lo = 0; lo < seg_nrecs[g]; lo += share) {
            if (pieces.size() <= P) pieces.emplace_back(new Piece);
            Piece &p = *pieces[P++];
            p.recs = segs[g];
            p.lo = lo;
            p.hi = lo + share < seg_nrecs[g] ? lo + share : seg_nrecs[g];
        }
    const modes_host_config cfg = h->cfg;
    // (more segments than the pool has workers for: the pieces of one worker in turn)
    const size_t W = P < 64 ? P : 64;
    pool.run(W, [&](size_t w) { for (size_t t = w; t < P; t += W) guess_piece(*pieces[t], cfg); });
    for (size_t t = 0; t < P; t++)
        for (uint32_t s = 0; s < kIcaoSlots; s++)
            if (pieces[t]->guess_set[s]) guess[s] = pieces[t]->guess_addr[s];
}

uint64_t modes_host_resolve_raw_spec(modes_host *h, const modes_record *const *segs, const uint64_t *seg_nrecs, uint32_t nsegs,
                                     char *out, uint64_t cap, uint64_t *nbytes, int threads,
                                     uint8_t *written, modes_icao_lookup *lookups, uint64_t lookup_cap, uint64_t *nlookups) {
    IcaoLog log;
    const uint64_t msgs = resolve_raw_mtv_impl(h, segs, seg_nrecs, nsegs, out, cap, nbytes, threads, &log);
    if (written)
        for (uint32_t s = 0; s < kIcaoSlots; s++) written[s] = log.written[s] ? 1 : 0;
    const uint64_t n = log.lookups.size();
    if (lookups)
        for (uint64_t i = 0; i < n && i < lookup_cap; i++) lookups[i] = modes_icao_lookup{log.lookups[i].addr, log.lookups[i].known ? 1u : 0u};
    if (nlookups) *nlookups = n;
    return msgs;
}

// The general resolve (any sink, candidates for --stats) with the log a later confirmation needs - what --resolve-on-ranks runs for
// the modes the lean --raw resolve does not serve (--stats: counters only; --onlyaddr / --raw-net: the sink formats).  One thread.
uint64_t modes_host_resolve_spec(modes_host *h, const modes_record *recs, uint64_t nrecs, const uint64_t *cands, uint64_t ncand,
                                 modes_sink_fn sink, void *user, uint8_t *written, modes_icao_lookup *lookups, uint64_t lookup_cap,
                                 uint64_t *nlookups) {
    IcaoLog log;
    uint64_t calls;
    {
        struct Log { modes_host *h; IcaoLog *was; ~Log() { h->log = was; } } keep{h, h->log};
        h->log = &log;
        calls = modes_host_resolve(h, recs, nrecs, cands, ncand, sink, user);
    }
    if (written)
        for (uint32_t s = 0; s < kIcaoSlots; s++) written[s] = log.written[s] ? 1 : 0;
    const uint64_t n = log.lookups.size();
    if (lookups)
        for (uint64_t i = 0; i < n && i < lookup_cap; i++) lookups[i] = modes_icao_lookup{log.lookups[i].addr, log.lookups[i].known ? 1u : 0u};
    if (nlookups) *nlookups = n;
    return calls;
}

// The counters as they stand (a speculative resolve that has to be repeated starts from the counters it found: modes_host_get_stats
// before, this after; a negative valid_preamble - "no candidates seen" - restores as 0).
void modes_host_set_stats(modes_host *h, const modes_host_stats *st) {
    h->st = *st;
    if (h->st.valid_preamble < 0) h->st.valid_preamble = 0;
}

int modes_host_whitelist_check(const modes_host *h, const modes_icao_lookup *lookups, uint64_t n) {
    modes_host probe = *h;                                                        // (no log: a check is not a lookup of the run)
    probe.log = nullptr;
    for (uint64_t i = 0; i < n; i++)
        if (icao_known(&probe, lookups[i].addr) != (lookups[i].known != 0)) return 0;
    return 1;
}

// What the kernels leave in modes_attempt.cls / .slot, for records of another producer (the oracle in the CPU tests, a
// capture replayed from disk): the same modes_classify() on the host, in place.
void modes_host_classify(const modes_host_config *cfg, modes_record *recs, uint64_t nrecs) {
    const uint32_t fix = cfg->fix_errors ? 1u : 0u, aggressive = cfg->aggressive ? 1u : 0u;
    for (uint64_t i = 0; i < nrecs; i++)
        for (int a = 0; a < 2; a++) {
            modes_attempt &at = recs[i].att[a];
            const uint32_t cls = modes_classify(at.msg[0] >> 3, at.errors, at.gate_ok, at.syndrome, at.nfix, fix, aggressive);
            at.cls = (uint8_t)cls;
            at.slot = (uint16_t)modes_class_slot(cls, at.msg[1], at.msg[2], at.msg[3], at.syndrome);
        }
}

// '*' + two hex digits per byte + ";\n".  A byte -> its two digits through a 256-entry table (one 2-byte store per message
// byte): the line formatter is half of the --raw resolve's time per message (rank 0 of an 8-GPU run formats 524,000 lines per
// step), and a digit-by-digit loop was 32 ns of its 62.
namespace {
struct HexPairs {
    char lower[256][2], upper[256][2];
    HexPairs() {
        const char *lo = "0123456789abcdef", *up = "0123456789ABCDEF";
        for (int b = 0; b < 256; b++) {
            lower[b][0] = lo[b >> 4]; lower[b][1] = lo[b & 15];
            upper[b][0] = up[b >> 4]; upper[b][1] = up[b & 15];
        }
    }
};
const HexPairs kHex;
inline int format_hex_line(const struct modesMessage *mm, char *buf, const char (*pairs)[2]) {
    const int nbytes = mm->msgbits / 8;
    buf[0] = '*';
    char *o = buf + 1;
    for (int b = 0; b < nbytes; b++, o += 2) memcpy(o, pairs[mm->msg[b]], 2);
    o[0] = ';';
    o[1] = '\n';
    o[2] = 0;
    return 3 + 2 * nbytes;
}
}  // namespace
int modes_format_raw(const struct modesMessage *mm, char *buf) {                  // :1324-1326
    return format_hex_line(mm, buf, kHex.lower);
}
int modes_format_raw_net(const struct modesMessage *mm, char *buf) {              // modesSendRawOutput, :2381-2393
    return format_hex_line(mm, buf, kHex.upper);
}

int modes_format_onlyaddr(const struct modesMessage *mm, char *buf) {             // :1319
    return snprintf(buf, 16, "%02x%02x%02x\n", mm->aa1, mm->aa2, mm->aa3);
}

int modes_format_stats(const modes_host_stats *st, char *buf) {                   // :2994-3005
    return snprintf(buf, 512,
                    "%lld valid preambles\n%lld demodulated again after phase correction\n"
                    "%lld demodulated with zero errors\n%lld with good crc\n%lld with bad crc\n"
                    "%lld errors corrected\n%lld single bit errors\n%lld two bits errors\n"
                    "%lld total usable messages\n",
                    (long long)st->valid_preamble, (long long)st->out_of_phase, (long long)st->demodulated,
                    (long long)st->goodcrc, (long long)st->badcrc, (long long)st->fixed, (long long)st->single_bit_fix,
                    (long long)st->two_bits_fix, (long long)(st->goodcrc + st->fixed));
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Verbose message dump: what the reference prints for one message when neither --raw nor
// --onlyaddr is given (displayModesMessage, dump1090.c:1314-1450, and the blank line
// useModesMessage adds, dump1090.c:1814).  Written as a table of text fragments plus a small
// appender; the strings are the output format and therefore have to be the reference's.
// ---------------------------------------------------------------------------------------------
namespace {

struct Out {
    char *buf;
    size_t cap, n;
    void put(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        if (n >= cap) return;
        va_list ap;
        va_start(ap, fmt);
        const int k = vsnprintf(buf + n, cap - n, fmt, ap);
        va_end(ap);
        if (k > 0) n += (size_t)k < cap - n ? (size_t)k : cap - n - 1;
    }
};

// dump1090.c:1033-1054
const char *const kCapability[8] = {
    "Level 1 (Survillance Only)",
    "Level 2 (DF0,4,5,11)",
    "Level 3 (DF0,4,5,11,20,21)",
    "Level 4 (DF0,4,5,11,20,21,24)",
    "Level 2+3+4 (DF0,4,5,11,20,21,24,code7 - is on ground)",
    "Level 2+3+4 (DF0,4,5,11,20,21,24,code7 - is on airborne)",
    "Level 2+3+4 (DF0,4,5,11,20,21,24,code7)",
    "Level 7 ???"};
const char *const kFlightStatus[8] = {
    "Normal, Airborne",
    "Normal, On the ground",
    "ALERT,  Airborne",
    "ALERT,  On the ground",
    "ALERT & Special Position Identification. Airborne or Ground",
    "Special Position Identification. Airborne or Ground",
    "Value 6 is not assigned",
    "Value 7 is not assigned"};
const char *const kAircraftType[4] = {"Aircraft Type D", "Aircraft Type C", "Aircraft Type B", "Aircraft Type A"};

// getMEDescription, dump1090.c:1060-1086: (type range, subtype range) -> name
struct MeName { int t0, t1, s0, s1; const char *name; };
const MeName kMeNames[] = {
    {1, 4, 0, 7, "Aircraft Identification and Category"},
    {5, 8, 0, 7, "Surface Position"},
    {9, 18, 0, 7, "Airborne Position (Baro Altitude)"},
    {19, 19, 1, 4, "Airborne Velocity"},
    {20, 22, 0, 7, "Airborne Position (GNSS Height)"},
    {23, 23, 0, 0, "Test Message"},
    {24, 24, 1, 1, "Surface System Status"},
    {28, 28, 1, 1, "Extended Squitter Aircraft Status (Emergency)"},
    {28, 28, 2, 2, "Extended Squitter Aircraft Status (1090ES TCAS RA)"},
    {29, 29, 0, 1, "Target State and Status Message"},
    {31, 31, 0, 1, "Aircraft Operational Status Message"},
};
const char *me_name(int type, int sub) {
    for (const MeName &m : kMeNames)
        if (type >= m.t0 && type <= m.t1 && sub >= m.s0 && sub <= m.s1) return m.name;
    return "Unknown";
}

// decodeMovementField, dump1090.c:2056-2066: ground speed in knots (truncated), -1 = not available.
// Piecewise linear: {last code of the segment, first code, knots per step, knots at the first code}.
int movement_knots(int movement) {
    if (movement == 0) return -1;
    if (movement == 1) return 0;
    static const struct { int last, first; double step, base; } seg[] = {
        {8, 2, 0.125, 0.125}, {12, 9, 0.25, 1}, {38, 13, 0.5, 2}, {93, 39, 1, 15}, {108, 94, 2, 70}, {123, 109, 5, 100}};
    for (const auto &g : seg)
        if (movement <= g.last) return (int)((movement - g.first) * g.step + g.base);
    return 175;
}

const char *unit_name(const struct modesMessage *mm) { return mm->unit == MODES_UNIT_METERS ? "meters" : "feet"; }

void put_squitter(Out &o, const struct modesMessage *mm) {                       // dump1090.c:1382-1433
    const int t = mm->metype, sub = mm->mesub;
    if (t >= 1 && t <= 4) {
        o.put("    Aircraft Type  : %s\n", kAircraftType[mm->aircraft_type & 3]);
        o.put("    Identification : %s\n", mm->flight);
    } else if (t >= 5 && t <= 18) {
        o.put("    F flag   : %s\n", mm->fflag ? "odd" : "even");
        o.put("    T flag   : %s\n", mm->tflag ? "UTC" : "non-UTC");
        if (t <= 8) {
            o.put("    Movement : %d", mm->movement);
            if (mm->movement_valid) o.put(" (%d kt)\n", movement_knots(mm->movement));
            else o.put(" (not available)\n");
            o.put("    Track    : %d degrees%s\n", mm->ground_track, mm->ground_track_valid ? "" : " (not valid)");
        } else {
            o.put("    Altitude : %d feet\n", mm->altitude);
        }
        o.put("    Latitude : %d (not decoded)\n", mm->raw_latitude);
        o.put("    Longitude: %d (not decoded)\n", mm->raw_longitude);
    } else if (t == 19 && sub >= 1 && sub <= 4) {
        if (sub <= 2) {
            o.put("    EW direction      : %d\n", mm->ew_dir);
            o.put("    EW velocity       : %d\n", mm->ew_velocity);
            o.put("    NS direction      : %d\n", mm->ns_dir);
            o.put("    NS velocity       : %d\n", mm->ns_velocity);
            o.put("    Vertical rate src : %d\n", mm->vert_rate_source);
            o.put("    Vertical rate sign: %d\n", mm->vert_rate_sign);
            o.put("    Vertical rate     : %d\n", mm->vert_rate);
        } else {                                                                   // no newlines in the reference either
            o.put("    Heading status: %d", mm->heading_is_valid);
            o.put("    Heading: %d", mm->heading);
        }
    } else {
        o.put("    Unrecognized ME type: %d subtype: %d\n", t, sub);
    }
}

}  // namespace

extern "C" int modes_format_verbose(const struct modesMessage *mm, int check_crc, char *buf, size_t cap) {
    Out o{buf, cap, 0};
    if (cap) buf[0] = 0;
    char raw[40];
    modes_format_raw(mm, raw);
    o.put("%s", raw);
    o.put("CRC: %06x (%s)\n", (int)mm->crc, mm->crcok ? "ok" : "wrong");
    if (mm->errorbit != -1) o.put("Single bit error fixed, bit %d\n", mm->errorbit);

    const int df = mm->msgtype;
    const bool altitude_reply = df == 4 || df == 20, identity_reply = df == 5 || df == 21;
    if (df == 0) {
        o.put("DF 0: Short Air-Air Surveillance.\n");
        o.put("  Altitude       : %d %s\n", mm->altitude, unit_name(mm));
        o.put("  ICAO Address   : %02x%02x%02x\n", mm->aa1, mm->aa2, mm->aa3);
    } else if (altitude_reply || identity_reply) {
        o.put("DF %d: %s, %s Reply.\n", df, df < 16 ? "Surveillance" : "Comm-B", altitude_reply ? "Altitude" : "Identity");
        o.put("  Flight Status  : %s\n", kFlightStatus[mm->fs & 7]);
        o.put("  DR             : %d\n", mm->dr);
        o.put("  UM             : %d\n", mm->um);
        if (altitude_reply) o.put("  Altitude       : %d %s\n", mm->altitude, unit_name(mm));
        else                o.put("  Squawk         : %d\n", mm->identity);
        o.put("  ICAO Address   : %02x%02x%02x\n", mm->aa1, mm->aa2, mm->aa3);
    } else if (df == 11) {
        o.put("DF 11: All Call Reply.\n");
        o.put("  Capability  : %s\n", kCapability[mm->ca & 7]);
        o.put("  ICAO Address: %02x%02x%02x\n", mm->aa1, mm->aa2, mm->aa3);
    } else if (df == 17 || df == 18) {
        if (df == 17) {
            o.put("DF 17: ADS-B message.\n");
            o.put("  Capability     : %d (%s)\n", mm->ca, kCapability[mm->ca & 7]);
        } else {
            o.put("DF 18: Extended Squitter.\n");
            o.put("  Control Field  : %d\n", mm->ca);
        }
        o.put("  ICAO Address   : %02x%02x%02x\n", mm->aa1, mm->aa2, mm->aa3);
        o.put("  Extended Squitter  Type: %d\n", mm->metype);
        o.put("  Extended Squitter  Sub : %d\n", mm->mesub);
        o.put("  Extended Squitter  Name: %s\n", me_name(mm->metype, mm->mesub));
        if (df == 17) put_squitter(o, mm);                                        // DF18 stops at the name (dump1090.c:1434-1443)
    } else if (check_crc) {
        o.put("DF %d with good CRC received (decoding still not implemented).\n", df);
    }
    o.put("\n");                                                                   // useModesMessage, dump1090.c:1814
    return (int)o.n;
}
