/* modes_oracle.c - CPU restatement of dump1090's IQ->message hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see modes_oracle.h).  Scalar, single-threaded C
 * that keeps the reference's one-position-at-a-time control flow so that it is
 * an independent check of the product's "records -> sequential resolve"
 * decomposition.  Pinned against the compiled reference: tests/test_oracle.py.
 *
 * All line numbers cite /root/reference/dump1090.c.
 */
#include "modes_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tables */

static uint16_t g_maglut[129 * 129];
static uint32_t g_crc_tab[112];

typedef struct {
    uint32_t syndrome;
    uint8_t  nbits;
    uint8_t  pos[2];       /* positions in the 112-bit frame */
} syn_entry;

#define N_SYN 5778         /* 107 single + 107*106/2 double (dump1090.c:71-75) */
static syn_entry g_syn[N_SYN];
static int g_ready;

/* dump1090.c:359-364: round(sqrt(i*i+q*q)*360) in double, i,q in [0,128]. */
void orc_build_maglut(uint16_t lut[129 * 129]) {
    for (int i = 0; i <= 128; i++)
        for (int q = 0; q <= 128; q++)
            lut[i * 129 + q] = (uint16_t)round(sqrt((double)(i * i + q * q)) * 360.0);
}

/* dump1090.c:683-698.  The reference embeds a 112-entry table; entry i is the
 * remainder of x^(111-i) modulo the Mode S generator 0x1FFF409 for the 88 data
 * bits, and 0 for the 24 parity positions.  Generated here instead of copied;
 * tests/test_oracle.py compares it with the reference's literal table. */
static void build_crc_table(void) {
    uint32_t r = 0xFFF409u;             /* x^24 mod G */
    for (int i = 87; i >= 0; i--) {
        g_crc_tab[i] = r;
        r <<= 1;
        if (r & 0x1000000u) r ^= 0x1FFF409u;
    }
    for (int i = 88; i < 112; i++) g_crc_tab[i] = 0;
}

static int syn_cmp(const void *a, const void *b) {
    uint32_t x = ((const syn_entry *)a)->syndrome, y = ((const syn_entry *)b)->syndrome;
    return (x > y) - (x < y);
}

static void flip_bit(uint8_t *msg, int pos) { msg[pos >> 3] ^= (uint8_t)(0x80u >> (pos & 7)); }

/* dump1090.c:795-841: syndromes of every 1- and 2-bit error over frame bits
 * 5..111, sorted by syndrome.  The syndrome of an error pattern is the
 * checksum of the all-zero message with those bits set. */
static void build_syndrome_table(void) {
    uint8_t z[14];
    int n = 0;
    memset(z, 0, sizeof z);
    for (int a = 5; a < 112; a++) {
        flip_bit(z, a);
        g_syn[n].syndrome = orc_checksum(z, 112);
        g_syn[n].nbits = 1; g_syn[n].pos[0] = (uint8_t)a; g_syn[n].pos[1] = 0xff;
        n++;
        for (int b = a + 1; b < 112; b++) {
            flip_bit(z, b);
            g_syn[n].syndrome = orc_checksum(z, 112);
            g_syn[n].nbits = 2; g_syn[n].pos[0] = (uint8_t)a; g_syn[n].pos[1] = (uint8_t)b;
            n++;
            flip_bit(z, b);
        }
        flip_bit(z, a);
    }
    qsort(g_syn, N_SYN, sizeof g_syn[0], syn_cmp);
}

static void ensure_ready(void) {
    if (g_ready) return;
    orc_build_maglut(g_maglut);
    build_crc_table();
    g_ready = 1;              /* orc_checksum below needs the crc table only */
    build_syndrome_table();
}

/* --------------------------------------------------------------- magnitude */

/* dump1090.c:1454-1469. */
void orc_magnitude(const uint8_t *iq, size_t nsamples, uint16_t *mag) {
    ensure_ready();
    for (size_t k = 0; k < nsamples; k++) {
        int i = (int)iq[2 * k] - 127, q = (int)iq[2 * k + 1] - 127;
        if (i < 0) i = -i;
        if (q < 0) q = -q;
        mag[k] = g_maglut[i * 129 + q];
    }
}

/* ----------------------------------------------------------------- framing */

/* dump1090.c:484-510: the reader publishes one buffer per 262144 bytes read,
 * and one more when read() hits EOF (possibly holding a partial tail). */
uint64_t orc_block_count(size_t nbytes) { return (uint64_t)(nbytes / ORC_DATA_LEN) + 1; }

/* dump1090.c:344 (initial fill 127), 481 (carry = last 476 bytes of the
 * previous buffer), 483-507 (new data then 127 padding).  Buffer k therefore
 * holds stream bytes [k*262144 - 476, (k+1)*262144), with 127 outside [0,n). */
void orc_frame_block(const uint8_t *stream, size_t nbytes, uint64_t k, uint8_t *out) {
    int64_t first = (int64_t)(k * (uint64_t)ORC_DATA_LEN) - (int64_t)ORC_CARRY_BYTES;
    for (uint32_t b = 0; b < ORC_BLOCK_BYTES; b++) {
        int64_t off = first + b;
        out[b] = (off >= 0 && (uint64_t)off < nbytes) ? stream[off] : 127;
    }
}

/* --------------------------------------------------------------------- CRC */

uint32_t orc_crc_table_entry(int i) { ensure_ready(); return g_crc_tab[i]; }

/* dump1090.c:746-753. */
int orc_len_by_type(int df) { return (df >= 16 && df <= 21) ? 112 : 56; }

/* dump1090.c:703-719: XOR of table words of the set data bits; 56-bit
 * messages use the last 56 entries. */
uint32_t orc_compute_crc(const uint8_t *msg, int bits) {
    ensure_ready();
    int base = 112 - bits;
    uint32_t c = 0;
    for (int k = 0; k < bits - 24; k++)
        if (msg[k >> 3] & (0x80u >> (k & 7))) c ^= g_crc_tab[base + k];
    return c & 0xFFFFFFu;
}

/* dump1090.c:733-742: computed parity XOR the 24 received parity bits. */
uint32_t orc_checksum(const uint8_t *msg, int bits) {
    int n = bits / 8;
    uint32_t rx = ((uint32_t)msg[n - 3] << 16) | ((uint32_t)msg[n - 2] << 8) | msg[n - 1];
    return (orc_compute_crc(msg, bits) ^ rx) & 0xFFFFFFu;
}

/* Lookup without side effects: which frame bits would dump1090.c:854-894 flip?
 * Returns count (0 if none / over maxfix / outside a short message). */
static int syndrome_lookup(uint32_t syn, int bits, int maxfix, int rel[2]) {
    ensure_ready();
    syn_entry key;
    key.syndrome = syn;
    const syn_entry *e = bsearch(&key, g_syn, N_SYN, sizeof g_syn[0], syn_cmp);
    if (!e || e->nbits > maxfix) return 0;                   /* :864-871 */
    int shift = 112 - bits;                                  /* :874     */
    for (int i = 0; i < e->nbits; i++) {
        int p = (int)e->pos[i] - shift;
        if (p < 0 || p >= bits) return 0;                    /* :877-879 */
        rel[i] = p;
    }
    return e->nbits;
}

/* dump1090.c:854-894. */
int orc_fix_bit_errors(uint8_t *msg, int bits, int maxfix, int *fixed) {
    int rel[2] = {-1, -1};
    int n = syndrome_lookup(orc_checksum(msg, bits), bits, maxfix, rel);
    for (int i = 0; i < n; i++) {
        flip_bit(msg, rel[i]);
        if (fixed) fixed[i] = rel[i];
    }
    return n;
}

/* ------------------------------------------------------------ demodulation */

/* dump1090.c:1602-1650. */
int orc_preamble_ok(const uint16_t *m) {
    /* :1602-1611 - shape of the four pulses (strict inequalities) */
    if (!(m[0] > m[1] && m[1] < m[2] && m[2] > m[3] && m[3] < m[0] &&
          m[4] < m[0] && m[5] < m[0] && m[6] < m[0] &&
          m[7] > m[8] && m[8] < m[9] && m[9] > m[6])) return 0;
    /* :1624-1626, :1639-1642 - quiet samples below 1/6 of the pulse sum */
    int level = ((int)m[0] + m[2] + m[7] + m[9]) / 6;
    if (m[4] >= level || m[5] >= level) return 0;
    if (m[11] >= level || m[12] >= level || m[13] >= level || m[14] >= level) return 0;
    return 1;
}

size_t orc_block_candidates(const uint16_t *m, uint32_t mlen, uint32_t *js, size_t cap) {
    size_t n = 0;
    if (mlen < ORC_FRAME_SAMPLES) return 0;
    for (uint32_t j = 0; j < mlen - ORC_FRAME_SAMPLES; j++)      /* :1593 */
        if (orc_preamble_ok(m + j)) { if (n < cap) js[n] = j; n++; }
    return n;
}

/* dump1090.c:1473-1476. */
static uint16_t scale_sample(uint16_t v, uint16_t factor) {
    uint32_t r = (uint32_t)v * factor / 16384u;
    return r > 65535u ? 65535u : (uint16_t)r;
}

/* dump1090.c:1498-1558 applied to a copy of m[16..239] (win[t-16] <-> m[t]).
 * The chain reads samples it has already rewritten, exactly as the in-place
 * original does. */
void orc_phase_corrected_window(const uint16_t *m, uint16_t win[224]) {
    memcpy(win, m + 16, 224 * sizeof(uint16_t));
    uint32_t on_time = (uint32_t)m[0] + m[2] + m[7] + m[9];          /* :1502 */
    uint32_t early   = ((uint32_t)m[-1] + m[6]) * 2;                 /* :1506 */
    uint32_t late    = ((uint32_t)m[3] + m[10]) * 2;                 /* :1510 */
#define W(t) win[(t) - 16]
    if (early > late) {                                              /* :1512 */
        uint16_t up = (uint16_t)(16384u + 16384u * early / (early + on_time));
        uint16_t dn = (uint16_t)(16384u - 16384u * early / (early + on_time));
        W(239) = scale_sample(W(239), up);                           /* :1519 */
        for (int t = 238; t > 16; t -= 2)                            /* :1523 */
            W(t - 1) = scale_sample(W(t - 1), (W(t) > W(t + 1)) ? dn : up);
    } else {
        uint16_t up = (uint16_t)(16384u + 16384u * late / (late + on_time));
        uint16_t dn = (uint16_t)(16384u - 16384u * late / (late + on_time));
        W(16) = scale_sample(W(16), up);                             /* :1542 */
        for (int t = 16; t < 238; t += 2)                            /* :1545 */
            W(t + 2) = scale_sample(W(t + 2), (W(t) > W(t + 1)) ? up : dn);
    }
#undef W
}

/* dump1090.c:1668-1706: slice 112 bit pairs from win[0..223] with the
 * "weak pair repeats the previous bit" rule, then pack MSB-first.  A bit value
 * of 2 (first pair equal) is OR-ed in unmasked, as the reference's
 * `bits[i]<<7 | ...` stored to unsigned char does.  Returns `errors`. */
static int slice_and_pack(const uint16_t *win, uint8_t msg[14]) {
    uint8_t bit[112];
    int errors = 0;
    for (int k = 0; k < 112; k++) {
        int lo = win[2 * k], hi = win[2 * k + 1];
        int d = lo > hi ? lo - hi : hi - lo;
        if (k > 0 && d < 256)      bit[k] = bit[k - 1];              /* :1675 */
        else if (lo == hi)       { bit[k] = 2; if (k < 56) errors++; } /* :1677 */
        else                       bit[k] = lo > hi;                 /* :1683 */
    }
    for (int b = 0; b < 14; b++) {
        unsigned v = 0;
        for (int t = 0; t < 8; t++) v |= (unsigned)bit[8 * b + t] << (7 - t);
        msg[b] = (uint8_t)v;
    }
    return errors;
}

/* dump1090.c:1708-1723: mean |lo-hi| over the message's own length, measured
 * on the UNcorrected samples; threshold 10*255. */
static int noise_gate_ok(const uint16_t *raw_win, const uint8_t msg[14]) {
    int nbytes = orc_len_by_type(msg[0] >> 3) / 8;
    int sum = 0;
    for (int k = 0; k < nbytes * 8; k++) {
        int d = (int)raw_win[2 * k] - (int)raw_win[2 * k + 1];
        sum += d < 0 ? -d : d;
    }
    return sum / (nbytes * 4) >= 10 * 255;
}

static void fill_attempt(const uint16_t *raw_win, const uint16_t *win, int maxfix, orc_attempt *a) {
    memset(a, 0, sizeof *a);
    a->errors  = (uint8_t)slice_and_pack(win, a->msg);
    a->gate_ok = (uint8_t)noise_gate_ok(raw_win, a->msg);
    int df = a->msg[0] >> 3, bits = orc_len_by_type(df);
    a->syndrome = orc_checksum(a->msg, bits);
    a->fixpos[0] = a->fixpos[1] = 0xff;
    /* dump1090.c:1112-1117: only DF11/17/18 with a non-zero syndrome are looked up */
    if (a->syndrome != 0 && maxfix > 0 && (df == 11 || df == 17 || df == 18)) {
        int rel[2] = {-1, -1};
        int n = syndrome_lookup(a->syndrome, bits, maxfix, rel);
        a->nfix = (uint8_t)n;
        for (int i = 0; i < n; i++) a->fixpos[i] = (uint8_t)rel[i];
    }
}

void orc_record_at(const uint16_t *m, uint32_t j, int maxfix, orc_record *out) {
    ensure_ready();
    uint16_t win[224];
    const uint16_t *raw = m + j + 16;
    out->j = j;
    fill_attempt(raw, raw, maxfix, &out->att[0]);
    if (j != 0) orc_phase_corrected_window(m + j, win);              /* :1660 */
    else        memcpy(win, raw, sizeof win);
    fill_attempt(raw, win, maxfix, &out->att[1]);
}

/* ------------------------------------------------------- stateful decoding */

#define ICAO_SLOTS 1024                                              /* :65 */

struct orc_state {
    orc_config cfg;
    orc_stats  st;
    uint32_t   icao[ICAO_SLOTS];     /* address per slot; 0 = empty          */
};

orc_state *orc_state_new(const orc_config *cfg) {
    ensure_ready();
    orc_state *s = calloc(1, sizeof *s);
    if (s) s->cfg = *cfg;
    return s;
}
void orc_state_free(orc_state *s) { free(s); }
void orc_state_stats(const orc_state *s, orc_stats *out) { *out = s->st; }

/* dump1090.c:898-905. */
static uint32_t icao_slot(uint32_t a) {
    a = ((a >> 16) ^ a) * 0x45d9f3bu;
    a = ((a >> 16) ^ a) * 0x45d9f3bu;
    a = ((a >> 16) ^ a);
    return a & (ICAO_SLOTS - 1);
}
/* dump1090.c:910-914 / 919-925 with the clock frozen (see header). */
static void icao_remember(orc_state *s, uint32_t addr) { s->icao[icao_slot(addr)] = addr; }
static int  icao_known(const orc_state *s, uint32_t addr) {
    return addr != 0 && s->icao[icao_slot(addr)] == addr;
}

/* The CRC / repair / whitelist part of decodeModesMessage (dump1090.c:1094-1128,
 * 1136-1138, 1181-1210).  Field decoding beyond what --raw/--onlyaddr print is
 * not needed to pin the path and is left out. */
static void decode_message(orc_state *s, const uint8_t raw[14], orc_message *mm) {
    memset(mm, 0, sizeof *mm);
    memcpy(mm->msg, raw, 14);
    uint8_t *msg = mm->msg;
    mm->msgtype = msg[0] >> 3;                                        /* :1099 */
    mm->msgbits = orc_len_by_type(mm->msgtype);                       /* :1100 */
    mm->crc = orc_checksum(msg, mm->msgbits);                         /* :1104 */
    mm->errorbit = -1;
    mm->crcok = (mm->crc == 0);
    int df = mm->msgtype;
    if (!mm->crcok && s->cfg.fix_errors && (df == 11 || df == 17 || df == 18)) {
        int fixed[2];
        int n = orc_fix_bit_errors(msg, mm->msgbits, s->cfg.aggressive ? 2 : 1, fixed);
        if (n > 0) {                                                  /* :1118 */
            mm->crc = orc_checksum(msg, mm->msgbits);
            mm->crcok = (mm->crc == 0);
            mm->errorbit = fixed[0];
            if (n == 1) s->st.single_bit_fix++; else s->st.two_bits_fix++;
        }
    }
    mm->aa1 = msg[1]; mm->aa2 = msg[2]; mm->aa3 = msg[3];             /* :1136 */
    if (df != 11 && df != 17 && df != 18) {                           /* :1183 */
        /* bruteForceAP, dump1090.c:942-983 */
        mm->crcok = 0;
        if (df == 0 || df == 4 || df == 5 || df == 16 || df == 20 || df == 21 || df == 24) {
            int last = mm->msgbits / 8 - 1;
            uint32_t c = orc_compute_crc(msg, mm->msgbits);
            uint32_t b0 = msg[last] ^ (c & 0xff), b1 = msg[last - 1] ^ ((c >> 8) & 0xff),
                     b2 = msg[last - 2] ^ ((c >> 16) & 0xff);
            uint32_t addr = b0 | (b1 << 8) | (b2 << 16);
            if (icao_known(s, addr)) {
                mm->aa1 = (int32_t)b2; mm->aa2 = (int32_t)b1; mm->aa3 = (int32_t)b0;
                mm->crcok = 1;
            }
        }
    } else {
        uint32_t addr = ((uint32_t)mm->aa1 << 16) | ((uint32_t)mm->aa2 << 8) | (uint32_t)mm->aa3;
        if (mm->crcok && mm->errorbit == -1) icao_remember(s, addr);  /* :1198 */
        if (df == 11 && !mm->crcok && mm->crc < 80 && icao_known(s, addr)) {
            mm->iid = (int32_t)mm->crc;                               /* :1204 */
            mm->crcok = 1;
        }
    }
    mm->phase_corrected = 0;
}

static void sink(const orc_state *s, const orc_message *mm, orc_message *msgs, size_t cap, size_t *n) {
    /* dump1090.c:1803 (the --stats suppression is the caller's business). */
    if (s->cfg.check_crc == 0 || mm->crcok) {
        if (*n < cap) msgs[*n] = *mm;
        (*n)++;
    }
}

/* dump1090.c:1563-1793.  Same loop shape as the reference: one position at a
 * time, `retry` is the reference's use_correction. */
void orc_detect_block(orc_state *s, const uint16_t *m, uint32_t mlen, uint32_t block,
                      orc_message *msgs, size_t cap, size_t *nmsgs) {
    uint16_t win[224];
    uint8_t  raw[14];
    int retry = 0;
    for (uint32_t j = 0; j < mlen - ORC_FRAME_SAMPLES; j++) {            /* :1593 */
        const uint16_t *p = m + j;
        if (!retry) {
            if (!orc_preamble_ok(p)) continue;                           /* :1602-1650 */
            s->st.valid_preamble++;                                      /* :1651 */
            memcpy(win, p + 16, sizeof win);
        } else {
            if (j != 0) {                                                /* :1660 */
                orc_phase_corrected_window(p, win);
                s->st.out_of_phase++;
            } else {
                memcpy(win, p + 16, sizeof win);
            }
        }
        int errors = slice_and_pack(win, raw);                           /* :1668-1706 */
        if (!noise_gate_ok(p + 16, raw)) { retry = 0; continue; }        /* :1713-1726 */

        int good = 0;
        if (errors == 0 || (s->cfg.aggressive && errors < 3)) {          /* :1731 */
            orc_message mm;
            decode_message(s, raw, &mm);
            mm.block = block; mm.j = j;
            if (mm.crcok || retry) {                                     /* :1738-1753 */
                if (errors == 0) s->st.demodulated++;
                if (mm.errorbit == -1) {
                    if (mm.crcok) s->st.goodcrc++; else s->st.badcrc++;
                } else {
                    s->st.badcrc++; s->st.fixed++;
                    if (mm.errorbit < 112) s->st.single_bit_fix++; else s->st.two_bits_fix++;
                }
            }
            uint32_t here = j;
            if (mm.crcok) {                                              /* :1769-1774 */
                j += (8u + (uint32_t)orc_len_by_type(raw[0] >> 3)) * 2u;
                good = 1;
                if (retry) mm.phase_corrected = 1;
            }
            mm.j = here;
            sink(s, &mm, msgs, cap, nmsgs);                              /* :1777 */
        }
        if (!good && !retry) { j--; retry = 1; }                         /* :1786-1791 */
        else retry = 0;
    }
}

int orc_run_stream(const orc_config *cfg, const uint8_t *stream, size_t nbytes,
                   orc_message *msgs, size_t cap, size_t *nmsgs, orc_stats *stats) {
    orc_state *s = orc_state_new(cfg);
    uint8_t  *buf = malloc(ORC_BLOCK_BYTES);
    uint16_t *mag = malloc(ORC_BLOCK_SAMPLES * sizeof(uint16_t));
    size_t n = 0;
    if (!s || !buf || !mag) { free(s); free(buf); free(mag); return -1; }
    uint64_t nblocks = orc_block_count(nbytes);
    for (uint64_t k = 0; k < nblocks; k++) {                             /* :2969-2990 */
        orc_frame_block(stream, nbytes, k, buf);
        orc_magnitude(buf, ORC_BLOCK_SAMPLES, mag);                      /* :2974 */
        orc_detect_block(s, mag, ORC_BLOCK_SAMPLES, (uint32_t)k, msgs, cap, &n); /* :2986 */
    }
    if (nmsgs) *nmsgs = n;
    if (stats) *stats = s->st;
    free(buf); free(mag); orc_state_free(s);
    return 0;
}

/* dump1090.c:1324-1326. */
int orc_format_raw(const orc_message *mm, char *buf) {
    static const char hex[] = "0123456789abcdef";
    int n = 0;
    buf[n++] = '*';
    for (int b = 0; b < mm->msgbits / 8; b++) {
        buf[n++] = hex[mm->msg[b] >> 4];
        buf[n++] = hex[mm->msg[b] & 15];
    }
    buf[n++] = ';'; buf[n++] = '\n'; buf[n] = 0;
    return n;
}
