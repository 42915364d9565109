/* modes_core.h - arithmetic shared by the gfx950 kernels (modes_gfx950.hip) and
 * the host resolve (modes_host.cpp).  Header-only, no HIP includes: every
 * function is MODES_HD (= __host__ __device__ under hipcc, nothing under g++), so
 * the exact code the kernels run can also be unit-tested on a CPU
 * (tests/native/core_shim.cpp) - that shim is test scaffolding, not a fallback:
 * the product's GPU entry points never route through host copies of these.
 *
 * Line numbers cite /root/reference/dump1090.c.
 */
#ifndef MODES_CORE_H
#define MODES_CORE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MODES_HD  __host__ __device__ __forceinline__
#define MODES_HDM __host__ __device__ __forceinline__      /* member functions */
#else
#define MODES_HD  static inline
#define MODES_HDM inline
#endif

/* ------------------------------------------------------------------ CRC-24 */

/* Mode S generator polynomial, x^24 + ... (the reference's table
 * dump1090.c:683-698 is x^(111-i) mod this; tests/test_oracle.py). */
#define MODES_CRC_POLY 0x1FFF409u

/* modesMessageLenByType, dump1090.c:746-753. */
MODES_HD int modes_len_by_df(int df) { return (df >= 16 && df <= 21) ? 112 : 56; }

/* modesChecksum (dump1090.c:733-742): the table-XOR of the data bits XOR the
 * received parity equals the remainder of the whole message (data + parity)
 * modulo the generator, computed here bitwise, MSB first. */
MODES_HD uint32_t modes_syndrome(const uint8_t *msg, int nbytes) {
    uint32_t r = 0;
    for (int b = 0; b < nbytes; b++) {
        for (int t = 7; t >= 0; t--) {          /* plain (non-augmented) long division */
            r = (r << 1) | ((uint32_t)(msg[b] >> t) & 1u);
            if (r & 0x1000000u) r ^= MODES_CRC_POLY;
        }
    }
    return r & 0xFFFFFFu;
}

/* Syndrome of a single flipped bit at frame position p of a 112-bit frame:
 * x^(111-p) mod G.  A 56-bit message's bit k is frame position k+56
 * (dump1090.c:874-880). */
MODES_HD uint32_t modes_bit_syndrome(int p) {
    uint32_t r = 1;
    for (int e = 0; e < 111 - p; e++) {
        r <<= 1;
        if (r & 0x1000000u) r ^= MODES_CRC_POLY;
    }
    return r;
}

/* fixBitErrors' table lookup (dump1090.c:795-841 builds it, 854-880 queries it)
 * without the table: the table holds the syndromes of all 1- and 2-bit error
 * patterns over frame bits 5..111, all 5778 of them distinct, so "bsearch the
 * syndrome, then require <= maxfix bits and every bit inside the message" is
 * the same as searching only patterns that satisfy those two conditions.
 * `esyn` = the 112 single-bit syndromes (modes_bit_syndrome).  Writes
 * message-relative positions; returns the number of bits (0 = no repair). */
MODES_HD int modes_find_fix(uint32_t syndrome, int bits, int maxfix, const uint32_t *esyn, uint8_t pos[2]) {
    const int first = (bits == 112) ? 5 : 56;       /* frame bits usable by this length */
    const int shift = 112 - bits;
    pos[0] = pos[1] = 0xff;
    if (maxfix < 1 || syndrome == 0) return 0;
    for (int p = first; p < 112; p++)
        if (esyn[p] == syndrome) { pos[0] = (uint8_t)(p - shift); return 1; }
    if (maxfix < 2) return 0;
    for (int p = first; p < 111; p++) {
        uint32_t want = syndrome ^ esyn[p];
        for (int q = p + 1; q < 112; q++)
            if (esyn[q] == want) { pos[0] = (uint8_t)(p - shift); pos[1] = (uint8_t)(q - shift); return 2; }
    }
    return 0;
}

/* --------------------------------------------- what the decoder makes of an attempt */

/* The MODES_CLS_* bytes and their meaning are part of the ABI: include/modes_gfx950.h. */
#include "../../include/modes_gfx950.h"

/* ICAOCacheHashAddress, dump1090.c:898-905 (MODES_ICAO_CACHE_LEN = 1024, dump1090.c:65). */
MODES_HD uint32_t modes_icao_slot(uint32_t a) {
    a = ((a >> 16) ^ a) * 0x45d9f3bu;
    a = ((a >> 16) ^ a) * 0x45d9f3bu;
    a = ((a >> 16) ^ a);
    return a & 1023u;
}

/* Which way an attempt goes through detectModeS()/decodeModesMessage(), as far as the whitelist does not decide it:
 * df = msg[0] >> 3 AS DEMODULATED (dump1090.c:1099: the type is read before any repair and never again), errors / gate_ok /
 * syndrome / nfix as in modes_attempt (nfix found with maxfix = fix ? (aggressive ? 2 : 1) : 0).
 *   :1723-1726  gate failed -> GATE;   :1731  errors > 0 (>= 3 with --aggressive) -> SKIP (not decoded)
 *   :1183       DF other than 11/17/18 -> crcok only through bruteForceAP (:942-983; the types of :948-950) -> AP, else BAD
 *   :1105       syndrome 0 -> CLEAN;   :1112-1128  repaired -> FIXED (the repaired frame's checksum is 0 by construction)
 *   :1204       DF11, syndrome < 80 -> IID;  otherwise BAD.
 * The one definition: the kernels (store_attempt) and the host's fallback for records without the byte both compile it. */
MODES_HD uint32_t modes_classify(int df, uint32_t errors, uint32_t gate_ok, uint32_t syndrome, uint32_t nfix, uint32_t fix,
                                 uint32_t aggressive) {
    uint32_t c = MODES_CLS_VALID | (fix ? MODES_CLS_FIX : 0u) | (aggressive ? MODES_CLS_AGGRESSIVE : 0u) |
                 (modes_len_by_df(df) == 112 ? MODES_CLS_LONG : 0u) | (errors == 0 ? MODES_CLS_NOERR : 0u);
    uint32_t kind;
    if (!gate_ok) kind = MODES_CLS_GATE;
    else if (!(errors == 0 || (aggressive && errors < 3))) kind = MODES_CLS_SKIP;
    else if (df != 11 && df != 17 && df != 18)
        kind = (df == 0 || df == 4 || df == 5 || df == 16 || df == 20 || df == 21 || df == 24) ? MODES_CLS_AP : MODES_CLS_BAD;
    else if (syndrome == 0) kind = MODES_CLS_CLEAN;
    else if (fix && nfix > 0 && nfix <= (aggressive ? 2u : 1u)) kind = MODES_CLS_FIXED;
    else if (df == 11 && syndrome < 80) kind = MODES_CLS_IID;
    else kind = MODES_CLS_BAD;
    return c | kind;
}
/* The whitelist slot that class touches: CLEAN / IID the address field msg[1..3], AP the address bruteForceAP recovers -
 * AP field XOR parity of the data bits (dump1090.c:960-975) = the syndrome itself.  0 for the classes that touch none. */
MODES_HD uint32_t modes_class_slot(uint32_t cls, uint32_t msg1, uint32_t msg2, uint32_t msg3, uint32_t syndrome) {
    const uint32_t kind = cls & MODES_CLS_KIND;
    if (kind == MODES_CLS_AP) return modes_icao_slot(syndrome);
    if (kind == MODES_CLS_CLEAN || kind == MODES_CLS_IID) return modes_icao_slot((msg1 << 16) | (msg2 << 8) | msg3);
    return 0u;
}

/* ------------------------------------------------------ magnitude / power */

/* s = (I-127)^2 + (Q-127)^2, 0..32768.  The reference's magnitude
 * (dump1090.c:1462-1467) is maglut[|I-127|*129 + |Q-127|] = round(360*sqrt(s)),
 * a strictly increasing function of s, so >,<,== between samples can be decided
 * on s. */
MODES_HD uint32_t modes_power(uint32_t i_byte, uint32_t q_byte) {
    int i = (int)i_byte - 127, q = (int)q_byte - 127;
    return (uint32_t)(i * i + q * q);
}
/* Index into OUR copy of the table.  The reference indexes maglut by (|I-127|, |Q-127|)
 * (129 x 129 entries); the value depends on s alone, so the device keeps it as a function table
 * of the saturated power  min(s, 32767)  - 32768 u16 = 64 KiB of LDS - and the index of a pair of
 * samples is modes_power_pair_sat: two subtracts, a multiply and a multiply-add (packed), no
 * absolute values.  s = 32768 occurs only for I = Q = 255 and 32767 is not a sum of two squares, so
 * entry 32767 holds maglut[128][128] and nothing collides. */
#define MODES_LUT_ENTRIES 32768
MODES_HD uint32_t modes_lut_index(uint32_t i_byte, uint32_t q_byte) {
    const uint32_t s = modes_power(i_byte, q_byte);
    return s > 32767u ? 32767u : s;
}

/* The reference's magnitude WITHOUT the table: maglut[...] = round(360 * sqrt(s)) (dump1090.c:362, in double) for the
 * saturated power s_sat = min(s, 32767) the kernels carry (32767 stands for 32768: I = Q = 255; 32767 itself is not a
 * sum of two squares).  360 sqrt(s) is never a half-integer, so m is the one integer with (m - 1/2)^2 < 129600 s <
 * (m + 1/2)^2, i.e. m^2 - m < N <= m^2 + m for N = 129600 s (< 2^32: 32-bit arithmetic throughout).  A single-precision
 * square root is within one of m; the two comparisons settle it.  The demod kernel uses this for powers beyond its small
 * LDS table (strong signals), so that the table fits next to the scan kernel's workgroups.  Exhaustively checked against the
 * table on the host (tests/test_core.py) and on the device (tests/test_gpu_parity.py). */
MODES_HD uint32_t modes_mag_exact(uint32_t s_sat) {
    const uint32_t s = s_sat >= 32767u ? 32768u : s_sat;
    const uint32_t n = s * 129600u;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t m = (uint32_t)(__builtin_sqrtf((float)n) + 0.5f);
    const uint32_t mm = __umul24(m, m);                      /* m <= 65167: the low 32 bits of the 24-bit product are exact */
#else
    uint32_t m = (uint32_t)(__builtin_sqrtf((float)n) + 0.5f);
    const uint32_t mm = m * m;
#endif
    if (mm - m >= n && m > 0) m -= 1;                        /* N <= m^2 - m : one too many */
    else if (mm + m < n) m += 1;                             /* N >  m^2 + m : one short    */
    return m;
}

/* ---------------------------------------------------- preamble predicates */

/* The full predicate of dump1090.c:1602-1650 on true magnitudes m[0..14]. */
template <class M>
MODES_HD bool modes_preamble_exact(const M &m) {
    const int m0 = m(0), m1 = m(1), m2 = m(2), m3 = m(3), m4 = m(4), m5 = m(5), m6 = m(6), m7 = m(7),
              m8 = m(8), m9 = m(9);
    if (!(m0 > m1 && m1 < m2 && m2 > m3 && m3 < m0 && m4 < m0 && m5 < m0 && m6 < m0 && m7 > m8 && m8 < m9 &&
          m9 > m6))
        return false;
    const int level = (m0 + m2 + m7 + m9) / 6;                               /* :1624 */
    if (m4 >= level || m5 >= level) return false;                             /* :1625 */
    return m(11) < level && m(12) < level && m(13) < level && m(14) < level;  /* :1639 */
}

/* ---- packed-u16 helpers: two samples per 32-bit register (v_pk_*_u16) ---- */
#if defined(__clang__)   /* hipcc (device + host) and the clang-built test shim */

typedef unsigned short modes_u16x2 __attribute__((ext_vector_type(2)));
#define MODES_PK(x) __builtin_bit_cast(modes_u16x2, (uint32_t)(x))
#define MODES_UN(x) __builtin_bit_cast(uint32_t, (modes_u16x2)(x))
MODES_HD uint32_t pk_max(uint32_t a, uint32_t b) { return MODES_UN(__builtin_elementwise_max(MODES_PK(a), MODES_PK(b))); }
MODES_HD uint32_t pk_min(uint32_t a, uint32_t b) { return MODES_UN(__builtin_elementwise_min(MODES_PK(a), MODES_PK(b))); }
MODES_HD uint32_t pk_subs(uint32_t a, uint32_t b) { return MODES_UN(__builtin_elementwise_sub_sat(MODES_PK(a), MODES_PK(b))); }
MODES_HD uint32_t pk_add(uint32_t a, uint32_t b) { return MODES_UN(MODES_PK(a) + MODES_PK(b)); }
MODES_HD uint32_t pk_sub(uint32_t a, uint32_t b) { return MODES_UN(MODES_PK(a) - MODES_PK(b)); }
MODES_HD uint32_t pk_mul(uint32_t a, uint32_t b) { return MODES_UN(MODES_PK(a) * MODES_PK(b)); }
MODES_HD uint32_t pk_shr2(uint32_t a) { return MODES_UN(MODES_PK(a) >> (unsigned short)2); }
MODES_HD uint32_t pk_shr1(uint32_t a) { return MODES_UN(MODES_PK(a) >> (unsigned short)1); }

/* Four bytes I0 Q0 I1 Q1 (little-endian dword) -> packed (s0, s1). */
MODES_HD uint32_t modes_power_pair(uint32_t w) {
    const uint32_t k127 = 0x007F007Fu;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t ip = __builtin_amdgcn_perm(0u, w, 0x0c020c00u);   /* (I0, I1): one v_perm_b32 each */
    uint32_t qp = __builtin_amdgcn_perm(0u, w, 0x0c030c01u);   /* (Q0, Q1)                      */
#else
    uint32_t ip = w & 0x00FF00FFu;              /* (I0, I1) */
    uint32_t qp = (w >> 8) & 0x00FF00FFu;       /* (Q0, Q1) */
#endif
    uint32_t ai = pk_sub(ip, k127), aq = pk_sub(qp, k127);   /* mod 2^16; squares are exact */
    return pk_add(pk_mul(ai, ai), pk_mul(aq, aq));            /* v_pk_mul_lo + v_pk_mad      */
}

/* The level tests of dump1090.c:1624-1642 as a NECESSARY condition on powers (what modes_level_bound evaluates):
 * a quiet sample x in {4,5,11..14} needs m_x < floor((m0+m2+m7+m9)/6) with m = round(360 sqrt(s)), hence
 * 360 sqrt(s_x) - 1/2 <= (360 SUM sqrt(s_k) + 2)/6 - 1, i.e. sqrt(s_x) < SUM sqrt(s_k)/6 <= sqrt(SUM s_k)/3
 * (Cauchy-Schwarz), i.e. 9 s_x < SUM s_k.  It never rejects a position the reference accepts; the false accepts are
 * removed by modes_preamble_exact in the demodulation kernels. */

/* Saturated powers for the production scan kernel: min(s, 32767).  32768 is attained only by
 * I = Q = 255 and 32767 is not a sum of two squares, so the clamp keeps every ordering relation
 * between samples exact; it buys one spare bit for the borrow trick of modes_order8_swar.
 * On the device the clamp is free: v_pk_mad_i16 ... clamp saturates at 0x7fff. */
MODES_HD uint32_t modes_power_pair_sat(uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t k127 = 0x007F007Fu;
    const uint32_t ip = w & 0x00FF00FFu;                              /* (I0, I1): full-rate v_and    */
    const uint32_t qp = __builtin_amdgcn_perm(0u, w, 0x0c030c01u);    /* (Q0, Q1): v_perm_b32         */
    const uint32_t ai = pk_sub(ip, k127), aq = pk_sub(qp, k127);
    uint32_t t, r;
    /* one block so that the wait state between a packed producer and its consumer is kept */
    asm("v_pk_mul_lo_u16 %0, %2, %2\n\ts_nop 0\n\tv_pk_mad_i16 %1, %3, %3, %0 clamp\n\ts_nop 0"
        : "=&v"(t), "=v"(r) : "v"(ai), "v"(aq));
    return r;
#else
    const uint32_t e = modes_power_pair(w);
    const uint32_t lo = e & 0xffffu, hi = e >> 16;
    return (lo > 32767u ? 32767u : lo) | ((hi > 32767u ? 32767u : hi) << 16);
#endif
}
/* Four dwords (8 samples) at once; on the device the eight packed multiplies are interleaved so
 * that no result is consumed by the next instruction (one s_nop instead of eight). */
MODES_HD void modes_power8_sat(const uint32_t w[4], uint32_t out[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t k127 = 0x007F007Fu;
    uint32_t ai[4], aq[4], t0, t1, t2, t3;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        ai[d] = pk_sub(w[d] & 0x00FF00FFu, k127);
        aq[d] = pk_sub(__builtin_amdgcn_perm(0u, w[d], 0x0c030c01u), k127);
    }
    asm("s_nop 0\n\t"
        "v_pk_mul_lo_u16 %4, %8, %8\n\t"
        "v_pk_mul_lo_u16 %5, %9, %9\n\t"
        "v_pk_mul_lo_u16 %6, %10, %10\n\t"
        "v_pk_mul_lo_u16 %7, %11, %11\n\t"
        "v_pk_mad_i16 %0, %12, %12, %4 clamp\n\t"
        "v_pk_mad_i16 %1, %13, %13, %5 clamp\n\t"
        "v_pk_mad_i16 %2, %14, %14, %6 clamp\n\t"
        "v_pk_mad_i16 %3, %15, %15, %7 clamp\n\t"
        "s_nop 0"
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(ai[0]), "v"(ai[1]), "v"(ai[2]), "v"(ai[3]), "v"(aq[0]), "v"(aq[1]), "v"(aq[2]), "v"(aq[3]));
#else
    for (int d = 0; d < 4; d++) out[d] = modes_power_pair_sat(w[d]);
#endif
}

/* The ordering relations of dump1090.c:1602-1611 for 8 positions, production form.  Same max/min
 * tree in packed ops (half rate), but the four final compares and their conjunction
 * are plain 32-bit subtracts and ANDs (full-rate ops on gfx950, tools/ubench_valu.hip):
 *     a > b   <=>   bit 15 of (b - a) mod 2^16          for a, b <= 32767.
 * A 32-bit subtract of two packed pairs gives the low half exactly; the high half sees the low
 * half's borrow, i.e. it tests a_hi > b_hi OR (a_hi == b_hi AND a_lo > b_lo): never a false
 * negative, and the extra accepts are removed downstream like every other scan false positive.
 * r[q]: bit 15 <-> position 2q, bit 31 <-> position 2q+1 (other bits are garbage).
 * E must hold saturated powers (modes_power_pair_sat). */
#define MODES_ORDER_FLAGS 0x80008000u
MODES_HD void modes_order8_swar(const uint32_t E[12], uint32_t r[4]) {
    uint32_t O[11];
#pragma unroll
#if defined(SCAN_ABL_NOPERM)
    /* ablation build (timing only; tools/ab_scan.py): the odd-aligned pairs cost nothing - some other dword of the window stands in
     * (ten distinct samples per position still, so noise survives at the same rate): what the scan would gain if the eight v_perm
     * per chunk were not needed */
    for (int t = 0; t < 11; t++) O[t] = E[(t + 6) % 12];
#else
    for (int t = 0; t < 11; t++) O[t] = (E[t] >> 16) | (E[t + 1] << 16);
#endif
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t x0 = E[q], x1 = O[q], x2 = E[q + 1], x3 = O[q + 1], x4 = E[q + 2], x5 = O[q + 2],
                       x6 = E[q + 3], x7 = O[q + 3], x8 = E[q + 4], x9 = O[q + 4];
        const uint32_t m13 = pk_max(x1, x3);
        const uint32_t big = pk_max(pk_max(m13, pk_max(x4, x5)), x6);
#if defined(__HIP_DEVICE_COMPILE__)
        /* three-input AND: one v_bitop3 (left to itself hipcc spends three two-input ANDs on half of the pairs) */
        r[q] = __builtin_amdgcn_bitop3_b32(big - x0, m13 - x2, x8 - pk_min(x7, x9), 0x80) & (x6 - x9);
#else
        r[q] = (big - x0) & (m13 - x2) & (x8 - pk_min(x7, x9)) & (x6 - x9);
#endif
    }
}
/* bit i of the result <-> position i of the window */
MODES_HD uint32_t modes_order8_mask(const uint32_t r[4]) {
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) m |= (((r[q] >> 15) & 1u) << (2 * q)) | (((r[q] >> 31) & 1u) << (2 * q + 1));
    return m;
}

/* Necessary condition for the level tests of dump1090.c:1624-1642 on powers, in plain integers
 * (the beta pass runs it on one position per lane):  9 * max(quiet) < s0 + s2 + s7 + s9.
 * Derivation above. */
MODES_HD bool modes_level_bound(uint32_t s0, uint32_t s2, uint32_t s7, uint32_t s9, uint32_t quiet_max) {
    return 9u * quiet_max < s0 + s2 + s7 + s9 + 4u;      /* + 4: the four pulses may be saturated powers (each <= 1 low) */
}

#endif /* __clang__ */

/* ----------------------------------------------------------- demodulation */

/* Result of one demodulation attempt; same fields as modes_attempt. */
struct modes_attempt_core {
    uint8_t msg[14];
    uint8_t errors;
    uint8_t gate_ok;
};

/* scaleSample, dump1090.c:1473-1476. */
MODES_HD uint32_t modes_scale(uint32_t v, uint32_t factor) {
    uint32_t r = v * factor / 16384u;
    return r > 65535u ? 65535u : r;
}

/* Bit slicing state machine of dump1090.c:1669-1689 fed one (lo, hi) pair at a
 * time, packing as dump1090.c:1696-1706 does (a bit value of 2 is OR-ed in
 * unmasked, then the byte is truncated to 8 bits). */
struct modes_slicer {
    uint32_t acc[14];
    int prev;
    int errors;
    MODES_HDM void reset() {
        for (int b = 0; b < 14; b++) acc[b] = 0;
        prev = 0; errors = 0;
    }
    MODES_HDM void push(int k, int lo, int hi) {
        int d = lo - hi;
        if (d < 0) d = -d;
        int bit;
        if (k > 0 && d < 256) bit = prev;
        else if (lo == hi) { bit = 2; if (k < 56) errors++; }
        else bit = lo > hi;
        prev = bit;
        acc[k >> 3] |= (uint32_t)bit << (7 - (k & 7));
    }
    MODES_HDM void finish(uint8_t msg[14]) const {
        for (int b = 0; b < 14; b++) msg[b] = (uint8_t)acc[b];
    }
};

/* Both attempts at a preamble position.  `mag(t)` returns the reference's
 * magnitude of the sample t places after the preamble start, t in [-1, 239]
 * (t = -1 only when with_phase).  with_phase = (block-local j != 0),
 * dump1090.c:1660.  The noise gate (dump1090.c:1713-1723) always uses the
 * uncorrected samples but the message length of the attempt it gates. */
template <class M>
MODES_HD void modes_demod_both(const M &mag, bool with_phase, modes_attempt_core out[2]) {
    modes_slicer sl;
    int sum56 = 0, sum112 = 0;
    /* attempt 0: samples as received */
    sl.reset();
    for (int k = 0; k < 112; k++) {
        int lo = mag(16 + 2 * k), hi = mag(17 + 2 * k);
        int d = lo > hi ? lo - hi : hi - lo;
        sum112 += d;
        if (k < 56) sum56 += d;
        sl.push(k, lo, hi);
    }
    sl.finish(out[0].msg);
    out[0].errors = (uint8_t)sl.errors;
    out[0].gate_ok = modes_len_by_df(out[0].msg[0] >> 3) == 112 ? (sum112 / 56 >= 2550) : (sum56 / 28 >= 2550);
    if (!out[0].gate_ok) { out[1] = out[0]; return; }     /* position ends here, dump1090.c:1723 */

    /* attempt 1: applyPhaseCorrection, dump1090.c:1498-1558 */
    if (!with_phase) { out[1] = out[0]; return; }
    const uint32_t on_time = (uint32_t)mag(0) + mag(2) + mag(7) + mag(9);
    const uint32_t early = ((uint32_t)mag(-1) + mag(6)) * 2u;
    const uint32_t late = ((uint32_t)mag(3) + mag(10)) * 2u;
    sl.reset();
    if (early > late) {
        /* Backward chain: the odd ("hi") sample of every pair is rescaled, starting
         * from the last one; the factor for pair k-1 depends on the already rescaled
         * pair k (lo_k > hi'_k ? down : up).  Record, per pair, what the slicer needs. */
        const uint32_t x = 16384u * early / (early + on_time);
        const uint32_t up = (16384u + x) & 0xFFFFu, dn = (16384u - x) & 0xFFFFu;
        uint16_t hi2[112];
        uint32_t h = modes_scale(mag(239), up);
        hi2[111] = (uint16_t)h;
        for (int k = 111; k >= 1; k--) {
            uint32_t lo = mag(16 + 2 * k);
            h = modes_scale(mag(15 + 2 * k), lo > h ? dn : up);
            hi2[k - 1] = (uint16_t)h;
        }
        for (int k = 0; k < 112; k++) sl.push(k, mag(16 + 2 * k), hi2[k]);
    } else {
        /* Forward chain: the even ("lo") sample of every pair is rescaled; the factor
         * for pair k+1 depends on the already rescaled pair k (lo'_k > hi_k ? up : down). */
        const uint32_t x = 16384u * late / (late + on_time);
        const uint32_t up = (16384u + x) & 0xFFFFu, dn = (16384u - x) & 0xFFFFu;
        uint32_t l = modes_scale(mag(16), up);
        for (int k = 0; k < 112; k++) {
            uint32_t hi = mag(17 + 2 * k);
            sl.push(k, (int)l, (int)hi);
            if (k < 111) l = modes_scale(mag(18 + 2 * k), l > hi ? up : dn);
        }
    }
    sl.finish(out[1].msg);
    out[1].errors = (uint8_t)sl.errors;
    out[1].gate_ok = modes_len_by_df(out[1].msg[0] >> 3) == 112 ? (sum112 / 56 >= 2550) : (sum56 / 28 >= 2550);
}

/* ------------------------------------------------------------------------
 * Data-parallel formulation of the same demodulation (used by the gfx950 demod
 * kernel: one wavefront per preamble, lane L owns bit pairs k = L and L + 64).
 *
 * Everything sequential in dump1090.c:1669-1689 and 1498-1558 is a first-order
 * boolean recurrence  r_k = G_k | (P_k & r_(k-1))  with G & P == 0, which is the
 * carry chain of an addition: with a = G|P and b = G, the carry out of bit k of
 * a + b is r_k.  Masks are 128-bit, bit k = bit pair k (0..111).
 *
 *  - bit slicing: a "weak" pair (k>0, |lo-hi| < 256) repeats the previous bit:
 *      bit_k = weak_k ? bit_(k-1) : (lo_k > hi_k)      G = strong & ~weak, P = weak
 *  - phase correction, early > late (dump1090.c:1512-1534, walks k = 111 -> 0 and
 *    rescales hi): c_k = lo_k > hi'_k, hi'_(k-1) = scale(hi_(k-1), c_k ? dn : up)
 *      c_(k-1) = Up_(k-1) | (Dn_(k-1) & c_k)   with Up_k = lo_k > scale(hi_k, up),
 *      Dn_k = lo_k > scale(hi_k, dn), Up subset of Dn    (same chain, bit-reversed)
 *  - phase correction, late >= early (dump1090.c:1535-1556, walks k = 0 -> 111 and
 *    rescales lo): c_k = lo'_k > hi_k, lo'_(k+1) = scale(lo_(k+1), c_k ? up : dn)
 *      c_k = Dn_k | (Up_k & c_(k-1))   with Up_k = scale(lo_k, up) > hi_k, Dn subset of Up
 * ------------------------------------------------------------------------ */
struct modes_m128 {
    uint64_t lo, hi;          /* bits 0..63, 64..127 */
};
MODES_HD modes_m128 m128_make(uint64_t lo, uint64_t hi) { modes_m128 r; r.lo = lo; r.hi = hi; return r; }
MODES_HD modes_m128 m128_and(modes_m128 a, modes_m128 b) { return m128_make(a.lo & b.lo, a.hi & b.hi); }
MODES_HD modes_m128 m128_or(modes_m128 a, modes_m128 b) { return m128_make(a.lo | b.lo, a.hi | b.hi); }
MODES_HD modes_m128 m128_xor(modes_m128 a, modes_m128 b) { return m128_make(a.lo ^ b.lo, a.hi ^ b.hi); }
MODES_HD modes_m128 m128_andn(modes_m128 a, modes_m128 b) { return m128_make(a.lo & ~b.lo, a.hi & ~b.hi); }   /* a & ~b */
MODES_HD modes_m128 m128_add(modes_m128 a, modes_m128 b) {
    modes_m128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
    return r;
}
MODES_HD modes_m128 m128_shr1(modes_m128 a) { return m128_make((a.lo >> 1) | (a.hi << 63), a.hi >> 1); }
MODES_HD uint64_t m128_rev64(uint64_t x) {
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    return (x >> 32) | (x << 32);
}
/* bit k <-> bit 111-k, for masks that only use bits 0..111 */
MODES_HD modes_m128 m128_rev112(modes_m128 a) {
    const uint64_t rl = m128_rev64(a.lo), rh = m128_rev64(a.hi);      /* bit k -> 127-k: {hi=rl, lo=rh} */
    return m128_make((rh >> 16) | (rl << 48), rl >> 16);              /* then >> 16: 127-k -> 111-k     */
}
/* r_k = G_k | (P_k & r_(k-1)), r_(-1) = 0.  Requires G & P == 0 and bits >= 112 clear. */
MODES_HD modes_m128 modes_chain(modes_m128 G, modes_m128 P) {
    const modes_m128 sum = m128_add(m128_or(G, P), G);
    return m128_shr1(m128_xor(sum, P));
}
/* same recurrence running from bit 111 down to bit 0 */
MODES_HD modes_m128 modes_chain_down(modes_m128 G, modes_m128 P) {
    return m128_rev112(modes_chain(m128_rev112(G), m128_rev112(P)));
}

/* Per-pair quantities of one slicing pass (elementwise; lane L computes k = L, L+64). */
MODES_HD void modes_pair_flags(int k, int lo, int hi, bool *weak, bool *strong, int *delta) {
    const int d = lo > hi ? lo - hi : hi - lo;
    *delta = d;
    *weak = k > 0 && d < 256;                      /* dump1090.c:1675 */
    *strong = lo > hi;                             /* dump1090.c:1683 */
}

/* msg[0] >> 3 of the FIRST slicing pass (uncorrected magnitudes) from its first six bit pairs alone
 * (dump1090.c:1669-1711): bit k of `weak` = |lo_k - hi_k| < 256, of `gt` = lo_k > hi_k, k = 0..5;
 * eq0 = (lo_0 == hi_0).  Pair 0 never repeats (:1675, i > 0); the value 2 it takes when lo == hi (:1682)
 * travels down a run of weak pairs and ORs into the neighbouring bit when packed (:1696-1706).
 * Pairs 6 and 7 land in bits 1, 0 (and 2 for a value 2 of pair 6) of msg[0]: never in the DF. */
MODES_HD int modes_df_first6(uint32_t weak, uint32_t gt, bool eq0) {
    uint32_t b = eq0 ? 2u : (gt & 1u);
    uint32_t m = b << 7;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 1; k < 6; k++) {
        b = ((weak >> k) & 1u) ? b : ((gt >> k) & 1u);
        m |= b << (7 - k);
    }
    return (int)((m & 0xffu) >> 3);
}

/* From the masks of one slicing pass to the packed message, dump1090.c:1669-1706.
 * `first_equal` = (lo_0 == hi_0): the reference stores the value 2 for that pair and for the
 * weak pairs that repeat it; packing ORs `2 << (7 - t)` into byte k/8 (t = k%8), i.e. sets the
 * bit of pair k-1 unless t == 0, where it falls off the byte (dump1090.c:1696-1706). */
MODES_HD modes_m128 modes_pack_bits(modes_m128 weak, modes_m128 strong, bool first_equal, uint8_t *errors) {
    modes_m128 bits = modes_chain(m128_andn(strong, weak), weak);
    bits.hi &= 0x0000FFFFFFFFFFFFull;
    *errors = first_equal ? 1 : 0;                 /* only pair 0 can take the lo==hi branch (:1677-1682) */
    if (first_equal) {
        /* pairs 0..r hold the value 2, r = length of the weak run that follows pair 0 */
        const modes_m128 w1 = m128_shr1(weak);     /* bit i = weak_(i+1) */
        int r = 0;
        {
            const uint64_t inv_lo = ~w1.lo;
            if (inv_lo) r = __builtin_ctzll(inv_lo);
            else {
                const uint64_t inv_hi = ~w1.hi;
                r = 64 + (inv_hi ? __builtin_ctzll(inv_hi) : 64);
            }
            if (r > 111) r = 111;
        }
        /* ones at bits 0..r-1 = the positions k-1 for k = 1..r */
        modes_m128 twos;
        if (r == 0) twos = m128_make(0, 0);
        else if (r < 64) twos = m128_make((1ull << r) - 1, 0);
        else if (r == 64) twos = m128_make(~0ull, 0);
        else twos = m128_make(~0ull, (1ull << (r - 64)) - 1);
        /* k % 8 == 0 falls off its byte: drop positions p = k-1 with p % 8 == 7 */
        const modes_m128 keep = m128_make(0x7F7F7F7F7F7F7F7Full, 0x7F7F7F7F7F7F7F7Full);
        bits = m128_or(bits, m128_and(twos, keep));
    }
    return bits;                                   /* bit k of the mask = message bit k (MSB first) */
}
/* mask (bit k = message bit k) -> the 14 message bytes */
MODES_HD void modes_bits_to_msg(modes_m128 bits, uint8_t msg[14]) {
    for (int b = 0; b < 14; b++) {
        const uint32_t v = (uint32_t)((b < 8 ? bits.lo >> (8 * b) : bits.hi >> (8 * (b - 8))) & 0xFF);
        /* pair 8b+t is bit t of v and bit 7-t of the byte */
        uint32_t rev = ((v & 0xF0) >> 4) | ((v & 0x0F) << 4);
        rev = ((rev & 0xCC) >> 2) | ((rev & 0x33) << 2);
        rev = ((rev & 0xAA) >> 1) | ((rev & 0x55) << 1);
        msg[b] = (uint8_t)rev;
    }
}
MODES_HD void modes_pack_message(modes_m128 weak, modes_m128 strong, bool first_equal, uint8_t msg[14], uint8_t *errors) {
    modes_bits_to_msg(modes_pack_bits(weak, strong, first_equal, errors), msg);
}

/* Scale factors of applyPhaseCorrection (dump1090.c:1502-1516 / 1538-1539).
 * m_1 = m[-1], m0..m10 as named.  Returns true for the early > late branch. */
MODES_HD bool modes_phase_factors(uint32_t m_1, uint32_t m0, uint32_t m2, uint32_t m3, uint32_t m6, uint32_t m7,
                                  uint32_t m9, uint32_t m10, uint32_t *up, uint32_t *dn) {
    const uint32_t on_time = m0 + m2 + m7 + m9;
    const uint32_t early = (m_1 + m6) * 2u, late = (m3 + m10) * 2u;
    const bool backward = early > late;
    const uint32_t e = backward ? early : late;
    const uint32_t x = 16384u * e / (e + on_time);
    *up = (16384u + x) & 0xFFFFu;
    *dn = (16384u - x) & 0xFFFFu;
    return backward;
}

#endif /* MODES_CORE_H */
