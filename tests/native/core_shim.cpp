// Test scaffolding: exposes the MODES_HD functions of dump1090_amd/csrc/modes_core.h
// (the arithmetic the gfx950 kernels execute per lane) to the CPU test-suite.
// Built with clang++ by tests/native/build.py.  NOT part of the product: nothing in
// dump1090_amd/ loads this library, and it is not a CPU fallback of the GPU path.
#include <cstring>
#include "modes_core.h"

extern "C" {

// modes_mag_exact over n saturated powers (and, with err = -1 / +1, the same correction step fed a square root that is
// one too small / too large - what a 1-ulp device sqrt may produce).
void shim_mag_exact(const uint32_t *s_sat, uint64_t n, uint32_t *out) {
    for (uint64_t k = 0; k < n; k++) out[k] = modes_mag_exact(s_sat[k]);
}

// modes_power_pair_sat + modes_order8_swar (the production scan kernel's alpha pass) over a stream.
// flags[p] = 1 if position p survives the ordering test.
void shim_order_stream(const uint8_t *iq, uint64_t nsamples, uint8_t *flags) {
    auto power2 = [&](uint64_t first) -> uint32_t {
        uint8_t b[4] = {127, 127, 127, 127};
        for (int t = 0; t < 4; t++) {
            uint64_t off = 2 * first + t;
            if (off < 2 * nsamples) b[t] = iq[off];
        }
        uint32_t w;
        memcpy(&w, b, 4);
        return modes_power_pair_sat(w);
    };
    for (uint64_t p0 = 0; p0 < nsamples; p0 += 8) {
        uint32_t E[12], r[4];
        for (int t = 0; t < 12; t++) E[t] = power2(p0 + 2 * t);
        modes_order8_swar(E, r);
        const uint32_t m = modes_order8_mask(r);
        for (int i = 0; i < 8 && p0 + i < nsamples; i++) flags[p0 + i] = (m >> i) & 1u;
    }
}

void shim_power_sat(const uint8_t *iq, uint64_t nsamples, uint16_t *s) {
    for (uint64_t k = 0; k < nsamples; k += 2) {
        uint8_t b[4] = {iq[2 * k], iq[2 * k + 1], 127, 127};
        if (k + 1 < nsamples) { b[2] = iq[2 * k + 2]; b[3] = iq[2 * k + 3]; }
        uint32_t w;
        memcpy(&w, b, 4);
        uint32_t pr = modes_power_pair_sat(w);
        s[k] = (uint16_t)pr;
        if (k + 1 < nsamples) s[k + 1] = (uint16_t)(pr >> 16);
    }
}

int shim_level_bound(uint32_t s0, uint32_t s2, uint32_t s7, uint32_t s9, uint32_t quiet) {
    return modes_level_bound(s0, s2, s7, s9, quiet) ? 1 : 0;
}

void shim_power(const uint8_t *iq, uint64_t nsamples, uint16_t *s) {
    for (uint64_t k = 0; k + 1 < nsamples + 1 && k < nsamples; k += 2) {
        uint8_t b[4] = {iq[2 * k], iq[2 * k + 1], 127, 127};
        if (k + 1 < nsamples) { b[2] = iq[2 * k + 2]; b[3] = iq[2 * k + 3]; }
        uint32_t w;
        memcpy(&w, b, 4);
        uint32_t pr = modes_power_pair(w);
        s[k] = (uint16_t)pr;
        if (k + 1 < nsamples) s[k + 1] = (uint16_t)(pr >> 16);
    }
}

struct MagPtr {
    const uint16_t *m;                 // m[0] is offset -1
    int operator()(int t) const { return m[t + 1]; }
};

// win = magnitudes at offsets -1..239 (241 values).  out = 2 x {msg[14], errors, gate_ok}
void shim_demod_both(const uint16_t *win, int with_phase, uint8_t *out) {
    modes_attempt_core a[2];
    modes_demod_both(MagPtr{win}, with_phase != 0, a);
    memcpy(out, a, sizeof a);
}

// DF of the first slicing pass from its first six bit pairs alone (the demod kernel's gate pre-test
// uses it to pick the message length).  win as above.
int shim_df_first6(const uint16_t *win) {
    uint32_t weak = 0, gt = 0;
    for (int k = 0; k < 6; k++) {
        const int lo = win[1 + 16 + 2 * k], hi = win[1 + 17 + 2 * k];
        const int d = lo > hi ? lo - hi : hi - lo;
        weak |= (d < 256 ? 1u : 0u) << k;
        gt |= (lo > hi ? 1u : 0u) << k;
    }
    return modes_df_first6(weak, gt, win[1 + 16] == win[1 + 17]);
}

// The data-parallel formulation the gfx950 demod kernel uses (carry chains, modes_core.h), with the
// 64 lanes emulated by loops: lane L owns pairs L and L+64, "ballots" become mask-building loops.
// Mirrors demod_kernel's stage 2 statement by statement.
namespace {
struct Lanes {
    int lo[112], hi[112];
};
modes_m128 ballot_pairs(const bool *f) {
    modes_m128 m = m128_make(0, 0);
    for (int k = 0; k < 112; k++)
        if (f[k]) { if (k < 64) m.lo |= 1ull << k; else m.hi |= 1ull << (k - 64); }
    return m;
}
bool bit_of(modes_m128 m, int k) { return ((k < 64 ? m.lo >> k : m.hi >> (k - 64)) & 1ull) != 0; }
void slice_pass_emul(const Lanes &v, uint8_t msg[14], uint8_t *errors, int *sum56, int *sum112) {
    bool w[112], s[112];
    int s56 = 0, s112 = 0;
    for (int k = 0; k < 112; k++) {
        int d;
        modes_pair_flags(k, v.lo[k], v.hi[k], &w[k], &s[k], &d);
        s112 += d;
        if (k < 56) s56 += d;
    }
    modes_pack_message(ballot_pairs(w), ballot_pairs(s), v.lo[0] == v.hi[0], msg, errors);
    if (sum56) { *sum56 = s56; *sum112 = s112; }
}
}  // namespace

void shim_demod_parallel(const uint16_t *win, int with_phase, uint8_t *out) {
    MagPtr mag{win};
    Lanes v;
    for (int k = 0; k < 112; k++) { v.lo[k] = mag(16 + 2 * k); v.hi[k] = mag(17 + 2 * k); }
    modes_attempt_core a[2];
    memset(a, 0, sizeof a);
    int sum56, sum112;
    slice_pass_emul(v, a[0].msg, &a[0].errors, &sum56, &sum112);
    a[0].gate_ok = modes_len_by_df(a[0].msg[0] >> 3) == 112 ? (sum112 / 56 >= 2550) : (sum56 / 28 >= 2550);
    a[1] = a[0];
    if (a[0].gate_ok && with_phase) {
        uint32_t up, dn;
        const bool backward = modes_phase_factors(mag(-1), mag(0), mag(2), mag(3), mag(6), mag(7), mag(9), mag(10), &up, &dn);
        Lanes n = v;
        bool U[112], D[112];
        if (backward) {
            int hu[112], hd[112];
            for (int k = 0; k < 112; k++) {
                hu[k] = (int)modes_scale((uint32_t)v.hi[k], up); hd[k] = (int)modes_scale((uint32_t)v.hi[k], dn);
                U[k] = v.lo[k] > hu[k]; D[k] = v.lo[k] > hd[k];
            }
            const modes_m128 Up = ballot_pairs(U);
            modes_m128 Pm = m128_andn(ballot_pairs(D), Up);
            Pm.hi &= ~(1ull << 47);
            const modes_m128 cm = modes_chain_down(Up, Pm);
            for (int k = 0; k < 112; k++) n.hi[k] = (k == 111) ? hu[k] : (bit_of(cm, k + 1) ? hd[k] : hu[k]);
        } else {
            int lu[112], ld[112];
            for (int k = 0; k < 112; k++) {
                lu[k] = (int)modes_scale((uint32_t)v.lo[k], up); ld[k] = (int)modes_scale((uint32_t)v.lo[k], dn);
                U[k] = lu[k] > v.hi[k]; D[k] = ld[k] > v.hi[k];
            }
            const modes_m128 Up = ballot_pairs(U), Dn = ballot_pairs(D);
            modes_m128 Gm = Dn, Pm = m128_andn(Up, Dn);
            Gm.lo = (Gm.lo & ~1ull) | (Up.lo & 1ull);
            Pm.lo &= ~1ull;
            const modes_m128 cm = modes_chain(Gm, Pm);
            for (int k = 0; k < 112; k++) n.lo[k] = (k == 0) ? lu[k] : (bit_of(cm, k - 1) ? lu[k] : ld[k]);
        }
        slice_pass_emul(n, a[1].msg, &a[1].errors, nullptr, nullptr);
        a[1].gate_ok = modes_len_by_df(a[1].msg[0] >> 3) == 112 ? (sum112 / 56 >= 2550) : (sum56 / 28 >= 2550);
    }
    memcpy(out, a, sizeof a);
}

int shim_preamble_exact(const uint16_t *m15) {
    struct M { const uint16_t *m; int operator()(int t) const { return m[t]; } };
    return modes_preamble_exact(M{m15}) ? 1 : 0;
}

uint32_t shim_syndrome(const uint8_t *msg, int nbytes) { return modes_syndrome(msg, nbytes); }
uint32_t shim_bit_syndrome(int p) { return modes_bit_syndrome(p); }

int shim_find_fix(uint32_t syndrome, int bits, int maxfix, uint8_t *pos) {
    static uint32_t esyn[112];
    static bool ready = false;
    if (!ready) { for (int p = 0; p < 112; p++) esyn[p] = modes_bit_syndrome(p); ready = true; }
    return modes_find_fix(syndrome, bits, maxfix, esyn, pos);
}

int shim_sizeof_attempt_core() { return (int)sizeof(modes_attempt_core); }
}
