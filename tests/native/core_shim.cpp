// Test scaffolding: exposes the MODES_HD functions of dump1090_amd/csrc/modes_core.h
// (the arithmetic the gfx950 kernels execute per lane) to the CPU test-suite.
// Built with clang++ by tests/native/build.py.  NOT part of the product: nothing in
// dump1090_amd/ loads this library, and it is not a CPU fallback of the GPU path.
#include <cstring>
#include "modes_core.h"

extern "C" {

// modes_power_pair + modes_scan8 over a whole stream, 8 positions per call, as the
// scan kernel's lanes do.  flags[p] = 1 if position p is forwarded.
void shim_scan_stream(const uint8_t *iq, uint64_t nsamples, uint8_t *flags) {
    auto power2 = [&](uint64_t first) -> uint32_t {   // (s[first], s[first+1]), 127 beyond the end
        uint8_t b[4] = {127, 127, 127, 127};
        for (int t = 0; t < 4; t++) {
            uint64_t off = 2 * first + t;
            if (off < 2 * nsamples) b[t] = iq[off];
        }
        uint32_t w;
        memcpy(&w, b, 4);
        return modes_power_pair(w);
    };
    for (uint64_t p0 = 0; p0 < nsamples; p0 += 8) {
        uint32_t E[12];
        for (int t = 0; t < 12; t++) E[t] = power2(p0 + 2 * t);
        uint32_t hit = modes_scan8(E);
        for (int i = 0; i < 8 && p0 + i < nsamples; i++) flags[p0 + i] = (hit & modes_scan8_bit(i)) != 0;
    }
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
