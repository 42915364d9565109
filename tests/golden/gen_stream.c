/* gen_stream.c - tests/synth.py's SparseFrameStream written to stdout, fast enough for BASELINE's 8 - 64 GiB configs
 * (numpy makes 12 MB/s of this noise; this makes ~1 GB/s on four threads).  TEST INFRASTRUCTURE: used only by
 * tests/golden/make_listings.py to feed the compiled reference (oracle/_ref/dump1090_ref --ifile -).
 *
 *   gen_stream <seed> <nbytes> <sigma_q16> <patch file>
 *
 * noise: byte idx = clip(127 + (((sum of the 8 bytes of mix64(seed + idx * GOLD)) - 1020) * sigma_q16 + 58982 >> 16))
 *        (tests/synth.py:noise_at; dump1090_amd/csrc/modes_gfx950.hip:synth_noise_kernel is the device's copy)
 * patch file: int64 n, int64 width, int64 first_byte[n] (ascending), uint8 data[n][width] - the final bytes of every
 *        frame's footprint (SparseFrameStream.patches()); the last 480 bytes of the stream are 127.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t mix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    if (argc != 5) { fprintf(stderr, "usage: gen_stream seed nbytes sigma_q16 patches\n"); return 2; }
    const uint64_t seed = strtoull(argv[1], 0, 0), nbytes = strtoull(argv[2], 0, 0);
    const int32_t sigma = (int32_t)strtol(argv[3], 0, 0);
    FILE *pf = fopen(argv[4], "rb");
    int64_t n = 0, width = 0;
    if (!pf || fread(&n, 8, 1, pf) != 1 || fread(&width, 8, 1, pf) != 1) { perror("patches"); return 1; }
    int64_t *first = malloc(sizeof(int64_t) * (size_t)(n + 1));
    uint8_t *data = malloc((size_t)(n * width + 1));
    if (fread(first, 8, (size_t)n, pf) != (size_t)n || fread(data, 1, (size_t)(n * width), pf) != (size_t)(n * width)) {
        fprintf(stderr, "short patch file\n");
        return 1;
    }
    fclose(pf);
    const uint64_t chunk = 1ull << 26;
    uint8_t *buf = malloc(chunk);
    int64_t k = 0;                                   /* first patch that may still touch the current chunk */
    for (uint64_t lo = 0; lo < nbytes; lo += chunk) {
        const uint64_t len = nbytes - lo < chunk ? nbytes - lo : chunk;
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < len; i++) {
            const uint64_t h = mix64(seed + (lo + i) * 0x9E3779B97F4A7C15ull);
            uint64_t t = (h & 0x00FF00FF00FF00FFull) + ((h >> 8) & 0x00FF00FF00FF00FFull);
            t = (t & 0x0000FFFF0000FFFFull) + ((t >> 16) & 0x0000FFFF0000FFFFull);
            const int32_t g = (int32_t)((t & 0xFFFFFFFFull) + (t >> 32));
            int32_t v = 127 + (((g - 1020) * sigma + 58982) >> 16);
            buf[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
        while (k < n && first[k] + width <= (int64_t)lo) k++;
        for (int64_t p = k; p < n && first[p] < (int64_t)(lo + len); p++) {
            for (int64_t b = 0; b < width; b++) {
                const int64_t o = first[p] + b - (int64_t)lo;
                if (o >= 0 && o < (int64_t)len) buf[o] = data[p * width + b];
            }
        }
        if (lo + len > nbytes - 480) {
            const uint64_t t0 = nbytes - 480 > lo ? nbytes - 480 - lo : 0;
            memset(buf + t0, 127, len - t0);
        }
        if (fwrite(buf, 1, len, stdout) != len) return 1;   /* the reader went away */
    }
    return 0;
}
