"""Seeded synthetic uint8 I/Q streams in dump1090's "2 Msps" file format.

Everything is integer arithmetic on a counter-based hash, so the bytes are a
pure function of (seed, byte index) on any machine / numpy version.  The same
noise definition is implemented on the device by the HIP fill kernel
(dump1090_amd/csrc/modes_gfx950.hip: synth_noise_kernel) and checked against
this module in tests/test_gpu_parity.py.

Signal model: SURVEY.md 3.6 (preamble pulses at samples 0,2,7,9 as drawn in
dump1090.c:1570-1592; bit k puts its pulse in sample 16+2k for a 1 and 17+2k
for a 0, dump1090.c:1669-1688).
"""
from __future__ import annotations

import numpy as np

DATA_LEN = 262144            # bytes per reader buffer        (dump1090.c:54)
BLOCK_STRIDE = DATA_LEN // 2 # 131072 samples between buffer starts
CARRY = 238                  # samples carried into the next buffer (dump1090.c:331)

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)

# round(1024*cos(2*pi*p/64)), p = 0..63 (sin is the same table shifted by 48)
COS64 = np.array([
    1024, 1019, 1004, 980, 946, 903, 851, 792, 724, 650, 569, 483, 392, 297, 200, 100,
    0, -100, -200, -297, -392, -483, -569, -650, -724, -792, -851, -903, -946, -980, -1004, -1019,
    -1024, -1019, -1004, -980, -946, -903, -851, -792, -724, -650, -569, -483, -392, -297, -200, -100,
    0, 100, 200, 297, 392, 483, 569, 650, 724, 792, 851, 903, 946, 980, 1004, 1019], dtype=np.int64)
SIN64 = np.roll(COS64, 16)


def mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays."""
    with np.errstate(over="ignore"):
        z = (x + _GOLD) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_at(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return mix64((np.uint64(seed) + idx.astype(np.uint64) * _GOLD) & _M64)


def noise_at(seed: int, idx: np.ndarray, sigma_q16: int = 941) -> np.ndarray:
    """The noise byte at every stream byte index in `idx` (any shape)."""
    h = hash_at(seed, idx)
    g = np.zeros(h.shape, dtype=np.int64)
    for b in range(8):
        g += ((h >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.int64)
    v = 127 + (((g - 1020) * sigma_q16 + 58982) >> 16)
    return np.clip(v, 0, 255).astype(np.uint8)


def noise_bytes(seed: int, first_byte: int, nbytes: int, sigma_q16: int = 941) -> np.ndarray:
    """Approximately Gaussian bytes around 127.4.

    g = sum of the 8 bytes of hash(seed, byte index) (mean 1020, sigma 209.02);
    value = clip(127 + floor(((g-1020)*sigma_q16 + 58982) / 65536), 0, 255).
    sigma_q16 = round(sigma/209.0215*65536): 941 -> sigma 3.0, 627 -> sigma 2.0.
    """
    out = np.empty(nbytes, dtype=np.uint8)
    step = 1 << 22
    for lo in range(0, nbytes, step):
        hi = min(nbytes, lo + step)
        out[lo:hi] = noise_at(seed, np.arange(first_byte + lo, first_byte + hi, dtype=np.uint64), sigma_q16)
    return out


# ----------------------------------------------------------------- Mode S CRC

def _crc_table():
    t = [0] * 112
    r = 0xFFF409
    for i in range(87, -1, -1):
        t[i] = r
        r <<= 1
        if r & 0x1000000:
            r ^= 0x1FFF409
    return t


CRC_TABLE = _crc_table()


def modes_crc(data: bytes, bits: int) -> int:
    """Parity of the first bits-24 bits (dump1090.c:703-719 semantics)."""
    base = 112 - bits
    c = 0
    for k in range(bits - 24):
        if data[k >> 3] & (0x80 >> (k & 7)):
            c ^= CRC_TABLE[base + k]
    return c


def make_frame(df: int, payload: bytes, xor_parity: int = 0) -> bytes:
    """DF byte + payload + 24-bit parity (optionally XOR an address / IID)."""
    nbytes = 14 if 16 <= df <= 21 else 7
    body = bytearray(nbytes)
    body[0] = (df << 3) | (payload[0] & 7)
    body[1:nbytes - 3] = payload[1:nbytes - 3]
    c = modes_crc(bytes(body), nbytes * 8) ^ xor_parity
    body[-3:] = bytes([(c >> 16) & 0xFF, (c >> 8) & 0xFF, c & 0xFF])
    return bytes(body)


def frame_envelope(frame: bytes) -> np.ndarray:
    """Pulse envelope (0/1 per half-microsecond sample), 16 + 2*bits samples."""
    nbits = len(frame) * 8
    e = np.zeros(16 + 2 * nbits, dtype=np.int64)
    e[[0, 2, 7, 9]] = 1
    for k in range(nbits):
        bit = (frame[k >> 3] >> (7 - (k & 7))) & 1
        e[16 + 2 * k + (0 if bit else 1)] = 1
    return e


def add_frame(iq: np.ndarray, sample: int, frame: bytes, amp: int, phase: int, smear16: int = 0) -> None:
    """Add a frame to an interleaved u8 I/Q array in place.

    amplitude per sample = amp * ((16-smear16)*e[t] + smear16*e[t-1]) / 16, then
    I += (a*cos+512)>>10, Q += (a*sin+512)>>10, clipped to u8.
    """
    e = frame_envelope(frame)
    e = np.concatenate([e, [0]])
    prev = np.concatenate([[0], e[:-1]])
    a = (amp * ((16 - smear16) * e + smear16 * prev)) >> 4
    di = (a * COS64[phase & 63] + 512) >> 10
    dq = (a * SIN64[phase & 63] + 512) >> 10
    n = len(e)
    nsamp = len(iq) // 2
    lo = max(0, -sample)
    hi = min(n, nsamp - sample)
    if hi <= lo:
        return
    idx = 2 * (sample + np.arange(lo, hi))
    iq[idx] = np.clip(iq[idx].astype(np.int64) + di[lo:hi], 0, 255).astype(np.uint8)
    iq[idx + 1] = np.clip(iq[idx + 1].astype(np.int64) + dq[lo:hi], 0, 255).astype(np.uint8)


def finish_stream(iq: np.ndarray) -> np.ndarray:
    """Oracle-safe tail: length a multiple of 262144 B, last 480 B = 127
    (neutralises the reference's EOF race)."""
    assert len(iq) % DATA_LEN == 0
    iq[-480:] = 127
    return iq


def _payload(seed: int, n: int, k: int) -> bytes:
    h = hash_at(seed ^ 0xABCDEF, np.arange(k * 2, k * 2 + 2, dtype=np.uint64))
    return (int(h[0]).to_bytes(8, "little") + int(h[1]).to_bytes(8, "little"))[:n]


def frames_stream(seed: int, nblocks: int, *, spacing: int = 3000, sigma_q16: int = 627,
                  amp=(40, 100), smear=(0,), flip1: int = 0, flip2: int = 0,
                  extra_offsets=(), df_cycle=(17, 11, 17, 17, 11, 4, 20, 0, 5, 21)) -> tuple[np.ndarray, list]:
    """Noise plus a deterministic train of Mode S frames.

    Every `spacing` samples (jittered by the hash) one frame is placed; DFs cycle
    through df_cycle; AP-type frames (DF0/4/5/20/21) reuse the ICAO address of an
    earlier DF17 so the whitelist path (dump1090.c:942-983) is exercised.
    flip1 / flip2: one in `flip1` (`flip2`) frames gets 1 (2) data bits inverted.
    Returns (stream, placements) with placements = [(sample, frame_bytes), ...].
    """
    nbytes = nblocks * DATA_LEN
    iq = noise_bytes(seed, 0, nbytes, sigma_q16)
    nsamp = nbytes // 2
    placed = []
    addrs = []
    k = 0
    pos = 500
    offsets = []
    while pos < nsamp - 600:
        offsets.append(pos)
        h = int(hash_at(seed ^ 0x5EED, np.array([k], dtype=np.uint64))[0])
        pos += spacing + (h % 257)
        k += 1
    offsets = sorted(set(offsets) | {o for o in extra_offsets if 0 <= o < nsamp - 600})
    for k, o in enumerate(offsets):
        h = int(hash_at(seed ^ 0xF00D, np.array([k], dtype=np.uint64))[0])
        df = df_cycle[k % len(df_cycle)]
        pay = bytearray(_payload(seed, 14, k))
        if df in (11, 17, 18):
            addr = (pay[1] << 16) | (pay[2] << 8) | pay[3]
            addrs.append(addr)
            frame = make_frame(df, bytes(pay))
        else:
            addr = addrs[(h >> 8) % len(addrs)] if addrs else 0x123456
            frame = make_frame(df, bytes(pay), xor_parity=addr)
        fb = bytearray(frame)
        nb = len(fb) * 8
        if flip2 and (h >> 20) % flip2 == 0:
            b1 = 5 + (h >> 24) % (nb - 5)
            b2 = 5 + (h >> 34) % (nb - 5)
            fb[b1 >> 3] ^= 0x80 >> (b1 & 7)
            if b2 != b1:
                fb[b2 >> 3] ^= 0x80 >> (b2 & 7)
        elif flip1 and (h >> 20) % flip1 == 0:
            b1 = 5 + (h >> 24) % (nb - 5)
            fb[b1 >> 3] ^= 0x80 >> (b1 & 7)
        a = amp[0] + (h >> 44) % (amp[1] - amp[0] + 1)
        sm = smear[(h >> 52) % len(smear)]
        add_frame(iq, o, bytes(fb), int(a), int((h >> 56) & 63), int(sm))
        placed.append((o, bytes(fb)))
    return finish_stream(iq), placed


# ------------------------------------------------- big sparse streams (BASELINE configs 3-5)


class SparseFrameStream:
    """sigma-noise(seed) with frames added at given places, defined lazily so that it can be as big as
    BASELINE's 8 GiB / 64 GiB configs: the device builds it as synth_noise + a scatter of `patches()`,
    the host materialises any byte range with `window()`; both are the same bytes.

    placements: list of (sample, frame_bytes, amp, phase, smear16); frames must not overlap.
    The last 480 bytes are 127 (finish_stream)."""

    SPAN = 16 + 224 + 2          # samples a frame can touch (long frame + smear tail), rounded up

    def __init__(self, seed: int, nbytes: int, sigma_q16: int, placements):
        assert nbytes % DATA_LEN == 0
        self.seed, self.nbytes, self.sigma_q16 = seed, nbytes, sigma_q16
        self.placements = sorted(placements, key=lambda p: p[0])
        self._starts = np.array([p[0] for p in self.placements], dtype=np.int64)
        assert len(self._starts) < 2 or int(np.diff(self._starts).min()) >= self.SPAN, "frames overlap"

    def window(self, lo: int, hi: int) -> np.ndarray:
        """Bytes [lo, hi) of the stream (lo even)."""
        assert lo % 2 == 0 and 0 <= lo <= hi <= self.nbytes
        out = noise_bytes(self.seed, lo, hi - lo, self.sigma_q16)
        s_lo, s_hi = lo // 2, (hi + 1) // 2
        a = int(np.searchsorted(self._starts, s_lo - self.SPAN, side="left"))
        b = int(np.searchsorted(self._starts, s_hi, side="left"))
        for sample, frame, amp, phase, smear in self.placements[a:b]:
            add_frame(out, sample - s_lo, frame, amp, phase, smear)
        tail = self.nbytes - 480
        if hi > tail:
            out[max(0, tail - lo):] = 127
        return out

    def patches(self):
        """-> (first_byte int64[n], data uint8[n, 2*SPAN]): the final bytes of every frame's footprint."""
        n = len(self.placements)
        first = 2 * self._starts
        width = 2 * self.SPAN
        data = np.empty((n, width), dtype=np.uint8)
        tail = self.nbytes - 480
        for lo in range(0, n, 16384):                    # noise for many footprints at once, frames one by one
            hi = min(n, lo + 16384)
            idx = first[lo:hi, None].astype(np.uint64) + np.arange(width, dtype=np.uint64)[None, :]
            data[lo:hi] = noise_at(self.seed, idx, self.sigma_q16)
            for k in range(lo, hi):
                sample, frame, amp, phase, smear = self.placements[k]
                add_frame(data[k], 0, frame, amp, phase, smear)
                if first[k] + width > tail:
                    data[k, max(0, tail - int(first[k])):] = 127
        return first, data

    def deltas(self):
        """-> (first_byte int64[n], delta int16[n, 2*SPAN]): what every frame ADDS to the noise under its footprint
        (interleaved dI, dQ per sample; the byte is clip(noise + delta, 0, 255): add_frame's arithmetic).  No hash in
        here, so the frames of a 64 GiB stream take seconds instead of a minute: the device, which already holds the
        noise, applies them (bench.py:build_frames_shard, tests: build_on_device).  Bytes past the end of the stream
        and the 127-tail are the caller's business."""
        n = len(self.placements)
        first = 2 * self._starts
        width = 2 * self.SPAN
        delta = np.zeros((n, width), dtype=np.int16)
        for lo in range(0, n, 32768):
            hi = min(n, lo + 32768)
            m = hi - lo
            fr = np.zeros((m, 14), dtype=np.uint8)
            nbits = np.empty(m, dtype=np.int64)
            for k in range(m):
                f = self.placements[lo + k][1]
                fr[k, :len(f)] = np.frombuffer(f, dtype=np.uint8)
                nbits[k] = 8 * len(f)
            bits = np.unpackbits(fr, axis=1).astype(np.int32)                    # (m, 112), MSB first
            live = (np.arange(112)[None, :] < nbits[:, None]).astype(np.int32)
            e = np.zeros((m, self.SPAN + 1), dtype=np.int32)                     # frame_envelope, one row per frame
            e[:, [0, 2, 7, 9]] = 1
            e[:, 16:16 + 224:2] = bits * live
            e[:, 17:17 + 224:2] = (1 - bits) * live
            amp = np.array([p[2] for p in self.placements[lo:hi]], dtype=np.int32)[:, None]
            phase = np.array([p[3] for p in self.placements[lo:hi]], dtype=np.int64)
            sm = np.array([p[4] for p in self.placements[lo:hi]], dtype=np.int32)[:, None]
            prev = np.concatenate([np.zeros((m, 1), dtype=np.int32), e[:, :-1]], axis=1)
            a = ((amp * ((16 - sm) * e + sm * prev)) >> 4)[:, :self.SPAN]       # add_frame's amplitude per sample
            delta[lo:hi, 0::2] = (a * COS64[phase & 63][:, None].astype(np.int32) + 512) >> 10
            delta[lo:hi, 1::2] = (a * SIN64[phase & 63][:, None].astype(np.int32) + 512) >> 10
        return first, delta


def config3_stream(seed: int, nblocks: int, *, per: int = 65536, sigma_q16: int = 941, amp=(40, 100),
                   flip1: int = 10, edge_every: int = 997, smear=(0,), flip2: int = 0,
                   only_samples=None) -> SparseFrameStream:
    """BASELINE config 3: sigma=3 noise + DF11/DF17 frames with valid parity, about one per `per`
    samples at hashed offsets, amplitude 40..100, random carrier phase, one in `flip1` with one
    flipped data bit; every `edge_every`-th frame sits at one of the buffer-seam offsets of
    SURVEY.md 3.3 (EDGE_DELTAS) instead."""
    nsamp = nblocks * BLOCK_STRIDE
    placements = []
    clean = {}
    # only_samples = (lo, hi): just the frames that can touch samples [lo, hi) of the same stream (a rank's shard of
    # a multi-GPU run builds its part without walking the other ranks' frames)
    k_lo, k_hi = 0, max(0, (nsamp - per + per - 1) // per)
    if only_samples is not None:
        k_lo = max(k_lo, (only_samples[0] - BLOCK_STRIDE) // per - 1)     # a seam frame sits up to a buffer behind its slot
        k_hi = min(k_hi, only_samples[1] // per + 2)
    hashes = hash_at(seed ^ 0xC0F3, np.arange(k_lo, max(k_lo, k_hi), dtype=np.uint64))
    for k in range(k_lo, k_hi):
        base = k * per
        h = int(hashes[k - k_lo])
        o = base + 300 + h % (per - 900)
        if edge_every and k % edge_every == edge_every - 1:
            seam = (base // BLOCK_STRIDE + 1) * BLOCK_STRIDE - CARRY
            if seam + 300 < nsamp:
                o = seam + EDGE_DELTAS[(k // edge_every) % len(EDGE_DELTAS)]
        df = 17 if (h >> 9) & 1 else 11
        pay = bytearray(_payload(seed, 14, k))
        fb = bytearray(make_frame(df, bytes(pay)))
        nb = len(fb) * 8
        if flip2 and (h >> 20) % flip2 == 0:
            for sh in (24, 34):
                bit = 5 + (h >> sh) % (nb - 5)
                fb[bit >> 3] ^= 0x80 >> (bit & 7)
        elif flip1 and (h >> 20) % flip1 == 0:
            bit = 5 + (h >> 24) % (nb - 5)
            fb[bit >> 3] ^= 0x80 >> (bit & 7)
        a = amp[0] + (h >> 44) % (amp[1] - amp[0] + 1)
        sm = smear[(h >> 52) % len(smear)]
        placements.append((o, bytes(fb), int(a), int((h >> 56) & 63), int(sm)))
        clean[o] = make_frame(df, bytes(pay))
    # Frames must not overlap (SparseFrameStream): of two whose footprints touch - a seam frame can land next to the
    # regular frame of the slot it falls into - the LATER one is dropped.  Decided on the raw placements, so a rank that
    # builds only its own part (only_samples) drops the same frames as the whole stream does.
    placements.sort(key=lambda p: p[0])
    starts = [p[0] for p in placements]
    dropped = {starts[i] for i in range(1, len(starts)) if starts[i] - starts[i - 1] < SparseFrameStream.SPAN}
    if dropped:
        placements = [p for p in placements if p[0] not in dropped]
        clean = {o: f for o, f in clean.items() if o not in dropped}
    if only_samples is not None:
        keep = [p for p in placements if p[0] + SparseFrameStream.SPAN > only_samples[0] and p[0] < only_samples[1]]
        clean = {p[0]: clean[p[0]] for p in keep}
        placements = keep
    st = SparseFrameStream(seed, nblocks * DATA_LEN, sigma_q16, placements)
    st.clean = clean                 # sample -> the frame as transmitted before any bit flip
    return st


# ---------------------------------------------------------- named test streams

def case_uniform(seed: int = 11, nblocks: int = 3) -> np.ndarray:
    h = hash_at(seed, np.arange(nblocks * DATA_LEN // 8, dtype=np.uint64))
    return finish_stream(h.view(np.uint8).copy())


def case_coarse(seed: int = 12, nblocks: int = 3) -> np.ndarray:
    """7-level noise: many equal neighbours -> exercises bit value 2 / errors==1."""
    h = hash_at(seed, np.arange(nblocks * DATA_LEN, dtype=np.uint64))
    lv = (h % np.uint64(7)).astype(np.int64)
    return finish_stream((127 + (lv - 3) * 9).astype(np.uint8))


EDGE_DELTAS = (-3, -2, -1, 0, 1, -100, -239, -241)


def case_edges(seed: int = 13, smear16: int = 0) -> np.ndarray:
    """One clean (or smeared) frame next to each of 8 buffer seams.

    Seam s (between buffers s and s+1) gets a frame at file sample
    131072*(s+1) - 238 + EDGE_DELTAS[s], i.e. buffer-local j = delta of buffer
    s+1 (delta >= 0) or j = 131072 + delta of buffer s (delta < 0):
      -3 -> j=131069 (last tested offset), -2/-1 -> j=131070/131071 (never
      tested, Q1), 0 -> j=0 (tested, but no phase correction on retry, Q2),
      -100/-239/-241 -> frames whose body crosses the seam (skip window is not
      carried over, Q3).  SURVEY.md 3.3.
    """
    nblocks = len(EDGE_DELTAS) + 1
    iq = noise_bytes(seed, 0, nblocks * DATA_LEN, 627)
    for k, o in enumerate((2000, 5000, 9000)):           # warm the ICAO whitelist
        add_frame(iq, o, make_frame(17, _payload(seed, 14, 100 + k)), 70, 7 * k, 0)
    for s_i, delta in enumerate(EDGE_DELTAS):
        o = BLOCK_STRIDE * (s_i + 1) - CARRY + delta
        df = 17 if s_i % 2 == 0 else 11
        add_frame(iq, o, make_frame(df, _payload(seed, 14, s_i)), 60, 5 * s_i + 3, smear16)
    return finish_stream(iq)


def case_edges_at(delta: int, seed: int = 14, nblocks: int = 2, smear16: int = 0, df: int = 17) -> np.ndarray:
    """One frame at file sample (131072 - 238 + delta), i.e. local j = delta of
    buffer 1 (delta >= 0) or j = 131072 + delta of buffer 0 (delta < 0), plus a
    few ordinary frames so the ICAO cache is warm."""
    iq = noise_bytes(seed, 0, nblocks * DATA_LEN, 627)
    for k, o in enumerate((2000, 5000, 9000)):
        add_frame(iq, o, make_frame(17, _payload(seed, 14, k)), 70, 7 * k, 0)
    o = BLOCK_STRIDE - CARRY + delta
    add_frame(iq, o, make_frame(df, _payload(seed, 14, 99)), 60, 11, smear16)
    return finish_stream(iq)


def case_frames(seed: int = 15, nblocks: int = 3) -> np.ndarray:
    return frames_stream(seed, nblocks, spacing=2500, flip1=5, flip2=9)[0]


def case_smear(seed: int = 16, nblocks: int = 3) -> np.ndarray:
    """Inter-sample energy leak: CRC failures on the first attempt, recovered by
    the phase-corrected retry or by the 1-bit repair."""
    return frames_stream(seed, nblocks, spacing=2500, sigma_q16=941, amp=(30, 90),
                         smear=(0, 4, 5, 6, 7, 8), flip1=7, flip2=11)[0]


def case_lowsnr(seed: int = 17, nblocks: int = 3) -> np.ndarray:
    """BASELINE config 5 in miniature: weak frames over sigma=3 noise with
    20-40 % leak and 2-bit errors (run with --aggressive)."""
    return frames_stream(seed, nblocks, spacing=1500, sigma_q16=941, amp=(8, 20),
                         smear=(3, 4, 5, 6), flip1=10, flip2=20)[0]


def case_saturated(seed: int = 19, nblocks: int = 2) -> np.ndarray:
    """Clipped receiver: frames whose pulses sit at the ADC rails on a silent (127, 127) background.  Two kinds, at
    even and odd sample offsets alike:
      A  pulses (255,255) - the one byte pair whose power needs 16 bits (32768) - between samples of power 0: the
         largest difference two samples can have (the packed-halves compare of the scan kernel has to survive it
         together with a borrow from the other half);
      B  the same pulses with the unconstrained preamble samples 1, 3, 6, 8 at (255,254) / (254,255) - power 32513,
         the largest a non-saturated pair reaches: the preamble holds only because 32768 stays above 32513.
    Kind C puts the rail value on samples 1 and 3 and 32513 on the pulses: no preamble (sample 1 > sample 0)."""
    iq = np.full(nblocks * DATA_LEN, 127, dtype=np.uint8)
    rng = np.random.default_rng(seed)

    def put(sample, i, q):
        iq[2 * sample], iq[2 * sample + 1] = i, q

    pos = 1000
    k = 0
    while pos + 400 < nblocks * DATA_LEN // 2 - 300:
        kind = "ABC"[k % 3]
        frame = make_frame(17 if k % 2 else 11, _payload(seed, 14, k))
        e = frame_envelope(frame)
        hi = (255, 255) if kind != "C" else ((255, 254) if k % 2 else (254, 255))
        for t in np.nonzero(e)[0]:
            put(pos + int(t), *hi)
        if kind == "B":
            for t in (1, 3, 6, 8):
                put(pos + t, *((255, 254) if (t + k) % 2 else (254, 255)))
        if kind == "C":
            put(pos + 1, 255, 255)
            put(pos + 3, 255, 255)
        pos += 331 + int(rng.integers(0, 40))            # every parity and every lane offset comes up
        k += 1
    return finish_stream(iq)


def case_noise(seed: int = 18, nblocks: int = 4) -> np.ndarray:
    """BASELINE config 2 in miniature: sigma=3 noise only."""
    return finish_stream(noise_bytes(seed, 0, nblocks * DATA_LEN, 941))


def modes1_padded(path: str) -> np.ndarray:
    raw = np.fromfile(path, dtype=np.uint8)
    pad = (-len(raw)) % DATA_LEN
    return np.concatenate([raw, np.full(pad, 127, dtype=np.uint8)])


def random_stream(i):
    """Stream i of the randomized differential test: sigma, frame density, amplitude range (down to the noise, up to
    saturation), inter-sample leak, one- and two-bit errors, seam frames all drawn from a seeded generator; one stream
    in four gets a stretch of uniform random bytes (preambles everywhere), one in four a stretch of full-scale square wave."""
    rng = np.random.RandomState(1000 + i)
    kw = dict(per=int(rng.choice([2048, 4096, 16384])), sigma_q16=int(rng.choice([300, 941, 2000])),
              amp=[(6, 12), (10, 30), (40, 100), (110, 127)][rng.randint(4)], smear=[(0,), (3, 4, 5, 6), (0, 8)][rng.randint(3)],
              flip1=int(rng.choice([0, 3, 10])), flip2=int(rng.choice([0, 4])), edge_every=int(rng.choice([0, 7, 61])))
    st = config3_stream(5000 + i, int(rng.randint(3, 9)), **kw)
    data = st.window(0, st.nbytes).copy()
    if rng.randint(4) == 0:
        lo = 2 * int(rng.randint(0, data.size // 2 - 40000))
        data[lo:lo + 65536] = rng.randint(0, 256, 65536).astype(np.uint8)
    if rng.randint(4) == 0:
        lo = 2 * int(rng.randint(0, data.size // 2 - 40000))
        period = int(rng.choice([2, 4, 6]))
        data[lo:lo + 32768] = np.where((np.arange(32768) // period) % 2 == 0, 255, 0).astype(np.uint8)
    data[-480:] = 127
    return data, kw
