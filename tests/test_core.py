"""CPU tests of dump1090_amd/csrc/modes_core.h - the per-lane arithmetic the gfx950
kernels execute - through the clang-built shim tests/native/core_shim.cpp, against
the oracle.  (The kernels' data movement is covered by the -m gpu tests.)"""
import ctypes as C

import numpy as np
import pytest

import oracle as orc
import synth
from native.build import build as build_shim


@pytest.fixture(scope="module")
def shim():
    L = C.CDLL(build_shim())
    L.shim_power.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.shim_power_sat.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.shim_order_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.shim_level_bound.argtypes = [C.c_uint32] * 5
    L.shim_demod_both.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.shim_demod_parallel.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.shim_preamble_exact.argtypes = [C.c_void_p]
    L.shim_df_first6.argtypes = [C.c_void_p]
    L.shim_syndrome.argtypes = [C.c_void_p, C.c_int]
    L.shim_syndrome.restype = C.c_uint32
    L.shim_bit_syndrome.restype = C.c_uint32
    L.shim_find_fix.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_void_p]
    L.shim_mag_exact.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    assert L.shim_sizeof_attempt_core() == 16
    return L


def test_mag_exact_equals_the_reference_table_for_every_power(shim):
    """modes_mag_exact (the demod kernel's table-free magnitude for strong samples) == round(360 sqrt(s)) as the
    reference's LUT holds it (dump1090.c:359-364, double arithmetic), for every saturated power, not only the 5924
    that are sums of two squares; and the +-1 correction it applies is enough for any square root within one of the
    true value."""
    s = np.arange(32768, dtype=np.uint32)
    out = np.zeros_like(s)
    shim.shim_mag_exact(s.ctypes.data, s.size, out.ctypes.data)
    true_s = s.astype(np.float64)
    true_s[32767] = 32768.0                                  # the saturated value stands for I = Q = 255
    want = np.round(np.sqrt(true_s) * 360.0).astype(np.uint32)
    assert np.array_equal(out, want)
    assert out.max() == 65167
    # the correction step in isolation: m0 = want + d, d in {-1, 0, 1}, must come back to want
    n = (true_s * 129600).astype(np.uint64)
    for d in (-1, 0, 1):
        m = want.astype(np.int64) + d
        m = np.maximum(m, 0).astype(np.uint64)
        mm = m * m
        fixed = np.where((mm - m >= n) & (m > 0) & (mm >= m), m - 1, np.where(mm + m < n, m + 1, m))
        assert np.array_equal(fixed.astype(np.uint32), want), d


def numpy_forward_mask(iq):
    """What the scan kernel forwards, restated with numpy on exact powers: the ten ordering relations (dump1090.c:1602-1611;
    modes_order8_swar evaluates them exactly on even positions and as a superset on odd ones) and the level bound
    9 max(quiet) < s0 + s2 + s7 + s9 + 4 (modes_level_bound)."""
    i = iq[0::2].astype(np.int64) - 127
    q = iq[1::2].astype(np.int64) - 127
    s = np.concatenate([i * i + q * q, np.zeros(24, dtype=np.int64)])
    n = iq.size // 2
    S = lambda k: s[k:k + n]
    ok = (S(0) > np.maximum.reduce([S(1), S(3), S(4), S(5), S(6)])) & (S(2) > np.maximum(S(1), S(3))) \
        & (S(7) > S(8)) & (S(9) > np.maximum(S(8), S(6)))
    quiet = np.maximum.reduce([S(4), S(5), S(11), S(12), S(13), S(14)])
    return ok & (9 * quiet < S(0) + S(2) + S(7) + S(9) + 4)


def true_preamble_mask(iq):
    mag = np.concatenate([orc.magnitude(iq).astype(np.int64), np.zeros(24, dtype=np.int64)])
    n = iq.size // 2
    M = lambda k: mag[k:k + n]
    ok = (M(0) > M(1)) & (M(1) < M(2)) & (M(2) > M(3)) & (M(3) < M(0)) & (M(4) < M(0)) & (M(5) < M(0)) \
        & (M(6) < M(0)) & (M(7) > M(8)) & (M(8) < M(9)) & (M(9) > M(6))
    level = (M(0) + M(2) + M(7) + M(9)) // 6
    for k in (4, 5, 11, 12, 13, 14):
        ok &= M(k) < level
    return ok


def test_power_pair_all_bytes(shim):
    iq = np.stack(np.meshgrid(np.arange(256), np.arange(256), indexing="ij"), -1).astype(np.uint8).reshape(-1)
    s = np.zeros(iq.size // 2, dtype=np.uint16)
    shim.shim_power(iq.ctypes.data, s.size, s.ctypes.data)
    i = iq[0::2].astype(np.int64) - 127
    q = iq[1::2].astype(np.int64) - 127
    assert np.array_equal(s, (i * i + q * q).astype(np.uint16))


def numpy_order_mask(iq, relaxed_high=False):
    """The ten ordering relations of dump1090.c:1602-1611 on powers."""
    i = iq[0::2].astype(np.int64) - 127
    q = iq[1::2].astype(np.int64) - 127
    s = np.concatenate([i * i + q * q, np.zeros(24, dtype=np.int64)])
    n = iq.size // 2
    S = lambda k: s[k:k + n]
    return (S(0) > np.maximum.reduce([S(1), S(3), S(4), S(5), S(6)])) & (S(2) > np.maximum(S(1), S(3))) \
        & (np.minimum(S(7), S(9)) > S(8)) & (S(9) > S(6))


def numpy_order_mask_relaxed(iq):
    """Same with >= : an upper bound for what the borrow trick may additionally accept."""
    i = iq[0::2].astype(np.int64) - 127
    q = iq[1::2].astype(np.int64) - 127
    s = np.concatenate([i * i + q * q, np.zeros(24, dtype=np.int64)])
    n = iq.size // 2
    S = lambda k: s[k:k + n]
    return (S(0) >= np.maximum.reduce([S(1), S(3), S(4), S(5), S(6)])) & (S(2) >= np.maximum(S(1), S(3))) \
        & (np.minimum(S(7), S(9)) >= S(8)) & (S(9) >= S(6))


@pytest.mark.parametrize("case", ["modes1", "uniform", "coarse", "frames", "lowsnr", "noise", "extreme"])
def test_order8_swar_is_exact_on_even_positions_and_a_superset_on_odd_ones(shim, streams, case):
    if case == "extreme":
        rng = np.random.default_rng(6)
        vals = np.array([0, 1, 126, 127, 128, 254, 255], dtype=np.uint8)
        iq = vals[rng.integers(0, len(vals), 2 * 65536)]
    else:
        iq = streams[case]
    n = iq.size // 2
    flags = np.zeros(n, dtype=np.uint8)
    shim.shim_order_stream(iq.ctypes.data, n, flags.ctypes.data)
    got, exact, relaxed = flags.astype(bool), numpy_order_mask(iq), numpy_order_mask_relaxed(iq)
    assert np.array_equal(got[0::2], exact[0::2])                 # low halves: exact
    assert not np.any(exact & ~got), "alpha dropped a position that satisfies the ordering relations"
    assert not np.any(got & ~relaxed)                              # high halves: at most '>' -> '>='
    assert got.sum() <= 1.5 * exact.sum() + 64
    # every true preamble survives alpha, and the level bound keeps it
    truth = true_preamble_mask(iq)
    assert not np.any(truth & ~got)


def test_level_bound_keeps_every_true_preamble(shim, streams):
    for case in ("modes1", "frames", "coarse"):
        iq = streams[case]
        i = iq[0::2].astype(np.int64) - 127
        q = iq[1::2].astype(np.int64) - 127
        s = np.minimum(i * i + q * q, 32767)
        for p in np.flatnonzero(true_preamble_mask(iq))[:4000]:
            if p + 15 > s.size:
                continue
            quiet = max(s[p + 4], s[p + 5], s[p + 11], s[p + 12], s[p + 13], s[p + 14])
            assert shim.shim_level_bound(int(s[p]), int(s[p + 2]), int(s[p + 7]), int(s[p + 9]), int(quiet))


def test_power_sat_all_bytes(shim):
    iq = np.stack(np.meshgrid(np.arange(256), np.arange(256), indexing="ij"), -1).astype(np.uint8).reshape(-1)
    s = np.zeros(iq.size // 2, dtype=np.uint16)
    shim.shim_power_sat(iq.ctypes.data, s.size, s.ctypes.data)
    i = iq[0::2].astype(np.int64) - 127
    q = iq[1::2].astype(np.int64) - 127
    exact = i * i + q * q
    assert np.array_equal(s, np.minimum(exact, 32767).astype(np.uint16))
    # the clamp changes exactly one value and keeps the order of all values
    assert (exact > 32767).sum() == 1 and not np.any(exact == 32767)


def test_preamble_exact_matches_oracle(shim, streams):
    iq = streams["modes1"]
    mag = orc.magnitude(iq)
    truth = true_preamble_mask(iq)
    assert not np.any(truth & ~numpy_forward_mask(iq)), "the s-domain filter drops a position the reference accepts"
    idx = np.flatnonzero(numpy_forward_mask(iq))
    idx = idx[idx < mag.size - 16]
    got = np.array([shim.shim_preamble_exact(mag[p:p + 15].ctypes.data) for p in idx], dtype=bool)
    assert np.array_equal(got, truth[idx])
    assert got.sum() > 100 and (~got).sum() > 10


def test_chain_primitives(shim):
    """modes_chain / modes_chain_down / m128_rev112 against a bit-by-bit loop (via random demod windows:
    every window exercises the slicing chain; the phase branch exercises the other two)."""
    rng = np.random.default_rng(11)
    for trial in range(4000):
        style = trial % 4
        if style == 0:
            win = rng.integers(0, 65536, 241).astype(np.uint16)
        elif style == 1:                        # many weak pairs and equal neighbours
            win = (rng.integers(0, 6, 241) * 200 + 3000).astype(np.uint16)
        elif style == 2:                        # strong frame-like alternation with leak
            b = rng.integers(0, 2, 241)
            win = (b * 30000 + rng.integers(0, 9000, 241)).astype(np.uint16)
        else:                                   # first pair equal -> the value-2 packing quirk
            win = (rng.integers(0, 4, 241) * 150 + 20000).astype(np.uint16)
            win[17] = win[18]
        a = np.zeros(32, dtype=np.uint8)
        b2 = np.zeros(32, dtype=np.uint8)
        shim.shim_demod_both(win.ctypes.data, 1, a.ctypes.data)
        shim.shim_demod_parallel(win.ctypes.data, 1, b2.ctypes.data)
        assert np.array_equal(a[:16], b2[:16]), (trial, "attempt 0")
        if a[15]:
            assert np.array_equal(a[16:], b2[16:]), (trial, "attempt 1")


@pytest.mark.parametrize("case", ["modes1", "coarse", "smear", "lowsnr", "edges_smear"])
def test_demod_both_matches_oracle_records(shim, streams, case):
    data = streams[case]
    checked = 0
    for k in range(orc.block_count(data.size)):
        mag = orc.block_magnitude(data, k)
        js = orc.block_candidates(mag)
        want = orc.records(mag, js, 2)
        for j, w in zip(js, want):
            j = int(j)
            win = np.zeros(241, dtype=np.uint16)
            if j > 0:
                win[0] = mag[j - 1]
            win[1:] = mag[j:j + 240]
            out = np.zeros(32, dtype=np.uint8)
            shim.shim_demod_both(win.ctypes.data, int(j != 0), out.ctypes.data)
            par = np.zeros(32, dtype=np.uint8)
            shim.shim_demod_parallel(win.ctypes.data, int(j != 0), par.ctypes.data)
            assert np.array_equal(out[:16], par[:16]), (case, k, j, "parallel attempt 0")
            if out[15]:
                assert np.array_equal(out[16:], par[16:]), (case, k, j, "parallel attempt 1")
            a0, a1 = out[:16], out[16:]
            assert bytes(a0[:14]) == bytes(w["att"][0]["msg"]) and a0[14] == w["att"][0]["errors"] \
                and a0[15] == w["att"][0]["gate_ok"], (case, k, j)
            if w["att"][0]["gate_ok"]:
                assert bytes(a1[:14]) == bytes(w["att"][1]["msg"]) and a1[14] == w["att"][1]["errors"] \
                    and a1[15] == w["att"][1]["gate_ok"], (case, k, j)
                checked += 1
    assert checked > 5


def test_syndrome_and_fix_match_oracle(shim):
    rng = np.random.default_rng(7)
    tab = [orc.lib().orc_crc_table_entry(i) for i in range(112)]
    for p in range(112):
        assert shim.shim_bit_syndrome(p) == (tab[p] if p < 88 else 1 << (111 - p))
    L = orc.lib()
    L.orc_fix_bit_errors.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    for trial in range(3000):
        bits = 112 if trial % 2 else 56
        nb = bits // 8
        df = 17 if bits == 112 else 11
        frame = bytearray(synth.make_frame(df, bytes(rng.integers(0, 256, 14, dtype=np.uint8))))
        nflip = trial % 4                     # 0..3 flipped bits, anywhere (incl. DF field / parity)
        for b in rng.choice(bits, size=nflip, replace=False):
            frame[b >> 3] ^= 0x80 >> (b & 7)
        msg = np.frombuffer(bytes(frame) + bytes(14 - nb), dtype=np.uint8).copy()
        syn = shim.shim_syndrome(msg.ctypes.data, nb)
        assert syn == L.orc_checksum(msg.ctypes.data, bits)
        for maxfix in (1, 2):
            pos = np.zeros(2, dtype=np.uint8)
            n = shim.shim_find_fix(syn, bits, maxfix, pos.ctypes.data)
            ref_msg = msg.copy()
            fixed = (C.c_int * 2)(-1, -1)
            rn = L.orc_fix_bit_errors(ref_msg.ctypes.data, bits, maxfix, fixed)
            assert n == rn, (trial, maxfix)
            assert [int(x) for x in pos[:n]] == [fixed[i] for i in range(rn)]


def test_df_from_first_six_pairs_equals_df_of_the_full_slicing_pass(shim):
    """The gate pre-test picks the message length from modes_df_first6; it must equal msg[0] >> 3 of the whole
    first pass (dump1090.c:1669-1711) for every mix of clear, weak (|lo-hi| < 256) and equal pairs - the
    value 2 of an equal first pair travels down runs of weak pairs and ORs into the neighbouring bit."""
    rng = np.random.default_rng(7)
    out = np.zeros(32, dtype=np.uint8)
    seen = set()
    for trial in range(20000):
        win = rng.integers(0, 65168, size=241).astype(np.uint16)
        kind = rng.integers(0, 4, size=8)                    # per pair: 0 clear, 1 weak, 2 equal, 3 as drawn
        for k in range(8):
            lo = int(win[17 + 2 * k])
            if kind[k] == 1:
                win[18 + 2 * k] = np.uint16(min(65167, max(0, lo + int(rng.integers(-255, 256)))))
            elif kind[k] == 2:
                win[18 + 2 * k] = np.uint16(lo)
        shim.shim_demod_both(win.ctypes.data, 0, out.ctypes.data)
        df = int(out[0]) >> 3
        assert shim.shim_df_first6(win.ctypes.data) == df, (trial, kind[:6])
        seen.add(df)
    assert len(seen) == 32


def test_scan_power_by_byte_dot_product_formula():
    """The arithmetic of power16_scan (modes_gfx950.hip), restated with numpy for all 65,536 byte pairs: b ^ 0x7f read as a
    signed byte is 127 - b; the signed dot product of a sample's two bytes with themselves, accumulated onto 0x7fff8000
    with saturation at INT32_MAX, has 0x8000 | min(s, 32767) in its low half - and the level bound of the beta pass with
    that constant bit on every value is the same predicate as without it.  (The device's instruction is checked by
    tests/test_gpu_parity.py::test_magnitude_all_byte_pairs.)"""
    i, q = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    sb = lambda b: ((b ^ 0x7F).astype(np.uint8)).astype(np.int8).astype(np.int64)
    assert np.array_equal(sb(i), 127 - i) and sb(i).min() == -128 and sb(i).max() == 127
    acc = 0x7FFF8000 + sb(i) * sb(i) + sb(q) * sb(q)
    acc = np.minimum(acc, 0x7FFFFFFF)                                    # v_dot4_i32_i8 ... clamp
    s = (i - 127) ** 2 + (q - 127) ** 2
    assert np.array_equal(acc & 0xFFFF, 0x8000 | np.minimum(s, 32767))
    assert (s == 32768).sum() == 1 and not (s == 32767).any()            # only I = Q = 255 saturates; nothing collides with it
    rng = np.random.default_rng(9)
    v = rng.integers(0, 32768, size=(200000, 5))
    b = 0x8000
    plain = 9 * v[:, 4] < v[:, 0] + v[:, 1] + v[:, 2] + v[:, 3] + 4
    biased = 9 * (v[:, 4] + b) < (v[:, 0] + b) + (v[:, 1] + b) + (v[:, 2] + b) + (v[:, 3] + b + 5 * b) + 4
    assert np.array_equal(plain, biased)


def test_beta_pass_flag_gather_formula():
    """scan_beta (modes_gfx950.hip) gathers the eight ordering flags of a queue entry - bits 15 / 31 of the four result
    words r[q] <-> positions 2q / 2q + 1 - with two v_perm into the bytes of two words, and walks
    f = ((lo & 0x80808080) >> 7) | ((hi & 0x80808080) >> 3) bit by bit with position = (b >> 3) | (b & 4): every subset of
    positions must come back, whatever else the result words hold."""
    rng = np.random.default_rng(10)
    for mask in range(256):
        r = rng.integers(0, 1 << 32, size=4, dtype=np.uint64) & np.uint64(0x7FFF7FFF)          # garbage in the other bits
        for p in range(8):
            if mask >> p & 1:
                r[p >> 1] |= np.uint64(1 << (31 if p & 1 else 15))
        byte = lambda w, k: (int(w) >> (8 * k)) & 0xFF
        lo = byte(r[0], 1) | byte(r[0], 3) << 8 | byte(r[1], 1) << 16 | byte(r[1], 3) << 24   # v_perm(r1, r0, 0x07050301)
        hi = byte(r[2], 1) | byte(r[2], 3) << 8 | byte(r[3], 1) << 16 | byte(r[3], 3) << 24
        f = ((lo & 0x80808080) >> 7) | ((hi & 0x80808080) >> 3)
        got = 0
        while f:
            b = (f & -f).bit_length() - 1
            f &= f - 1
            got |= 1 << ((b >> 3) | (b & 4))
        assert got == mask


def test_scan_ring_layout_model():
    """The scan kernel's 66-entry LDS ring (modes_gfx950.hip: scan_run), modelled: a wavefront's LDS operations execute in order, so one
    chunk's step is write(all lanes) then read a, b(all lanes).  Every lane must see the two lanes before it - the previous chunk's
    lanes 62, 63 for lanes 0, 1 - and every 16-lane group of a b128 operation must touch 16 different bank quads
    (MI355X_MICROARCH: ds_read_b128 is served in four groups of 16 lanes, bank = (address / 4) mod 64)."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in g] for g in groups]
    wr = {0: lambda l: 2 + l, 1: lambda l: 2 + l if l < 62 else l - 62}
    rda = {0: lambda l: l, 1: lambda l: 64 + l if l < 2 else l}
    rdb = {0: lambda l: l + 1, 1: lambda l: 65 if l == 0 else (0 if l == 63 else l + 1)}
    ring = [None] * 66
    for lane in (62, 63):                                      # prologue: the look-back of chunk 0 where an odd chunk's lanes 62, 63 write
        ring[wr[1](lane)] = (-1, lane)
    for chunk in range(7):
        par = chunk & 1
        for fn in (wr[par], rda[par], rdb[par]):
            assert all(0 <= fn(l) < 66 for l in range(64))
            for g in groups:                                   # 16 entries, 16 different residues mod 16: no bank conflict
                assert len({fn(l) % 16 for l in g}) == 16, (chunk, g)
        for lane in range(64):
            ring[wr[par](lane)] = (chunk, lane)
        for lane in range(64):
            want_a = (chunk, lane - 2) if lane >= 2 else (chunk - 1, 62 + lane)
            want_b = (chunk, lane - 1) if lane >= 1 else (chunk - 1, 63)
            assert ring[rda[par](lane)] == want_a and ring[rdb[par](lane)] == want_b, (chunk, lane)
