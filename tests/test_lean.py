"""CPU tests of the LEAN --raw resolve (modes_host.cpp: lean_resolve) and of the class byte it runs on
(include/modes_gfx950.h MODES_CLS_*, dump1090_amd/csrc/modes_core.h modes_classify): the listing, the counters and the whitelist
it leaves must be those of the general resolve with a raw sink - the form every earlier round pinned to the reference - on every
named stream and flag set, whether the records arrive unclassified (the oracle's), classified the way the kernels do it
(modes_host_classify) or classified for ANOTHER configuration (the host must not believe the byte)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as orc
import synth
from dump1090_amd import HostResolver, _native as N
from helpers import maxfix_of, oracle_records

CASES = ["modes1", "uniform", "coarse", "edges", "edges_smear", "frames", "smear", "lowsnr", "saturated"]


def general_listing(recs, flags, threads=1):
    """The pre-round-6 path: modes_host_resolve + modesMessage + raw sink."""
    os.environ["MODES_HOST_NO_LEAN"] = "1"
    try:
        r = HostResolver(**flags)
        n, text = r.raw_listing(recs, None, threads=1)
        out = (n, text, r.stats(), r.whitelist())
        r.close()
        return out
    finally:
        del os.environ["MODES_HOST_NO_LEAN"]


def lean_listing(recs, flags, threads=1):
    r = HostResolver(**flags)
    n, text = r.raw_listing(recs, None, threads=threads)
    out = (n, text, r.stats(), r.whitelist())
    r.close()
    return out


def pieces_listing(recs, flags, threads):
    """modes_host_resolve_raw_pieces: the listing left in the pieces' own buffers (no gathering copy), joined here; the list cut in
    two segments at a buffer boundary where it has one."""
    r = HostResolver(**flags)
    cut = int(np.searchsorted(recs["block"], recs["block"][recs.size // 2])) if recs.size else 0
    segs = [recs[:cut], recs[cut:]] if 0 < cut < recs.size else [recs]
    n, text = r.raw_listing_segments(segs, threads=threads)
    out = (n, text, r.stats(), r.whitelist())
    r.close()
    return out


def same(a, b, ctx):
    assert a[0] == b[0] and a[1] == b[1], ctx
    assert a[2] == b[2], ctx
    assert np.array_equal(a[3][0], b[3][0]) and np.array_equal(a[3][1], b[3][1]), ctx


@pytest.mark.parametrize("case", CASES)
def test_lean_listing_equals_general_resolve(golden, streams, case):
    data = streams[case]
    for fs, flags in orc.FLAGSETS.items():
        mf = maxfix_of(flags)
        recs, _ = oracle_records(data, mf)
        assert not recs["att"]["cls"].any()
        want = general_listing(recs, flags)
        assert want[1].decode() == golden[case]["raw"][fs]["text"], (case, fs)
        same(lean_listing(recs, flags), want, (case, fs, "unclassified"))
        cls = N.classify_records(recs, fix=flags["fix"], aggressive=flags["aggressive"])
        assert (cls["att"]["cls"] & 0x80).all()
        same(lean_listing(cls, flags), want, (case, fs, "classified"))
        # classified for the other policies: the byte says so, the host classifies again
        for fix, aggr in ((True, False), (True, True), (False, False)):
            if (fix, aggr) == (flags["fix"], flags["aggressive"]):
                continue
            other = N.classify_records(recs, fix=fix, aggressive=aggr)
            same(lean_listing(other, flags), want, (case, fs, "foreign class", fix, aggr))
        for th in (-2, -5):
            same(lean_listing(cls, flags, threads=th), want, (case, fs, "pieces", th))
            same(pieces_listing(cls, flags, th), want, (case, fs, "pieces left in place", th))
        # (a host starts every piece from the batch's own start state until one is caught with a wrong answer; MODES_HOST_MT_GUESS: the
        #  guess pass from the first call on - both are exact)
        os.environ["MODES_HOST_MT_GUESS"] = "1"
        try:
            same(lean_listing(cls, flags, threads=-4), want, (case, fs, "pieces, guessing"))
            same(pieces_listing(recs, flags, -3), want, (case, fs, "pieces left in place, guessing, unclassified"))
        finally:
            del os.environ["MODES_HOST_MT_GUESS"]


def test_class_byte_against_the_decoder(streams):
    """modes_classify against modes_host_decode (the field-by-field decoder pinned by the verbose goldens): for every attempt
    of every record the class says what the decoder does on an EMPTY and on a FULL whitelist."""
    lib = N.host_lib()
    seen = set()
    for case in CASES:
        data = streams[case]
        for fs, flags in orc.FLAGSETS.items():
            recs, _ = oracle_records(data, maxfix_of(flags))
            cls = N.classify_records(recs, fix=flags["fix"], aggressive=flags["aggressive"])
            cfg = N.HostConfig(int(flags["fix"]), int(flags["aggressive"]), int(flags["check_crc"]), 0)
            h = lib.modes_host_create(C.byref(cfg))
            mm = N.ModesMessage()
            for i in range(min(cls.size, 400)):
                for a in (0, 1):
                    at = cls["att"][i, a]
                    c = int(at["cls"])
                    kind = c & 7
                    seen.add(kind)
                    assert c & 0x80 and bool(c & 0x20) == flags["fix"] and bool(c & 0x40) == flags["aggressive"]
                    df = int(at["msg"][0]) >> 3
                    assert bool(c & 0x08) == (16 <= df <= 21) and bool(c & 0x10) == (at["errors"] == 0)
                    if not at["gate_ok"]:
                        assert kind == 1
                        continue
                    reaches = at["errors"] == 0 or (flags["aggressive"] and at["errors"] < 3)
                    assert (kind == 2) == (not reaches)
                    if not reaches:
                        continue
                    # empty whitelist: crcok only for CLEAN / FIXED
                    z32, z64 = np.zeros(N.ICAO_SLOTS, np.uint32), np.zeros(N.ICAO_SLOTS, np.int64)
                    lib.modes_host_set_whitelist(h, z32.ctypes.data, z64.ctypes.data)
                    att = N.Attempt.from_buffer_copy(cls["att"][i, a].tobytes())
                    lib.modes_host_decode(h, C.byref(att), C.byref(mm))
                    assert bool(mm.crcok) == (kind in (3, 4)), (case, fs, i, a, kind)
                    assert (mm.errorbit != -1) == (kind == 4)
                    addr, seen_at = np.zeros(N.ICAO_SLOTS, np.uint32), np.zeros(N.ICAO_SLOTS, np.int64)
                    lib.modes_host_get_whitelist(h, addr.ctypes.data, seen_at.ctypes.data)
                    if kind == 3:       # the one write, at the slot the record names
                        want = (int(at["msg"][1]) << 16) | (int(at["msg"][2]) << 8) | int(at["msg"][3])
                        assert addr[int(at["slot"])] == want and np.count_nonzero(addr) == (1 if want else 0)
                    else:
                        assert not addr.any()
                    # a whitelist that knows the address the class names: IID / AP turn crcok
                    if kind in (5, 6):
                        a24 = int(at["syndrome"]) if kind == 6 else (int(at["msg"][1]) << 16) | (int(at["msg"][2]) << 8) | int(at["msg"][3])
                        if a24:
                            z32[int(at["slot"])] = a24
                            lib.modes_host_set_whitelist(h, z32.ctypes.data, z64.ctypes.data)
                            lib.modes_host_decode(h, C.byref(att), C.byref(mm))
                            assert mm.crcok == 1, (case, fs, i, a, kind)
                            assert (mm.aa1 << 16 | mm.aa2 << 8 | mm.aa3) == a24
            lib.modes_host_destroy(h)
    assert seen >= {1, 2, 3, 4, 6, 7}, seen          # (IID needs a DF11 with a tiny syndrome: the constructed stream below)


def test_iid_and_ap_frames_through_the_lean_resolve():
    """DF11 replies to an interrogator (parity overlaid with an IID < 80, dump1090.c:1204) and DF4 replies (parity overlaid with the
    address, :942-983) of an aircraft whose DF17 came first - and of one that never announced itself - through both resolves."""
    icao_a, icao_b = bytes([0x4B, 0x17, 0x01]), bytes([0x3C, 0x66, 0x02])
    df17 = synth.make_frame(17, b"\x05" + icao_a + bytes(10))
    df11_iid = synth.make_frame(11, b"\x05" + icao_a + bytes(10), xor_parity=5)   # II = 5
    df4_a = synth.make_frame(4, bytes([0x00, 0x05, 0x31]) + bytes(10), xor_parity=int.from_bytes(icao_a, "big"))
    df4_b = synth.make_frame(4, bytes([0x00, 0x05, 0x31]) + bytes(10), xor_parity=int.from_bytes(icao_b, "big"))
    df11_b = synth.make_frame(11, b"\x05" + icao_b + bytes(10), xor_parity=7)
    iq = synth.noise_bytes(77, 0, 3 * synth.DATA_LEN, 627)
    for k, fr in enumerate([df11_iid, df4_a, df17, df11_iid, df4_a, df4_b, df11_b, df4_a]):
        synth.add_frame(iq, 5000 + 4000 * k, fr, 90, 7 * k)
    data = synth.finish_stream(iq)
    flags = orc.FLAGSETS["default"]
    recs, _ = oracle_records(data, 1)
    cls = N.classify_records(recs, 1)
    kinds = set((cls["att"]["cls"][:, 0] & 7).tolist())
    assert {3, 5, 6} <= kinds, kinds
    want = general_listing(recs, flags)
    ref, _ = orc.run_stream(data, **flags)
    assert want[1].decode() == orc.raw_text(ref)
    # before the DF17 nothing of aircraft A validates; after it its IID reply and its DF4 do; B's never do
    assert want[0] == 4
    same(lean_listing(recs, flags), want, "unclassified")
    same(lean_listing(cls, flags), want, "classified")
    same(lean_listing(cls, flags, threads=-3), want, "pieces")
    same(pieces_listing(cls, flags, -3), want, "pieces left in place")
    # one host, the same batch twice: the second piece of the first call starts without a guess, is caught with a wrong answer (aircraft A's
    # DF17 lies in the piece before it) and resolved again; from then on the host guesses - both calls print the sequential listing
    r = HostResolver(**flags)
    first = r.raw_listing(cls, None, threads=-3)
    r2 = HostResolver(**flags)
    assert first == r2.raw_listing(cls, None, threads=1)
    for _ in range(2):
        a, b = r.raw_listing(cls, None, threads=-3), r2.raw_listing(cls, None, threads=1)      # (the whitelist now knows A from the start)
        assert a == b and a[0] >= first[0]
    r.close(), r2.close()


def test_lean_listing_into_a_short_buffer(streams):
    """cap smaller than the listing: whole lines as long as they fit, nothing behind the first that does not, the full length reported."""
    recs, _ = oracle_records(streams["frames"], 1)
    lib = N.host_lib()
    cfg = N.HostConfig(1, 0, 1, 0)
    h = lib.modes_host_create(C.byref(cfg))
    big = C.create_string_buffer(1 << 20)
    nb = C.c_uint64()
    n = lib.modes_host_resolve_raw(h, recs.ctypes.data, recs.size, None, 0, big, len(big), C.byref(nb))
    full = big.raw[: nb.value]
    assert n > 50 and full.count(b"\n") == n and big.raw[nb.value] == 0
    lib.modes_host_destroy(h)
    for cap in (0, 1, 17, 18, 31, 32, 40, 41, 100, 500, nb.value - 1, nb.value, nb.value + 1, nb.value + 39, nb.value + 40):
        for threads in (1, -3):
            h = lib.modes_host_create(C.byref(cfg))
            buf = C.create_string_buffer(b"\xAA" * (cap + 64), cap + 64)
            nb2 = C.c_uint64()
            if threads == 1:
                n2 = lib.modes_host_resolve_raw(h, recs.ctypes.data, recs.size, None, 0, buf if cap else None, cap, C.byref(nb2))
            else:
                n2 = lib.modes_host_resolve_raw_mt(h, recs.ctypes.data, recs.size, buf if cap else None, cap, C.byref(nb2), threads)
            lib.modes_host_destroy(h)
            assert n2 == n and nb2.value == nb.value, (cap, threads)
            raw = buf.raw
            assert raw[cap:] == b"\xAA" * 64, (cap, threads, "wrote behind cap")
            if cap:
                stored = raw[:cap].split(b"\0")[0]
                assert full.startswith(stored) and (stored == b"" or stored.endswith(b";\n")), (cap, threads)
                if cap > nb.value:
                    assert stored == full


def test_cpu_budget_is_positive_and_bounded(monkeypatch):
    lib = N.host_lib()
    b = lib.modes_host_cpu_budget()
    assert 1 <= b <= (os.cpu_count() or 1)
    assert b <= len(os.sched_getaffinity(0))
