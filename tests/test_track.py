"""CPU tests of the sink-side state: aircraft table, CPR positions and the two network line formats,
against what the compiled reference wrote to clients of its ports 30002 / 30003
(tests/golden/modes1_rawnet*.txt, modes1_sbs*.txt; generator: tests/golden/make_net_golden.py)."""
import ctypes as C
import os

import pytest

import oracle as orc
from dump1090_amd import HostResolver, Tracker, _native as N, raw_net_text
from helpers import maxfix_of, oracle_records

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_text(name):
    """a golden of tests/golden (the three big verbose dumps are kept gzipped)"""
    path = os.path.join(GOLD, name)
    if not os.path.exists(path) and os.path.exists(path + ".gz"):
        import gzip
        with gzip.open(path + ".gz", "rt") as f:
            return f.read()
    with open(path) as f:
        return f.read()


def modes1_messages(streams, flagset):
    flags = orc.FLAGSETS[flagset]
    recs, cands = oracle_records(streams["modes1"], maxfix_of(flags))
    r = HostResolver(**flags)
    msgs = r.resolve(recs, cands)
    r.close()
    return msgs


@pytest.mark.parametrize("flagset,tag", [("default", ""), ("aggressive", "_aggressive")])
def test_network_sinks_match_reference_capture(streams, flagset, tag):
    msgs = modes1_messages(streams, flagset)
    assert raw_net_text(msgs) == golden_text("modes1_rawnet%s.txt" % tag)
    t = Tracker()
    got = t.sbs_text(msgs, now_ms=1700000000000)
    want = golden_text("modes1_sbs%s.txt" % tag)
    assert got.count("\n") == want.count("\n")
    assert got == want
    assert "MSG,3" in got and ",37.1" in got          # positions were decoded, not just echoed as unknown
    t.close()


def test_tracker_table_and_reference_position(streams):
    msgs = modes1_messages(streams, "default")
    t = Tracker()
    t.sbs_text(msgs, now_ms=5000)
    rows = t.aircraft()
    assert sum(a["messages"] for a in rows) == len(msgs)
    assert len({a["addr"] for a in rows}) == len(rows)
    by_addr = {a["hexaddr"]: a for a in rows}
    a = by_addr["4d2023"]
    assert 36.0 < a["lat"] < 38.0 and 13.0 < a["lon"] < 15.0 and a["altitude"] > 0 and a["speed"] > 0
    lat, lon, n = t.reference_position()
    assert n > 10 and -90.0 <= lat <= 90.0 and -180.0 <= lon <= 180.0   # running mean of every decoded position
    # interactiveRemoveStaleAircrafts: nothing is older than the TTL yet; later everything is
    assert t.expire(now_ms=5000 + 60000) == 0
    assert t.expire(now_ms=5000 + 60001) == len(rows) and t.aircraft() == []
    t.close()


def test_cpr_pair_needs_both_frames_within_ten_seconds():
    """dump1090.c:2120: a position appears only when an even and an odd frame are <= 10 s apart; the newer
    frame decides which zone index is used (dump1090.c:1973)."""
    lib = N.host_lib()

    def frame(odd, lat, lon):
        mm = N.ModesMessage()
        mm.msgtype, mm.metype, mm.crcok = 17, 11, 1
        mm.aa1, mm.aa2, mm.aa3 = 0x40, 0x62, 0x1D
        mm.fflag, mm.raw_latitude, mm.raw_longitude, mm.altitude = odd, lat, lon, 38000
        return mm

    # the worked example of the CPR literature: even (93000, 51372), odd (74158, 50194) -> 52.2572 N, 3.91937 E
    even, odd = frame(0, 93000, 51372), frame(1, 74158, 50194)
    buf = C.create_string_buffer(256)
    tr = lib.modes_tracker_create()
    a = lib.modes_tracker_receive(tr, C.byref(even), 1, 1_000_000)
    assert a.contents.lat == 0 and a.contents.lon == 0
    assert lib.modes_format_sbs(C.byref(even), a, buf, 256) and b",,,,,,,38000,,,,,,,0,0,0,0" in buf.value
    a = lib.modes_tracker_receive(tr, C.byref(odd), 1, 1_020_000)          # 20 s later: no pairing
    assert a.contents.lat == 0
    a = lib.modes_tracker_receive(tr, C.byref(even), 1, 1_021_000)         # 1 s after the odd frame: even is newer
    assert abs(a.contents.lat - 52.2572) < 1e-3 and abs(a.contents.lon - 3.91937) < 1e-3
    assert lib.modes_format_sbs(C.byref(even), a, buf, 256) and b"52.25720,3.91937" in buf.value
    bad = frame(1, 74158, 50194)
    bad.crcok = 0
    assert not lib.modes_tracker_receive(tr, C.byref(bad), 1, 1_021_500)   # dump1090.c:2073
    assert lib.modes_tracker_receive(tr, C.byref(bad), 0, 1_021_500)
    lib.modes_tracker_destroy(tr)


def test_sbs_line_per_message_type():
    lib = N.host_lib()
    buf = C.create_string_buffer(256)
    tr = lib.modes_tracker_create()

    def line(**kw):
        mm = N.ModesMessage()
        mm.crcok, mm.aa1, mm.aa2, mm.aa3 = 1, 0xAB, 0xCD, 0xEF
        for k, v in kw.items():
            setattr(mm, k, v)
        a = lib.modes_tracker_receive(tr, C.byref(mm), 1, 0)
        n = lib.modes_format_sbs(C.byref(mm), a, buf, 256)
        return buf.value.decode() if n else None

    assert line(msgtype=0, altitude=1200) == "MSG,5,,,ABCDEF,,,,,,,1200,,,,,,,,,,\n"
    assert line(msgtype=4, altitude=1300, fs=3) == "MSG,5,,,ABCDEF,,,,,,,1300,,,,,,,-1,0,0,-1\n"
    assert line(msgtype=5, identity=7700, fs=5) == "MSG,6,,,ABCDEF,,,,,,,,,,,,,7700,0,-1,-1,0\n"
    assert line(msgtype=21, identity=1234, fs=0) == "MSG,6,,,ABCDEF,,,,,,,,,,,,,1234,0,0,0,0\n"
    assert line(msgtype=11) == "MSG,8,,,ABCDEF,,,,,,,,,,,,,,,,,\n"
    assert line(msgtype=17, metype=4, flight=b"KLM1023 ") == "MSG,1,,,ABCDEF,,,,,,KLM1023 ,,,,,,,,0,0,0,0\n"
    assert line(msgtype=17, metype=19, mesub=1, velocity=420, heading=271, vert_rate_sign=1, vert_rate=11) == \
        "MSG,4,,,ABCDEF,,,,,,,,420,271,,,-640,,0,0,0,0\n"
    assert line(msgtype=17, metype=19, mesub=3) is None and line(msgtype=16) is None and line(msgtype=20) is None
    lib.modes_tracker_destroy(tr)


# ---- frame-level parity: every DF / ME type, CPR all over the globe, surface positions, AP replies, bit errors ----
# tests/golden/make_frames_golden.py sent the lines of frames_in.txt to the compiled reference's raw-input port
# (decodeHexMessage, dump1090.c:2472-2502) and recorded its verbose stdout, its SBS port and its raw-output port.

def replay_frames(flagset):
    flags = orc.FLAGSETS[flagset]
    lib = N.host_lib()
    cfg = N.HostConfig(int(flags["fix"]), int(flags["aggressive"]), int(flags["check_crc"]), 0)
    h = lib.modes_host_create(C.byref(cfg))
    tr = lib.modes_tracker_create()
    verbose, sbs, rawnet = [], [], []
    vbuf, lbuf = C.create_string_buffer(1024), C.create_string_buffer(256)
    for line in golden_text("frames_in.txt").split():
        frame = bytes.fromhex(line[1:-1]).ljust(14, b"\0")
        mm = N.ModesMessage()
        lib.modes_host_decode_frame(h, frame, C.byref(mm))
        if not lib.modes_host_wants(h, C.byref(mm)):                      # dump1090.c:1803
            continue
        a = lib.modes_tracker_receive(tr, C.byref(mm), int(flags["check_crc"]), 1700000000000)
        if a and lib.modes_format_sbs(C.byref(mm), a, lbuf, 256):
            sbs.append(lbuf.value.decode())
        lib.modes_format_verbose(C.byref(mm), int(flags["check_crc"]), vbuf, 1024)
        verbose.append(vbuf.value.decode())
        lib.modes_format_raw_net(C.byref(mm), lbuf)
        rawnet.append(lbuf.value.decode())
    n = lib.modes_tracker_json(tr, 0, None, 0)                           # length first, then the text
    jbuf = C.create_string_buffer(n + 1)
    assert lib.modes_tracker_json(tr, 0, jbuf, n + 1) == n
    lib.modes_tracker_destroy(tr)
    lib.modes_host_destroy(h)
    return "".join(verbose), "".join(sbs), "".join(rawnet), jbuf.value.decode()


@pytest.mark.parametrize("flagset,tag", [("default", ""), ("aggressive", "_aggressive"), ("nofix", "_nofix")])
def test_scripted_frames_match_reference(flagset, tag):
    verbose, sbs, rawnet, table = replay_frames(flagset)
    want_raw = golden_text("frames_rawnet%s.txt" % tag)
    assert rawnet.count("\n") == want_raw.count("\n") > 1300
    assert rawnet == want_raw                                  # which frames are displayed, repaired bytes included
    want_sbs = golden_text("frames_sbs%s.txt" % tag)
    got_lines, want_lines = sbs.splitlines(), want_sbs.splitlines()
    for k, (g, w) in enumerate(zip(got_lines, want_lines)):
        assert g == w, "SBS line %d" % k
    assert len(got_lines) == len(want_lines) > 800
    assert sum("MSG,3" in ln and ln.split(",")[14] != "" for ln in got_lines) > 100     # decoded positions
    want_verbose = golden_text("frames_verbose%s.txt" % tag)
    got_v, want_v = verbose.splitlines(), want_verbose.splitlines()
    for k, (g, w) in enumerate(zip(got_v, want_v)):
        assert g == w, "verbose line %d" % k
    assert len(got_v) == len(want_v)
    # the aircraft table the reference then served as /data.json (aircraftsToJson, dump1090.c:2505-2552)
    assert table == golden_text("frames_aircraft%s.json" % tag)
    assert table.count("\"hex\"") >= 10


def test_tracker_json_of_an_empty_table_and_short_buffers():
    lib = N.host_lib()
    tr = lib.modes_tracker_create()
    buf = C.create_string_buffer(64)
    assert lib.modes_tracker_json(tr, 0, buf, 64) == 4 and buf.value == b"[\n]\n"
    assert lib.modes_tracker_json(tr, 1, buf, 3) == 4 and buf.value == b"[\n"        # truncated, still terminated
    lib.modes_tracker_destroy(tr)
