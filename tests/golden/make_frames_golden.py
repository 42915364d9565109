#!/usr/bin/env python3
"""Golden output of the reference for a scripted list of FRAMES (no samples involved).

The reference accepts "*<hex>;" lines on its raw-input TCP port and hands them to
decodeModesMessage + useModesMessage (decodeHexMessage, dump1090.c:2472-2502).  This script writes a
deterministic list of frames - every downlink format, every extended-squitter type the decoder knows,
even/odd airborne CPR pairs all over the globe, surface positions, AP-protected replies of known and
unknown aircraft, single- and double-bit errors - feeds it to the compiled reference
(oracle/_ref/dump1090_ref --net-only, constant clock) and records
    frames_in[_<tag>].txt        the lines sent
    frames_verbose[_<tag>].txt   its stdout: the verbose dump of every message it displays
    frames_sbs[_<tag>].txt       what it wrote to a client of its BaseStation port
    frames_rawnet[_<tag>].txt    what it wrote to a client of its raw-output port
    frames_aircraft[_<tag>].json the aircraft table it then served as /data.json
so that the host decoder, the whitelist, the tracker and the formatters are pinned on inputs the one
captured fixture (modes1.bin: a single aircraft) never produces.
"""
import os
import pty
import random
import select
import socket
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import oracle as orc                                  # noqa: E402
from dump1090_amd import _native as N                 # noqa: E402   (only modes_compute_crc: the parity generator)

PORTS = dict(ro=31012, ri=31011, http=31090, sbs=31013)
AIS = "?ABCDEFGHIJKLMNOPQRSTUVWXYZ????? ???????????????0123456789??????"


def parity(frame: bytearray, bits: int) -> int:
    return N.host_lib().modes_compute_crc(bytes(frame), bits)


def finish(frame: bytearray, bits: int, xor_addr: int = 0) -> bytes:
    """Write the parity field (XORed with an address for AP-protected formats)."""
    n = bits // 8
    crc = parity(frame, bits) ^ xor_addr
    frame[n - 3], frame[n - 2], frame[n - 1] = (crc >> 16) & 0xff, (crc >> 8) & 0xff, crc & 0xff
    return bytes(frame[:n])


def put_bits(frame: bytearray, first: int, width: int, value: int):
    """Message bits are numbered from 0 at the MSB of byte 0."""
    for k in range(width):
        bit = (value >> (width - 1 - k)) & 1
        byte, off = divmod(first + k, 8)
        frame[byte] = (frame[byte] & ~(0x80 >> off)) | ((0x80 >> off) if bit else 0)


def es_frame(rng, df, addr, me: int, ca=5) -> bytes:
    f = bytearray(14)
    f[0] = (df << 3) | ca
    f[1], f[2], f[3] = (addr >> 16) & 0xff, (addr >> 8) & 0xff, addr & 0xff
    put_bits(f, 32, 56, me)
    return finish(f, 112)


def me_ident(rng, metype, text):
    me = (metype << 51) | (rng.randrange(8) << 48)
    for i, ch in enumerate(text.ljust(8)[:8]):
        me |= AIS.index(ch) << (42 - 6 * i)
    return me


def me_airborne(rng, metype, odd, lat17, lon17, alt12):
    return (metype << 51) | (rng.randrange(4) << 49) | (rng.randrange(2) << 48) | (alt12 << 36) | (rng.randrange(2) << 35) \
        | (odd << 34) | (lat17 << 17) | lon17


def me_surface(rng, metype, odd, lat17, lon17):
    return (metype << 51) | (rng.randrange(128) << 44) | (rng.randrange(2) << 43) | (rng.randrange(128) << 36) \
        | (rng.randrange(2) << 35) | (odd << 34) | (lat17 << 17) | lon17


def me_velocity(rng, sub):
    return (19 << 51) | (sub << 48) | rng.getrandbits(48)


def short_frame(rng, df, addr, payload27, known=True) -> bytes:
    """DF0/4/5: 5 bits DF + 27 bits payload + 24 bits address/parity."""
    f = bytearray(14)
    put_bits(f, 0, 5, df)
    put_bits(f, 5, 27, payload27)
    return finish(f, 56, xor_addr=addr)


def long_ap_frame(rng, df, addr) -> bytes:
    f = bytearray(14)
    put_bits(f, 0, 5, df)
    put_bits(f, 5, 83, rng.getrandbits(83))
    return finish(f, 112, xor_addr=addr)


def df11_frame(rng, addr, iid=0) -> bytes:
    f = bytearray(14)
    f[0] = (11 << 3) | rng.randrange(8)
    f[1], f[2], f[3] = (addr >> 16) & 0xff, (addr >> 8) & 0xff, addr & 0xff
    return finish(f, 56, xor_addr=iid)               # the interrogator id rides on the parity (dump1090.c:1181-1193)


def flip(frame: bytes, *bits) -> bytes:
    b = bytearray(frame)
    for k in bits:
        b[k >> 3] ^= 0x80 >> (k & 7)
    return bytes(b)


def cpr_encode(lat, lon, odd):
    """Airborne CPR encoding (the inverse of dump1090.c:1952-1990) - only to get frames that decode to
    sensible places; the golden does not depend on it being right."""
    import math
    dlat = 360.0 / (59 if odd else 60)
    yz = int(math.floor(131072 * ((lat % dlat) / dlat) + 0.5))
    rlat = dlat * (yz / 131072.0 + math.floor(lat / dlat))
    nl = 1
    if abs(rlat) < 87:
        nl = int(math.floor(2 * math.pi / math.acos(1 - (1 - math.cos(math.pi / 30)) / math.cos(math.pi / 180 * abs(rlat)) ** 2)))
    dlon = 360.0 / max(nl - odd, 1)
    xz = int(math.floor(131072 * ((lon % dlon) / dlon) + 0.5))
    return yz & 0x1ffff, xz & 0x1ffff


def script(seed):
    rng = random.Random(seed)
    addrs = [rng.randrange(1, 1 << 24) for _ in range(24)]
    out = []
    # 1. squitters: acquisition + identification + airborne positions of a small fleet, interleaved
    places = [(52.25, 3.92), (37.1, 13.8), (-33.9, 151.2), (64.1, -21.9), (1.35, 103.99), (-54.8, -68.3), (78.2, 15.6),
              (0.01, -0.01), (35.6, 139.8), (-12.0, -77.1), (89.0, 10.0), (-89.5, -170.0), (45.0, 179.99), (45.0, -179.99)]
    for i, a in enumerate(addrs[:14]):
        out.append(df11_frame(rng, a))
        out.append(es_frame(rng, 17, a, me_ident(rng, 1 + i % 4, "TEST%03d" % i)))
    for rnd in range(6):
        for i, a in enumerate(addrs[:14]):
            lat, lon = places[i]
            lat = max(-89.9, min(89.9, lat + 0.02 * rnd))
            lon += 0.03 * rnd
            odd = (rnd + i) & 1
            yz, xz = cpr_encode(lat, lon, odd)
            alt12 = rng.randrange(1 << 12)
            out.append(es_frame(rng, 17 if i % 5 else 18, a, me_airborne(rng, 9 + (i + rnd) % 10, odd, yz, xz, alt12)))
            if rnd % 2:
                out.append(es_frame(rng, 17, a, me_velocity(rng, 1 + (i % 4))))
    # 2. arbitrary raw CPR values (band mismatches, poles, wrap-around)
    for k in range(300):
        a = addrs[rng.randrange(14)]
        out.append(es_frame(rng, 17, a, me_airborne(rng, rng.randrange(9, 19), rng.randrange(2), rng.getrandbits(17), rng.getrandbits(17),
                                                    rng.randrange(1 << 12))))
    # 3. surface positions (need the reference position the airborne ones produced), movement and track fields
    for k in range(200):
        a = addrs[rng.randrange(14)]
        if k % 3 == 0:
            lat, lon = places[rng.randrange(len(places))]
            yz, xz = cpr_encode(lat, lon, k & 1)
            yz, xz = (yz * 4) & 0x1ffff, (xz * 4) & 0x1ffff          # surface encoding covers a quarter of the range
        else:
            yz, xz = rng.getrandbits(17), rng.getrandbits(17)
        out.append(es_frame(rng, 17, a, me_surface(rng, rng.randrange(5, 9), k & 1, yz, xz)))
    # 4. every extended-squitter type / subtype with random payloads (operational status, TCAS, test, reserved ...)
    for metype in range(32):
        for sub in range(8):
            out.append(es_frame(rng, 17 + (sub & 1), addrs[rng.randrange(14)], (metype << 51) | (sub << 48) | rng.getrandbits(48)))
    # 5. AP-protected replies: surveillance altitude / identity (all flight statuses, emergency squawks), comm-B, ACAS
    for k in range(400):
        known = k % 4 != 3
        a = addrs[rng.randrange(14)] if known else rng.randrange(1, 1 << 24)
        kind = k % 6
        if kind == 0:
            out.append(short_frame(rng, 0, a, rng.getrandbits(27)))
        elif kind == 1:
            out.append(short_frame(rng, 4, a, rng.getrandbits(27)))
        elif kind == 2:
            out.append(short_frame(rng, 5, a, rng.getrandbits(27)))
        elif kind == 3:
            out.append(long_ap_frame(rng, 16, a))
        elif kind == 4:
            out.append(long_ap_frame(rng, 20, a))
        else:
            out.append(long_ap_frame(rng, 21, a))
    for squawk_bits in (0x0aaa, 0x1555, 0x1fff, 0x0000):
        for fs in range(8):
            out.append(short_frame(rng, 5, addrs[3], (fs << 24) | (rng.getrandbits(11) << 13) | squawk_bits))
    # 6. all-call replies with interrogator ids, of known and of new aircraft
    for k in range(120):
        a = addrs[rng.randrange(24)]
        out.append(df11_frame(rng, a, iid=rng.randrange(16) if k % 2 else 0))
    # 7. bit errors: one and two flipped bits in DF11 / DF17 / DF18 (repairable) and in other formats (not repaired)
    for k in range(300):
        a = addrs[rng.randrange(14)]
        base = [df11_frame(rng, a), es_frame(rng, 17, a, rng.getrandbits(56)), es_frame(rng, 18, a, rng.getrandbits(56)),
                short_frame(rng, 4, a, rng.getrandbits(27)), long_ap_frame(rng, 20, a)][k % 5]
        nb = len(base) * 8
        out.append(flip(base, rng.randrange(nb)) if k % 2 else flip(base, *rng.sample(range(nb), 2)))
    # 8. formats the decoder does not know, and pure junk
    for df in (1, 2, 3, 6, 7, 8, 9, 10, 12, 13, 14, 15, 19, 22, 23, 24, 25, 31):
        f = bytearray(rng.getrandbits(8) for _ in range(14))
        f[0] = (df << 3) | (f[0] & 7)
        out.append(bytes(f[:14 if 16 <= df <= 21 else 7]))
    for k in range(100):
        f = bytes(rng.getrandbits(8) for _ in range(14))
        out.append(f[:14 if 16 <= (f[0] >> 3) <= 21 else 7])
    return out


def run_reference(lines, flags):
    env = dict(os.environ, LD_PRELOAD=orc.FIXED_TIME)
    args = [orc.REF_BIN, "--net-only", "--net-ro-port", str(PORTS["ro"]), "--net-ri-port", str(PORTS["ri"]),
            "--net-http-port", str(PORTS["http"]), "--net-sbs-port", str(PORTS["sbs"])] + flags
    master, slave = pty.openpty()                        # a terminal: the reference's stdout becomes line buffered
    proc = subprocess.Popen(args, stdout=slave, stderr=subprocess.DEVNULL, env=env)
    os.close(slave)
    got = {"ro": b"", "sbs": b"", "out": b""}
    stop = threading.Event()

    def sock_reader(name):
        for _ in range(100):
            try:
                s = socket.create_connection(("127.0.0.1", PORTS[name]), timeout=5)
                break
            except OSError:
                time.sleep(0.05)
        s.settimeout(0.2)
        while not stop.is_set():
            try:
                chunk = s.recv(1 << 16)
            except socket.timeout:
                continue
            if not chunk:
                break
            got[name] += chunk
        s.close()

    def pty_reader():
        while not stop.is_set():
            r, _, _ = select.select([master], [], [], 0.2)
            if r:
                try:
                    chunk = os.read(master, 1 << 16)
                except OSError:
                    break
                if not chunk:
                    break
                got["out"] += chunk

    threads = [threading.Thread(target=sock_reader, args=("ro",)), threading.Thread(target=sock_reader, args=("sbs",)),
               threading.Thread(target=pty_reader)]
    for t in threads:
        t.start()
    time.sleep(0.8)                                      # the two output clients are accepted
    src = socket.create_connection(("127.0.0.1", PORTS["ri"]), timeout=5)
    for k in range(0, len(lines), 8):
        src.sendall("".join(lines[k:k + 8]).encode())
        time.sleep(0.004)
    time.sleep(1.0)
    src.close()
    time.sleep(0.3)
    # the aircraft table as the web map polls it (handleHTTPRequest -> aircraftsToJson, dump1090.c:2505-2552)
    web = socket.create_connection(("127.0.0.1", PORTS["http"]), timeout=5)
    web.sendall(b"GET /data.json HTTP/1.0\r\n\r\n")
    web.settimeout(5)
    reply = b""
    while True:
        chunk = web.recv(1 << 16)
        if not chunk:
            break
        reply += chunk
    web.close()
    got["json"] = reply.split(b"\r\n\r\n", 1)[1]
    time.sleep(0.2)
    stop.set()
    proc.terminate()
    proc.wait()
    for t in threads:
        t.join()
    os.close(master)
    return got["out"].decode().replace("\r\n", "\n"), got["sbs"].decode(), got["ro"].decode(), got["json"].decode()


def main():
    frames = script(20260922)
    lines = ["*%s;\n" % f.hex() for f in frames]
    for tag, flags in (("", []), ("_aggressive", ["--aggressive"]), ("_nofix", ["--no-fix"])):
        verbose, sbs, ro, table = run_reference(lines, flags)
        open(os.path.join(HERE, "frames_in.txt"), "w").write("".join(lines))
        import gzip
        with gzip.open(os.path.join(HERE, "frames_verbose%s.txt.gz" % tag), "wt") as gz:      # 48,000 lines: kept compressed
            gz.write(verbose)
        open(os.path.join(HERE, "frames_sbs%s.txt" % tag), "w").write(sbs)
        open(os.path.join(HERE, "frames_rawnet%s.txt" % tag), "w").write(ro)
        open(os.path.join(HERE, "frames_aircraft%s.json" % tag), "w").write(table)
        print(tag or "default", len(lines), "frames ->", ro.count("\n"), "displayed,", sbs.count("\n"), "SBS lines,",
              verbose.count("\n"), "lines of verbose text")


if __name__ == "__main__":
    main()
