"""ctypes view of oracle/liboracle.so (CPU restatement) and oracle/_ref (the
compiled reference).  TEST INFRASTRUCTURE ONLY: imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the
product package dump1090_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_BIN = os.path.join(HERE, "_ref", "dump1090_ref")
FIXED_TIME = os.path.join(HERE, "_ref", "libfixedtime.so")

BLOCK_BYTES = 262620
BLOCK_SAMPLES = 131310
DATA_LEN = 262144


class Config(C.Structure):
    _fields_ = [("fix_errors", C.c_int), ("aggressive", C.c_int), ("check_crc", C.c_int)]


class Message(C.Structure):
    _fields_ = [("msg", C.c_uint8 * 14), ("msgbits", C.c_int32), ("msgtype", C.c_int32),
                ("crcok", C.c_int32), ("crc", C.c_uint32), ("errorbit", C.c_int32),
                ("aa1", C.c_int32), ("aa2", C.c_int32), ("aa3", C.c_int32),
                ("phase_corrected", C.c_int32), ("iid", C.c_int32),
                ("block", C.c_uint32), ("j", C.c_uint32)]


STAT_NAMES = ("valid_preamble", "out_of_phase", "demodulated", "goodcrc", "badcrc",
              "fixed", "single_bit_fix", "two_bits_fix")


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in STAT_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in STAT_NAMES}


ATTEMPT_DTYPE = np.dtype([("msg", np.uint8, 14), ("errors", np.uint8), ("gate_ok", np.uint8),
                          ("nfix", np.uint8), ("fixpos", np.uint8, 2), ("pad", np.uint8, 5),
                          ("syndrome", np.uint32)])
RECORD_DTYPE = np.dtype([("j", np.uint32), ("att", ATTEMPT_DTYPE, 2)])
assert ATTEMPT_DTYPE.itemsize == 28 and RECORD_DTYPE.itemsize == 60


def build():
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", HERE, "all"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_run_stream.argtypes = [C.POINTER(Config), C.c_void_p, C.c_size_t, C.POINTER(Message),
                                     C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Stats)]
        L.orc_run_stream.restype = C.c_int
        L.orc_magnitude.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_frame_block.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
        L.orc_block_count.argtypes = [C.c_size_t]
        L.orc_block_count.restype = C.c_uint64
        L.orc_block_candidates.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        L.orc_block_candidates.restype = C.c_size_t
        L.orc_record_at.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_crc_table_entry.argtypes = [C.c_int]
        L.orc_crc_table_entry.restype = C.c_uint32
        L.orc_build_maglut.argtypes = [C.c_void_p]
        L.orc_checksum.argtypes = [C.c_void_p, C.c_int]
        L.orc_checksum.restype = C.c_uint32
        _lib = L
    return _lib


def config(fix=True, aggressive=False, check_crc=True) -> Config:
    return Config(int(fix), int(aggressive), int(check_crc))


FLAGSETS = {
    "default": dict(fix=True, aggressive=False, check_crc=True),
    "nofix": dict(fix=False, aggressive=False, check_crc=True),
    "aggressive": dict(fix=True, aggressive=True, check_crc=True),
    "nocrc": dict(fix=True, aggressive=False, check_crc=False),
    "nofix_nocrc": dict(fix=False, aggressive=False, check_crc=False),
    "aggressive_nocrc": dict(fix=True, aggressive=True, check_crc=False),
}


def run_stream(stream: np.ndarray, cap: int = 1 << 20, **flags):
    """-> (list[Message], stats dict) for a whole byte stream."""
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    cfg = config(**flags)
    msgs = (Message * cap)()
    n = C.c_size_t()
    st = Stats()
    rc = lib().orc_run_stream(C.byref(cfg), stream.ctypes.data, stream.size, msgs, cap, C.byref(n), C.byref(st))
    assert rc == 0 and n.value <= cap
    return list(msgs[: n.value]), st.as_dict()


def raw_text(msgs) -> str:
    """--raw listing (dump1090.c:1324-1326)."""
    return "".join("*" + bytes(m.msg[: m.msgbits // 8]).hex() + ";\n" for m in msgs)


def onlyaddr_text(msgs) -> str:
    """--onlyaddr listing (dump1090.c:1318-1320)."""
    return "".join("%02x%02x%02x\n" % (m.aa1, m.aa2, m.aa3) for m in msgs)


def stats_text(st: dict) -> str:
    """--stats summary (dump1090.c:2993-3006)."""
    return ("%d valid preambles\n%d demodulated again after phase correction\n"
            "%d demodulated with zero errors\n%d with good crc\n%d with bad crc\n"
            "%d errors corrected\n%d single bit errors\n%d two bits errors\n"
            "%d total usable messages\n" % (
                st["valid_preamble"], st["out_of_phase"], st["demodulated"], st["goodcrc"], st["badcrc"],
                st["fixed"], st["single_bit_fix"], st["two_bits_fix"], st["goodcrc"] + st["fixed"]))


def block_count(nbytes: int) -> int:
    return int(lib().orc_block_count(nbytes))


def frame_block(stream: np.ndarray, k: int) -> np.ndarray:
    out = np.empty(BLOCK_BYTES, dtype=np.uint8)
    lib().orc_frame_block(stream.ctypes.data, stream.size, k, out.ctypes.data)
    return out


def magnitude(iq: np.ndarray) -> np.ndarray:
    iq = np.ascontiguousarray(iq, dtype=np.uint8)
    out = np.empty(iq.size // 2, dtype=np.uint16)
    lib().orc_magnitude(iq.ctypes.data, out.size, out.ctypes.data)
    return out


def block_magnitude(stream: np.ndarray, k: int) -> np.ndarray:
    return magnitude(frame_block(stream, k))


def block_candidates(mag: np.ndarray) -> np.ndarray:
    js = np.empty(mag.size, dtype=np.uint32)
    n = lib().orc_block_candidates(mag.ctypes.data, mag.size, js.ctypes.data, js.size)
    return js[:n].copy()


def records(mag: np.ndarray, js, maxfix: int) -> np.ndarray:
    out = np.zeros(len(js), dtype=RECORD_DTYPE)
    for i, j in enumerate(js):
        lib().orc_record_at(mag.ctypes.data, int(j), maxfix, out[i:i + 1].ctypes.data)
    return out


def maglut() -> np.ndarray:
    out = np.empty(129 * 129, dtype=np.uint16)
    lib().orc_build_maglut(out.ctypes.data)
    return out


def have_ref() -> bool:
    return os.path.exists(REF_BIN) and os.path.exists(FIXED_TIME)


def run_ref(path: str, flags: list[str]) -> str:
    """stdout of the compiled reference under the constant-clock interposer."""
    env = dict(os.environ, LD_PRELOAD=FIXED_TIME)
    return subprocess.run([REF_BIN, "--ifile", path] + flags, capture_output=True, env=env, check=True).stdout.decode()


def run_ref_bytes(data: np.ndarray, flags: list[str]) -> bytes:
    """stdout of the compiled reference fed through --ifile - (stdin), constant clock.  For streams too
    big for a temp file; `data` must be oracle-safe (multiple of 262144 bytes, >= 480 trailing 127s)."""
    import threading
    env = dict(os.environ, LD_PRELOAD=FIXED_TIME)
    proc = subprocess.Popen([REF_BIN, "--ifile", "-"] + flags, stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
    view = memoryview(np.ascontiguousarray(data, dtype=np.uint8)).cast("B")

    def feed():
        step = 1 << 24
        try:
            for lo in range(0, len(view), step):
                proc.stdin.write(view[lo:lo + step])
        finally:
            proc.stdin.close()

    t = threading.Thread(target=feed, daemon=True)
    t.start()
    out = proc.stdout.read()
    t.join()
    assert proc.wait() == 0
    return out
