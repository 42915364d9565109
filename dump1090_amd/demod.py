"""Host-side mirror of the reference's two hot-path calls, on top of the C ABI.

    reference (dump1090.c)                      here
    ------------------------------------------  -------------------------------------------
    computeMagnitudeVector()          :1454     Demodulator.compute_magnitude_vector(iq)
    detectModeS(m, mlen)              :1563     Demodulator.detect(iq, ...) + .fetch()   (GPU)
      ... decodeModesMessage(&mm,msg) :1735       HostResolver.resolve(records)           (CPU, in order)
      ... useModesMessage(&mm)        :1777       -> list[Message] (what the sink would show)
    main loop over buffers            :2969     Demodulator.demodulate(stream)

torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as N


def block_count(nbytes: int) -> int:
    """Buffers the reference's reader publishes for an nbytes stream (dump1090.c:484-510)."""
    return nbytes // N.DATA_LEN + 1


def shard_blocks(nblocks: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous partition of buffers [0, nblocks) over `world` ranks -> (first_block, count)."""
    base, extra = divmod(nblocks, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def shard_byte_range(first_block: int, nblocks: int, stream_nbytes: int) -> tuple[int, int]:
    """Stream bytes a rank needs for its buffers: [start - 476, end), clipped to the stream
    (the 476-byte carry of dump1090.c:481 is the only overlap between shards)."""
    lo = max(0, first_block * N.DATA_LEN - N.CARRY_BYTES)
    hi = min(stream_nbytes, (first_block + nblocks) * N.DATA_LEN)
    return lo, max(lo, hi)


@dataclass
class Message:
    """One message the reference's sink would display (fields of struct modesMessage)."""
    msg: bytes
    msgbits: int
    msgtype: int
    crcok: int
    crc: int
    errorbit: int
    aa1: int
    aa2: int
    aa3: int
    phase_corrected: int
    iid: int
    block: int
    j: int
    fields: dict
    verbose: str = ""            # modes_format_verbose(): the reference's default (non --raw) dump of this message

    def raw_line(self) -> str:
        return "*" + self.msg[: self.msgbits // 8].hex() + ";\n"         # dump1090.c:1324-1326

    def addr_line(self) -> str:
        return "%02x%02x%02x\n" % (self.aa1, self.aa2, self.aa3)          # dump1090.c:1319


_EXTRA_FIELDS = [f for f, _ in N.ModesMessage._fields_ if f not in (
    "msg", "msgbits", "msgtype", "crcok", "crc", "errorbit", "aa1", "aa2", "aa3", "phase_corrected", "iid")]


def _to_message(e: N.Emitted, check_crc: bool = True) -> Message:
    mm = e.mm
    buf = C.create_string_buffer(1024)
    N.host_lib().modes_format_verbose(C.byref(mm), int(check_crc), buf, 1024)
    extra = {}
    for f in _EXTRA_FIELDS:
        v = getattr(mm, f)
        extra[f] = v.decode("ascii", "replace") if isinstance(v, bytes) else int(v)
    return Message(bytes(mm.msg), mm.msgbits, mm.msgtype, mm.crcok, mm.crc, mm.errorbit, mm.aa1, mm.aa2, mm.aa3,
                   mm.phase_corrected, mm.iid, e.block, e.j, extra, buf.value.decode("ascii", "replace"))


class HostResolver:
    """The sequential half: records -> messages (libmodes_host.so).  No GPU needed."""

    def __init__(self, fix: bool = True, aggressive: bool = False, check_crc: bool = True, text_buffer=None):
        """text_buffer: the grow-only listing buffer of a resolver that is being replaced (take_text_buffer()) - a step loop
        that starts every step with a fresh whitelist must not also pay for a fresh buffer (34 MB, zero-filled and
        faulted in under the GIL, for the 524,000 lines of an 8-GPU step: 7 ms)."""
        self._lib = N.host_lib()
        self.check_crc = check_crc
        cfg = N.HostConfig(int(fix), int(aggressive), int(check_crc), 0)
        self._h = self._lib.modes_host_create(C.byref(cfg))
        if not self._h:
            raise N.ModesError(-3, "modes_host_create failed")
        self._rawbuf = text_buffer   # grow-only text buffer of raw_listing*: per resolver (ctypes drops the GIL inside the C call,
                                     # two resolvers on two threads must not format into one buffer)

    def take_text_buffer(self):
        """Hand the listing buffer to a successor (HostResolver(text_buffer=...)); this resolver allocates anew if used again."""
        buf, self._rawbuf = self._rawbuf, None
        return buf

    def _text_buffer(self, cap):
        if self._rawbuf is None or len(self._rawbuf) < cap:
            self._rawbuf = (C.c_char * (cap + cap // 4))()
        return self._rawbuf

    def close(self):
        if getattr(self, "_h", None):
            self._lib.modes_host_destroy(self._h)
            self._h = None

    __del__ = close

    def set_time(self, now_seconds: int):
        """Advance the clock behind the ICAO whitelist's 60 s TTL (dump1090.c:913,924); a file run never does."""
        self._lib.modes_host_set_time(self._h, int(now_seconds))

    def resolve(self, records: np.ndarray, candidates: np.ndarray | None = None) -> list[Message]:
        """Records (RECORD_DTYPE, ascending (block, j)) -> messages that pass the display filter."""
        records = np.ascontiguousarray(records, dtype=N.RECORD_DTYPE)
        cptr, ncand = None, 0
        if candidates is not None:
            candidates = np.ascontiguousarray(candidates, dtype=np.uint64)
            cptr, ncand = candidates.ctypes.data, candidates.size
        cap = 2 * records.size + 16                       # at most two sink calls per record
        out = (N.Emitted * cap)()
        n = self._lib.modes_host_resolve_to_array(self._h, records.ctypes.data, records.size, cptr, ncand, out, cap)
        assert n <= cap
        return [_to_message(out[i], self.check_crc) for i in range(n)]

    def count(self, records: np.ndarray, candidates: np.ndarray | None = None) -> int:
        """Like resolve() but only counts the messages (no Python objects; used by the bench)."""
        records = np.ascontiguousarray(records, dtype=N.RECORD_DTYPE)
        cptr, ncand = None, 0
        if candidates is not None:
            candidates = np.ascontiguousarray(candidates, dtype=np.uint64)
            cptr, ncand = candidates.ctypes.data, candidates.size
        return int(self._lib.modes_host_resolve_to_array(self._h, records.ctypes.data, records.size, cptr, ncand,
                                                         None, 0))

    def raw_listing(self, records: np.ndarray, candidates: np.ndarray | None = None, threads: int = 1, text: bool = True):
        """(number of lines, the --raw listing) of a batch, formatted in C (modes_host_resolve_raw; with threads > 1 and
        no candidates: modes_host_resolve_raw_mt, the same listing from several threads).  text=False: the listing is
        formatted all the same but stays in the resolver's buffer - (number of lines, None); a step loop that only checks
        its last listing saves the copy into a Python object (made under the GIL) on all the others."""
        records = np.ascontiguousarray(records, dtype=N.RECORD_DTYPE)
        cptr, ncand = None, 0
        if candidates is not None:
            candidates = np.ascontiguousarray(candidates, dtype=np.uint64)
            cptr, ncand = candidates.ctypes.data, candidates.size
        if (threads > 1 or threads < 0) and candidates is None and not text:
            return self._pieces([records], threads)[0], None        # nothing to hand back: the listing stays in the library's pieces
        buf = self._text_buffer(62 * records.size + 64)   # at most two 31-byte lines per record
        nbytes = C.c_uint64()
        if (threads > 1 or threads < 0) and candidates is None:
            n = self._lib.modes_host_resolve_raw_mt(self._h, records.ctypes.data, records.size, buf, len(buf),
                                                    C.byref(nbytes), threads)
        else:
            n = self._lib.modes_host_resolve_raw(self._h, records.ctypes.data, records.size, cptr, ncand, buf, len(buf),
                                                 C.byref(nbytes))
        return int(n), (C.string_at(buf, nbytes.value) if text else None)     # copies the listing only, not the whole buffer

    def raw_listing_segments(self, segments, threads: int = 1, text: bool = True):
        """raw_listing of a batch that lies in several record arrays (in stream order, whole buffers each), resolved as ONE
        batch by up to `threads` threads without concatenating them (modes_host_resolve_raw_mtv)."""
        segs = [np.ascontiguousarray(a, dtype=N.RECORD_DTYPE) for a in segments if len(a)]
        if not segs:
            return 0, (b"" if text else None)
        n, pieces = self._pieces(segs, max(1, threads))
        return n, (b"".join(C.string_at(b, ln) for b, ln in pieces) if text else None)

    def _pieces(self, segs, threads):
        """modes_host_resolve_raw_pieces: (lines, [(address, length) of the listing's pieces in stream order]) - the listing stays where
        the resolve's threads wrote it (no gathering copy); valid until this thread's next multi-threaded listing call."""
        ptrs = (C.c_void_p * len(segs))(*[a.ctypes.data for a in segs])
        lens = (C.c_uint64 * len(segs))(*[a.size for a in segs])
        out = (N.TextPiece * 80)()
        npieces, nbytes = C.c_uint32(), C.c_uint64()
        n = self._lib.modes_host_resolve_raw_pieces(self._h, ptrs, lens, len(segs), out, len(out), C.byref(npieces), C.byref(nbytes), threads)
        pieces = [(out[i].base, out[i].len) for i in range(npieces.value)]
        assert sum(ln for _, ln in pieces) == nbytes.value
        return int(n), pieces

    # ---- resolve on the ranks that demodulated (include/modes_host.h; distributed.RankResolve is the protocol) ----
    @staticmethod
    def _segments(segments):
        segs = [np.ascontiguousarray(a, dtype=N.RECORD_DTYPE) for a in segments if len(a)]
        ptrs = (C.c_void_p * max(1, len(segs)))(*[a.ctypes.data for a in segs])
        lens = (C.c_uint64 * max(1, len(segs)))(*[a.size for a in segs])
        return segs, ptrs, lens

    def whitelist(self):
        """(addr uint32[ICAO_SLOTS], seen int64[ICAO_SLOTS]): the ICAO whitelist as it stands (copies)."""
        addr = np.empty(N.ICAO_SLOTS, dtype=np.uint32)
        seen = np.empty(N.ICAO_SLOTS, dtype=np.int64)
        self._lib.modes_host_get_whitelist(self._h, addr.ctypes.data, seen.ctypes.data)
        return addr, seen

    def set_whitelist(self, addr, seen):
        addr = np.ascontiguousarray(addr, dtype=np.uint32)
        seen = np.ascontiguousarray(seen, dtype=np.int64)
        assert addr.size == N.ICAO_SLOTS and seen.size == N.ICAO_SLOTS
        self._lib.modes_host_set_whitelist(self._h, addr.ctypes.data, seen.ctypes.data)

    def whitelist_guess(self, segments, threads: int = 1) -> np.ndarray:
        """uint32[ICAO_SLOTS]: what the clean DF11/17/18 frames of these records would leave on the whitelist (ICAO_NONE: nothing)."""
        segs, ptrs, lens = self._segments(segments)
        guess = np.empty(N.ICAO_SLOTS, dtype=np.uint32)
        self._lib.modes_host_whitelist_guess(self._h, ptrs, lens, len(segs), guess.ctypes.data, max(1, threads))
        return guess

    def raw_listing_spec(self, segments, threads: int = 1):
        """raw_listing_segments from the whitelist this resolver holds NOW (a guess the caller has set), with the log a
        later confirmation needs: -> (lines, nbytes, written uint8[ICAO_SLOTS], lookups LOOKUP_DTYPE[]).  The listing
        stays in the resolver's buffer (text_view(nbytes))."""
        segs, ptrs, lens = self._segments(segments)
        total = sum(a.size for a in segs)
        buf = self._text_buffer(62 * total + 64)
        nbytes, nlook = C.c_uint64(), C.c_uint64()
        written = np.zeros(N.ICAO_SLOTS, dtype=np.uint8)
        lookups = np.empty(2 * total + 16, dtype=N.LOOKUP_DTYPE)
        n = self._lib.modes_host_resolve_raw_spec(self._h, ptrs, lens, len(segs), buf, len(buf), C.byref(nbytes), max(1, threads),
                                                  written.ctypes.data, lookups.ctypes.data, lookups.size, C.byref(nlook))
        assert nlook.value <= lookups.size and nbytes.value < len(buf)
        return int(n), int(nbytes.value), written, lookups[: nlook.value]

    def text_view(self, nbytes: int) -> np.ndarray:
        """The first nbytes of the listing buffer as uint8 (a view: valid until the next listing call of this resolver)."""
        return np.frombuffer(self._rawbuf, dtype=np.uint8, count=nbytes) if nbytes else np.zeros(0, dtype=np.uint8)

    def whitelist_check(self, lookups) -> bool:
        """Does this resolver's whitelist (at its clock) give every one of the logged answers?"""
        lookups = np.ascontiguousarray(lookups, dtype=N.LOOKUP_DTYPE)
        return bool(self._lib.modes_host_whitelist_check(self._h, lookups.ctypes.data, lookups.size))

    def stats(self) -> dict:
        st = N.HostStats()
        self._lib.modes_host_get_stats(self._h, C.byref(st))
        return st.as_dict()

    def stats_text(self) -> str:
        st = N.HostStats()
        self._lib.modes_host_get_stats(self._h, C.byref(st))
        buf = C.create_string_buffer(512)
        self._lib.modes_format_stats(C.byref(st), buf)
        return buf.value.decode()


def raw_text(msgs) -> str:
    return "".join(m.raw_line() for m in msgs)


def verbose_text(msgs) -> str:
    """The reference's default listing (no --raw): dump1090.c:1314-1450 + :1814."""
    return "".join(m.verbose for m in msgs)


def raw_net_text(msgs) -> str:
    """What the reference writes to its raw-output TCP clients (port 30002, dump1090.c:2381-2393)."""
    return "".join("*" + m.msg[: m.msgbits // 8].hex().upper() + ";\n" for m in msgs)


def _to_struct(m: Message) -> N.ModesMessage:
    mm = N.ModesMessage()
    C.memmove(mm.msg, m.msg, min(len(m.msg), 14))
    for f in ("msgbits", "msgtype", "crcok", "crc", "errorbit", "aa1", "aa2", "aa3", "phase_corrected", "iid"):
        setattr(mm, f, getattr(m, f))
    for f, v in m.fields.items():
        setattr(mm, f, v.encode("ascii", "replace") if isinstance(v, str) else v)
    return mm


class Tracker:
    """Aircraft table + CPR positions + SBS lines (libmodes_host.so: modes_tracker_*, modes_format_sbs):
    the state behind the reference's useModesMessage() while an SBS client is connected
    (dump1090.c:1806-1808, 2069-2167, 2397-2448).  No GPU needed."""

    def __init__(self, check_crc: bool = True):
        self._lib = N.host_lib()
        self.check_crc = check_crc
        self._h = self._lib.modes_tracker_create()
        if not self._h:
            raise N.ModesError(-3, "modes_tracker_create failed")

    def close(self):
        if self._h:
            self._lib.modes_tracker_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def sbs_text(self, msgs, now_ms: int = 0) -> str:
        """The lines port 30003 would carry for these messages, in order."""
        out = []
        buf = C.create_string_buffer(256)
        for m in msgs:
            mm = _to_struct(m)
            a = self._lib.modes_tracker_receive(self._h, C.byref(mm), int(self.check_crc), now_ms)
            if a and self._lib.modes_format_sbs(C.byref(mm), a, buf, 256):
                out.append(buf.value.decode("ascii"))
        return "".join(out)

    def aircraft(self) -> list[dict]:
        """The table, newest aircraft first."""
        rows = []
        for i in range(self._lib.modes_tracker_count(self._h)):
            a = self._lib.modes_tracker_get(self._h, i).contents
            rows.append({f: (getattr(a, f).decode("ascii", "replace") if isinstance(getattr(a, f), bytes) else getattr(a, f))
                         for f, _ in N.Aircraft._fields_})
        return rows

    def reference_position(self):
        lat, lon, n = C.c_double(), C.c_double(), C.c_int()
        self._lib.modes_tracker_reference(self._h, C.byref(lat), C.byref(lon), C.byref(n))
        return lat.value, lon.value, n.value

    def expire(self, now_ms: int, ttl_ms: int = 60000) -> int:
        return int(self._lib.modes_tracker_expire(self._h, now_ms, ttl_ms))


def onlyaddr_text(msgs) -> str:
    return "".join(m.addr_line() for m in msgs)


class Demodulator:
    """GPU scan + demod + host resolve for one device.  Raises if the HIP library or a GPU is missing."""

    def __init__(self, device: int = 0, fix: bool = True, aggressive: bool = False, check_crc: bool = True,
                 keep_candidates: bool = False, run_chunks: int = 0, slot_cap: int = 0, max_records: int = 0,
                 scan_variant: int = 0, overlap: int = 0, no_retry: bool = False, direct_records: int = 0,
                 demod_variant: int = 0, order_in_stream: bool = False):
        self._lib = N.gpu_lib()
        self.flags = dict(fix=fix, aggressive=aggressive, check_crc=check_crc)
        self.device = device
        cfg = N.GpuConfig(device, int(fix), int(aggressive), int(keep_candidates), run_chunks, slot_cap, max_records,
                          scan_variant, int(overlap), (N.GPU_NO_RETRY if no_retry else 0) | (N.GPU_ORDER_IN_STREAM if order_in_stream else 0),
                          direct_records, demod_variant)
        h = C.c_void_p()
        rc = self._lib.modes_gpu_create(C.byref(cfg), C.byref(h))
        if rc != N.MODES_OK:
            raise N.ModesError(rc, self._lib.modes_gpu_last_error(None).decode())
        self._h = h
        self.keep_candidates = keep_candidates
        self.last = {}
        self._out = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.modes_gpu_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != N.MODES_OK:
            raise N.ModesError(rc, self._lib.modes_gpu_last_error(self._h).decode())

    @staticmethod
    def _stream_ptr(stream):
        if stream is None:
            return None
        return C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))

    # ---- computeMagnitudeVector, dump1090.c:1454 -------------------------------------------
    def compute_magnitude_vector(self, iq, stream=None):
        """uint8 CUDA tensor of interleaved I/Q -> uint16 CUDA tensor of magnitudes."""
        import torch
        assert iq.is_cuda and iq.dtype == torch.uint8 and iq.is_contiguous()
        n = iq.numel() // 2
        out = torch.empty(n, dtype=torch.uint16, device=iq.device)
        st = stream if stream is not None else torch.cuda.current_stream(iq.device)
        self._check(self._lib.modes_gpu_compute_magnitude(self._h, iq.data_ptr(), n, out.data_ptr(),
                                                          self._stream_ptr(st)))
        return out

    def compute_power(self, iq, stream=None):
        import torch
        assert iq.is_cuda and iq.dtype == torch.uint8 and iq.is_contiguous()
        n = iq.numel() // 2
        out = torch.empty(n, dtype=torch.uint16, device=iq.device)
        st = stream if stream is not None else torch.cuda.current_stream(iq.device)
        self._check(self._lib.modes_gpu_compute_power(self._h, iq.data_ptr(), n, out.data_ptr(), self._stream_ptr(st)))
        return out

    def debug_tables(self):
        """(the device's magnitude table by saturated power, modes_mag_exact of every index computed on the device)"""
        import torch
        dev = torch.device("cuda", self.device)
        lut = torch.empty(32768, dtype=torch.uint16, device=dev)
        exact = torch.empty(32768, dtype=torch.uint16, device=dev)
        self._check(self._lib.modes_gpu_debug_tables(self._h, lut.data_ptr(), exact.data_ptr(),
                                                     self._stream_ptr(torch.cuda.current_stream(dev))))
        torch.cuda.synchronize(dev)
        return lut.cpu().numpy(), exact.cpu().numpy()

    # ---- detectModeS (stateless part), dump1090.c:1563 -----------------------------------
    def detect(self, iq, stream_byte0: int = 0, first_block: int = 0, nblocks: int | None = None, stream=None):
        """Launch scan + demod for buffers [first_block, first_block+nblocks) of a stream whose bytes
        [stream_byte0, stream_byte0+len(iq)) sit in the CUDA uint8 tensor `iq`.  Asynchronous."""
        import torch
        assert iq.is_cuda and iq.dtype == torch.uint8 and iq.is_contiguous()
        if nblocks is None:
            nblocks = block_count(stream_byte0 + iq.numel()) - first_block
        span = N.Span(iq.data_ptr(), iq.numel(), stream_byte0, first_block, nblocks)
        st = stream if stream is not None else torch.cuda.current_stream(iq.device)
        self._keepalive = iq
        self._check(self._lib.modes_gpu_detect(self._h, C.byref(span), self._stream_ptr(st)))

    def fetch(self, copy: bool = True):
        """Wait for detect(); -> (records ndarray, candidates ndarray | None, info dict).  copy=False: the arrays are
        views of the library's own (pinned) buffers, valid until the next detect on this context."""
        res = N.GpuResult()
        self._check(self._lib.modes_gpu_fetch(self._h, C.byref(res)))
        return self._unpack(res, copy)

    def set_output(self, records=None, count=None):
        """Caller-owned device output (modes_gpu_set_output): `records` a CUDA uint8 tensor of capacity * 64 bytes
        that receives the ordered list, `count` a CUDA int64 tensor of one element that receives the number of
        records - both written by the kernels of detect(), in stream order.  None, None: the context's own list."""
        if records is None:
            self._check(self._lib.modes_gpu_set_output(self._h, None, 0, None))
            self._out = None
            return
        import torch
        assert records.is_cuda and records.dtype == torch.uint8 and records.is_contiguous() and records.numel() % 64 == 0
        assert count is None or (count.is_cuda and count.dtype == torch.int64 and count.numel() == 1)
        self._check(self._lib.modes_gpu_set_output(self._h, records.data_ptr(), records.numel() // 64,
                                                   count.data_ptr() if count is not None else None))
        self._out = (records, count)

    def set_timing(self, on: bool):
        """Kernel times in the result of every detect that follows (default on; ~9 us of idle GPU per kernel boundary)."""
        self._check(self._lib.modes_gpu_set_timing(self._h, int(bool(on))))

    def host_profile(self, reset: bool = False) -> dict:
        """Host seconds inside modes_gpu_detect so far, by section (modes_gpu_host_profile)."""
        out = (C.c_double * 8)()
        self._check(self._lib.modes_gpu_host_profile(self._h, out, int(reset)))
        names = ("set_device", "setup", "launch_scan", "launch_demod", "launch_finalize", "rest")
        d = {n: float(out[i]) for i, n in enumerate(names)}
        d["calls"] = int(out[6])
        return d

    def stream_ceiling(self, iq, launches: int = 96, time_every: int = 4, stream=None):
        """(average, shortest) ms of a read-only pass over the CUDA uint8 tensor `iq` with the scan kernel's loads and
        nothing else (modes_gpu_stream_ceiling): the measured HBM streaming ceiling of this box."""
        import torch
        assert iq.is_cuda and iq.dtype == torch.uint8 and iq.is_contiguous()
        st = stream if stream is not None else torch.cuda.current_stream(iq.device)
        avg, best = C.c_float(), C.c_float()
        n = iq.numel() // 1024 * 1024
        self._check(self._lib.modes_gpu_stream_ceiling(self._h, iq.data_ptr(), n, launches, time_every, C.byref(avg), C.byref(best),
                                                       self._stream_ptr(st)))
        return float(avg.value), float(best.value), n

    def stream_wait(self, stream):
        """Make `stream` wait for the results of the detect in flight (modes_gpu_stream_wait)."""
        self._check(self._lib.modes_gpu_stream_wait(self._h, self._stream_ptr(stream)))

    def fetch_device(self):
        """Wait for detect() without copying the records: -> (number of records, info dict).  The ordered list
        is in the tensor given to set_output() (its first n * 64 bytes)."""
        res = N.GpuResult()
        self._check(self._lib.modes_gpu_fetch_device(self._h, C.byref(res)))
        self.last = dict(n_records=int(res.n_records), n_forwarded=int(res.n_forwarded), n_preambles=int(res.n_preambles),
                         scan_ms=float(res.scan_ms), demod_ms=float(res.demod_ms), order_ms=float(res.order_ms))
        return int(res.n_records), dict(self.last)

    def _unpack(self, res, copy: bool = True):
        n = int(res.n_records)
        if n:
            raw = (C.c_uint8 * (n * 64)).from_address(C.cast(res.records, C.c_void_p).value)
            recs = np.frombuffer(raw, dtype=N.RECORD_DTYPE)
            if copy:
                recs = recs.copy()                                        # the library reuses its buffer on the next call
        else:
            recs = np.zeros(0, dtype=N.RECORD_DTYPE)
        cands = None
        if self.keep_candidates:
            nc = int(res.n_candidates)
            cands = (np.frombuffer((C.c_uint8 * (nc * 8)).from_address(C.cast(res.candidates, C.c_void_p).value),
                                   dtype=np.uint64).copy() if nc else np.zeros(0, dtype=np.uint64))
        self.last = dict(n_records=n, n_forwarded=int(res.n_forwarded), n_preambles=int(res.n_preambles),
                         scan_ms=float(res.scan_ms), demod_ms=float(res.demod_ms), order_ms=float(res.order_ms))
        return recs, cands, dict(self.last)

    # ---- host buffers: what the C host does (modes_gpu_submit_host + fetch) -------------------
    def host_alloc(self, nbytes: int):
        """Pinned host memory as a numpy uint8 array (modes_gpu_host_alloc); free with host_free(arr)."""
        p = C.c_void_p()
        self._check(self._lib.modes_gpu_host_alloc(self._h, nbytes, C.byref(p)))
        arr = np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=np.uint8)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = self._pinned.pop(arr.ctypes.data)
        self._lib.modes_gpu_host_free(self._h, p)

    def submit_host(self, data: np.ndarray, stream_byte0: int = 0, first_block: int = 0, nblocks: int | None = None):
        """Asynchronous host -> HBM copy + kernels (modes_gpu_submit_host); fetch() later."""
        assert data.dtype == np.uint8 and data.flags.c_contiguous
        if nblocks is None:
            nblocks = block_count(stream_byte0 + data.size) - first_block
        self._check(self._lib.modes_gpu_submit_host(self._h, data.ctypes.data, data.size, stream_byte0, first_block, nblocks))

    def records_from_host(self, data: np.ndarray, stream_byte0: int = 0, first_block: int = 0,
                          nblocks: int | None = None):
        """modes_gpu_demod_host: stage a host buffer, detect, fetch (what the C host calls)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if nblocks is None:
            nblocks = block_count(stream_byte0 + data.size) - first_block
        res = N.GpuResult()
        self._check(self._lib.modes_gpu_demod_host(self._h, data.ctypes.data, data.size, stream_byte0, first_block,
                                                   nblocks, C.byref(res)))
        return self._unpack(res)

    # ---- the whole path -----------------------------------------------------------------------
    def demodulate(self, data, resolver: HostResolver | None = None, batch_blocks: int = 8192):
        """Whole stream (numpy uint8 on the host, or a CUDA uint8 tensor) -> list[Message].
        The stream is walked `batch_blocks` buffers (2 GiB by default) per GPU call, like the
        reference's main loop walks it one buffer at a time (dump1090.c:2969-2990); the resolver
        carries the ICAO whitelist from batch to batch."""
        own = resolver is None
        if own:
            resolver = HostResolver(**self.flags)
        try:
            n = data.size if isinstance(data, np.ndarray) else data.numel()
            total = block_count(n)
            msgs, tot = [], dict(n_records=0, n_forwarded=0, n_preambles=0, scan_ms=0.0, demod_ms=0.0, order_ms=0.0)
            for b0 in range(0, total, batch_blocks):
                nb = min(batch_blocks, total - b0)
                lo, hi = shard_byte_range(b0, nb, n)
                if isinstance(data, np.ndarray):
                    recs, cands, info = self.records_from_host(data[lo:hi], stream_byte0=lo, first_block=b0, nblocks=nb)
                else:
                    self.detect(data[lo:hi], stream_byte0=lo, first_block=b0, nblocks=nb)
                    recs, cands, info = self.fetch()
                msgs += resolver.resolve(recs, cands)
                for k in tot:
                    tot[k] += info[k]
            self.last = dict(tot)
            self.last["stats"] = resolver.stats()
            self.last["stats_text"] = resolver.stats_text()
            return msgs
        finally:
            if own:
                resolver.close()

    # ---- synthetic input ----------------------------------------------------------------------
    def synth_noise(self, out, first_byte: int, seed: int, sigma_q16: int = 941, stream=None):
        """Fill a CUDA uint8 tensor with tests/synth.py:noise_bytes(seed, first_byte, len(out))."""
        import torch
        assert out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous()
        st = stream if stream is not None else torch.cuda.current_stream(out.device)
        self._check(self._lib.modes_gpu_synth_noise(self._h, out.data_ptr(), first_byte, out.numel(), seed, sigma_q16,
                                                    self._stream_ptr(st)))
        return out

    def fill(self, out, value: int = 127, stream=None):
        import torch
        st = stream if stream is not None else torch.cuda.current_stream(out.device)
        self._check(self._lib.modes_gpu_fill(self._h, out.data_ptr(), out.numel(), value, self._stream_ptr(st)))
        return out
