"""ctypes bindings of the two product libraries (include/modes_gfx950.h, include/modes_host.h).

No fallback: if libmodes_gfx950.so is missing or no HIP device is present, creating a
GPU context raises ModesError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
GPU_LIB = os.environ.get("MODES_GPU_LIB") or os.path.join(PKG_DIR, "libmodes_gfx950.so")      # (MODES_GPU_LIB: another build of the library - tools/ab_scan.py, experiments)
HOST_LIB = os.environ.get("MODES_HOST_LIB") or os.path.join(PKG_DIR, "libmodes_host.so")       # (MODES_HOST_LIB: another build, for A/B timings)
GATHER_LIB = os.path.join(PKG_DIR, "libmodes_gather.so")      # the C hosts' record gather over RCCL (include/modes_gather.h)

DATA_LEN = 262144
CARRY_BYTES = 476
BLOCK_STRIDE = 131072
BLOCK_POSITIONS = 131070
CARRY_SAMPLES = 238

MODES_OK = 0
ERRORS = {-1: "MODES_ERR_ARG", -2: "MODES_ERR_HIP", -3: "MODES_ERR_NOMEM", -4: "MODES_ERR_OVERFLOW",
          -5: "MODES_ERR_STATE"}


class ModesError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s: %s" % (ERRORS.get(code, code), text))
        self.code = code


class GpuConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("fix_errors", C.c_int32), ("aggressive", C.c_int32),
                ("keep_candidates", C.c_int32), ("run_chunks", C.c_uint32), ("slot_cap", C.c_uint32),
                ("max_records", C.c_uint32), ("scan_variant", C.c_uint32), ("overlap", C.c_uint32),
                ("flags", C.c_uint32), ("direct_records", C.c_uint32), ("demod_variant", C.c_uint32)]


GPU_NO_RETRY = 1
GPU_ORDER_IN_STREAM = 2


class Attempt(C.Structure):
    _fields_ = [("msg", C.c_uint8 * 14), ("errors", C.c_uint8), ("gate_ok", C.c_uint8), ("nfix", C.c_uint8),
                ("fixpos", C.c_uint8 * 2), ("cls", C.c_uint8), ("slot", C.c_uint16), ("pad", C.c_uint8 * 2),
                ("syndrome", C.c_uint32)]


class Record(C.Structure):
    _fields_ = [("block", C.c_uint32), ("j", C.c_uint32), ("att", Attempt * 2)]


ATTEMPT_DTYPE = np.dtype([("msg", np.uint8, 14), ("errors", np.uint8), ("gate_ok", np.uint8), ("nfix", np.uint8),
                          ("fixpos", np.uint8, 2), ("cls", np.uint8), ("slot", np.uint16), ("pad", np.uint8, 2),
                          ("syndrome", np.uint32)])
RECORD_DTYPE = np.dtype([("block", np.uint32), ("j", np.uint32), ("att", ATTEMPT_DTYPE, 2)])
assert C.sizeof(Record) == 64 and RECORD_DTYPE.itemsize == 64


class Span(C.Structure):
    _fields_ = [("iq", C.c_void_p), ("nbytes", C.c_uint64), ("stream_byte0", C.c_uint64),
                ("first_block", C.c_uint64), ("nblocks", C.c_uint64)]


class GpuResult(C.Structure):
    _fields_ = [("records", C.POINTER(Record)), ("n_records", C.c_uint64),
                ("candidates", C.POINTER(C.c_uint64)), ("n_candidates", C.c_uint64),
                ("n_forwarded", C.c_uint64), ("n_preambles", C.c_uint64),
                ("scan_ms", C.c_float), ("demod_ms", C.c_float), ("order_ms", C.c_float), ("reserved", C.c_float)]


class HostConfig(C.Structure):
    _fields_ = [("fix_errors", C.c_int32), ("aggressive", C.c_int32), ("check_crc", C.c_int32),
                ("reserved", C.c_int32)]


STAT_NAMES = ("valid_preamble", "out_of_phase", "demodulated", "goodcrc", "badcrc", "fixed", "single_bit_fix",
              "two_bits_fix")


class HostStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in STAT_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in STAT_NAMES}


class ModesMessage(C.Structure):
    """struct modesMessage - dump1090.c:211-260, field for field."""
    _fields_ = [("msg", C.c_ubyte * 14), ("msgbits", C.c_int), ("msgtype", C.c_int), ("crcok", C.c_int),
                ("crc", C.c_uint32), ("errorbit", C.c_int), ("aa1", C.c_int), ("aa2", C.c_int), ("aa3", C.c_int),
                ("phase_corrected", C.c_int), ("ca", C.c_int), ("iid", C.c_int), ("metype", C.c_int),
                ("mesub", C.c_int), ("heading_is_valid", C.c_int), ("heading", C.c_int), ("aircraft_type", C.c_int),
                ("fflag", C.c_int), ("tflag", C.c_int), ("raw_latitude", C.c_int), ("raw_longitude", C.c_int),
                ("flight", C.c_char * 9), ("ew_dir", C.c_int), ("ew_velocity", C.c_int), ("ns_dir", C.c_int),
                ("ns_velocity", C.c_int), ("vert_rate_source", C.c_int), ("vert_rate_sign", C.c_int),
                ("vert_rate", C.c_int), ("velocity", C.c_int), ("movement", C.c_int), ("movement_valid", C.c_int),
                ("ground_track", C.c_int), ("ground_track_valid", C.c_int), ("fs", C.c_int), ("dr", C.c_int),
                ("um", C.c_int), ("identity", C.c_int), ("altitude", C.c_int), ("unit", C.c_int)]


class Emitted(C.Structure):
    _fields_ = [("mm", ModesMessage), ("block", C.c_uint32), ("j", C.c_uint32)]


class TextPiece(C.Structure):                      # modes_text_piece
    _fields_ = [("base", C.c_void_p), ("len", C.c_uint64)]


class Aircraft(C.Structure):
    """modes_aircraft (include/modes_host.h), the reference's struct aircraft without the list link."""
    _fields_ = [("addr", C.c_uint32), ("hexaddr", C.c_char * 7), ("flight", C.c_char * 9), ("altitude", C.c_int),
                ("speed", C.c_int), ("track", C.c_int), ("odd_cprlat", C.c_int), ("odd_cprlon", C.c_int),
                ("even_cprlat", C.c_int), ("even_cprlon", C.c_int), ("lat", C.c_double), ("lon", C.c_double),
                ("odd_cprtime", C.c_int64), ("even_cprtime", C.c_int64), ("seen_ms", C.c_int64), ("messages", C.c_long)]


ICAO_SLOTS = 1024                  # MODES_ICAO_SLOTS
ICAO_NONE = 0xFFFFFFFF             # MODES_ICAO_NONE
LOOKUP_DTYPE = np.dtype([("addr", np.uint32), ("known", np.uint32)])      # modes_icao_lookup

SINK_FN = C.CFUNCTYPE(None, C.POINTER(ModesMessage), C.c_uint32, C.c_uint32, C.c_void_p)

# every symbol include/*.h declares (tests/test_abi.py checks the libraries export them)
GPU_SYMBOLS = ("modes_gpu_create", "modes_gpu_destroy", "modes_gpu_last_error", "modes_gpu_compute_magnitude",
               "modes_gpu_detect", "modes_gpu_fetch", "modes_gpu_fetch_device", "modes_gpu_set_output", "modes_gpu_stream_wait", "modes_gpu_set_timing", "modes_gpu_demod_host", "modes_gpu_submit_host",
               "modes_gpu_host_alloc", "modes_gpu_host_free", "modes_gpu_compute_power", "modes_gpu_debug_tables",
               "modes_gpu_synth_noise", "modes_gpu_fill", "modes_gpu_abi_version", "modes_gpu_host_profile",
               "modes_gpu_stream_ceiling")
HOST_SYMBOLS = ("modes_host_create", "modes_host_destroy", "modes_host_set_time", "modes_host_resolve", "modes_host_resolve_to_array",
                "modes_host_resolve_raw", "modes_host_resolve_raw_mt", "modes_host_resolve_raw_mtv", "modes_host_wants",
                "modes_host_get_stats", "modes_host_decode", "modes_host_decode_frame", "modes_format_raw", "modes_format_raw_net",
                "modes_format_onlyaddr", "modes_format_verbose", "modes_format_stats", "modes_checksum", "modes_compute_crc", "modes_message_len_by_type",
                "modes_block_count",
                "modes_host_get_whitelist", "modes_host_set_whitelist", "modes_host_whitelist_guess", "modes_host_resolve_raw_spec",
                "modes_host_whitelist_check", "modes_host_cpu_budget", "modes_host_classify", "modes_host_resolve_spec", "modes_host_set_stats", "modes_host_resolve_raw_pieces",
                "modes_tracker_create", "modes_tracker_destroy", "modes_tracker_receive", "modes_tracker_expire",
                "modes_tracker_count", "modes_tracker_get", "modes_tracker_reference", "modes_tracker_json", "modes_format_sbs")

GATHER_SYMBOLS = ("modes_gather_unique_id", "modes_gather_create", "modes_gather_destroy", "modes_gather_last_error",
                  "modes_gather_output", "modes_gather_set_empty", "modes_gather_counts", "modes_gather_records", "modes_gather_wait",
                  "modes_gather_get_stats", "modes_gather_abi_version", "modes_gather_set_candidates", "modes_gather_candidates")


class GatherConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32), ("cap_records", C.c_uint32),
                ("nslots", C.c_uint32), ("cap_candidates", C.c_uint32)]


class GatherStats(C.Structure):
    _fields_ = [("nranks", C.c_int32), ("rank", C.c_int32), ("rccl_version", C.c_int32), ("reserved", C.c_uint32),
                ("calls", C.c_uint64), ("p2p_ops", C.c_uint64), ("bytes_received", C.c_uint64), ("bytes_sent", C.c_uint64),
                ("gather_ms", C.c_double)]


_gpu = None
_host = None
_gather = None


def gather_lib():
    """libmodes_gather.so - what a C host binds for the N-GPU gather (Python hosts use dump1090_amd/distributed.py); loaded
    here for the tests.  Needs librccl (torch's copy is picked up when torch is already imported)."""
    global _gather
    if _gather is None:
        if not os.path.exists(GATHER_LIB):
            raise ModesError(-2, "%s is not built" % GATHER_LIB)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(GATHER_LIB)
        L.modes_gather_unique_id.argtypes = [C.c_void_p]
        L.modes_gather_create.argtypes = [C.POINTER(GatherConfig), C.c_void_p, C.POINTER(C.c_void_p)]
        L.modes_gather_destroy.argtypes = [C.c_void_p]
        L.modes_gather_destroy.restype = None
        L.modes_gather_last_error.argtypes = [C.c_void_p]
        L.modes_gather_last_error.restype = C.c_char_p
        L.modes_gather_output.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
        for f in (L.modes_gather_set_empty, L.modes_gather_counts, L.modes_gather_records):
            f.argtypes = [C.c_void_p, C.c_uint32]
        L.modes_gather_wait.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint64))]
        L.modes_gather_get_stats.argtypes = [C.c_void_p, C.POINTER(GatherStats)]
        _gather = L
    return _gather


def gpu_lib():
    """libmodes_gfx950.so (loading it needs the HIP runtime, not a GPU)."""
    global _gpu
    if _gpu is None:
        if not os.path.exists(GPU_LIB):
            raise ModesError(-2, "%s is not built (python -c 'import __graft_entry__ as g; g.build()')" % GPU_LIB)
        # torch brings its own copy of the HIP runtime: load it first, so that this library binds to the runtime torch's
        # streams and tensors live in (loaded the other way round the process ends up with two runtimes, and the second
        # one sees no device)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(GPU_LIB)
        L.modes_gpu_create.argtypes = [C.POINTER(GpuConfig), C.POINTER(C.c_void_p)]
        L.modes_gpu_destroy.argtypes = [C.c_void_p]
        L.modes_gpu_destroy.restype = None
        L.modes_gpu_last_error.argtypes = [C.c_void_p]
        L.modes_gpu_last_error.restype = C.c_char_p
        L.modes_gpu_compute_magnitude.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.modes_gpu_compute_power.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.modes_gpu_debug_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.modes_gpu_detect.argtypes = [C.c_void_p, C.POINTER(Span), C.c_void_p]
        L.modes_gpu_fetch.argtypes = [C.c_void_p, C.POINTER(GpuResult)]
        L.modes_gpu_fetch_device.argtypes = [C.c_void_p, C.POINTER(GpuResult)]
        L.modes_gpu_stream_wait.argtypes = [C.c_void_p, C.c_void_p]
        L.modes_gpu_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.modes_gpu_set_output.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.modes_gpu_submit_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.modes_gpu_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.modes_gpu_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.modes_gpu_host_free.restype = None
        L.modes_gpu_demod_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                           C.POINTER(GpuResult)]
        L.modes_gpu_synth_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
                                            C.c_void_p]
        L.modes_gpu_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint8, C.c_void_p]
        L.modes_gpu_host_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        L.modes_gpu_stream_ceiling.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_float),
                                               C.POINTER(C.c_float), C.c_void_p]
        _gpu = L
    return _gpu


def classify_records(records: np.ndarray, maxfix: int = 1, fix=None, aggressive=None) -> np.ndarray:
    """A copy of `records` with the class byte and whitelist slot of every attempt filled in (modes_host_classify): what the
    kernels write for a context of that configuration - for records of another producer (the oracle in the CPU tests)."""
    fix = (maxfix > 0) if fix is None else fix
    aggressive = (maxfix > 1) if aggressive is None else aggressive
    out = np.ascontiguousarray(records, dtype=RECORD_DTYPE).copy()
    cfg = HostConfig(int(fix), int(aggressive), 1, 0)
    host_lib().modes_host_classify(C.byref(cfg), out.ctypes.data, out.size)
    return out


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise ModesError(-2, "%s is not built" % HOST_LIB)
        L = C.CDLL(HOST_LIB)
        L.modes_host_create.argtypes = [C.POINTER(HostConfig)]
        L.modes_host_create.restype = C.c_void_p
        L.modes_host_destroy.argtypes = [C.c_void_p]
        L.modes_host_destroy.restype = None
        L.modes_host_set_time.argtypes = [C.c_void_p, C.c_int64]
        L.modes_host_set_time.restype = None
        L.modes_host_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, SINK_FN,
                                         C.c_void_p]
        L.modes_host_resolve.restype = C.c_uint64
        L.modes_host_resolve_to_array.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                  C.POINTER(Emitted), C.c_uint64]
        L.modes_host_resolve_to_array.restype = C.c_uint64
        L.modes_host_resolve_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                             C.POINTER(C.c_uint64)]
        L.modes_host_resolve_raw.restype = C.c_uint64
        L.modes_host_resolve_raw_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]
        L.modes_host_resolve_raw_mt.restype = C.c_uint64
        L.modes_host_resolve_raw_mtv.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p, C.c_uint64,
                                                 C.POINTER(C.c_uint64), C.c_int]
        L.modes_host_resolve_raw_mtv.restype = C.c_uint64
        # resolve on the ranks that demodulated (include/modes_host.h)
        L.modes_host_get_whitelist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.modes_host_get_whitelist.restype = None
        L.modes_host_set_whitelist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.modes_host_set_whitelist.restype = None
        L.modes_host_whitelist_guess.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p, C.c_int]
        L.modes_host_whitelist_guess.restype = None
        L.modes_host_resolve_raw_spec.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p, C.c_uint64,
                                                  C.POINTER(C.c_uint64), C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.modes_host_resolve_raw_spec.restype = C.c_uint64
        L.modes_host_resolve_raw_pieces.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(TextPiece), C.c_uint32,
                                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_int]
        L.modes_host_resolve_raw_pieces.restype = C.c_uint64
        L.modes_host_resolve_spec.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, SINK_FN, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_uint64, C.POINTER(C.c_uint64)]
        L.modes_host_resolve_spec.restype = C.c_uint64
        L.modes_host_set_stats.argtypes = [C.c_void_p, C.POINTER(HostStats)]
        L.modes_host_set_stats.restype = None
        L.modes_host_whitelist_check.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.modes_host_whitelist_check.restype = C.c_int
        L.modes_host_cpu_budget.argtypes = []
        L.modes_host_cpu_budget.restype = C.c_int
        L.modes_host_classify.argtypes = [C.POINTER(HostConfig), C.c_void_p, C.c_uint64]
        L.modes_host_classify.restype = None
        L.modes_host_wants.argtypes = [C.c_void_p, C.POINTER(ModesMessage)]
        L.modes_host_get_stats.argtypes = [C.c_void_p, C.POINTER(HostStats)]
        L.modes_host_get_stats.restype = None
        L.modes_host_decode.argtypes = [C.c_void_p, C.POINTER(Attempt), C.POINTER(ModesMessage)]
        L.modes_host_decode.restype = None
        L.modes_host_decode_frame.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(ModesMessage)]
        L.modes_host_decode_frame.restype = None
        L.modes_format_raw.argtypes = [C.POINTER(ModesMessage), C.c_char_p]
        L.modes_format_raw_net.argtypes = [C.POINTER(ModesMessage), C.c_char_p]
        L.modes_format_onlyaddr.argtypes = [C.POINTER(ModesMessage), C.c_char_p]
        L.modes_format_verbose.argtypes = [C.POINTER(ModesMessage), C.c_int, C.c_char_p, C.c_size_t]
        L.modes_format_stats.argtypes = [C.POINTER(HostStats), C.c_char_p]
        L.modes_checksum.argtypes = [C.c_void_p, C.c_int]
        L.modes_checksum.restype = C.c_uint32
        L.modes_compute_crc.argtypes = [C.c_void_p, C.c_int]
        L.modes_compute_crc.restype = C.c_uint32
        L.modes_block_count.argtypes = [C.c_uint64]
        L.modes_block_count.restype = C.c_uint64
        L.modes_tracker_create.restype = C.c_void_p
        L.modes_tracker_destroy.argtypes = [C.c_void_p]
        L.modes_tracker_destroy.restype = None
        L.modes_tracker_receive.argtypes = [C.c_void_p, C.POINTER(ModesMessage), C.c_int, C.c_int64]
        L.modes_tracker_receive.restype = C.POINTER(Aircraft)
        L.modes_tracker_expire.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.modes_tracker_expire.restype = C.c_uint64
        L.modes_tracker_count.argtypes = [C.c_void_p]
        L.modes_tracker_count.restype = C.c_uint64
        L.modes_tracker_get.argtypes = [C.c_void_p, C.c_uint64]
        L.modes_tracker_get.restype = C.POINTER(Aircraft)
        L.modes_tracker_reference.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.modes_tracker_reference.restype = None
        L.modes_tracker_json.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
        L.modes_tracker_json.restype = C.c_size_t
        L.modes_format_sbs.argtypes = [C.POINTER(ModesMessage), C.POINTER(Aircraft), C.c_char_p, C.c_size_t]
        _host = L
    return _host
