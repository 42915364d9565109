"""Per-wavefront timeline of record_kernel on the low-SNR workload (-DMODES_TRACE build; tools/trace_demod.py has the build line):
    python tools/trace_record.py dump1090_amd/libmodes_gfx950_trace.so [lowsnr|frames]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from dump1090_amd import _native as N
N.GPU_LIB = os.path.abspath(sys.argv[1])
from dump1090_amd import Demodulator
import bench
wl = sys.argv[2] if len(sys.argv) > 2 else "lowsnr"
kw = bench.LOWSNR if wl == "lowsnr" else {}
iq, _ = bench.build_frames_shard(torch, torch.device("cuda", 0), 4096, 0, 1 << 30, seed=5 if wl == "lowsnr" else 3, **kw)
d = Demodulator(fix=True, aggressive=(wl == "lowsnr"), demod_variant=2)
for _ in range(4):
    d.detect(iq); recs, _, info = d.fetch()
t = np.zeros(8192 * 8, dtype=np.uint64)
assert N.gpu_lib().modes_gpu_trace(t.ctypes.data_as(C.c_void_p)) == 0
t = t.reshape(-1, 8).astype(np.int64)[4096:]
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
us = lambda v: v / 100.0
print("record_kernel %.4f ms by events, %d records, %d wavefronts traced" % (info["order_ms"], recs.size, len(t)))
for name, v in (("start", us(t[:, 0] - t0)), ("end", us(t[:, 2] - t0)), ("life", us(t[:, 2] - t[:, 0])), ("counts", us(t[:, 1])), ("table", us(t[:, 4])),
                ("records", us(t[:, 5])), ("records of the wavefront", t[:, 3]), ("records of its batches", t[:, 6])):
    print("%-26s min %8.2f p10 %8.2f p50 %8.2f p90 %8.2f max %8.2f" % (name, v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
has = t[:, 3] > 0
print("us per record (wavefronts with records): p50 %.2f  mean %.2f" % (np.median(us(t[has, 5]) / t[has, 3]), (us(t[has, 5]) / t[has, 3]).mean()))
for k in range(1, int(t[:, 3].max()) + 1):
    sel = t[:, 3] == k
    if sel.any():
        print("   wavefronts with %d record(s): %4d, records phase p50 %.2f us, life p50 %.2f us" % (k, sel.sum(), np.median(us(t[sel, 5])), np.median(us(t[sel, 2] - t[sel, 0]))))
