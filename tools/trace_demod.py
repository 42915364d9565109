"""Per-wavefront timeline of demod_kernel.  Needs a -DMODES_TRACE build of the library:
    (cd dump1090_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMODES_TRACE -I../../include -shared \
        -o ../libmodes_gfx950_trace.so modes_gfx950.hip -lpthread)
    python tools/trace_demod.py dump1090_amd/libmodes_gfx950_trace.so [MiB] [demod_variant]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dump1090_amd import _native as N
N.GPU_LIB = sys.argv[1]
from dump1090_amd import Demodulator
d = Demodulator(fix=False, demod_variant=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
if os.environ.get('MODES_TRACE_NOTIMING'):
    d.set_timing(False)
iq = torch.empty((int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 20, dtype=torch.uint8, device="cuda:0")
d.synth_noise(iq, 0, seed=20260922, sigma_q16=941)
for _ in range(3):
    d.detect(iq); d.fetch()
t = np.zeros(8192 * 8, dtype=np.uint64)
lib = None
lib = N.gpu_lib()
assert lib.modes_gpu_trace(t.ctypes.data_as(C.c_void_p)) == 0
t = t.reshape(-1, 8).astype(np.int64)
t = t[t[:, 0] != 0]                      # wavefronts that ran
print("wavefronts", len(t))
t0 = t[:, 0].min()
start, lut, end, cand = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0, t[:, 3]
print("demod_ms", d.last["demod_ms"])
for name, v in (("start", start), ("lut done", lut), ("end", end), ("life", end - start), ("work", end - lut), ("cands", cand),
                ("batch setup", t[:, 4] / 100.0), ("stage 1", t[:, 5] / 100.0), ("stage 2a", t[:, 6] / 100.0), ("stage 2b+3", t[:, 7] / 100.0)):
    print("%-11s min %8.2f p10 %8.2f p50 %8.2f p90 %8.2f max %8.2f  (us; cands: count)" % (name, v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))

# who is slow?  (8 wavefronts per workgroup share their workgroup's numbers; one line per workgroup)
wpw = 16 if (len(sys.argv) > 3 and sys.argv[3] == "2") else 8            # select_kernel: 16 wavefronts per workgroup       # wavefronts per workgroup: select_kernel has 16
wg = t[::wpw] if len(t) % wpw == 0 else t
life = (wg[:, 2] - wg[:, 0]) / 100.0
order = np.argsort(-life)
print("corr(life, cands) = %.2f" % np.corrcoef(life, wg[:, 3])[0, 1])
print("slowest workgroups: index, life, cands, setup, stage1, stage2a, stage2b+3 (us)")
for i in order[:12]:
    print("  wg %4d  life %6.2f  cands %4d  %5.2f %6.2f %6.2f %6.2f" % (i, life[i], wg[i, 3], wg[i, 4] / 100.0, wg[i, 5] / 100.0, wg[i, 6] / 100.0, wg[i, 7] / 100.0))
print("fastest:")
for i in order[-4:]:
    print("  wg %4d  life %6.2f  cands %4d  %5.2f %6.2f %6.2f %6.2f" % (i, life[i], wg[i, 3], wg[i, 4] / 100.0, wg[i, 5] / 100.0, wg[i, 6] / 100.0, wg[i, 7] / 100.0))
for m in (8, 32, 256):
    g = [life[np.arange(len(life)) % m == r].mean() for r in range(min(m, 8))]
    print("mean life by wg index mod %d (first 8 residues): %s" % (m, " ".join("%.1f" % x for x in g)))
