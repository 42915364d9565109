"""bench.py against another build of the GPU library (A/B experiments):  python tools/bench_lib.py <lib.so> [bench args]"""
import sys, os
sys.path.insert(0, os.getcwd())
from dump1090_amd import _native as N
N.GPU_LIB = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-live-traffic"] + sys.argv[2:]
import runpy
runpy.run_path("bench.py", run_name="__main__")
