#!/bin/bash
# Runs the CPU tests of the host library and of the header-only device arithmetic against UBSan builds
# (-fno-sanitize-recover: any finding aborts the test process).  The host library is built with gcc, the
# shims around modes_core.h / modes_order.h need clang (vector extensions) and its shared UBSan runtime.
# The normal builds are restored whatever happens.
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
CLANG=/opt/rocm/lib/llvm/bin/clang++
RT=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
SAN="-fsanitize=undefined -fno-sanitize-recover=undefined"
cp dump1090_amd/libmodes_host.so /tmp/libmodes_host_orig.so
restore() {
    cp /tmp/libmodes_host_orig.so "$R/dump1090_amd/libmodes_host.so"
    (cd "$R" && python -c "import sys; sys.path.insert(0, 'tests'); from native.build import build, build_order; build(force=True); build_order(force=True)")
}
trap restore EXIT
set -e
(cd dump1090_amd/csrc && g++ -O1 -g -std=c++17 -fPIC $SAN -I../../include -shared -o ../libmodes_host.so modes_host.cpp modes_track.cpp -lm)
(cd tests/native && $CLANG -O1 -g -std=c++17 -fPIC -shared $SAN -shared-libsan -Wl,-rpath,$RT -I ../../dump1090_amd/csrc -o libcore_shim.so core_shim.cpp \
  && $CLANG -O1 -g -std=c++17 -fPIC -shared -pthread $SAN -shared-libsan -Wl,-rpath,$RT -I ../../include -o liborder_shim.so order_shim.cpp)
python -m pytest tests/test_host.py tests/test_track.py tests/test_core.py tests/test_order.py -x -q
