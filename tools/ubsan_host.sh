#!/bin/bash
# Runs the CPU tests of the host library and of the header-only device arithmetic with UBSan builds
# (gcc's runtime; -fno-sanitize-recover: any finding aborts the test process).  Restores the normal builds.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
cp dump1090_amd/libmodes_host.so /tmp/libmodes_host_orig.so
(cd dump1090_amd/csrc && g++ -O1 -g -std=c++17 -fPIC -fsanitize=undefined -fno-sanitize-recover=undefined -I../../include -shared \
    -o ../libmodes_host.so modes_host.cpp modes_track.cpp -lm)
(cd tests/native && g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=undefined -fno-sanitize-recover=undefined -I ../../dump1090_amd/csrc \
    -o libcore_shim.so core_shim.cpp && g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=undefined -fno-sanitize-recover=undefined \
    -I ../../include -o liborder_shim.so order_shim.cpp)
rc=0
python -m pytest tests/test_host.py tests/test_track.py tests/test_core.py tests/test_order.py -x -q || rc=$?
cp /tmp/libmodes_host_orig.so dump1090_amd/libmodes_host.so
python -c "import sys; sys.path.insert(0, 'tests'); from native.build import build, build_order; build(force=True); build_order(force=True)"
exit $rc
