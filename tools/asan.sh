#!/bin/bash
# usage: tools/asan.sh [pytest args]  -- the host simulation of libwrhip (the SAME sources: host state tracker + every kernel body, compiled
# by g++) built with AddressSanitizer + UndefinedBehaviorSanitizer and run under the hostsim parity suite.  GPU sanitizers are not available
# on the pool; an out-of-bounds read in a kernel body is silent on the device, here it stops the test.  (float-cast-overflow and
# float-divide-by-zero are off: swgl's own arithmetic relies on both, e.g. 1 / w of a clipped vertex.)
cd "$(dirname "$0")/.." || exit 1
so=webrender_amd/csrc/build/libwrhip_hostsim_asan.so
if [ ! -e $so ] || [ webrender_amd/csrc/wrhip.hip -nt $so ] || [ webrender_amd/csrc/wrhip_k_setup.h -nt $so ] || [ webrender_amd/csrc/wrhip_k_raster.h -nt $so ]; then
  mkdir -p webrender_amd/csrc/build
  (cd webrender_amd/csrc && g++ -x c++ -DWRHIP_HOSTSIM -O1 -g -fsanitize=address,undefined -fno-sanitize=float-cast-overflow,float-divide-by-zero \
     -fno-omit-frame-pointer -std=c++17 -fPIC -shared -ffp-contract=off -Wl,-Bsymbolic -Wno-unused-result wrhip.hip -o build/libwrhip_hostsim_asan.so) || exit 1
fi
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=${UBSAN_OPTIONS:-print_stacktrace=1:halt_on_error=1}
export WRHIP_HOSTSIM_LIB=$PWD/$so
if [ $# -eq 0 ]; then set -- tests/test_hostsim_parity.py tests/test_abi_surface.py tests/test_text_reftest_relations.py tests/test_dist.py -q -x -p no:cacheprovider; fi
exec python -m pytest -m "not gpu" "$@"
