#!/bin/bash
# usage: tools/tsan.sh  -- the host simulation built with ThreadSanitizer, and the paths that hand work between threads (the submit helper
# wrq, the copy pool, the staging ring's fence-based reuse) driven DIRECTLY by python scripts: a pytest parent that forks the driver as a
# subprocess hangs in TSan's fork interceptor (multi-threaded fork), which is the tool's limit, not the library's.
cd "$(dirname "$0")/.." || exit 1
so=webrender_amd/csrc/build/libwrhip_hostsim_tsan.so
if [ ! -e $so ] || [ webrender_amd/csrc/wrhip.hip -nt $so ] || [ webrender_amd/csrc/wrhip_rt.h -nt $so ]; then
  mkdir -p webrender_amd/csrc/build
  (cd webrender_amd/csrc && g++ -x c++ -DWRHIP_HOSTSIM -O1 -g -fsanitize=thread -fno-omit-frame-pointer -std=c++17 -fPIC -shared -ffp-contract=off \
     -Wl,-Bsymbolic -Wno-unused-result wrhip.hip -o build/libwrhip_hostsim_tsan.so) || exit 1
fi
export LD_PRELOAD="$(gcc -print-file-name=libtsan.so)" TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0"
n=0
for env in "" "WRHIP_STAGING_BYTES=1048576" "WRHIP_STAGING_BYTES=1048576 WRHIP_RING_DRAIN=1"; do
  n=$((n + $(env $env timeout 600 python tests/ring_wrap_driver.py $PWD/$so 4 2>&1 >/dev/null | grep -c "WARNING: ThreadSanitizer")))
done
cat > /tmp/wr_tsan_scenes.py <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from webrender_amd import scenes
from webrender_amd.harness import render_direct
# (texture uploads above 1 MB go through the copy pool's helpers)
for make in (lambda: scenes.image_grid(), lambda: scenes.cfg3_text(width=1024, height=512, lines=24, glyphs_per_line=60), lambda: scenes.mix_blend_grid(seed=213, perspective="clip")):
    render_direct(sys.argv[1], make())
PY
m=$(timeout 900 python /tmp/wr_tsan_scenes.py $PWD/$so 2>&1 >/dev/null | grep -c "WARNING: ThreadSanitizer")
n=$((n + m))
echo "ThreadSanitizer warnings: $n"
[ $n -eq 0 ]
