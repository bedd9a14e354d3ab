#!/usr/bin/env python3
"""usage: WRHIP_HOSTSIM_NOEXEC=1 python tools/host_profile.py <workload> [frames]
The host side of a frame (recording, staging copies, flush) timed in the GPU-less container: the host-simulation build with every
kernel launch skipped replays the workload's GL trace natively, streamed, and the library's own phase timers say where the time
goes.  Absolute numbers are this container's CPU; the shares carry over to the GPU box's host."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("WRHIP_HOSTSIM_NOEXEC", "1")
from webrender_amd import glapi
from webrender_amd.harness import record_scene, ScenePlayer
from bench import make_frame

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = os.path.join(ROOT, "webrender_amd", "csrc", "libwrhip_hostsim.so")
rec, _ = record_scene(lib, make_frame(wl, encoding="quad"))
p = ScenePlayer(lib, rec)
p.stream(3)
get = C.CFUNCTYPE(None, C.c_void_p)(p.symbol("WrhipGetStats"))
reset = C.CFUNCTYPE(None)(p.symbol("WrhipResetStats"))
best = None
for _ in range(3):
    reset()
    t0 = time.perf_counter()
    p.stream(n)
    wall = time.perf_counter() - t0
    hs = glapi.WrhipStats()
    get(C.byref(hs))
    row = {"wall": 1e6 * wall / n, "record_draws": hs.host_record_ns / n / 1e3, "stage_uploads": hs.host_upload_ns / n / 1e3,
           "flush_and_launch": hs.host_flush_ns / n / 1e3, "blocked": hs.host_wait_ns / n / 1e3}
    if best is None or row["wall"] < best["wall"]:
        best = row
best["other"] = best["wall"] - sum(v for k, v in best.items() if k != "wall")
print(wl, {k: round(v, 1) for k, v in best.items()})
