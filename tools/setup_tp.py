"""usage: WRHIP_LIB_PATH=ab/libwrhip_timing.so WRHIP_PRIM_TIMES=<file> python tools/setup_tp.py [workload]  -- latency-mode frames only
(the standalone setup kernel), then the time points of its last launch (WR_TP in wrhip_kernels.h, WRHIP_TIMING build)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
torch.cuda.init()
from webrender_amd import glapi, scenes
from webrender_amd.harness import record_scene, ScenePlayer
w = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
lib = glapi.wrhip_path()
rec, _ = record_scene(lib, scenes.make_workload(w, encoding="quad"))
p = ScenePlayer(lib, rec)
ms = p.frames(5, 10)
print(w, "latency-mode ms per frame", np.round(ms, 4))
del p
import gc; gc.collect()
