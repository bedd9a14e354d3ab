"""N>1 path on CPU: world-size-2 (and 3) gloo runs of the strip-sharding +
framebuffer all-gather harness, with the host-simulation backend standing in
for the GPU.  The reassembled window must equal the unsharded render."""
import os
import socket
import subprocess
import sys
import numpy as np
import pytest
from conftest import ROOT, hostsim_lib

WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, os.environ["WR_ROOT"])
import torch, torch.distributed as dist
from webrender_amd import scenes
from webrender_amd.dist import ShardedFramePlayer, strip_rows
from webrender_amd.harness import render_direct
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=rank, world_size=world)
lib = os.environ["WR_LIB"]
make = lambda: scenes.cfg2_overlapping_rects(width=1024, height=1000, n=120, seed=21, fractional=True)
p = ShardedFramePlayer(lib, "custom", "quad", rank, world, device="cpu", frame=make())
p.frames(1, 2)
got = p.assembled()
# per-rank host / setup work: what this rank recorded, staged and set up for two frames ...
import ctypes as C
from webrender_amd.glapi import WrhipStats
st = WrhipStats()
reset = C.CFUNCTYPE(None)(p.symbol("WrhipResetStats")); get = C.CFUNCTYPE(None, C.c_void_p)(p.symbol("WrhipGetStats"))
reset(); p.frames(0, 2); get(C.byref(st))
mine = (st.prims, st.h2d_bytes)
# ... against an unsharded player of the same frame
from webrender_amd.harness import record_scene, ScenePlayer
rec, _ = record_scene(lib, make())
full = ScenePlayer(lib, rec)
full.frames(1, 0)
C.CFUNCTYPE(None)(full.symbol("WrhipResetStats"))(); full.frames(0, 2)
sf = WrhipStats(); C.CFUNCTYPE(None, C.c_void_p)(full.symbol("WrhipGetStats"))(C.byref(sf))
print(f"RANK{rank} prims {mine[0]} of {sf.prims}, h2d {mine[1]} of {sf.h2d_bytes}")
if world == 2:      # the strips are whole tile rows: each rank records, stages and sets up about half of the frame's prims
    # (the data textures -- prim headers, GPU buffers -- are uploaded by every rank: SURVEY section 8e, "broadcast H2D")
    assert mine[0] < 0.65 * sf.prims and mine[1] < sf.h2d_bytes, (mine, sf.prims, sf.h2d_bytes)
want, _ = render_direct(lib, make())
ok = np.array_equal(got, want)
# PER-RANK FRAME BUILDING (SURVEY section 8e: per-rank H2D): the rank builds the frame for the tiles of its strip only, so its data
# textures hold its share of the prims -- same window, and at world 2 at most 0.6 of the unsharded frame's host-to-device bytes
# (a frame with enough prims for the data textures to outweigh the fixed uploads: 8000 rects, BASELINE configs[4]'s generator)
big = lambda tf=None: scenes.cfg5_many_rects(width=1024, height=1000, n=8000, seed=5, **({"tile_filter": tf(1000)} if tf else {}))
want_big, _ = render_direct(lib, big())          # (before the player: render_direct leaves no context current)
pr = ShardedFramePlayer(lib, "custom", "quad", rank, world, device="cpu", frame=big)
pr.frames(1, 1)
ok = ok and np.array_equal(pr.assembled(), want_big)
sr = WrhipStats()
C.CFUNCTYPE(None)(pr.symbol("WrhipResetStats"))(); pr.frames(0, 2); C.CFUNCTYPE(None, C.c_void_p)(pr.symbol("WrhipGetStats"))(C.byref(sr))
del pr
rec_big, _ = record_scene(lib, big())
full_big = ScenePlayer(lib, rec_big)
full_big.frames(1, 0)
C.CFUNCTYPE(None)(full_big.symbol("WrhipResetStats"))(); full_big.frames(0, 2)
sfb = WrhipStats(); C.CFUNCTYPE(None, C.c_void_p)(full_big.symbol("WrhipGetStats"))(C.byref(sfb))
print(f"RANK{rank} per-rank frame: prims {sr.prims} of {sfb.prims}, h2d {sr.h2d_bytes} of {sfb.h2d_bytes} unsharded")
if world == 2:
    assert sr.h2d_bytes <= 0.6 * sfb.h2d_bytes and sr.prims <= 0.6 * sfb.prims, (sr.h2d_bytes, sfb.h2d_bytes, sr.prims, sfb.prims)
# the same frames with the window gathered on the presenting rank only (SURVEY section 8e: "or gather to the presenting GPU")
p2 = ShardedFramePlayer(lib, "custom", "quad", rank, world, device="cpu", frame=make(), gather="root")
p2.frames(1, 2)
if rank == 0:
    ok = ok and np.array_equal(p2.assembled(), want)
else:
    p2._drain()
# the NATIVE per-frame loop (csrc/wr_replay.c wr_shard_stream: what bench.py runs with the RCCL transport), here with its CPU
# stand-in transport: strips written straight into the receivers' windows, no Python per frame
port = os.environ["MASTER_PORT"]
for mode in ("all", "root"):
    pn = ShardedFramePlayer(lib, "custom", "quad", rank, world, device="cpu", frame=make(), gather=mode, native="shm",
                            shm_name=f"/wrshard_{port}_{mode}")
    pn.frames(1, 1)
    pn.stream(3)
    if mode == "all" or rank == 0:
        ok = ok and np.array_equal(pn.assembled(), want)
    dist.barrier()
    pn.close()
# ALTERNATING, distinct frames through the native loop, in the unpipelined and in the pipelined order (WrhipFlushHeld: frame k's
# strips move after frame k + 1's flush): every exchange has to move the strips of ITS frame -- the window after an odd / even number
# of frames is the other / this scene's (ADVICE r4: the pipelined branch had no test)
make_b = lambda: scenes.cfg2_overlapping_rects(width=1024, height=1000, n=120, seed=22, fractional=True)
rec_b, _ = record_scene(lib, make_b())
want_b, _ = render_direct(lib, make_b())
assert not np.array_equal(want_b, want)
for pipelined in ("0", "1"):
    if pipelined == "1":
        os.environ["WRHIP_SHARD_PIPELINE_SHM"] = "1"
    for iters, expect in ((4, want_b), (5, want), (1, want), (2, want_b)):
        pn = ShardedFramePlayer(lib, "custom", "quad", rank, world, device="cpu", frame=make(), gather="all", native="shm",
                                shm_name=f"/wrshard_{port}_alt{pipelined}_{iters}")
        pn.stream_alternating(rec_b.stream, iters)
        got_alt = pn.assembled()
        if not np.array_equal(got_alt, expect):
            print(f"RANK{rank} alternating frames: pipelined={pipelined} iters={iters} differ in {int((got_alt != expect).sum())} bytes")
            ok = False
        dist.barrier()
        pn.close()
os.environ.pop("WRHIP_SHARD_PIPELINE_SHM", None)
# each rank only rasterised its own strip: pixels it does not own stay at the clear colour in its window
y0, y1 = p.fb_rows
flags = torch.tensor([1 if ok else 0])
dist.all_reduce(flags, op=dist.ReduceOp.MIN)
if rank == 0:
    print("DIST_OK" if int(flags.item()) == 1 else "DIST_MISMATCH", got.shape, int((got != want).sum()))
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 3])
def test_strip_sharding_allgather_gloo(world, tmp_path):
    lib = hostsim_lib()
    if lib is None:
        pytest.skip("hostsim library not built")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), WR_ROOT=ROOT, WR_LIB=lib)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0], outs[0]


def test_strip_rows_partition():
    from webrender_amd.dist import strip_rows
    for H in (2160, 4320, 1000, 64, 63):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for r in range(world):
                y0, y1, strip = strip_rows(H, r, world)
                assert y0 <= y1 <= H and (y1 - y0) <= strip and (y0 % 64 == 0 or y0 == H)
                covered.append((y0, y1))
            assert covered[0][0] == 0 and covered[-1][1] == H or any(c[1] == H for c in covered)
            for a, b in zip(covered, covered[1:]):
                assert a[1] == b[0] or b[0] == b[1] == H


# ---- the RCCL transport of the native loop on real GPUs: first contact before the multi-GPU bench -------------------------------
RCCL_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, os.environ["WR_ROOT"])
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=rank, world_size=world,
                        device_id=torch.device("cuda", rank))
from webrender_amd import scenes
from webrender_amd.dist import ShardedFramePlayer
from webrender_amd.harness import render_direct
lib = os.environ["WR_LIB"]
ok = True
# alternating, distinct frames would need two traces; the pipelined order is exercised by streaming MANY frames of one trace and
# comparing the window after every stream length (an exchange one frame late or early shows as a strip of another frame only when
# frames differ -- so the second scene below is streamed after the first on the SAME window size and compared as well)
for make in (lambda: scenes.cfg2_overlapping_rects(width=2048, height=1080, n=150, seed=31),
             lambda: scenes.image_grid(width=2048, height=1024, n=60, seed=32)):
    want, _ = render_direct(lib, make())
    for mode in ("root", "all"):
        pn = ShardedFramePlayer(lib, "custom", "quad", rank, world, device="cuda", frame=make(), gather=mode, native="rccl")
        pn.frames(1, 2)
        for n in (1, 2, 5):
            pn.stream(n)
            if mode == "all" or rank == 0:
                ok = ok and np.array_equal(pn.assembled(), want)
        dist.barrier()
        pn.close()
flags = torch.tensor([1 if ok else 0], device="cuda")
dist.all_reduce(flags, op=dist.ReduceOp.MIN)
if rank == 0:
    print("RCCL_OK" if int(flags.item()) == 1 else "RCCL_MISMATCH")
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_native_shard_loop_over_rccl_two_gpus(tmp_path):
    """wr_shard_open_rccl / grouped ncclSend + ncclRecv between two ranks' windows, both gather modes, pipelined exchange: the
    assembled window equals the unsharded render.  Skips on a box with one GPU (the round's gpurun boxes); the driver's
    multi-GPU node runs it before the N > 1 bench does."""
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from conftest import wrhip_lib
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   WR_ROOT=ROOT, WR_LIB=wrhip_lib(), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "RCCL_OK" in outs[0], outs[0]
