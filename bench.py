#!/usr/bin/env python3
"""bench.py -- frames/s of the WebRender batched-primitive raster path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one frame: the complete C-ABI call stream WebRender's render thread
issues for the workload (per-frame data-texture + instance uploads, every
picture-cache tile rasterised, the composite pass, Finish), replayed natively
(csrc/wr_replay.c) against libwrhip.  Workload at N=1 is BASELINE.json
configs[1]: 1000 overlapping translucent rects at 3840x2160 ("cfg2").  At N>1
it is configs[4] -- 100 k rects at 7680x4320 ("cfg5"), the configuration the
tile sharding is for: one frame split N ways ("strong" scaling), the window
reassembled on rank 0 -- the presenting GPU -- by an RCCL gather of the ranks'
strips; the line also carries the same workload on one GPU measured in the same
run (`single_gpu_same_workload`), the all-gather variant (every rank ends up
with the window) and every rank's prims / upload bytes per frame.

The timed region (K frames issued back to back + one Finish, between barrier +
synchronize) is repeated REPEATS times and the median region is reported, so a
short driver run (--steps 20 = 2 ms of cfg2) is a steady-state number and not
one pipeline fill.

One JSON line on rank 0 with metric/value plus
  roofline      the DOMINANT kernel -- the raster-kernel variant with the largest
                summed GPU time per frame: its own algorithmic bytes per launch /
                its own average launch duration (hipEvents on the context's
                stream around every launch, WrhipSetProfiling), against the
                8 TB/s HBM peak; `per_kernel` lists every kernel of the frame
                the same way (upload scatter, setup stage, each raster variant)
  cpu_baseline  the reference's own swgl rasteriser (oracle/_ref/libswgl_ref_gen_clang.so:
                gl.cc compiled from /root/reference with the shader headers
                oracle/gen generates from the reference's GLSL, clang build =
                the shipping flags of swgl/build.rs) replaying the same
                call stream on one host core, plus an N-process run (one
                process per core, frames replayed independently) with the core
                count stated
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def measured_ceiling():
    """What a streaming kernel achieves on this chip in ONE launch of the headline pass's size (tools/ubench/stream.hip, committed as
    profiles/*_stream_ubench.json): the write-only figure is the practical denominator of a tile pass that starts from a clear."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_stream_ubench.json")))
    if not files:
        return None
    r = {e["name"]: e for e in json.load(open(files[-1]))["results"]}
    pick = lambda n: r[n]["median_GBps"] if n in r else None
    return {"store_only_75MB_GBps": pick("store_wg2_75.8MB"), "copy_75MB_GBps": pick("copy_75.8MB"), "triad_75MB_GBps": pick("triad_75.8MB"),
            "store_only_1GiB_GBps": pick("store_wg_1GiB"), "copy_1GiB_GBps": pick("copy_1GiB"), "empty_launch_us": r.get("empty_launch", {}).get("median_us"),
            "source": "profiles/" + os.path.basename(files[-1])}
REPEATS = 7            # timed regions per run; the median is reported


def pin_to_gpu_numa_node(local_rank):
    """Run this process (and the threads the backend starts) on the CPU socket the GPU hangs off: the render thread writes the pinned
    staging ring and rings doorbells for every frame, and a two-socket host places a new process on either socket (what `numactl
    --cpunodebind` does for wrench on a multi-socket box).  Only when the launcher left the process free to run anywhere; WRHIP_BENCH_NO_PIN=1
    turns it off.  Returns a description for the bench line."""
    if os.environ.get("WRHIP_BENCH_NO_PIN"):
        return "not pinned (WRHIP_BENCH_NO_PIN)"
    try:
        import ctypes as C
        import torch
        hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(local_rank)) != 0:
            return "not pinned (no PCI bus id)"
        bus = buf.value.decode().lower()
        node = int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read())
        if node < 0:
            return "not pinned (device reports no NUMA node)"
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        allowed = os.sched_getaffinity(0)
        if len(allowed) < os.cpu_count():
            return f"not pinned (launcher restricted the process to {len(allowed)} CPUs)"
        os.sched_setaffinity(0, cpus & allowed)
        return f"NUMA node {node} of GPU {bus} ({len(cpus & allowed)} CPUs)"
    except Exception as e:          # noqa: BLE001 -- placement is an optimisation, never a failure
        return f"not pinned ({type(e).__name__}: {e})"


def make_frame(workload, **kw):
    from webrender_amd import scenes
    return scenes.make_workload(workload, **kw)


def kernel_label(k):
    if k.kind == 0:
        return "wr_upload_kernel"
    if k.kind == 1:
        return "wr_setup_kernel"
    if k.kind == 3:
        return "wr_mask_rows_kernel"
    if k.kind == 4:      # a run of thin R8 levels in one launch (k.depth = levels)
        return f"wr_raster_chain_kernel<{k.feat}> x{k.depth} levels"
    if k.kind == 5:      # glyph levels: the 128-VGPR instantiation of the textured variant
        return f"wr_raster_dense_kernel<{k.fmt}, {'true' if k.depth else 'false'}, 4, {k.feat}>"
    if k.kind == 6:      # throughput mode: the tile pass with the NEXT flush's setup stage in front (one launch)
        return f"wr_setup_raster_kernel<{k.fmt}, {'true' if k.depth else 'false'}, 4, {k.feat}>"
    if k.kind == 7:
        return f"wr_setup_raster_dense_kernel<{k.fmt}, {'true' if k.depth else 'false'}, 4, {k.feat}>"
    if k.kind == 8:
        return "wr_setup_rows_kernel"
    if k.kind == 9:      # the span-rows targets of a level (cs_blur / cs_scale passes): one wave per target row piece
        return f"wr_span_rows_kernel<{k.fmt}>"
    if k.kind == 10:     # picture targets of a few large gradient / image prims: one wave per tile row piece
        return "wr_tile_rows_kernel"
    if k.kind == 11:     # ... with the next flush's setup stage in front
        return "wr_setup_tile_rows_kernel"
    if k.kind == 12:     # thin launches: the R = 1 instantiation (64 x 4 pixels per wave, several workgroups per bin)
        return f"wr_raster_kernel<{k.fmt}, false, 1, {k.feat}>"
    if k.kind == 13:     # ... with the next flush's setup stage in front
        return f"wr_setup_raster_thin_kernel<{k.fmt}, false, 1, {k.feat}>"
    return f"wr_raster_kernel<{k.fmt}, {'true' if k.depth else 'false'}, 4, {k.feat}>"


def pmc_traffic(workload, encoding, label):
    """HBM bytes per launch of kernel `label` from the rocprofv3 PMC passes committed under
    profiles/ (FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on
    gfx950).  None when no such profile exists for the workload: counters cannot be collected
    from inside the timed process.  In throughput mode the tile pass runs fused with the next
    frame's setup stage (wr_setup_raster_kernel<...>): both spellings are matched."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_hbm_{workload}.json")))
    if not files or encoding != "quad":
        return None
    d = json.load(open(files[-1]))
    # the counters describe the kernels they were collected with: a profile of other kernel sources says nothing about these
    import hashlib
    cur = hashlib.sha256(b"".join(open(os.path.join(ROOT, "webrender_amd", "csrc", f), "rb").read()
                                  for f in ("wrhip_kernels.h", "wrhip_k_setup.h", "wrhip_k_pixels.h", "wrhip_k_rows.h", "wrhip_k_raster.h", "wrhip_types.h"))).hexdigest()[:16]
    if d.get("kernel_sources_sha256_16") != cur:
        return None
    tag = label[label.index("<"):] if "<" in label else label
    tot = n = 0
    for name, k in d["kernels"].items():
        if (label.split("<")[0] + tag in name) if "<" in label else (label in name):
            tot += k["hbm_bytes_per_launch"] * k["launches"]
            n += k["launches"]
    if not n:
        return None
    return int(tot / n), "profiles/" + os.path.basename(files[-1])


def _cpu_worker(args):
    lib, rec, n = args
    from webrender_amd.harness import ScenePlayer
    p = ScenePlayer(lib, rec)
    return list(p.frames(0, n))


def cpu_baseline(rec, budget_s=12.0):
    """Reference swgl rasteriser on the host cores, bounded sample of the same frame trace: one core
    (swgl is single-threaded by design, swgl/README.md:6), then one replaying process per core."""
    from webrender_amd.harness import ScenePlayer
    kind = "reference"
    for name in ("libswgl_ref_gen_clang.so", "libswgl_ref_gen.so", "libswgl_ref_clang.so", "libswgl_ref_gcc.so"):
        lib = os.path.join(ROOT, "oracle", "_ref", name)
        if os.path.exists(lib):
            break
    else:
        return None
    generated = "_gen" in os.path.basename(lib)
    p = ScenePlayer(lib, rec)          # setup + first frame (untimed)
    t0 = time.perf_counter()
    ms = list(p.frames(0, 1))
    per = ms[0] / 1e3
    n = int(max(2, min(60, budget_s / max(per, 1e-3))))
    ms += list(p.frames(0, n - 1))
    ms = np.array(ms)
    out = {"value": round(1e3 / ms.mean(), 4), "unit": "frames/s", "cores": 1, "kind": kind,
           "ms_per_frame": round(float(ms.mean()), 3),
           "oracle": ("swgl's gl.cc compiled unmodified from /root/reference with the shader headers (82 program keys) generated "
                      "from webrender/res/*.glsl by oracle/gen (swgl/build.rs + glsl-to-cxx restated; clang, build.rs's flags)"
                      if generated else
                      "swgl's gl.cc compiled unmodified from /root/reference; the shader headers (48 program keys) it includes are "
                      "this repo's hand-written restatements of webrender/res/*.glsl"),
           "sample": f"{len(ms)} frames of the same trace replayed by swgl ({os.path.basename(lib)}), "
                     f"{time.perf_counter() - t0:.1f} s"}
    # N processes, one per core, each replaying whole frames (how a tile-parallel swgl would use the box)
    try:
        import multiprocessing as mp
        cores = len(os.sched_getaffinity(0))
        procs = max(1, min(cores, 64))
        per_proc = int(max(2, min(30, 8.0 / max(ms.mean() / 1e3, 1e-3))))
        t1 = time.perf_counter()
        with mp.get_context("spawn").Pool(procs) as pool:
            res = pool.map(_cpu_worker, [(lib, rec, per_proc)] * procs)
        wall = max(sum(r) for r in res) / 1e3
        out["multi_process"] = {"value": round(procs * per_proc / wall, 3), "unit": "frames/s", "cores": procs,
                                "host_cores": cores, "sample": f"{procs} processes x {per_proc} frames, {time.perf_counter() - t1:.1f} s wall"}
    except Exception as e:          # noqa: BLE001 -- the single-core figure is the contract; this one is extra
        out["multi_process"] = {"error": str(e)}
    return out


XGMI_LINK_GBS = 153.0        # one xGMI link, one direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point)
XGMI_EFFICIENCY = 0.7        # ASSUMED payload efficiency of a large ncclSend / ncclRecv over one link (not measured: no multi-GPU node)


def emulate_world(args):
    """bench.py --emulate-world W [--workload cfg5]: what each rank of a W-rank run would do, one rank after the other on this one
    GPU -- the rank's frame (built for its tiles: webrender_amd/dist.py build_rank_frame), its row restriction, its uploads -- each
    streamed and timed like the N = 1 bench; then the projection: the frame rate of the slowest rank against the time the presenting
    GPU needs to receive the other ranks' strips, each over its own xGMI link (the native loop moves frame k's strips while frame
    k + 1 is flushed, so the two overlap: frame time = max of the two).  Everything under "projected" is a model, labelled so."""
    import torch
    import ctypes as C
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libwrhip has no CPU path")
    torch.cuda.set_device(0)
    from webrender_amd import glapi
    from webrender_amd.harness import record_scene, ScenePlayer
    from webrender_amd.dist import ShardedFramePlayer, strip_rows
    W = args.emulate_world
    wl = args.workload or "cfg5"
    lib = glapi.wrhip_path()

    def timed(player, stats_of):
        player.frames(args.warmup, 0)
        player.stream(max(2, args.warmup))
        reg = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            player.stream(args.steps)
            torch.cuda.synchronize()
            reg.append(time.perf_counter() - t0)
        st = glapi.WrhipStats()
        C.CFUNCTYPE(None)(stats_of.symbol("WrhipResetStats"))()
        player.stream(args.steps)
        C.CFUNCTYPE(None, C.c_void_p)(stats_of.symbol("WrhipGetStats"))(C.byref(st))
        return float(np.median(reg)) / args.steps, st

    rec, _ = record_scene(lib, make_frame(wl, encoding=args.encoding))
    full = ScenePlayer(lib, rec)
    t1, s1 = timed(full, full)
    width, height = rec.width, rec.height
    del full
    ranks = []
    for r in range(W):
        p = ShardedFramePlayer(lib, wl, args.encoding, r, W, device="cuda", native=None)
        t, st = timed(p.player, p)          # (the rank's frames through the plain replayer: no exchange, that is the model's part)
        sy0, sy1, _ = strip_rows(height, r, W)
        ranks.append({"rank": r, "screen_rows": [int(sy0), int(sy1)], "us_per_frame": round(1e6 * t, 1), "prims_per_frame": int(st.prims // args.steps),
                      "h2d_bytes_per_frame": int(st.h2d_bytes // args.steps)})
        del p
    slow = max(x["us_per_frame"] for x in ranks)
    strip_bytes = max(x["screen_rows"][1] - x["screen_rows"][0] for x in ranks[1:]) * width * 4 if W > 1 else 0
    gather_us = 1e6 * strip_bytes / (XGMI_LINK_GBS * 1e9 * XGMI_EFFICIENCY)
    frame_us = max(slow, gather_us)
    out = {"metric": f"PROJECTED frames/sec of {wl} split over {W} GPUs, from every rank's share measured on ONE GPU (bench.py --emulate-world)",
           "value": round(1e6 / frame_us, 2), "unit": "frames/s", "projected": True, "n_gpus": 1, "emulated_world": W,
           "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": wl, "encoding": args.encoding, "target": f"{width}x{height}"},
           "single_gpu": {"us_per_frame": round(1e6 * t1, 1), "frames_per_s": round(1 / t1, 2), "h2d_bytes_per_frame": int(s1.h2d_bytes // args.steps),
                          "prims_per_frame": int(s1.prims // args.steps)},
           "per_rank": ranks,
           "model": {"slowest_rank_us": slow, "gather_us": round(gather_us, 1), "strip_bytes": int(strip_bytes),
                     "xgmi_link_GBps": XGMI_LINK_GBS, "assumed_efficiency": XGMI_EFFICIENCY,
                     "statement": "rank 0 receives the W - 1 other strips concurrently, each over its own point-to-point xGMI link, while the next frame is "
                                  "flushed (the native loop's pipelined order): frame time = max(slowest rank, one strip / (link x efficiency)); "
                                  "RCCL launch latency, the host's share of a rank's frame under W processes and PCIe contention between ranks are NOT modelled"},
           "projected_speedup_vs_single_gpu": round(t1 * 1e6 / frame_us, 3),
           "max_rank_h2d_frac": round(max(x["h2d_bytes_per_frame"] for x in ranks) / max(1, int(s1.h2d_bytes // args.steps)), 3)}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None,
                    help="default: cfg2 (BASELINE configs[1], the headline) on one GPU, cfg5 (configs[4]: 100 k rects at 8K, "
                         "the configuration the tile sharding is for) on several")
    ap.add_argument("--encoding", default="quad", choices=["quad", "brush"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true", help="take the N > 1 code path (sharded player, collectives) even with one rank: a self-test")
    ap.add_argument("--emulate-world", type=int, default=0, metavar="W",
                    help="on ONE GPU: every rank's share of a W-rank run (its strip's rows, its frame's uploads) timed back to back, "
                         "and a PROJECTED W-GPU rate under a stated xGMI gather model (no collective runs)")
    args = ap.parse_args()
    if args.emulate_world:
        return emulate_world(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    sharded = world > 1 or args.sharded
    if args.workload is None:
        args.workload = "cfg5" if sharded else "cfg2"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libwrhip has no CPU path")
    torch.cuda.set_device(local_rank)
    affinity = pin_to_gpu_numa_node(local_rank)
    if not sharded and os.environ.get("WRHIP_BENCH_INIT_DIST"):
        # A/B: the UNSHARDED bench inside a process that initialised torch.distributed with the given backend
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        be = os.environ["WRHIP_BENCH_INIT_DIST"]
        dist.init_process_group(be, **({"device_id": torch.device("cuda", local_rank)} if be == "nccl" else {}))
        if os.environ.get("WRHIP_BENCH_DIST_WARM"):      # ... and made it create its communicator (one collective)
            t = torch.zeros(1, device="cuda" if be == "nccl" else "cpu"); dist.all_reduce(t); torch.cuda.synchronize()
    if sharded:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from webrender_amd import glapi
    from webrender_amd.harness import record_scene, ScenePlayer
    lib = glapi.wrhip_path()
    if not os.path.exists(lib):
        raise SystemExit("libwrhip.so missing; run __graft_entry__.build()")

    single = None            # N > 1: the same workload unsharded on one GPU (rank 0), measured in this run: the 1-GPU point of the curve
    if sharded:
        from webrender_amd.dist import ShardedFramePlayer
        if rank == 0:
            rec1, _ = record_scene(lib, make_frame(args.workload, encoding=args.encoding))
            p1 = ScenePlayer(lib, rec1)
            p1.frames(args.warmup, 0)
            reg = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                p1.stream(args.steps)
                torch.cuda.synchronize()
                reg.append(time.perf_counter() - t0)
            single = {"value": round(args.steps / float(np.median(reg)), 2), "unit": "frames/s", "n_gpus": 1,
                      "note": "same workload, unsharded, rank 0's GPU, this run"}
            del p1
        # (value: the window assembled on rank 0, the presenting GPU -- SURVEY section 8e's "gather to the presenting GPU"; the
        # all-gather variant, every rank ending up with the whole window, is timed alongside: at 8K it moves 132 MB per frame
        # into EVERY GPU, which bounds it near 0.4 k frames/s whatever the raster rate)
        # (the native loop owns its RCCL communicator; if it cannot be opened -- librccl not where torch keeps it, a symbol missing --
        # every rank gets the same refusal and takes the torch.distributed loop the native one is tested against: slower per frame
        # (Python per frame), same pixels; the line says which transport ran)
        transport = os.environ.get("WRHIP_BENCH_TRANSPORT", "rccl")
        def open_sharded(gather):
            if transport == "rccl":
                try:
                    return ShardedFramePlayer(lib, args.workload, args.encoding, rank, world, gather=gather, native="rccl"), "native loop, RCCL"
                except RuntimeError as e:
                    print(f"bench: native sharded loop unavailable on rank {rank} ({e}); torch.distributed loop instead", file=sys.stderr)
            return ShardedFramePlayer(lib, args.workload, args.encoding, rank, world, gather=gather, native=None), "torch.distributed loop (fallback)"
        player, transport_used = open_sharded("root")
        frame_w, frame_h = player.width, player.height
        rec = None
    else:
        frame = make_frame(args.workload, encoding=args.encoding)
        frame_w, frame_h = frame.width, frame.height
        rec, _ = record_scene(lib, frame)
        player = ScenePlayer(lib, rec)

    def barrier():
        if sharded:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region: K frames issued back to back (the GL contract does not ask for
    # a glFinish per frame; the backend pipelines host recording with GPU
    # execution), one Finish at the end, bracketed by barrier + synchronize.
    player.frames(args.warmup, 0)
    player.stream(max(2, args.warmup))      # (the streamed path has kernels of its own -- the fused tile pass -- : load them untimed)
    regions = []
    for _ in range(REPEATS):
        barrier()
        t0 = time.perf_counter()
        player.stream(args.steps)
        barrier()
        regions.append(time.perf_counter() - t0)
    if sharded:
        import torch.distributed as dist
        t = torch.tensor(regions, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)        # every region: the slowest rank
        regions = [float(v) for v in t.tolist()]
    elapsed = float(np.median(regions))
    # per-frame latency with a Finish after every frame (what `wrench perf` samples)
    lat = player.frames(0, min(args.steps, 50))

    # ---- N > 1: per-rank work, and the same frames with the window gathered on rank 0 only ------------------------------------
    multi = None
    if sharded:
        import ctypes as C
        import torch.distributed as dist
        st = glapi.WrhipStats()
        C.CFUNCTYPE(None)(player.symbol("WrhipResetStats"))()
        player.stream(args.steps)
        C.CFUNCTYPE(None, C.c_void_p)(player.symbol("WrhipGetStats"))(C.byref(st))
        mine = {"rank": rank, "prims_per_frame": int(st.prims // args.steps), "h2d_bytes_per_frame": int(st.h2d_bytes // args.steps),
                "strip_rows": int(max(0, player.fb_rows[1] - player.fb_rows[0]))}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        del player
        root, _ = open_sharded("all")
        root.frames(args.warmup, 0)
        rr = []
        for _ in range(3):
            barrier()
            t0 = time.perf_counter()
            root.stream(args.steps)
            barrier()
            rr.append(time.perf_counter() - t0)
        t = torch.tensor(rr, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        multi = {"rccl_ranks": world, "transport": transport_used, "collective": "window strips to rank 0, the presenting GPU (value): grouped ncclSend / ncclRecv on the backend's stream, "
                                                    "in place between the ranks' windows, per-frame loop in native code (csrc/wr_replay.c wr_shard_stream); every rank to every rank alongside",
                 "all_gather": {"value": round(args.steps / float(np.median(t.tolist())), 2), "unit": "frames/s"},
                 "window_bytes_per_frame": int(root.height * root.row_bytes), "per_rank": per_rank}
        player = root

    # ---- where the host side of a frame goes: the library's own phase timers over one more streamed region -------
    import ctypes as C
    host = None
    if not sharded:
        _get = C.CFUNCTYPE(None, C.c_void_p)(player.symbol("WrhipGetStats"))
        _reset = C.CFUNCTYPE(None)(player.symbol("WrhipResetStats"))
        barrier()
        _reset()
        t0 = time.perf_counter()
        player.stream(args.steps)
        issued = time.perf_counter() - t0           # (stream() ends with the one Finish of the region)
        barrier()
        hs = glapi.WrhipStats()
        _get(C.byref(hs))
        n = float(args.steps)
        host = {"unit": "us per frame", "wall": round(1e6 * issued / n, 2),
                "record_draws": round(hs.host_record_ns / n / 1e3, 2), "stage_uploads": round(hs.host_upload_ns / n / 1e3, 2),
                "flush_and_launch": round(hs.host_flush_ns / n / 1e3, 2), "blocked_on_stream": round(hs.host_wait_ns / n / 1e3, 2)}
        host["other_calls_and_replayer"] = round(host["wall"] - host["record_draws"] - host["stage_uploads"] - host["flush_and_launch"]
                                                 - host["blocked_on_stream"], 2)

    # ---- per-kernel rooflines (separate, event-timed pass: one event pair + wait per launch) -------
    roof = None
    if not sharded:
        get_stats = C.CFUNCTYPE(None, C.c_void_p)(player.symbol("WrhipGetStats"))
        get_kstats = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)(player.symbol("WrhipGetKernelStats"))
        reset = C.CFUNCTYPE(None)(player.symbol("WrhipResetStats"))
        prof = C.CFUNCTYPE(None, C.c_int)(player.symbol("WrhipSetProfiling"))
        # mode 2: the launches stay where the TIMED mode issues them (raster launches held back and fused with the next frame's
        # setup stage), so the kernels listed here are the ones that produced `value`
        prof(2)
        player.stream(3)
        reset()
        nprof = min(args.steps, 50)
        player.stream(nprof)
        st = glapi.WrhipStats()
        get_stats(C.byref(st))
        ks = (glapi.WrhipKernelStat * 32)()
        nk = get_kstats(ks, 32)
        prof(0)
        per_kernel = []
        for k in list(ks)[:nk]:
            if not k.launches or not k.ns:
                continue
            us = k.ns / k.launches / 1e3
            ab = k.algo_bytes / k.launches
            e = {"name": kernel_label(k), "launches_per_frame": round(k.launches / nprof, 2), "us": round(us, 2),
                 "us_per_frame": round(k.ns / nprof / 1e3, 2), "workgroups": int(k.workgroups / k.launches),
                 "algo_bytes": int(ab), "GBps": round(ab / (k.ns / k.launches), 1),
                 "frac": round(ab / (k.ns / k.launches) / HBM_PEAK_GBS, 4), "traffic": None}
            tr = pmc_traffic(args.workload, args.encoding, e["name"])
            if tr:
                e["traffic"], e["traffic_source"] = tr
            per_kernel.append(e)
        rasters = [e for e in per_kernel if e["name"].startswith("wr_raster_") or e["name"].startswith("wr_setup_raster") or e["name"].startswith("wr_span_rows") or e["name"].startswith("wr_tile_rows") or e["name"].startswith("wr_setup_tile_rows")]
        if rasters:
            dom = max(rasters, key=lambda e: e["us_per_frame"])
            roof = {"bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
                    "traffic": dom["traffic"], "kernel": dom["name"], "avg_launch_us": dom["us"],
                    "algo_bytes_per_launch": dom["algo_bytes"],
                    "raster_us_per_frame": round(sum(e["us_per_frame"] for e in rasters), 2),
                    "kernel_us_per_frame": round(sum(e["us_per_frame"] for e in per_kernel), 2),
                    "frame_algo_bytes": int(sum(e["algo_bytes"] * e["launches_per_frame"] for e in per_kernel)),
                    "mode": "throughput (frames streamed, launches event-timed where that mode issues them)",
                    "per_kernel": per_kernel}
            if dom.get("traffic_source"):
                roof["traffic_source"] = dom["traffic_source"]
            mc = measured_ceiling()
            if mc:
                roof["peak_measured"] = mc
                if mc.get("store_only_75MB_GBps"):
                    roof["frac_of_measured_store_only"] = round(dom["GBps"] / mc["store_only_75MB_GBps"], 4)

    if rank == 0:
        from webrender_amd.wrench_scenes import DESCRIPTIONS as wrench_desc
        fps = args.steps / elapsed
        out = {
            "metric": "frames/sec + Mpixels/sec on wrench benchmarks, 4K target, 1/2/4/8 GPU",
            "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "mpixels_per_s": round(fps * frame_w * frame_h / 1e6, 1),
            "frame_latency_ms": round(float(np.mean(lat)), 4),
            "timed_regions_ms": [round(1e3 * r, 3) for r in regions],
            "config": {"workload": f"{args.workload}: " + wrench_desc.get(args.workload, "") + {
                "cfg2": "1000 overlapping translucent rects (ps_quad_textured + premultiplied-alpha blend), "
                        "3840x2160, 20 picture-cache tiles + composite, seed 2",
                "cfg5": "100k rects (50% opaque), 7680x4320, 72 tiles + composite, seed 5 (BASELINE configs[4])",
                "cfg4": "box-shadow-large.yaml at device_pixel_scale 2 (shadow 1840^2 px): cs_clip_rectangle mask -> "
                        "2 cs_scale halvings -> cs_blur V/H -> cs_clip_box_shadow x clip-out mask -> masked "
                        "brush_solid -> composite, 3840x2160",
                "cfg3": "text: 200 lines x 250 glyphs (ps_text_run, R8 glyph atlas 2048^2, premultiplied-alpha "
                        "blend), 3840x2160, 20 tiles + composite, seed 3",
                "transforms": "wrench benchmarks/transforms-simple.yaml: 11 full-size translucent rects under rotate(45), 1024x1024",
                "simple-batching": "wrench benchmarks/simple-batching.yaml: 14 x rect [0,0,512,512] green, 3840x2160",
                "cfg1": "16x16 opaque rect grid 1024x1024"}.get(args.workload, ""),
                "encoding": args.encoding, "target": f"{frame_w}x{frame_h}",
                "parallelism": "single GPU" if not sharded else
                f"tile rows sharded over {world} GPUs + RCCL gather of framebuffer strips to rank 0"},
        }
        if multi:
            out["multi_gpu"] = multi
        if single:
            out["single_gpu_same_workload"] = single
            out["speedup_vs_single_gpu"] = round(fps / single["value"], 3)
            # N > 1 runs another workload than the N = 1 headline (configs[4], the one the sharding is for): the metric names it, and
            # the baseline of a strong-scaling point is the same frame on one GPU, measured in this very run
            out["metric"] += f" [N > 1: {args.workload}, one frame split over the ranks]"
            out["vs_baseline"] = out["speedup_vs_single_gpu"]
            out["vs_baseline_note"] = "value / single_gpu_same_workload.value (same workload, unsharded, rank 0's GPU, this run); BASELINE.md has no published number"
        out["host_affinity"] = affinity
        if host:
            out["host"] = host
            if roof:
                # is the GPU the bound?  kernel time per frame (event-timed pass) against the step, and the host's own work against it
                out["gpu_busy_frac"] = round(roof["kernel_us_per_frame"] / (1e3 * out["ms_per_step"]), 3)
                host_work = host["wall"] - host["blocked_on_stream"]
                out["host_bound"] = bool(host_work > roof["kernel_us_per_frame"])
                out["host_work_us_per_frame"] = round(host_work, 2)
        if roof:
            out["roofline"] = roof
        if not sharded and not args.no_cpu_baseline:
            cb = cpu_baseline(rec)
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if sharded:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
