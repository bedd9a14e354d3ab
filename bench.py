#!/usr/bin/env python3
"""bench.py -- frames/s of the WebRender batched-primitive raster path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one frame: the complete C-ABI call stream WebRender's render thread
issues for the workload (per-frame data-texture + instance uploads, every
picture-cache tile rasterised, the composite pass, Finish), replayed natively
(csrc/wr_replay.c) against libwrhip.  Workload at N=1 is BASELINE.json
configs[1]: 1000 overlapping translucent rects at 3840x2160 ("cfg2").

One JSON line on rank 0 with metric/value plus
  roofline      dominant kernel (wr_raster_kernel): algorithmic bytes per launch /
                average launch duration measured with hipEvents on the
                context's stream, against the 8 TB/s HBM peak
  cpu_baseline  the reference's own swgl (oracle/_ref, clang build = what ships)
                replaying the same call stream on one host core
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


def make_frame(workload, **kw):
    from webrender_amd import scenes
    if workload == "cfg2":
        return scenes.cfg2_overlapping_rects(**kw)
    if workload == "cfg5":
        return scenes.cfg5_many_rects(**kw)
    if workload == "cfg1":
        return scenes.cfg1_solid_colors(**kw)
    if workload == "cfg4":
        kw.pop("encoding", None)
        return scenes.cfg4_box_shadow(dps=2.0, **kw)
    if workload == "cfg3":
        kw.pop("encoding", None)
        return scenes.cfg3_text(**kw)
    raise SystemExit(f"unknown workload {workload}")


def pmc_traffic(workload, encoding):
    """HBM bytes per raster launch from the rocprofv3 PMC passes committed under
    profiles/ (FETCH_SIZE and WRITE_SIZE collected in separate runs of this
    same command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    16 B/lane streaming reads on gfx950).  None when no such profile exists for
    the workload: counters cannot be collected from inside the timed process."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                          f"*pmc_hbm_{workload}.json")))
    if not files or encoding != "quad":
        return None
    d = json.load(open(files[-1]))
    v = [k["hbm_bytes_per_launch"] for n, k in d["kernels"].items() if "wr_raster_kernel" in n]
    if not v:
        return None
    return int(sum(v) / len(v)), "profiles/" + os.path.basename(files[-1])


def cpu_baseline(rec, budget_s=12.0):
    """Reference swgl on one host core, bounded sample of the same frame trace."""
    from webrender_amd.harness import ScenePlayer
    lib = os.path.join(ROOT, "oracle", "_ref", "libswgl_ref_clang.so")
    kind = "reference"
    if not os.path.exists(lib):
        lib = os.path.join(ROOT, "oracle", "_ref", "libswgl_ref_gcc.so")
    if not os.path.exists(lib):
        return None
    p = ScenePlayer(lib, rec)          # setup + first frame (untimed)
    t0 = time.perf_counter()
    ms = list(p.frames(0, 1))
    per = ms[0] / 1e3
    n = int(max(2, min(60, budget_s / max(per, 1e-3))))
    ms += list(p.frames(0, n - 1))
    ms = np.array(ms)
    return {"value": round(1e3 / ms.mean(), 4), "unit": "frames/s", "cores": 1, "kind": kind,
            "ms_per_frame": round(float(ms.mean()), 3),
            "sample": f"{len(ms)} frames of the same trace replayed by swgl ({os.path.basename(lib)}), "
                      f"{time.perf_counter() - t0:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--encoding", default="quad", choices=["quad", "brush"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libwrhip has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from webrender_amd import glapi
    from webrender_amd.harness import record_scene, ScenePlayer
    lib = glapi.wrhip_path()
    if not os.path.exists(lib):
        raise SystemExit("libwrhip.so missing; run __graft_entry__.build()")

    if world > 1:
        from webrender_amd.dist import ShardedFramePlayer
        player = ShardedFramePlayer(lib, args.workload, args.encoding, rank, world)
        frame_w, frame_h = player.width, player.height
        rec = None
    else:
        frame = make_frame(args.workload, encoding=args.encoding)
        frame_w, frame_h = frame.width, frame.height
        rec, _ = record_scene(lib, frame)
        player = ScenePlayer(lib, rec)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region: K frames issued back to back (the GL contract does not ask for
    # a glFinish per frame; the backend pipelines host recording with GPU
    # execution), one Finish at the end, bracketed by barrier + synchronize.
    player.frames(args.warmup, 0)
    barrier()
    t0 = time.perf_counter()
    player.stream(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    # per-frame latency with a Finish after every frame (what `wrench perf` samples)
    lat = player.frames(0, min(args.steps, 50))
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (separate, event-timed pass) -------
    import ctypes as C
    roof = None
    if world == 1:
        get_stats = C.CFUNCTYPE(None, C.c_void_p)(player.symbol("WrhipGetStats"))
        reset = C.CFUNCTYPE(None)(player.symbol("WrhipResetStats"))
        prof = C.CFUNCTYPE(None, C.c_int)(player.symbol("WrhipSetProfiling"))
        prof(1)
        player.frames(3, 0)
        reset()
        nprof = min(args.steps, 50)
        player.frames(0, nprof)
        st = glapi.WrhipStats()
        get_stats(C.byref(st))
        prof(0)
        if st.raster_launches and st.raster_ns:
            avg_ns = st.raster_ns / st.raster_launches
            bytes_per_launch = st.raster_algo_bytes / st.raster_launches
            achieved = bytes_per_launch / avg_ns  # GB/s
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "kernel": "wr_raster_kernel<RGBA8>", "avg_launch_us": round(avg_ns / 1e3, 2),
                    "algo_bytes_per_launch": int(bytes_per_launch),
                    "launches_per_frame": st.raster_launches / nprof,
                    "raster_us_per_frame": round(st.raster_ns / nprof / 1e3, 2)}
            tr = pmc_traffic(args.workload, args.encoding)
            if tr:
                roof["traffic"], roof["traffic_source"] = tr

    if rank == 0:
        fps = args.steps / elapsed
        out = {
            "metric": "frames/sec + Mpixels/sec on wrench benchmarks, 4K target, 1/2/4/8 GPU",
            "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "mpixels_per_s": round(fps * frame_w * frame_h / 1e6, 1),
            "frame_latency_ms": round(float(np.mean(lat)), 4),
            "config": {"workload": f"{args.workload}: " + {
                "cfg2": "1000 overlapping translucent rects (ps_quad_textured + premultiplied-alpha blend), "
                        "3840x2160, 20 picture-cache tiles + composite, seed 2",
                "cfg5": "100k rects (50% opaque), 7680x4320, 72 tiles + composite, seed 5",
                "cfg4": "box-shadow-large.yaml at device_pixel_scale 2 (shadow 1840^2 px): cs_clip_rectangle mask -> "
                        "2 cs_scale halvings -> cs_blur V/H -> cs_clip_box_shadow x clip-out mask -> masked "
                        "brush_solid -> composite, 3840x2160",
                "cfg3": "text: 200 lines x 250 glyphs (ps_text_run, R8 glyph atlas 2048^2, premultiplied-alpha "
                        "blend), 3840x2160, 20 tiles + composite, seed 3",
                "cfg1": "16x16 opaque rect grid 1024x1024"}[args.workload],
                "encoding": args.encoding, "target": f"{frame_w}x{frame_h}",
                "parallelism": "single GPU" if world == 1 else
                f"tile rows sharded over {world} GPUs + RCCL all-gather of framebuffer strips"},
        }
        if roof:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(rec)
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
