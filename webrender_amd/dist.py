"""Multi-GPU: screen-space strip sharding of one frame + RCCL all-gather of the
window framebuffer.

One process per GPU.  Every rank replays the *same* call stream, but tells its
backend which pixel rows of each render target it owns (`WrhipSetTargetRows`):
rank r owns the screen rows [Y0_r, Y1_r) -- multiples of 64 so they coincide
with raster bins -- of every picture-cache tile and of the window.  Rows outside
are neither rasterised nor stored, and a tile with NO row in the rank's strip is
dropped when its draws are recorded: its instance bytes are not snapshotted,
staged or uploaded and its prims never reach the setup stage, so the per-rank
host and setup work shrinks with the number of ranks, not only the raster work
(libwrhip: an empty row range makes DrawElementsInstanced / Clear on that target
no-ops).  Off-screen mask / blur targets are left unrestricted (replicated on
every rank: they are tiny compared with tiles and would otherwise have to be
exchanged between passes).

The only collective is the all-gather that reassembles the framebuffer: each
rank contributes its strip (equal-sized, padded) and receives the others, over
xGMI with RCCL ("nccl" backend of torch.distributed) or gloo on CPU tests.
"""
import ctypes as C
import numpy as np


def strip_rows(height, rank, world, align=64):
    """Screen rows [y0, y1) owned by `rank`, and the (padded) strip height every rank contributes to the all-gather.
    The `align`-row groups (raster bin rows) are dealt out as evenly as they go: rank r gets groups
    [r * G // world, (r + 1) * G // world), so no rank is left without work while another holds two extra groups."""
    groups = (height + align - 1) // align
    g0, g1 = rank * groups // world, (rank + 1) * groups // world
    y0, y1 = min(height, g0 * align), min(height, g1 * align)
    per = max((r + 1) * groups // world - r * groups // world for r in range(world))
    return y0, y1, per * align


def strip_tile_filter(height, rank, world):
    """tile_filter(tx, ty) of the scene builders for `rank`: the picture-cache tiles with a row in the rank's strip.  A rank BUILDS
    its frame with it -- the frame builder of a strip-sharded WebRender only visits the tiles it draws (picture/tile.rs culls per
    tile anyway) -- so the data textures it uploads (GPU buffers, prim headers: one entry per prim per visited tile) hold the
    rank's share, not the frame's: SURVEY section 8e's per-rank H2D.  The kept prims keep their z ids and their order."""
    from .scenes import TILE_H
    sy0, sy1, _ = strip_rows(height, rank, world)

    def keep(tx, ty):
        return ty * TILE_H < sy1 and (ty + 1) * TILE_H > sy0
    keep.rows = (sy0, sy1)          # ... and within a kept tile, the prims with a row in the strip (the builders cull against it)
    return keep


def build_rank_frame(workload, encoding, rank, world):
    """The rank's frame of a named workload: built for its tiles only where the workload's builder takes a tile filter and the
    frame's height is known up front (BASELINE configs 1 / 2 / 5: tile grids of rects), the whole frame otherwise (the row
    restriction still drops the other tiles' draws when they are recorded; the data textures are then the whole frame's)."""
    from .scenes import make_workload
    heights = {"cfg1": 1024, "cfg2": 2160, "cfg5": 4320}
    if world > 1 and workload in heights:
        return make_workload(workload, encoding=encoding, tile_filter=strip_tile_filter(heights[workload], rank, world))
    return make_workload(workload, encoding=encoding)


def target_rows_for_rank(rec, rank, world):
    """{texture name: (y0, y1)} in texture rows for this rank, plus the window
    strip in framebuffer rows.  The window projection is y-flipped
    (renderer/mod.rs:4861-4866): screen row y is framebuffer row H-1-y."""
    H = rec.height
    sy0, sy1, _ = strip_rows(H, rank, world)
    rows = {}
    for name, (x0, y0, x1, y1) in rec.tile_rects.items():
        ty0 = int(max(sy0 - y0, 0))
        ty1 = int(min(sy1 - y0, y1 - y0))
        rows[name] = (ty0, ty1) if ty1 > ty0 else (0, -1)   # (0,-1): owns nothing
    fb = (H - sy1, H - sy0)
    return rows, fb


class DeviceArray:
    """Minimal __cuda_array_interface__ wrapper so torch can view HBM owned by libwrhip."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {
            "shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 3, "strides": None}


class ShardedFramePlayer:
    """Renders `workload` with rows sharded over `world` ranks and all-gathers
    the window.  Interface mirrors harness.ScenePlayer (frames / stream)."""

    def __init__(self, lib, workload, encoding, rank, world, device="cuda", frame=None, gather="all", native=None, shm_name=None):
        """gather: "all" -- every rank receives the whole window (all-gather; north_star's reassembly step) -- or "root" --
        only rank 0, the presenting GPU, does ("or gather to the presenting GPU if only one consumer").
        native: "rccl" -- the per-frame loop runs in native code (csrc/wr_replay.c wr_shard_stream: replay, WrhipFlush, grouped
        ncclSend / ncclRecv on the backend's stream straight between the ranks' windows, no Python and no staging copy per
        frame) -- or "shm" -- the same loop with the CPU stand-in transport of the tests (shared-memory window).  None: the
        torch.distributed loop below (what the native loop is tested against)."""
        import torch
        import torch.distributed as dist
        from .harness import record_scene, ScenePlayer
        self.torch, self.dist = torch, dist
        self.rank, self.world = rank, world
        self.gather = gather
        if frame is None:
            frame = build_rank_frame(workload, encoding, rank, world)
        elif callable(frame):                 # frame(tile_filter_for(height)): the caller's builder, for this rank's tiles
            frame = frame(lambda height: strip_tile_filter(height, rank, world))
        self.width, self.height = frame.width, frame.height
        rec, _ = record_scene(lib, frame)
        self.rec = rec
        self.player = ScenePlayer(lib, rec)
        sym = self.player.symbol
        set_rows = C.CFUNCTYPE(None, C.c_uint32, C.c_int32, C.c_int32)(sym("WrhipSetTargetRows"))
        fb_tex = C.CFUNCTYPE(C.c_uint32, C.c_uint32)(sym("WrhipGetFramebufferTexture"))(0)
        get_ptr = C.CFUNCTYPE(C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p)(sym("WrhipGetTextureDevicePtr"))
        rows, fb_rows = target_rows_for_rank(rec, rank, world)
        for name, (y0, y1) in rows.items():
            tid = rec.texture_ids[name]
            if y1 > y0:
                set_rows(tid, y0, y1)
            else:
                set_rows(tid, 1, 0)                         # owns no row of this tile: its draws are dropped at record time
        if fb_rows[1] > fb_rows[0]:
            set_rows(fb_tex, fb_rows[0], fb_rows[1])
        else:
            set_rows(fb_tex, 1, 0)
        self.fb_rows = fb_rows
        self._flush = C.CFUNCTYPE(None)(sym("WrhipFlush"))
        self._get_stream = C.CFUNCTYPE(C.c_void_p)(sym("WrhipGetStream"))
        _, _, self.strip = strip_rows(self.height, rank, world)
        self.row_bytes = self.width * 4
        self.device = device
        if device == "cuda":
            ptr = get_ptr(fb_tex, None, None, None)
            self.fb = torch.as_tensor(DeviceArray(ptr, self.height * self.row_bytes), device="cuda")
            # torch enqueues the strip copy and the all-gather behind the backend's own work on the
            # backend's HIP stream: no host sync per frame, frames stay pipelined
            sp = self._get_stream()
            self.ext_stream = torch.cuda.ExternalStream(int(sp)) if sp else None
        else:
            self.fb = None       # CPU/gloo test path reads the strip back through ReadPixels
            self.ext_stream = None
        self.native = None
        if native:
            self._open_native(native, lib, get_ptr, fb_tex, shm_name)
            return
        chunk = self.strip * self.row_bytes
        self.send = [torch.zeros(chunk, dtype=torch.uint8, device=device) for _ in range(2)]
        # (gather to rank 0: only the presenting rank holds the assembled window)
        self.gathered = [torch.zeros(chunk * world if (gather == "all" or rank == 0) else 0, dtype=torch.uint8, device=device) for _ in range(2)]
        self.pending = None
        self.k = 0

    # -- the native loop ----------------------------------------------------------
    def _open_native(self, kind, lib, get_ptr, fb_tex, shm_name):
        import os
        rl = self.player.rp.lib
        rl.wr_shard_open_rccl.restype = C.c_void_p
        rl.wr_shard_open_rccl.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
        rl.wr_shard_open_shm.restype = C.c_void_p
        rl.wr_shard_open_shm.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_size_t]
        rl.wr_shard_set_window.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        rl.wr_shard_stream.restype = C.c_int
        rl.wr_shard_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        rl.wr_shard_stream2.restype = C.c_int
        rl.wr_shard_stream2.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        rl.wr_shard_unique_id.argtypes = [C.c_char_p, C.c_char_p]
        rl.wr_shard_close.argtypes = [C.c_void_p, C.c_char_p]
        mode = 1 if self.gather == "all" else 0
        h = self.player.rp.h
        if kind == "rccl":
            librccl = os.path.join(os.path.dirname(self.torch.__file__), "lib", "librccl.so").encode()
            ids = [None, 0]
            if self.world > 1:
                if self.rank == 0:
                    buf = C.create_string_buffer(128)
                    rc = rl.wr_shard_unique_id(librccl, buf)        # (1: librccl not found, 2: no ncclGetUniqueId)
                    ids = [buf.raw if rc == 0 else None, rc]
                # (setup, once: the id of the communicator the native loop owns -- and rank 0's status WITH it, so that a rank 0 that
                # cannot get an id does not leave the others waiting in this broadcast: every rank raises the same RuntimeError)
                self.dist.broadcast_object_list(ids, src=0)
                if ids[1] != 0:
                    raise RuntimeError(f"wr_shard_unique_id failed on rank 0 ({ids[1]})")
            self.native = rl.wr_shard_open_rccl(h, librccl, self.rank, self.world, mode, ids[0] or b"\0" * 128)
        else:
            self.shm_name = shm_name.encode()
            if self.rank == 0:                  # a stale segment of a crashed run (non-zero arrival counter) must not be picked up
                rl.wr_shard_shm_reset.argtypes = [C.c_char_p]
                rl.wr_shard_shm_reset(self.shm_name)
            if self.world > 1:
                self.dist.barrier()
            self.native = rl.wr_shard_open_shm(h, self.shm_name, self.rank, self.world, mode, self.height * self.row_bytes)
        if self.world > 1:
            # every rank takes the same transport: one that could not open its side tells the others before anybody enters a collective
            oks = [None] * self.world
            self.dist.all_gather_object(oks, bool(self.native))
            if not all(oks):
                if self.native:
                    rl.wr_shard_close(self.native, None)
                    self.native = None
                raise RuntimeError(f"wr_shard_open failed on rank(s) {[r for r, ok in enumerate(oks) if not ok]}")
        if not self.native:
            raise RuntimeError("wr_shard_open failed")
        rl.wr_shard_probe.restype = C.c_int
        rl.wr_shard_probe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        rl.wr_shard_set_pipelined.argtypes = [C.c_void_p, C.c_int]
        self._probed = False
        rows = []
        for r in range(self.world):
            sy0, sy1, _ = strip_rows(self.height, r, self.world)
            rows += [self.height - sy1, self.height - sy0]
        ptr = get_ptr(fb_tex, None, None, None)
        rl.wr_shard_set_window(self.native, ptr, self.height, self.row_bytes, (C.c_int * len(rows))(*rows))
        self._rl = rl

    def _probe(self, t):
        """Once per player, before its first streamed frame: the pipelined exchange order of the native loop is decided by ALL ranks
        together (flush counts differ per rank; ranks in different orders would combine strips of different frames)."""
        if self._probed:
            return
        self._probed = True
        n = self._rl.wr_shard_probe(self.native, t, len(t))
        ns = [n]
        if self.world > 1:
            ns = [None] * self.world
            self.dist.all_gather_object(ns, int(n))
        if min(ns) < 0:
            raise RuntimeError(f"wr_shard_probe failed ({ns})")
        self._rl.wr_shard_set_pipelined(self.native, 1 if max(ns) == 1 else 0)

    def _native_stream(self, iters):
        ms = C.c_double(0.0)
        t = self.rec.stream
        self._probe(t)
        rc = self._rl.wr_shard_stream(self.native, t, len(t), iters, C.byref(ms))
        if rc != 0:
            raise RuntimeError(f"wr_shard_stream failed ({rc})")
        return ms.value

    def stream_alternating(self, other_stream, iters):
        """native loop only: `iters` frames alternating between this player's frame trace (even frames) and `other_stream` -- the
        frame trace of another scene recorded with the same resources (same window, same texture ids) -- exchanged after every one"""
        ms = C.c_double(0.0)
        t = self.rec.stream
        self._probe(t)
        rc = self._rl.wr_shard_stream2(self.native, t, len(t), other_stream, len(other_stream), iters, C.byref(ms))
        if rc != 0:
            raise RuntimeError(f"wr_shard_stream2 failed ({rc})")
        return ms.value

    def _collect(self, i):
        """Start the collective that reassembles the window from every rank's strip (asynchronous)."""
        if self.gather == "all":
            return self.dist.all_gather_into_tensor(self.gathered[i], self.send[i], async_op=True)
        chunk = self.send[i].numel()
        outs = [self.gathered[i][r * chunk:(r + 1) * chunk] for r in range(self.world)] if self.rank == 0 else None
        return self.dist.gather(self.send[i], outs, dst=0, async_op=True)

    # -- one frame: render own strips, then contribute to the all-gather ------
    def _frame(self):
        i = self.k & 1
        self.k += 1
        y0, y1 = self.fb_rows
        n = max(0, y1 - y0) * self.row_bytes
        if self.device == "cuda" and self.ext_stream is not None:
            self.player.rp.exec(self.rec.stream)     # the frame without a Finish
            self._flush()                            # ... submitted (composite included), not waited for
            with self.torch.cuda.stream(self.ext_stream):
                if n:
                    self.send[i][:n].copy_(self.fb[y0 * self.row_bytes:y0 * self.row_bytes + n])
                if self.pending is not None:
                    self.pending.wait()              # stream-side wait: the send buffer of two frames ago is free again
                self.pending = self._collect(i)
            return
        self.player.rp.exec(self.rec.frame)          # includes Finish(): strip is in HBM
        if self.device == "cuda":
            if n:
                self.send[i][:n].copy_(self.fb[y0 * self.row_bytes:y0 * self.row_bytes + n])
        else:
            px = self.player.read_pixels()             # RGBA bytes; strips are byte-exact either way
            if n:
                self.send[i][:n] = self.torch.from_numpy(px[y0:y1].reshape(-1).copy())
        if self.pending is not None:
            self.pending.wait()
        self.pending = self._collect(i)

    def frames(self, warmup, iters):
        import time
        if self.native:
            self._native_stream(warmup) if warmup else None
            return np.array([self._native_stream(1) for _ in range(iters)])
        out = []
        for it in range(warmup + iters):
            t0 = time.perf_counter()
            self._frame()
            if it >= warmup:
                out.append((time.perf_counter() - t0) * 1e3)
        self._drain()
        return np.array(out)

    def stream(self, iters):
        if self.native:
            self._native_stream(iters)
            return
        for _ in range(iters):
            self._frame()
        self._drain()

    def _drain(self):
        if self.native:
            return
        if self.pending is not None:
            if self.ext_stream is not None:
                with self.torch.cuda.stream(self.ext_stream):
                    self.pending.wait()
            else:
                self.pending.wait()
            self.pending = None
        if self.device == "cuda":
            self.torch.cuda.synchronize()

    def assembled(self):
        """The reassembled window (uint8 [H, W, 4]) from the last gather, in
        framebuffer row order (bottom-up, as ReadPixels returns it)."""
        if self.native:       # the strips landed in this rank's own window: an ordinary readback
            return self.player.read_pixels()
        self._drain()
        g = self.gathered[(self.k - 1) & 1].cpu().numpy()
        chunk = self.strip * self.row_bytes
        out = np.zeros((self.height, self.width, 4), np.uint8)
        for r in range(self.world):
            sy0, sy1, _ = strip_rows(self.height, r, self.world)
            if sy1 <= sy0:
                continue
            f0, f1 = self.height - sy1, self.height - sy0
            out[f0:f1] = g[r * chunk:r * chunk + (f1 - f0) * self.row_bytes].reshape(f1 - f0, self.width, 4)
        return out

    def symbol(self, name):
        return self.player.symbol(name)

    def close(self):
        if self.native:
            self._rl.wr_shard_close(self.native, getattr(self, "shm_name", None) if self.rank == 0 else None)
            self.native = None
