"""Frame harness: records the C-ABI call stream of rendering a scene and
replays it natively -- the stand-in for `wrench png` / `wrench perf`
(wrench/src/main.rs:517-814, perf.rs:198-270).

  setup trace  CreateContext .. first frame .. Finish   (resource creation, untimed)
  frame trace  one steady-state frame .. Finish         (what `perf` times: the
               full tile-raster + composite stream is re-issued every iteration,
               see BASELINE.md §1 on why wrench's own loop would only re-composite)
  read trace   ReadPixels of the window (reftest.rs:306-319)
"""
import numpy as np
from . import glconst as G
from .glapi import GL
from .renderer import Renderer
from .trace import Trace, NativeReplayer, TAG_SCRATCH


class RecordedScene:
    def __init__(self, width, height, setup, frame, read, read_offset, stream=None):
        self.width, self.height = width, height
        self.setup, self.frame, self.read, self.read_offset = setup, frame, read, read_offset
        self.stream = stream or frame
        self.texture_ids = {}     # TextureRef.name -> backend texture id (deterministic across replays)
        self.tile_rects = {}      # TextureRef.name -> composite rect (x0,y0,x1,y1) in device px


def record_scene(recording_backend_path, frame):
    """Run `frame` twice through the Python Renderer mirror against
    `recording_backend_path`, capturing the call stream."""
    tr = Trace()
    gl = GL(recording_backend_path, tr)
    r = Renderer(gl, frame.width, frame.height)
    r.render(frame)
    r.finish()
    setup = tr.serialize()
    tr2 = Trace()
    tr2.scratch = 0
    gl.trace = tr2
    r.render(frame)
    stream_bytes = tr2.serialize()      # the frame without a trailing Finish (throughput loops)
    r.finish()
    frame_bytes = tr2.serialize()
    tr3 = Trace()
    gl.trace = tr3
    px = r.read_pixels()
    read_bytes = tr3.serialize()
    read_off = [v for (_, args) in tr3.calls for (tag, _, v) in args if tag == TAG_SCRATCH][-1]
    gl.trace = None
    rec = RecordedScene(frame.width, frame.height, setup, frame_bytes, read_bytes, read_off, stream_bytes)
    rec.texture_ids = {name: t.id for name, t in r.textures.items()}
    rec.tile_rects = {t.texture.name: t.rect for t in frame.composite_tiles}
    r.destroy()
    return rec, px


class ScenePlayer:
    """Replays a RecordedScene against a backend library natively."""

    def __init__(self, backend_path, scene):
        self.scene = scene
        self.rp = NativeReplayer(backend_path)
        self.rp.exec(scene.setup)

    def frames(self, warmup, iters):
        return self.rp.loop(self.scene.frame, warmup, iters)

    def stream(self, iters):
        """`iters` frames back to back, one Finish at the end; total wall ms."""
        return self.rp.stream(self.scene.stream, iters)

    def read_pixels(self):
        self.rp.exec(self.scene.read)
        n = self.scene.width * self.scene.height * 4
        buf = self.rp.scratch(self.scene.read_offset, n)
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.scene.height, self.scene.width, 4).copy()

    def symbol(self, name):
        return self.rp.symbol(name)


def render_direct(backend_path, frame, frames=1):
    """Render through the Python mirror without tracing; returns RGBA8 window
    pixels, or {texture name: pixels, "window": ...} when the frame asks for
    render-target readbacks (Frame.readback)."""
    gl = GL(backend_path)
    r = Renderer(gl, frame.width, frame.height)
    for _ in range(frames):
        r.render(frame)
    r.finish()
    err = gl.GetError()            # (libwrhip: GL_INVALID_OPERATION when a prim was on a path it cannot draw exactly)
    px = r.read_pixels()
    extra = {ref.name: r.device.read_texture(r.resolve(ref)) for ref in getattr(frame, "readback", [])}
    stats = gl.stats() if gl.is_wrhip else None
    if stats is not None:
        stats["gl_error"] = int(err)
    r.destroy()
    if extra:
        extra["window"] = px
        return extra, stats
    return px, stats


def render_pipelined(backend_path, frames):
    """Issue `frames` (same window size) back to back WITHOUT a Finish in between; after each
    frame the window is blitted (device-side, asynchronous) into a keeper texture.  One Finish at
    the end, then the keepers are read back: [BGRA8 pixels per frame].  Exercises the backend's
    cross-frame pipelining (uploads of frame k+1 against the raster work of frame k)."""
    gl = GL(backend_path)
    w, h = frames[0].width, frames[0].height
    r = Renderer(gl, w, h)
    d = r.device
    keepers = [d.create_texture(w, h, G.GL_RGBA8, render_target=True) for _ in frames]
    for f, k in zip(frames, keepers):
        r.render(f)
        gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, 0)
        gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, k.fbo)
        gl.BlitFramebuffer(0, 0, w, h, 0, 0, w, h, G.GL_COLOR_BUFFER_BIT, G.GL_NEAREST)
        gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, 0)
    r.finish()
    out = [d.read_texture(k) for k in keepers]
    r.destroy()
    return out
