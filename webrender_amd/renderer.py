"""Host-side mirror of the part of `webrender::Renderer` that turns a built
`Frame` into the GL call stream (webrender/src/renderer/mod.rs):

  bind_frame_data + gpu buffer textures    mod.rs:4418-4430, 4540-4558
  draw_frame pass / target loop            mod.rs:4525-4841
  draw_picture_cache_target                mod.rs:2669-2803
  draw_alpha_batch_container               mod.rs:2804-2969
  draw_instanced_batch                     mod.rs:2022-2065
  composite_frame / composite_simple       mod.rs:4843-4905, 3340-3484
  draw_tile_list                           mod.rs:3126-3334

The call order is the contract the backend must honour; this module is the
stand-in for the Rust caller (which cannot be built here) and is what both the
oracle and libwrhip are driven by in tests and bench.
"""
import numpy as np
from . import glconst as G
from .device import Device, ortho, SAMPLER_SLOTS
from .frame import TEX_W


class Renderer:
    def __init__(self, gl, width, height):
        self.gl = gl
        self.device = Device(gl)
        self.width, self.height = width, height
        self.device.init_default_framebuffer(width, height)
        self.textures = {}       # TextureRef.name -> device.Texture
        self.data_tex = {}       # sampler name -> (device.Texture, rows)
        self.frame_count = 0
        d = self.device
        # dummy 1x1 white texture bound for TextureSource::Invalid/Dummy
        # (renderer/mod.rs:1063-1090 dummy cache texture)
        self.dummy = d.create_texture(1, 1, G.GL_RGBA8)
        d.upload_texture(self.dummy, 0, 0, 1, 1, G.GL_BGRA, G.GL_UNSIGNED_BYTE,
                         np.array([255, 255, 255, 255], dtype=np.uint8))

    # ---- resources -------------------------------------------------------
    def resolve(self, ref):
        if ref is None:
            return self.dummy
        tex = self.textures.get(ref.name)
        if tex is None or (tex.width, tex.height, tex.format) != (ref.w, ref.h, ref.fmt):
            if tex is not None:
                self.device.delete_texture(tex)
            tex = self.device.create_texture(ref.w, ref.h, ref.fmt, ref.filter,
                                             ref.render_target, ref.with_depth)
            self.textures[ref.name] = tex
            if ref.pixels is not None:
                fmt = ref.upload_format or (G.GL_RED if ref.fmt == G.GL_R8 else G.GL_BGRA)
                self.device.upload_texture(tex, 0, 0, ref.w, ref.h, fmt,
                                           getattr(ref, "upload_type", None) or G.GL_UNSIGNED_BYTE, ref.pixels)
        return tex

    def _update_data_texture(self, sampler, store, fmt, upload_fmt, upload_ty, min_rows=1):
        d = self.device
        data = store.texture_data(min_rows)
        rows = data.shape[0]
        cur = self.data_tex.get(sampler)
        if cur is None or cur[1] < rows:
            if cur is not None:
                d.delete_texture(cur[0])
            tex = d.create_texture(TEX_W, rows, fmt, G.GL_NEAREST)
            self.data_tex[sampler] = (tex, rows)
        tex = self.data_tex[sampler][0]
        d.upload_texture(tex, 0, 0, TEX_W, rows, upload_fmt, upload_ty, data)
        d.bind_texture(SAMPLER_SLOTS[sampler], tex.id)

    def bind_frame_data(self, frame):
        F, I = (G.GL_RGBA32F, G.GL_RGBA, G.GL_FLOAT), (G.GL_RGBA32I, G.GL_RGBA_INTEGER, G.GL_INT)
        self._update_data_texture("sPrimitiveHeadersF", frame.prim_headers_f, *F)
        self._update_data_texture("sPrimitiveHeadersI", frame.prim_headers_i, *I)
        self._update_data_texture("sTransformPalette", frame.transforms, *F)
        self._update_data_texture("sRenderTasks", frame.render_tasks, *F)
        # GPU cache: persistent 1024 x >=20 RGBAF32 texture (gpu_cache.rs:46)
        self._update_gpu_cache(frame.gpu_cache, *F)

    def _update_gpu_cache(self, store, fmt, upload_fmt, upload_ty):
        """GpuCacheTexture::flush over the PixelBuffer bus (renderer/gpu_cache.rs:324-356; the bus a device without scatter shaders --
        swgl -- gets, :104-150): the rows are kept on the CPU side, a row remembers the span of blocks that changed since it was last
        uploaded (CacheRow::add_dirty, :51-54) and only that span of every dirty row goes to the texture; a frame that changed no block
        uploads nothing.  A texture that had to grow is uploaded whole (:239-249)."""
        d = self.device
        data = store.texture_data(20)
        rows = data.shape[0]
        cur = self.data_tex.get("sGpuCache")
        if cur is None or cur[1] < rows:
            if cur is not None:
                d.delete_texture(cur[0])
            tex = d.create_texture(TEX_W, rows, fmt, G.GL_NEAREST)
            self.data_tex["sGpuCache"] = (tex, rows)
            self._gpu_cache_cpu = None
        tex = self.data_tex["sGpuCache"][0]
        old = getattr(self, "_gpu_cache_cpu", None)
        if old is None or old.shape != data.shape:
            d.upload_texture(tex, 0, 0, TEX_W, rows, upload_fmt, upload_ty, data)
        else:
            changed = (old.view(np.uint32) != data.view(np.uint32)).any(axis=2)       # [rows, TEX_W] blocks (bit patterns: NaN payloads too)
            for r in np.nonzero(changed.any(axis=1))[0]:
                cols = np.nonzero(changed[r])[0]
                x0, x1 = int(cols[0]), int(cols[-1]) + 1
                d.upload_texture(tex, x0, int(r), x1 - x0, 1, upload_fmt, upload_ty, np.ascontiguousarray(data[r, x0:x1]))
        self._gpu_cache_cpu = data.copy()
        d.bind_texture(SAMPLER_SLOTS["sGpuCache"], tex.id)

    def _create_gpu_buffer_texture(self, sampler, store, fmt, upload_fmt, upload_ty):
        d = self.device
        data = store.texture_data()
        tex = d.create_texture(TEX_W, data.shape[0], fmt, G.GL_NEAREST)
        d.upload_texture(tex, 0, 0, TEX_W, data.shape[0], upload_fmt, upload_ty, data)
        d.bind_texture(SAMPLER_SLOTS[sampler], tex.id)
        return tex

    # ---- drawing -----------------------------------------------------------
    def _bind_step_textures(self, step):
        d = self.device
        rect = "TEXTURE_RECT" in step.shader       # sColor0-2 are sampler2DRect (external / IOSurface images)
        for slot in (0, 1, 2, 9):  # Color0-2 + ClipMask, mod.rs:2066-2100
            ref = step.textures.get(slot)
            d.bind_texture(slot, self.resolve(ref).id if ref is not None else self.dummy.id, rect and slot != 9)

    def _draw_step(self, step, projection):
        d = self.device
        desc = step.desc
        prog = d.create_program(step.shader, desc)
        vao = d.create_vao(desc)
        d.bind_program(prog, projection)
        self._bind_step_textures(step)
        d.draw_instanced_batch(vao, step.instances)

    def draw_picture_cache_target(self, target, valid_rect=None):
        """Renderer::draw_picture_cache_target + draw_alpha_batch_container (renderer/mod.rs:2669-2968).  `valid_rect`: the
        tile's valid rect in tile pixels (what the compositor shows of it: a tile at the edge of the window is only valid
        inside the window) -- the picture task's scissor rect is dirty rect & valid rect (picture.rs:5213-5217), so the
        batches are drawn scissored to it (mod.rs:2813-2820); the clear stays unscissored while dirty == valid (mod.rs:2701-2707)."""
        d = self.device
        tex = self.resolve(target.texture)
        d.bind_draw_target(tex.fbo_with_depth or tex.fbo, tex.width, tex.height)
        projection = ortho(0.0, tex.width, 0.0, tex.height)
        d.enable_depth_write()
        d.set_blend(False)
        d.clear_target(target.clear_color, 1.0, target.clear_rect)
        d.disable_depth_write()
        # draw_alpha_batch_container
        uses_scissor = valid_rect is not None and tuple(valid_rect) != (0, 0, tex.width, tex.height)
        if uses_scissor:
            self.gl.Enable(G.GL_SCISSOR_TEST)
            self.gl.SetScissor(valid_rect[0], valid_rect[1], valid_rect[2] - valid_rect[0], valid_rect[3] - valid_rect[1])
        if target.opaque:
            d.set_blend(False)
            d.enable_depth(G.GL_LEQUAL)
            d.enable_depth_write()
            for step in target.opaque:
                self._draw_step(step, projection)
            d.disable_depth_write()
        else:
            d.disable_depth()
        if target.alpha:
            d.set_blend(True)
            prev = None
            for step in target.alpha:
                if step.blend != prev:
                    d.set_blend_mode(step.blend)
                    prev = step.blend
                self._draw_step(step, projection)
            d.set_blend(False)
        d.disable_depth()
        if uses_scissor:
            self.gl.Disable(G.GL_SCISSOR_TEST)
        d.invalidate_depth_target()

    def draw_offscreen_target(self, target):
        """Alpha / colour / texture-cache targets: generic ordered steps
        (draw_alpha_target mod.rs:3754-3929, draw_texture_cache_target 3931-)."""
        d = self.device
        tex = self.resolve(target.texture)
        d.bind_draw_target(tex.fbo, tex.width, tex.height)
        projection = ortho(0.0, tex.width, 0.0, tex.height)
        d.disable_depth()
        d.disable_depth_write()
        d.set_blend(False)
        if target.clear_color is not None:
            d.clear_target(target.clear_color, None, target.clear_rect)
        blend_on, prev = False, None
        for step in target.steps:
            want = step.blend is not None
            if want != blend_on:
                d.set_blend(want)
                blend_on = want
            if want and step.blend != prev:
                d.set_blend_mode(step.blend)
                prev = step.blend
            self._draw_step(step, projection)
        if blend_on:
            d.set_blend(False)

    def composite_simple(self, frame):
        d = self.device
        d.bind_draw_target(0, frame.width, frame.height)
        d.disable_depth_write()
        d.disable_depth()
        # wrench: surface_origin_is_top_left == false -> flipped ortho (mod.rs:4861-4866)
        projection = ortho(0.0, frame.width, frame.height, 0.0)
        d.clear_target(frame.clear_color, None, None)
        opaque = [t for t in frame.composite_tiles if t.opaque]
        alpha = [t for t in frame.composite_tiles if not t.opaque]

        rect = getattr(frame, "texture_rect", False)     # scenes.texture_rect: the surfaces are rectangle textures (texel uv)

        def draw_tile_list(tiles):
            # one batch per texture / program change (mod.rs:3260-3334)
            batch, cur = [], None
            def flush():
                if not batch:
                    return
                key = "composite FAST_PATH,TEXTURE_2D" if cur[1] else ("composite TEXTURE_2D,YUV" if cur[2] else "composite TEXTURE_2D")
                if rect:
                    key = key.replace("TEXTURE_2D", "TEXTURE_RECT")
                prog = d.create_program(key, "COMPOSITE")
                vao = d.create_vao("COMPOSITE")
                d.bind_program(prog, projection)
                d.bind_texture(0, self.resolve(cur[0]).id, rect)
                if cur[2]:            # the chroma planes of a YUV surface (sColor1, sColor2)
                    for slot, ref in enumerate(cur[2][1:], start=1):
                        d.bind_texture(slot, self.resolve(ref).id, rect)
                d.draw_instanced_batch(vao, np.stack(batch))
            for t in tiles:
                k = (t.texture, t.fast, tuple(t.yuv["planes"]) if t.yuv else None)
                if cur is not None and (k[0] is not cur[0] or k[1] != cur[1] or k[2] != cur[2]):
                    flush()
                    batch.clear()
                cur = k
                if t.yuv:
                    batch.append(frame.composite_instance(t.rect, t.clip_rect, flip=t.flip, yuv=t.yuv))
                elif t.fast and not rect:
                    batch.append(frame.composite_instance(t.rect, t.clip_rect))
                else:
                    uv = t.uv_rect or (0.0, 0.0, float(t.texture.w), float(t.texture.h))
                    batch.append(frame.composite_instance(t.rect, t.clip_rect, t.color or (1, 1, 1, 1), uv,
                                                          uv_type=1, flip=t.flip))
            flush()

        if opaque:
            d.set_blend(False)
            draw_tile_list(opaque)
        if alpha:
            d.set_blend(True)
            d.set_blend_mode("PremultipliedAlpha")
            draw_tile_list(list(reversed(alpha)))
            d.set_blend(False)

    def render(self, frame):
        """Renderer::render_impl / draw_frame for one frame."""
        d = self.device
        for ref in frame.static_textures:
            self.resolve(ref)
        d.disable_depth_write()
        d.set_blend(False)
        self.bind_frame_data(frame)
        gbf = self._create_gpu_buffer_texture("sGpuBufferF", frame.gpu_buffer_f,
                                              G.GL_RGBA32F, G.GL_RGBA, G.GL_FLOAT)
        gbi = self._create_gpu_buffer_texture("sGpuBufferI", frame.gpu_buffer_i,
                                              G.GL_RGBA32I, G.GL_RGBA_INTEGER, G.GL_INT)
        # valid rect of every composited tile, in tile pixels: clip rect & tile rect, relative to the tile origin
        valid = {}
        for t in frame.composite_tiles:
            r_, c_ = t.rect, t.clip_rect
            if t.uv_rect is None and all(float(v).is_integer() for v in tuple(r_) + tuple(c_)):
                valid[t.texture.name] = (int(max(c_[0], r_[0]) - r_[0]), int(max(c_[1], r_[1]) - r_[1]),
                                         int(min(c_[2], r_[2]) - r_[0]), int(min(c_[3], r_[3]) - r_[1]))
        for targets in frame.passes:
            for target in targets:
                if target.kind == "picture_tile":
                    self.draw_picture_cache_target(target, valid.get(target.texture.name))
                else:
                    self.draw_offscreen_target(target)
        self.composite_simple(frame)
        # end_frame (mod.rs:4826-4841)
        d.delete_texture(gbf)
        d.delete_texture(gbi)
        self.frame_count += 1

    def read_pixels(self):
        """wrench reftest readback: glReadPixels(RGBA) of the whole window
        (reftest.rs:306-319); rows are bottom-up in GL terms, returned as-is."""
        return self.device.read_pixels_rgba8(0, 0, self.width, self.height)

    def finish(self):
        self.gl.Finish()

    def destroy(self):
        self.device.destroy()
