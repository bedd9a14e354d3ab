"""Drives a backend (the reference's swgl or libwrhip) through the parts of the 99-function ABI that the frame
scenes never reach -- queries, pixel-buffer-object uploads and readbacks, buffer mapping, TexImage2D, texture copies,
clears of textures / rects, scaled and flipped blits, externally backed textures, locked resources, a depth buffer that
has to survive a flush in the middle of a target, overlapping uploads -- and returns everything observable as a dict of
numpy arrays / ints / bytes, so that two backends can be compared entry by entry (swgl_fns.rs:23-322;
gl.cc:1383-1395, 1518-1581, 1690-1828, 2007-2047, 2322-2360, 2370-2532, 2556-2659; composite.h:432-593)."""
import ctypes as C
import numpy as np
from webrender_amd import glconst as G, scenes
from webrender_amd.glapi import GL
from webrender_amd.renderer import Renderer
from webrender_amd.device import ortho


def _read_tex(d, tex):
    return d.read_texture(tex).copy()


def _ptr_bytes(addr, n):
    return np.frombuffer(C.string_at(addr, n), dtype=np.uint8).copy()


def run(backend_path):
    out = {}
    gl = GL(backend_path)
    W, H = 512, 512
    r = Renderer(gl, W, H)
    d = r.device
    rng = np.random.default_rng(2024)

    # ---- identity / limits the caller reads at start-up --------------------------------------------------
    out["renderer_is_software_webrender"] = int(gl.GetString(G.GL_RENDERER).startswith(b"Software WebRender"))
    out["version"] = gl.GetString(G.GL_VERSION)
    out["glsl"] = gl.GetString(G.GL_SHADING_LANGUAGE_VERSION)
    out["vendor"] = gl.GetString(G.GL_VENDOR)
    ints = {}
    for name in ("GL_MAX_TEXTURE_SIZE", "GL_MAX_TEXTURE_IMAGE_UNITS", "GL_NUM_EXTENSIONS", "GL_MAJOR_VERSION", "GL_MINOR_VERSION",
                 "GL_MAX_ARRAY_TEXTURE_LAYERS", "GL_DRAW_FRAMEBUFFER_BINDING", "GL_READ_FRAMEBUFFER_BINDING"):
        v = C.c_int32(-1)
        gl.GetIntegerv(getattr(G, name), v)
        ints[name] = v.value
    out["integers"] = repr(sorted(ints.items()))
    out["extensions"] = repr(sorted(gl.GetStringi(G.GL_EXTENSIONS, i) for i in range(ints["GL_NUM_EXTENSIONS"])))
    b = C.c_uint8(7)
    gl.DepthMask(0); gl.GetBooleanv(G.GL_DEPTH_WRITEMASK, b); out["depth_writemask_0"] = b.value
    gl.DepthMask(1); gl.GetBooleanv(G.GL_DEPTH_WRITEMASK, b); out["depth_writemask_1"] = b.value

    # ---- queries around whole frames ------------------------------------------------------------------
    def samples_of(frame):
        q = gl.gen("GenQueries")
        gl.BeginQuery(G.GL_SAMPLES_PASSED, q)
        r.render(frame)
        gl.EndQuery(G.GL_SAMPLES_PASSED)
        v = C.c_uint64(0)
        gl.GetQueryObjectui64v(q, G.GL_QUERY_RESULT, v)
        gl.DeleteQuery(q)
        return int(v.value)
    small = dict(width=W, height=H)
    out["samples_rects"] = samples_of(scenes.cfg2_overlapping_rects(n=40, seed=5, fractional=True, **small))
    out["samples_rects_aa_brush"] = samples_of(scenes.cfg2_overlapping_rects(n=30, seed=6, fractional=True, encoding="brush", aa_edges=15, **small))
    out["samples_rotated"] = samples_of(scenes.rotated_rects(n=20, seed=7, **small))
    out["samples_occluded_images"] = samples_of(scenes.add_occluders(scenes.image_grid(n=30, seed=8, **small), n=12, zmax=34, seed=3))
    out["samples_text"] = samples_of(scenes.cfg3_text(lines=6, glyphs_per_line=20, run_len=10, **small))
    q2 = gl.gen("GenQueries")
    gl.BeginQuery(G.GL_SAMPLES_PASSED, q2); gl.EndQuery(G.GL_SAMPLES_PASSED)
    v = C.c_uint64(99); gl.GetQueryObjectui64v(q2, G.GL_QUERY_RESULT, v)
    out["samples_empty_query"] = int(v.value)
    qt = gl.gen("GenQueries")
    gl.BeginQuery(G.GL_TIME_ELAPSED, qt)
    r.render(scenes.cfg2_overlapping_rects(n=40, seed=5, **small))
    gl.EndQuery(G.GL_TIME_ELAPSED)
    v = C.c_uint64(0); gl.GetQueryObjectui64v(qt, G.GL_QUERY_RESULT, v)
    out["time_elapsed_positive"] = int(v.value > 0)
    # seventy sample queries whose results are only read at the end (more than the backend has device slots for: a slot is
    # handed on only after its previous owner's count was collected), and an id that counted samples reused as a timer
    qs = []
    for k in range(70):
        q = gl.gen("GenQueries")
        gl.BeginQuery(G.GL_SAMPLES_PASSED, q)
        r.render(scenes.cfg2_overlapping_rects(n=1 + k % 5, seed=300 + k, fractional=True, **small))
        gl.EndQuery(G.GL_SAMPLES_PASSED)
        qs.append(q)
    late = []
    for q in qs:
        v = C.c_uint64(0); gl.GetQueryObjectui64v(q, G.GL_QUERY_RESULT, v)
        late.append(int(v.value))
    out["samples_late_reads"] = repr(late)
    gl.BeginQuery(G.GL_TIME_ELAPSED, qs[3])
    r.render(scenes.cfg2_overlapping_rects(n=3, seed=5, **small))
    gl.EndQuery(G.GL_TIME_ELAPSED)
    v = C.c_uint64(0); gl.GetQueryObjectui64v(qs[3], G.GL_QUERY_RESULT, v)
    out["reused_query_is_a_timer"] = int(v.value > 0 and v.value != late[3])
    for q in qs:
        gl.DeleteQuery(q)
    r.finish()
    out["window_after_queries"] = r.read_pixels().copy()

    # ---- texture uploads: TexImage2D, PBO, mapped buffers, row length, RGBA swizzle, overlapping writes ----------
    tw, th = 64, 48
    t1 = d.create_texture(tw, th, G.GL_RGBA8, render_target=True)
    img = rng.integers(0, 256, size=(th, tw, 4), dtype=np.uint8)
    gl.ActiveTexture(G.GL_TEXTURE0); gl.BindTexture(G.GL_TEXTURE_2D, t1.id)
    gl.TexImage2D(G.GL_TEXTURE_2D, 0, G.GL_RGBA8, tw, th, 0, G.GL_BGRA, G.GL_UNSIGNED_BYTE, img)
    out["teximage2d"] = _read_tex(d, t1)
    pbo = gl.gen("GenBuffers")
    sub = rng.integers(0, 256, size=(20, 24, 4), dtype=np.uint8)
    gl.BindBuffer(G.GL_PIXEL_UNPACK_BUFFER, pbo)
    gl.BufferData(G.GL_PIXEL_UNPACK_BUFFER, sub.nbytes + 64, None, G.GL_STREAM_DRAW)
    gl.BufferSubData(G.GL_PIXEL_UNPACK_BUFFER, 64, sub.nbytes, sub)
    gl.BindTexture(G.GL_TEXTURE_2D, t1.id)
    gl.TexSubImage2D(G.GL_TEXTURE_2D, 0, 5, 7, 24, 20, G.GL_BGRA, G.GL_UNSIGNED_BYTE, 64)         # offset into the PBO
    out["pbo_upload"] = _read_tex(d, t1)
    p = gl.MapBufferRange(G.GL_PIXEL_UNPACK_BUFFER, 64, 16 * 4 * 4, 0x0002)                    # GL_MAP_WRITE_BIT
    patch = rng.integers(0, 256, size=(4, 16, 4), dtype=np.uint8)
    C.memmove(p, patch.ctypes.data, patch.nbytes)
    out["unmap_ok"] = int(gl.UnmapBuffer(G.GL_PIXEL_UNPACK_BUFFER))
    gl.TexSubImage2D(G.GL_TEXTURE_2D, 0, 40, 40, 16, 4, G.GL_RGBA, G.GL_UNSIGNED_BYTE, 64)         # GL_RGBA: swizzled on the way in
    out["mapped_upload_rgba"] = _read_tex(d, t1)
    out["map_out_of_range_null"] = int(not gl.MapBufferRange(G.GL_PIXEL_UNPACK_BUFFER, sub.nbytes, 4096, 0x0002))
    p2 = gl.MapBuffer(G.GL_PIXEL_UNPACK_BUFFER, G.GL_READ_ONLY)
    out["mapbuffer_contents"] = _ptr_bytes(p2 + 64, 64)
    gl.UnmapBuffer(G.GL_PIXEL_UNPACK_BUFFER)
    gl.BindBuffer(G.GL_PIXEL_UNPACK_BUFFER, 0)
    wide = rng.integers(0, 256, size=(10, 40, 4), dtype=np.uint8)
    gl.PixelStorei(G.GL_UNPACK_ROW_LENGTH, 40)
    gl.TexSubImage2D(G.GL_TEXTURE_2D, 0, 0, 30, 12, 10, G.GL_BGRA, G.GL_UNSIGNED_BYTE, wide)      # 12 of 40 texels per row
    gl.PixelStorei(G.GL_UNPACK_ROW_LENGTH, 0)
    out["row_length_upload"] = _read_tex(d, t1)
    a = rng.integers(0, 256, size=(16, 16, 4), dtype=np.uint8)
    b2 = rng.integers(0, 256, size=(12, 12, 4), dtype=np.uint8)
    gl.TexSubImage2D(G.GL_TEXTURE_2D, 0, 10, 10, 16, 16, G.GL_BGRA, G.GL_UNSIGNED_BYTE, a)
    gl.TexSubImage2D(G.GL_TEXTURE_2D, 0, 14, 12, 12, 12, G.GL_BGRA, G.GL_UNSIGNED_BYTE, b2)        # overlaps the previous write: last wins
    gl.TexSubImage2D(G.GL_TEXTURE_2D, 0, 10, 10, 4, 4, G.GL_BGRA, G.GL_UNSIGNED_BYTE, b2)
    out["overlapping_uploads"] = _read_tex(d, t1)
    # uploads of a megabyte and more (libwrhip splits the staging copy over helper threads and sizes the scatter launch by the
    # segment), BGRA rows, RGBA rows that are swizzled on the way in, and a row-length upload into the middle
    tbig = d.create_texture(1024, 600, G.GL_RGBA8, render_target=True)
    gl.ActiveTexture(G.GL_TEXTURE0); gl.BindTexture(G.GL_TEXTURE_2D, tbig.id)
    bigpx = rng.integers(0, 256, size=(600, 1024, 4), dtype=np.uint8)
    gl.TexSubImage2D(G.GL_TEXTURE_2D, 0, 0, 0, 1024, 600, G.GL_BGRA, G.GL_UNSIGNED_BYTE, bigpx)
    out["big_upload_bgra"] = _read_tex(d, tbig)
    bigpx2 = rng.integers(0, 256, size=(520, 1024, 4), dtype=np.uint8)
    gl.TexSubImage2D(G.GL_TEXTURE_2D, 0, 0, 40, 1024, 520, G.GL_RGBA, G.GL_UNSIGNED_BYTE, bigpx2)
    out["big_upload_rgba"] = _read_tex(d, tbig)
    gl.PixelStorei(G.GL_UNPACK_ROW_LENGTH, 1024)
    gl.TexSubImage2D(G.GL_TEXTURE_2D, 0, 100, 10, 800, 500, G.GL_BGRA, G.GL_UNSIGNED_BYTE, bigpx)
    gl.PixelStorei(G.GL_UNPACK_ROW_LENGTH, 0)
    out["big_upload_row_length"] = _read_tex(d, tbig)
    gl.BindTexture(G.GL_TEXTURE_2D, t1.id)
    t_r8 = d.create_texture(40, 24, G.GL_R8, render_target=True)
    r8 = rng.integers(0, 256, size=(24, 40), dtype=np.uint8)
    gl.BindTexture(G.GL_TEXTURE_2D, t_r8.id)
    gl.TexSubImage2D(G.GL_TEXTURE_2D, 0, 0, 0, 40, 24, G.GL_RED, G.GL_UNSIGNED_BYTE, r8)
    out["r8_upload"] = _read_tex(d, t_r8)
    gl.GenerateMipmap(G.GL_TEXTURE_2D)

    # ---- clears of textures and rects -------------------------------------------------------------------
    col = (C.c_float * 4)(0.25, 0.5, 0.75, 1.0)
    gl.ClearTexSubImage(t1.id, 0, 8, 4, 0, 20, 10, 1, G.GL_RGBA, G.GL_FLOAT, col)
    ub = (C.c_uint8 * 4)(10, 200, 30, 128)
    gl.ClearTexSubImage(t1.id, 0, 30, 20, 0, 9, 9, 1, G.GL_RGBA, G.GL_UNSIGNED_BYTE, ub)
    gl.ClearTexSubImage(t1.id, 0, 50, 2, 0, 10, 5, 1, G.GL_RG, G.GL_UNSIGNED_BYTE, ub)
    out["clear_tex_sub_image"] = _read_tex(d, t1)
    gl.ClearColorRect(t1.fbo, 2, 30, 30, 12, 0.9, 0.1, 0.4, 0.6)
    gl.ClearColorRect(t1.fbo, 50, 40, 100, 100, 0.0, 1.0, 0.0, 1.0)          # clipped to the texture
    out["clear_color_rect"] = _read_tex(d, t1)
    red = (C.c_float * 4)(0.6, 0.0, 0.0, 0.0)
    gl.ClearTexSubImage(t_r8.id, 0, 3, 3, 0, 11, 7, 1, G.GL_RED, G.GL_FLOAT, red)
    out["clear_r8"] = _read_tex(d, t_r8)
    t2 = d.create_texture(tw, th, G.GL_RGBA8, render_target=True)
    gl.ClearTexImage(t2.id, 0, G.GL_RGBA, G.GL_FLOAT, col)
    out["clear_tex_image"] = _read_tex(d, t2)

    # ---- copies -------------------------------------------------------------------------------------------
    gl.CopyImageSubData(t1.id, G.GL_TEXTURE_2D, 0, 4, 6, 0, t2.id, G.GL_TEXTURE_2D, 0, 20, 10, 0, 30, 25, 1)
    out["copy_image_sub_data"] = _read_tex(d, t2)
    gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, t1.fbo)
    gl.BindTexture(G.GL_TEXTURE_2D, t2.id)
    gl.CopyTexSubImage2D(G.GL_TEXTURE_2D, 0, 0, 30, 30, 20, 25, 15)
    out["copy_tex_sub_image_2d"] = _read_tex(d, t2)

    # ---- blits: 1:1, scaled up / down, flipped, linear, across formats ---------------------------------------
    def blit(src, dst, s, dd, filt=G.GL_NEAREST):
        gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, src.fbo)
        gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, dst.fbo)
        gl.BlitFramebuffer(s[0], s[1], s[2], s[3], dd[0], dd[1], dd[2], dd[3], G.GL_COLOR_BUFFER_BIT, filt)
        gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, 0)
        gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, 0)
    big = d.create_texture(150, 110, G.GL_RGBA8, render_target=True)
    gl.ClearTexImage(big.id, 0, G.GL_RGBA, G.GL_FLOAT, red)
    # (read back before the partial blits: swgl keeps a whole-texture clear pending and, when a blit then names a skip rect,
    # force_clear (gl.cc:2207-2253) leaves the skip COLUMNS unwritten in every row of the 32-row groups the rect touches --
    # stale allocator memory, not a behaviour to reproduce; the readback resolves the pending clear first)
    out["clear_big"] = _read_tex(d, big)
    blit(t1, big, (0, 0, tw, th), (10, 5, 10 + tw, 5 + th))
    out["blit_1to1"] = _read_tex(d, big)
    blit(t1, big, (0, 0, tw, th), (3, 3, 3 + 2 * tw, 3 + 2 * th))
    out["blit_upscale_nearest"] = _read_tex(d, big)
    blit(t1, big, (4, 2, 60, 44), (80, 60, 80 + 23, 60 + 17))
    out["blit_downscale_nearest"] = _read_tex(d, big)
    blit(t1, big, (0, 0, tw, th), (10, 5 + th, 10 + tw, 5))                      # dest flipped in y
    out["blit_flip_dst"] = _read_tex(d, big)
    blit(t1, big, (0, th, tw, 0), (70, 10, 70 + 50, 10 + 70))                    # source flipped + scaled
    out["blit_flip_src_scaled"] = _read_tex(d, big)
    blit(t1, big, (0, 0, tw, th), (-20, -10, 100, 90))                           # dest partly outside
    out["blit_clipped_dst"] = _read_tex(d, big)
    blit(t1, big, (-10, -5, tw + 10, th + 5), (0, 0, 150, 110))                  # source request larger than the texture
    out["blit_clipped_src"] = _read_tex(d, big)
    blit(t1, big, (0, 0, tw, th), (5, 5, 5 + 131, 5 + 97), G.GL_LINEAR)
    out["blit_upscale_linear"] = _read_tex(d, big)
    blit(t1, big, (2, 2, 62, 46), (20, 20, 20 + 31, 20 + 19), G.GL_LINEAR)
    out["blit_downscale_linear"] = _read_tex(d, big)
    blit(t1, big, (0, 0, tw, th), (10, 100, 10 + 90, 30), G.GL_LINEAR)           # linear, flipped
    out["blit_linear_flip"] = _read_tex(d, big)
    blit(t_r8, big, (0, 0, 40, 24), (100, 2, 140, 26))                           # R8 -> RGBA8
    out["blit_r8_to_rgba8"] = _read_tex(d, big)
    r8b = d.create_texture(70, 50, G.GL_R8, render_target=True)
    gl.ClearTexImage(r8b.id, 0, G.GL_RED, G.GL_FLOAT, red)
    out["clear_r8b"] = _read_tex(d, r8b)
    blit(t_r8, r8b, (0, 0, 40, 24), (4, 4, 4 + 60, 4 + 40))
    blit(t1, r8b, (0, 0, 20, 20), (50, 30, 70, 50))                              # RGBA8 -> R8
    out["blit_r8_scaled_and_from_rgba8"] = _read_tex(d, r8b)
    blit(t_r8, r8b, (0, 0, 40, 24), (0, 0, 70, 50), G.GL_LINEAR)
    out["blit_r8_linear"] = _read_tex(d, r8b)

    # ---- readbacks: ReadPixels into a PBO, clipped ReadPixels, GetColorBuffer, locked resources -----------------
    pack = gl.gen("GenBuffers")
    gl.BindBuffer(G.GL_PIXEL_PACK_BUFFER, pack)
    gl.BufferData(G.GL_PIXEL_PACK_BUFFER, 32 * 16 * 4 + 32, None, G.GL_STREAM_DRAW)
    gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, t1.fbo)
    gl.ReadPixels(8, 8, 32, 16, G.GL_RGBA, G.GL_UNSIGNED_BYTE, 32)
    pm = gl.MapBuffer(G.GL_PIXEL_PACK_BUFFER, G.GL_READ_ONLY)
    out["readpixels_pbo_rgba"] = _ptr_bytes(pm + 32, 32 * 16 * 4)
    gl.UnmapBuffer(G.GL_PIXEL_PACK_BUFFER)
    gl.BindBuffer(G.GL_PIXEL_PACK_BUFFER, 0)
    clip = np.full((30, 40, 4), 77, np.uint8)
    gl.ReadPixels(40, 30, 40, 30, G.GL_BGRA, G.GL_UNSIGNED_BYTE, clip)          # partly outside the 64x48 texture
    out["readpixels_clipped"] = clip.copy()
    gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, 0)
    w_, h_, s_ = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    pc = gl.GetColorBuffer(t1.fbo, 1, w_, h_, s_)
    out["getcolorbuffer_dims"] = (w_.value, h_.value)
    out["getcolorbuffer"] = np.stack([_ptr_bytes(pc + y * s_.value, w_.value * 4) for y in range(h_.value)])
    lt = gl.LockTexture(t2.id)
    gl.LockResource(lt)
    pr = gl.GetResourceBuffer(lt, w_, h_, s_)
    out["locked_texture"] = np.stack([_ptr_bytes(pr + y * s_.value, w_.value * 4) for y in range(h_.value)])
    gl.UnlockResource(lt); gl.UnlockResource(lt)
    lf = gl.LockFramebuffer(0)
    pf = gl.GetResourceBuffer(lf, w_, h_, s_)
    out["locked_default_framebuffer"] = np.stack([_ptr_bytes(pf + y * s_.value, w_.value * 4) for y in range(h_.value)])
    gl.UnlockResource(lf)

    # ---- Composite(): locked RGBA8 -> locked RGBA8, copied or blended, scaled / flipped / clipped (composite.h:542-589)
    ld = gl.LockTexture(big.id)
    ls = gl.LockTexture(t1.id)
    def composite(src, dd, opaque, flip_x=0, flip_y=0, filt=G.GL_NEAREST, clip=None):
        clip = clip or dd
        gl.Composite(ld, ls, src[0], src[1], src[2], src[3], dd[0], dd[1], dd[2], dd[3], opaque, flip_x, flip_y, filt,
                     clip[0], clip[1], clip[2], clip[3])
        pd = gl.GetResourceBuffer(ld, w_, h_, s_)
        return np.stack([_ptr_bytes(pd + y * s_.value, w_.value * 4) for y in range(h_.value)])
    out["composite_opaque_1to1"] = composite((0, 0, tw, th), (20, 30, tw, th), 1)
    out["composite_blend_1to1"] = composite((0, 0, tw, th), (70, 40, tw, th), 0)
    out["composite_blend_clipped"] = composite((0, 0, tw, th), (-10, -8, tw, th), 0, clip=(0, 0, 30, 25))
    out["composite_opaque_scaled"] = composite((3, 2, 50, 40), (5, 60, 90, 45), 1)
    out["composite_blend_scaled_flip_y"] = composite((0, 0, tw, th), (40, 10, 100, 90), 0, flip_y=1)
    out["composite_opaque_linear"] = composite((0, 0, tw, th), (2, 2, 140, 100), 1, filt=G.GL_LINEAR)
    out["composite_blend_linear_clip"] = composite((4, 4, 40, 30), (10, 10, 120, 95), 0, filt=G.GL_LINEAR, clip=(30, 25, 70, 50))
    out["composite_flip_x"] = composite((0, 0, tw, th), (60, 50, tw, th), 1, flip_x=1)
    out["composite_blend_flip_xy_scaled"] = composite((0, 0, tw, th), (0, 0, 150, 110), 0, flip_x=1, flip_y=1, filt=G.GL_LINEAR)
    out["composite_src_outside"] = composite((-6, -4, tw + 12, th + 8), (30, 20, 100, 80), 1)
    gl.UnlockResource(ls); gl.UnlockResource(ld)
    out["composite_texture_after"] = _read_tex(d, big)

    # ---- CompositeYUV(): three locked planes -> locked RGBA8 through the fixed-point colour matrix (composite.h:1160-1386):
    # 4:2:0 and 4:4:4 R8 planes and 10-bit R16 ones, every YUVRangedColorSpace, 1:1 / up / down scales (the half-resolution
    # chroma fast path and the generic row walk), flips, clips, requests reaching outside the planes and the destination
    rngv = np.random.default_rng(97)
    vw, vh = 96, 64
    def plane(w, h, bits=8):
        yy, xx = np.mgrid[0:h, 0:w]
        base = ((xx * 255 // max(w - 1, 1)) ^ rngv.integers(0, 256, size=(h, w))).astype(np.uint16) & 0xFF
        return base.astype(np.uint8) if bits == 8 else ((base << (bits - 8)) | rngv.integers(0, 1 << (bits - 8), size=(h, w))).astype(np.uint16)
    def video(cw, ch, bits=8):
        fmt, ty = (G.GL_R8, G.GL_UNSIGNED_BYTE) if bits == 8 else (G.GL_R16, G.GL_UNSIGNED_SHORT)
        texs = []
        for (w, h) in ((vw, vh), (cw, ch), (cw, ch)):
            t = d.create_texture(w, h, fmt)
            d.upload_texture(t, 0, 0, w, h, G.GL_RED, ty, plane(w, h, bits))
            texs.append(t)
        return texs
    yuv_dst = d.create_texture(200, 160, G.GL_RGBA8, render_target=True)
    gl.ClearColorRect(yuv_dst.fbo, 0, 0, 200, 160, 0.25, 0.5, 0.75, 1.0)      # (fresh texture storage is not defined: start from a colour)
    lyd = gl.LockTexture(yuv_dst.id)
    def composite_yuv(planes, space, depth, src, dd, flip_x=0, flip_y=0, clip=None):
        clip = clip or dd
        locks = [gl.LockTexture(t.id) for t in planes]
        gl.CompositeYUV(lyd, locks[0], locks[1], locks[2], space, depth, src[0], src[1], src[2], src[3], dd[0], dd[1], dd[2], dd[3],
                        flip_x, flip_y, clip[0], clip[1], clip[2], clip[3])
        for l in locks:
            gl.UnlockResource(l)
        pd = gl.GetResourceBuffer(lyd, w_, h_, s_)
        return np.stack([_ptr_bytes(pd + y * s_.value, w_.value * 4) for y in range(h_.value)])
    v420, v444, v420_10 = video(vw // 2, vh // 2), video(vw, vh), video(vw // 2, vh // 2, 10)
    for space in range(7):
        out[f"composite_yuv_420_1to1_space{space}"] = composite_yuv(v420, space, 8, (0, 0, vw, vh), (10 + space, 8, vw, vh))
    out["composite_yuv_420_upscaled"] = composite_yuv(v420, 2, 8, (0, 0, vw, vh), (3, 5, 190, 150))
    out["composite_yuv_420_upscaled_frac"] = composite_yuv(v420, 0, 8, (5, 3, 70, 50), (7, 9, 171, 131))
    out["composite_yuv_420_downscaled"] = composite_yuv(v420, 3, 8, (0, 0, vw, vh), (20, 20, 41, 29))
    out["composite_yuv_420_flip_xy"] = composite_yuv(v420, 1, 8, (0, 0, vw, vh), (30, 10, 120, 100), flip_x=1, flip_y=1)
    out["composite_yuv_420_clipped"] = composite_yuv(v420, 4, 8, (0, 0, vw, vh), (-20, -12, 180, 140), clip=(10, 6, 100, 90))
    out["composite_yuv_420_src_outside"] = composite_yuv(v420, 2, 8, (-9, -5, vw + 20, vh + 12), (15, 12, 150, 110))
    out["composite_yuv_444_1to1"] = composite_yuv(v444, 5, 8, (0, 0, vw, vh), (40, 50, vw, vh))
    out["composite_yuv_444_scaled"] = composite_yuv(v444, 2, 8, (2, 1, 80, 60), (0, 0, 133, 101))
    out["composite_yuv_420_10bit"] = composite_yuv(v420_10, 2, 10, (0, 0, vw, vh), (12, 14, 150, 120))
    out["composite_yuv_420_10bit_1to1"] = composite_yuv(v420_10, 4, 10, (0, 0, vw, vh), (50, 40, vw, vh), flip_y=1)
    gl.UnlockResource(lyd)
    out["composite_yuv_texture_after"] = _read_tex(d, yuv_dst)

    # ---- externally backed texture: the caller's memory holds the result after ResolveFramebuffer -------------
    ext = np.zeros((40, 64, 4), np.uint8)
    ext[..., 1] = 200
    te = gl.gen("GenTextures")
    gl.SetTextureBuffer(te, G.GL_RGBA8, 50, 40, 64 * 4, ext, 0, 0)
    fe = gl.gen("GenFramebuffers")
    gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, fe)
    gl.FramebufferTexture2D(G.GL_DRAW_FRAMEBUFFER, G.GL_COLOR_ATTACHMENT0, G.GL_TEXTURE_2D, te, 0)
    out["ext_fb_status"] = int(gl.CheckFramebufferStatus(G.GL_DRAW_FRAMEBUFFER) == G.GL_FRAMEBUFFER_COMPLETE)
    gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, 0)
    gl.ClearColorRect(fe, 5, 5, 30, 20, 0.2, 0.4, 0.6, 0.8)
    blit_src = t1
    gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, blit_src.fbo); gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, fe)
    gl.BlitFramebuffer(0, 0, 20, 20, 28, 18, 48, 38, G.GL_COLOR_BUFFER_BIT, G.GL_NEAREST)
    gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, 0); gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, 0)
    gl.ResolveFramebuffer(fe)
    gl.Finish()
    pce = gl.GetColorBuffer(fe, 1, w_, h_, s_)
    out["external_texture"] = np.stack([_ptr_bytes(pce + y * s_.value, w_.value * 4) for y in range(h_.value)])
    out["external_texture_dims"] = (w_.value, h_.value, s_.value)

    # ---- a depth buffer that has to survive a flush in the middle of a target ------------------------------
    fr = scenes.add_occluders(scenes.image_grid(n=40, seed=77, **small), n=14, zmax=44, seed=5, first=lambda tx, ty: True)
    for ref in fr.static_textures:
        r.resolve(ref)
    d.disable_depth_write(); d.set_blend(False)
    r.bind_frame_data(fr)
    gbf = r._create_gpu_buffer_texture("sGpuBufferF", fr.gpu_buffer_f, G.GL_RGBA32F, G.GL_RGBA, G.GL_FLOAT)
    gbi = r._create_gpu_buffer_texture("sGpuBufferI", fr.gpu_buffer_i, G.GL_RGBA32I, G.GL_RGBA_INTEGER, G.GL_INT)
    target = fr.passes[0][0]
    tex = r.resolve(target.texture)
    d.bind_draw_target(tex.fbo_with_depth, tex.width, tex.height)
    proj = ortho(0.0, tex.width, 0.0, tex.height)
    d.enable_depth_write(); d.set_blend(False)
    d.clear_target(target.clear_color, 1.0, None)
    d.enable_depth(G.GL_LEQUAL)
    mids = []
    for i, step in enumerate(target.opaque):
        r._draw_step(step, proj)
        # the target is read back between its batches: everything recorded so far has to run, and the depth it leaves
        # behind is still needed by the batches that follow
        gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, tex.fbo)
        px = np.empty((64, 64, 4), np.uint8)
        gl.ReadPixels(0, 0, 64, 64, G.GL_BGRA, G.GL_UNSIGNED_BYTE, px)
        gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, 0)
        mids.append(px)
    d.disable_depth_write()
    d.set_blend(True)
    for step in target.alpha:
        d.set_blend_mode(step.blend)
        r._draw_step(step, proj)
        qx = gl.gen("GenQueries")                      # a timer query also drains the pipeline mid-target
        gl.BeginQuery(G.GL_TIME_ELAPSED, qx); gl.EndQuery(G.GL_TIME_ELAPSED); gl.DeleteQuery(qx)
    d.set_blend(False); d.disable_depth(); d.invalidate_depth_target()
    gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, 0)
    out["mid_target_readbacks"] = np.stack(mids)
    out["mid_target_flush_tile"] = _read_tex(d, tex)
    d.delete_texture(gbf); d.delete_texture(gbi)

    # ---- programs / attributes / errors / context reference counting -----------------------------------
    prog = d.create_program("brush_solid", "PRIM_INSTANCES")
    out["attrib_locations"] = (gl.GetAttribLocation(prog.id, b"aPosition"), gl.GetAttribLocation(prog.id, b"aData"),
                               gl.GetAttribLocation(prog.id, b"aNoSuchAttribute"))
    out["no_error"] = int(gl.GetError())
    size_of = C.CFUNCTYPE(C.c_size_t, C.c_void_p)(lambda ptr: 4096)          # (swgl asks the caller's allocator; libwrhip sums its HBM storage)
    out["report_memory_positive"] = int(gl.ReportMemory(d.ctx, C.cast(size_of, C.c_void_p).value) > 0)
    gl.ReferenceContext(d.ctx)
    gl.DestroyContext(d.ctx)                 # drops the extra reference only: the context keeps working
    gl.ClearTexImage(t2.id, 0, G.GL_RGBA, G.GL_FLOAT, col)
    out["after_reference_drop"] = _read_tex(d, t2)
    for t in (t1, t2, big, t_r8, r8b):
        d.delete_texture(t)
    gl.DeleteBuffer(pbo); gl.DeleteBuffer(pack); gl.DeleteFramebuffer(fe); gl.DeleteTexture(te)
    r.destroy()
    return out


def digest_of(res):
    import hashlib
    h = {}
    for k, v in sorted(res.items()):
        if isinstance(v, np.ndarray):
            h[k] = hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()
        elif isinstance(v, bytes):
            h[k] = v.decode()
        else:
            h[k] = v if isinstance(v, (int, str)) else list(v)
    return h


# entries that are expected to differ between backends (names / version strings of the implementation)
BACKEND_SPECIFIC = ("report_memory_positive",)


def compare(got, want):
    bad = []
    for k in want:
        if k in BACKEND_SPECIFIC:
            continue
        a, b = got.get(k), want[k]
        same = np.array_equal(a, b) if isinstance(b, np.ndarray) else a == b
        if not same:
            if isinstance(b, np.ndarray) and isinstance(a, np.ndarray) and a.shape == b.shape:
                diff = np.abs(a.astype(int) - b.astype(int))
                bad.append(f"{k}: {int((diff > 0).sum())} values differ, max {int(diff.max())}")
            else:
                bad.append(f"{k}: {a!r} != {b!r}")
    return bad
