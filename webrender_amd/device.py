"""Host-side mirror of `webrender::Device` (webrender/src/device/gl.rs) for the
draw path: the thin GL state cache that sits between `Renderer` and the C ABI.

Only what the batched-primitive path needs is mirrored, with the same method
names and the same GL call sequences, so that the call stream reaching the
backend is the one the Rust `Device` would produce:

  create_program / link        device/gl.rs:2326-2362, 2145-2252
  VertexDescriptor.bind        device/gl.rs:284-327
  create_vao / shared quad     renderer/vertex.rs:1077-1133
  update_vao_instances         device/gl.rs:3552-3608
  draw_indexed_triangles_instanced_u16   device/gl.rs:3728-3750
  bind_draw_target / clear_target        device/gl.rs:2574-2660, 3779-3822
  set_blend_mode_*             device/gl.rs:3901-4025
  ortho planes                 device/gl.rs:2036-2046
"""
import numpy as np
from . import glconst as G

RESERVE_DEPTH_BITS = 2  # device/gl.rs:1061
MAX_DEPTH_IDS = 1 << (24 - RESERVE_DEPTH_BITS)
ORTHO_NEAR = float(-MAX_DEPTH_IDS)
ORTHO_FAR = float(MAX_DEPTH_IDS - 1)

# Sampler -> texture unit table, renderer/mod.rs:371-385.
SAMPLER_SLOTS = {
    "sColor0": 0, "sColor1": 1, "sColor2": 2, "sGpuCache": 3,
    "sTransformPalette": 4, "sRenderTasks": 5, "sDither": 6,
    "sPrimitiveHeadersF": 7, "sPrimitiveHeadersI": 8, "sClipMask": 9,
    "sGpuBufferF": 10, "sGpuBufferI": 11,
}

F32, U8N, U16N, I32, U16 = "F32", "U8Norm", "U16Norm", "I32", "U16"
_KIND_SIZE = {F32: 4, U8N: 1, U16N: 2, I32: 4, U16: 2}


class VertexDescriptor:
    """renderer/vertex.rs:27-826 `desc::*`: (name, count, kind) lists."""

    def __init__(self, vertex_attributes, instance_attributes):
        self.vertex_attributes = vertex_attributes
        self.instance_attributes = instance_attributes

    def instance_stride(self):
        return sum(c * _KIND_SIZE[k] for _, c, k in self.instance_attributes)


_POS = [("aPosition", 2, U8N)]
DESC = {
    # vertex.rs:31-43
    "PRIM_INSTANCES": VertexDescriptor(_POS, [("aData", 4, I32)]),
    # vertex.rs:44-73
    "BLUR": VertexDescriptor(_POS, [
        ("aBlurRenderTaskAddress", 1, I32), ("aBlurSourceTaskAddress", 1, I32),
        ("aBlurDirection", 1, I32), ("aBlurParams", 3, F32)]),
    # vertex.rs:279-332 (BorderInstance, gpu_types.rs:193-202)
    "BORDER": VertexDescriptor(_POS, [
        ("aTaskOrigin", 2, F32), ("aRect", 4, F32), ("aColor0", 4, F32), ("aColor1", 4, F32), ("aFlags", 1, I32),
        ("aWidths", 2, F32), ("aRadii", 2, F32), ("aClipParams1", 4, F32), ("aClipParams2", 4, F32)]),
    # vertex.rs:74-108 (LineDecorationJob, render_target.rs:1184-1190)
    "LINE": VertexDescriptor(_POS, [
        ("aTaskRect", 4, F32), ("aLocalSize", 2, F32), ("aWavyLineThickness", 1, F32), ("aStyle", 1, I32), ("aAxisSelect", 1, F32)]),
    # vertex.rs:109-138 (FastLinearGradientInstance, prim_store/gradient/linear.rs:689-694)
    "FAST_LINEAR_GRADIENT": VertexDescriptor(_POS, [
        ("aTaskRect", 4, F32), ("aColor0", 4, F32), ("aColor1", 4, F32), ("aAxisSelect", 1, F32)]),
    # vertex.rs:139-180 (LinearGradientInstance, prim_store/gradient/linear.rs:727-734)
    "LINEAR_GRADIENT": VertexDescriptor(_POS, [
        ("aTaskRect", 4, F32), ("aStartPoint", 2, F32), ("aEndPoint", 2, F32), ("aScale", 2, F32),
        ("aExtendMode", 1, I32), ("aGradientStopsAddress", 1, I32)]),
    # vertex.rs:181-229 (RadialGradientInstance, prim_store/gradient/radial.rs)
    "RADIAL_GRADIENT": VertexDescriptor(_POS, [
        ("aTaskRect", 4, F32), ("aCenter", 2, F32), ("aScale", 2, F32), ("aStartRadius", 1, F32), ("aEndRadius", 1, F32),
        ("aXYRatio", 1, F32), ("aExtendMode", 1, I32), ("aGradientStopsAddress", 1, I32)]),
    # vertex.rs:231-279 (ConicGradientInstance, prim_store/gradient/conic.rs)
    "CONIC_GRADIENT": VertexDescriptor(_POS, [
        ("aTaskRect", 4, F32), ("aCenter", 2, F32), ("aScale", 2, F32), ("aStartOffset", 1, F32), ("aEndOffset", 1, F32),
        ("aAngle", 1, F32), ("aExtendMode", 1, I32), ("aGradientStopsAddress", 1, I32)]),
    # vertex.rs:334-358
    "SCALE": VertexDescriptor(_POS, [
        ("aScaleTargetRect", 4, F32), ("aScaleSourceRect", 4, F32),
        ("aSourceRectType", 1, F32)]),
    # vertex.rs:359-445
    "CLIP_RECT": VertexDescriptor(_POS, [
        ("aClipDeviceArea", 4, F32), ("aClipOrigins", 4, F32), ("aDevicePixelScale", 1, F32),
        ("aTransformIds", 2, I32), ("aClipLocalPos", 2, F32), ("aClipLocalRect", 4, F32),
        ("aClipMode", 1, F32), ("aClipRect_TL", 4, F32), ("aClipRadii_TL", 4, F32),
        ("aClipRect_TR", 4, F32), ("aClipRadii_TR", 4, F32), ("aClipRect_BL", 4, F32),
        ("aClipRadii_BL", 4, F32), ("aClipRect_BR", 4, F32), ("aClipRadii_BR", 4, F32)]),
    # vertex.rs:447-498
    "CLIP_BOX_SHADOW": VertexDescriptor(_POS, [
        ("aClipDeviceArea", 4, F32), ("aClipOrigins", 4, F32), ("aDevicePixelScale", 1, F32),
        ("aTransformIds", 2, I32), ("aClipDataResourceAddress", 2, U16), ("aClipSrcRectSize", 2, F32),
        ("aClipMode", 1, I32), ("aStretchMode", 2, I32), ("aClipDestRect", 4, F32)]),
    # vertex.rs:732-780
    "COMPOSITE": VertexDescriptor(_POS, [
        ("aDeviceRect", 4, F32), ("aDeviceClipRect", 4, F32), ("aColor", 4, F32),
        ("aParams", 4, F32), ("aUvRect0", 4, F32), ("aUvRect1", 4, F32),
        ("aUvRect2", 4, F32), ("aFlip", 2, F32)]),
    # vertex.rs:782-802
    "CLEAR": VertexDescriptor(_POS, [("aRect", 4, F32), ("aColor", 4, F32)]),
    # vertex.rs:802-826 (CopyInstance, gpu_types.rs:169-175)
    "COPY": VertexDescriptor(_POS, [("a_src_rect", 4, F32), ("a_dst_rect", 4, F32), ("a_dst_texture_size", 2, F32)]),
    # vertex.rs:532-581 (SvgFilterInstance, gpu_types.rs:148-158)
    "SVG_FILTER": VertexDescriptor(_POS, [
        ("aFilterRenderTaskAddress", 1, I32), ("aFilterInput1TaskAddress", 1, I32), ("aFilterInput2TaskAddress", 1, I32),
        ("aFilterKind", 1, U16), ("aFilterInputCount", 1, U16), ("aFilterGenericInt", 1, U16), ("aUnused", 1, U16),
        ("aFilterExtraDataAddress", 2, U16)]),
    # vertex.rs:582-631 (SVGFEFilterInstance, gpu_types.rs:160-170)
    "SVG_FILTER_NODE": VertexDescriptor(_POS, [
        ("aFilterTargetRect", 4, F32), ("aFilterInput1ContentScaleAndOffset", 4, F32), ("aFilterInput2ContentScaleAndOffset", 4, F32),
        ("aFilterInput1TaskAddress", 1, I32), ("aFilterInput2TaskAddress", 1, I32), ("aFilterKind", 1, U16),
        ("aFilterInputCount", 1, U16), ("aFilterExtraDataAddress", 2, U16)]),
    # vertex.rs:633-650 (MaskInstance, gpu_types.rs:618-624)
    "MASK": VertexDescriptor(_POS, [("aData", 4, I32), ("aClipData", 4, I32)]),
}


def ortho(left, right, bottom, top, near=ORTHO_NEAR, far=ORTHO_FAR):
    """euclid Transform3D::ortho, column-major float32[16] as sent to
    glUniformMatrix4fv (renderer/mod.rs:4705-4712)."""
    f = np.float32
    tx = -(f(right) + f(left)) / (f(right) - f(left))
    ty = -(f(top) + f(bottom)) / (f(top) - f(bottom))
    tz = -(f(far) + f(near)) / (f(far) - f(near))
    m = np.zeros(16, dtype=np.float32)
    m[0] = f(2.0) / (f(right) - f(left))
    m[5] = f(2.0) / (f(top) - f(bottom))
    m[10] = f(-2.0) / (f(far) - f(near))
    m[12], m[13], m[14], m[15] = tx, ty, tz, 1.0
    return m


class Program:
    def __init__(self, pid, key, u_transform):
        self.id, self.key, self.u_transform = pid, key, u_transform


class VAO:
    def __init__(self, vid, instance_vbo, stride):
        self.id, self.instance_vbo, self.instance_stride = vid, instance_vbo, stride


class Texture:
    def __init__(self, tid, w, h, fmt, fbo=0, fbo_with_depth=0):
        self.id, self.width, self.height, self.format = tid, w, h, fmt
        self.fbo, self.fbo_with_depth = fbo, fbo_with_depth


class Device:
    def __init__(self, gl):
        self.gl = gl
        self.ctx = gl.CreateContext()
        gl.MakeCurrent(self.ctx)
        self.programs = {}
        self.vaos = {}
        self.bound_program = 0
        self.quad_vbo = self.quad_ibo = 0
        self.depth_rbos = {}

    # -- context / default framebuffer (wrench/src/main.rs:283-288) --------
    def init_default_framebuffer(self, w, h):
        self.gl.InitDefaultFramebuffer(0, 0, w, h, 0, None)
        self.fb_size = (w, h)

    def is_software_webrender(self):  # device/gl.rs:1645
        return self.gl.GetString(G.GL_RENDERER).startswith(b"Software WebRender")

    # -- programs ----------------------------------------------------------
    def create_program(self, key, desc_name):
        if key in self.programs:
            return self.programs[key]
        gl = self.gl
        desc = DESC[desc_name]
        vs = gl.CreateShader(G.GL_VERTEX_SHADER)
        fs = gl.CreateShader(G.GL_FRAGMENT_SHADER)
        gl.ShaderSourceByName(vs, key.encode())
        gl.ShaderSourceByName(fs, key.encode())
        pid = gl.CreateProgram()
        gl.AttachShader(pid, vs)
        gl.AttachShader(pid, fs)
        for i, (name, _, _) in enumerate(desc.vertex_attributes + desc.instance_attributes):
            gl.BindAttribLocation(pid, i, name.encode())
        gl.LinkProgram(pid)
        if not gl.GetLinkStatus(pid):
            raise RuntimeError(f"backend has no shader '{key}'")
        gl.DeleteShader(vs)
        gl.DeleteShader(fs)
        gl.UseProgram(pid)
        self.bound_program = pid
        for name, slot in SAMPLER_SLOTS.items():
            loc = gl.GetUniformLocation(pid, name.encode())
            if loc != -1:
                gl.Uniform1i(loc, slot)
        u_transform = gl.GetUniformLocation(pid, b"uTransform")
        prog = Program(pid, key, u_transform)
        self.programs[key] = prog
        return prog

    def bind_program(self, prog, projection):
        gl = self.gl
        if self.bound_program != prog.id:
            gl.UseProgram(prog.id)
            self.bound_program = prog.id
        gl.UniformMatrix4fv(prog.u_transform, 1, 0, projection)

    # -- VAOs --------------------------------------------------------------
    def _bind_attributes(self, attrs, start_index, divisor, vbo):
        gl = self.gl
        gl.BindBuffer(G.GL_ARRAY_BUFFER, vbo)
        stride = sum(c * _KIND_SIZE[k] for _, c, k in attrs)
        offset = 0
        for i, (_, count, kind) in enumerate(attrs):
            idx = start_index + i
            gl.EnableVertexAttribArray(idx)
            gl.VertexAttribDivisor(idx, divisor)
            if kind == F32:
                gl.VertexAttribPointer(idx, count, G.GL_FLOAT, 0, stride, offset)
            elif kind == U8N:
                gl.VertexAttribPointer(idx, count, G.GL_UNSIGNED_BYTE, 1, stride, offset)
            elif kind == U16N:
                gl.VertexAttribPointer(idx, count, G.GL_UNSIGNED_SHORT, 1, stride, offset)
            elif kind == I32:
                gl.VertexAttribIPointer(idx, count, G.GL_INT, stride, offset)
            else:
                gl.VertexAttribIPointer(idx, count, G.GL_UNSIGNED_SHORT, stride, offset)
            offset += count * _KIND_SIZE[kind]

    def create_vao(self, desc_name):
        if desc_name in self.vaos:
            return self.vaos[desc_name]
        gl = self.gl
        desc = DESC[desc_name]
        vid = gl.gen("GenVertexArrays")
        gl.BindVertexArray(vid)
        if not self.quad_vbo:
            # Shared unit quad, renderer/vertex.rs:1077-1096.
            self.quad_vbo = gl.gen("GenBuffers")
            self.quad_ibo = gl.gen("GenBuffers")
            verts = np.array([[0, 0], [255, 0], [0, 255], [255, 255]], dtype=np.uint8)
            idx = np.array([0, 1, 2, 2, 1, 3], dtype=np.uint16)
            gl.BindBuffer(G.GL_ARRAY_BUFFER, self.quad_vbo)
            gl.BufferData(G.GL_ARRAY_BUFFER, verts.nbytes, verts, G.GL_STATIC_DRAW)
            gl.BindBuffer(G.GL_ELEMENT_ARRAY_BUFFER, self.quad_ibo)
            gl.BufferData(G.GL_ELEMENT_ARRAY_BUFFER, idx.nbytes, idx, G.GL_STATIC_DRAW)
        instance_vbo = gl.gen("GenBuffers")
        gl.BindBuffer(G.GL_ELEMENT_ARRAY_BUFFER, self.quad_ibo)
        self._bind_attributes(desc.vertex_attributes, 0, 0, self.quad_vbo)
        self._bind_attributes(desc.instance_attributes, len(desc.vertex_attributes), 1,
                              instance_vbo)
        vao = VAO(vid, instance_vbo, desc.instance_stride())
        self.vaos[desc_name] = vao
        return vao

    def draw_instanced_batch(self, vao, instances):
        """Renderer::draw_instanced_batch (renderer/mod.rs:2022-2065): bind VAO,
        upload the instance array, one DrawElementsInstanced."""
        gl = self.gl
        data = np.ascontiguousarray(instances).view(np.uint8).reshape(-1)
        n = data.nbytes // vao.instance_stride
        assert n * vao.instance_stride == data.nbytes
        gl.BindVertexArray(vao.id)
        gl.BindBuffer(G.GL_ARRAY_BUFFER, vao.instance_vbo)
        gl.BufferData(G.GL_ARRAY_BUFFER, data.nbytes, data, G.GL_STREAM_DRAW)
        gl.DrawElementsInstanced(G.GL_TRIANGLES, 6, G.GL_UNSIGNED_SHORT, 0, n)

    # -- textures / targets ------------------------------------------------
    def create_texture(self, w, h, internal_format, filter_=G.GL_LINEAR,
                       render_target=False, with_depth=False):
        gl = self.gl
        tid = gl.gen("GenTextures")
        gl.ActiveTexture(G.GL_TEXTURE0)
        gl.BindTexture(G.GL_TEXTURE_2D, tid)
        gl.TexParameteri(G.GL_TEXTURE_2D, G.GL_TEXTURE_MAG_FILTER, filter_)
        gl.TexParameteri(G.GL_TEXTURE_2D, G.GL_TEXTURE_MIN_FILTER, filter_)
        gl.TexParameteri(G.GL_TEXTURE_2D, G.GL_TEXTURE_WRAP_S, G.GL_CLAMP_TO_EDGE)
        gl.TexParameteri(G.GL_TEXTURE_2D, G.GL_TEXTURE_WRAP_T, G.GL_CLAMP_TO_EDGE)
        gl.TexStorage2D(G.GL_TEXTURE_2D, 1, internal_format, w, h)
        tex = Texture(tid, w, h, internal_format)
        if render_target:
            tex.fbo = gl.gen("GenFramebuffers")
            gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, tex.fbo)
            gl.FramebufferTexture2D(G.GL_DRAW_FRAMEBUFFER, G.GL_COLOR_ATTACHMENT0,
                                    G.GL_TEXTURE_2D, tid, 0)
            if with_depth:
                rbo = self.depth_rbos.get((w, h))
                if rbo is None:  # shared depth target per size, device/gl.rs:2776-2800
                    rbo = gl.gen("GenRenderbuffers")
                    gl.BindRenderbuffer(G.GL_RENDERBUFFER, rbo)
                    gl.RenderbufferStorage(G.GL_RENDERBUFFER, G.GL_DEPTH_COMPONENT24, w, h)
                    self.depth_rbos[(w, h)] = rbo
                tex.fbo_with_depth = gl.gen("GenFramebuffers")
                gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, tex.fbo_with_depth)
                gl.FramebufferTexture2D(G.GL_DRAW_FRAMEBUFFER, G.GL_COLOR_ATTACHMENT0,
                                        G.GL_TEXTURE_2D, tid, 0)
                gl.FramebufferRenderbuffer(G.GL_DRAW_FRAMEBUFFER, G.GL_DEPTH_ATTACHMENT,
                                           G.GL_RENDERBUFFER, rbo)
            gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, 0)
        return tex

    def upload_texture(self, tex, x, y, w, h, fmt, ty, data):
        gl = self.gl
        gl.ActiveTexture(G.GL_TEXTURE0)
        gl.BindTexture(G.GL_TEXTURE_2D, tex.id)
        gl.TexSubImage2D(G.GL_TEXTURE_2D, 0, x, y, w, h, fmt, ty, data)

    def delete_texture(self, tex):
        gl = self.gl
        if tex.fbo:
            gl.DeleteFramebuffer(tex.fbo)
        if tex.fbo_with_depth:
            gl.DeleteFramebuffer(tex.fbo_with_depth)
        gl.DeleteTexture(tex.id)

    def bind_texture(self, slot, tex_id, rect=False):
        """rect: through GL_TEXTURE_RECTANGLE -- the binding a TEXTURE_RECT program's sampler2DRect reads (gl.cc:853-855)"""
        self.gl.ActiveTexture(G.GL_TEXTURE0 + slot)
        self.gl.BindTexture(G.GL_TEXTURE_RECTANGLE if rect else G.GL_TEXTURE_2D, tex_id)

    def bind_draw_target(self, fbo, w, h):
        self.gl.BindFramebuffer(G.GL_DRAW_FRAMEBUFFER, fbo)
        self.gl.SetViewport(0, 0, w, h)

    def clear_target(self, color, depth, rect=None):
        gl = self.gl
        bits = 0
        if color is not None:
            gl.ClearColor(*[float(c) for c in color])
            bits |= G.GL_COLOR_BUFFER_BIT
        if depth is not None:
            gl.ClearDepth(float(depth))
            bits |= G.GL_DEPTH_BUFFER_BIT
        if bits:
            if rect is not None:
                gl.Enable(G.GL_SCISSOR_TEST)
                gl.SetScissor(*rect)
                gl.Clear(bits)
                gl.Disable(G.GL_SCISSOR_TEST)
            else:
                gl.Clear(bits)

    # -- blend / depth -------------------------------------------------------
    def set_blend(self, on):
        (self.gl.Enable if on else self.gl.Disable)(G.GL_BLEND)

    def set_blend_factors(self, color, alpha):
        self.gl.BlendEquation(G.GL_FUNC_ADD)
        self.gl.BlendFunc(color[0], color[1], alpha[0], alpha[1])

    BLEND_MODES = {
        "Alpha": ((G.GL_SRC_ALPHA, G.GL_ONE_MINUS_SRC_ALPHA), (G.GL_ONE, G.GL_ONE_MINUS_SRC_ALPHA)),
        "PremultipliedAlpha": ((G.GL_ONE, G.GL_ONE_MINUS_SRC_ALPHA), (G.GL_ONE, G.GL_ONE_MINUS_SRC_ALPHA)),
        "PremultipliedDestOut": ((G.GL_ZERO, G.GL_ONE_MINUS_SRC_ALPHA), (G.GL_ZERO, G.GL_ONE_MINUS_SRC_ALPHA)),
        "Multiply": ((G.GL_ZERO, G.GL_SRC_COLOR), (G.GL_ZERO, G.GL_SRC_ALPHA)),
        "SubpixelDualSource": ((G.GL_ONE, G.GL_ONE_MINUS_SRC1_COLOR), (G.GL_ONE, G.GL_ONE_MINUS_SRC1_ALPHA)),
        "PlusLighter": ((G.GL_ONE, G.GL_ONE), (G.GL_ONE, G.GL_ONE)),
        "Screen": ((G.GL_ONE, G.GL_ONE_MINUS_SRC_COLOR), (G.GL_ONE, G.GL_ONE_MINUS_SRC_ALPHA)),
    }

    ADVANCED = {"Multiply": G.GL_MULTIPLY_KHR, "Screen": G.GL_SCREEN_KHR, "Overlay": G.GL_OVERLAY_KHR, "Darken": G.GL_DARKEN_KHR,
                "Lighten": G.GL_LIGHTEN_KHR, "ColorDodge": G.GL_COLORDODGE_KHR, "ColorBurn": G.GL_COLORBURN_KHR,
                "HardLight": G.GL_HARDLIGHT_KHR, "SoftLight": G.GL_SOFTLIGHT_KHR, "Difference": G.GL_DIFFERENCE_KHR,
                "Exclusion": G.GL_EXCLUSION_KHR, "Hue": G.GL_HSL_HUE_KHR, "Saturation": G.GL_HSL_SATURATION_KHR,
                "Color": G.GL_HSL_COLOR_KHR, "Luminosity": G.GL_HSL_LUMINOSITY_KHR}

    def set_blend_mode(self, name):
        """device/gl.rs:3901-4025.  "Advanced:<MixBlendMode>" = set_blend_mode_advanced (KHR_blend_equation_advanced, what the
        renderer uses for mix-blend-mode pictures when the backend advertises the extension); "SubpixelConstantTextColor:r,g,b,a"
        = set_blend_mode_subpixel_constant_text_color; "Min" / "Max": the plain GL equations (in swgl's key table, gl.cc:627-628)."""
        if name.startswith("Advanced:"):
            self.gl.BlendEquation(self.ADVANCED[name.split(":", 1)[1]])
        elif name in ("Min", "Max"):
            self.gl.BlendEquation(G.GL_MIN if name == "Min" else G.GL_MAX)
        elif name.startswith("SubpixelConstantTextColor:"):
            r, g, b, a = (float(v) for v in name.split(":", 1)[1].split(","))
            self.gl.BlendColor(r, g, b, a)
            self.set_blend_factors((G.GL_CONSTANT_COLOR, G.GL_ONE_MINUS_SRC_COLOR), (G.GL_CONSTANT_ALPHA, G.GL_ONE_MINUS_SRC_ALPHA))
        else:
            self.set_blend_factors(*self.BLEND_MODES[name])

    def enable_depth(self, func=G.GL_LEQUAL):
        self.gl.Enable(G.GL_DEPTH_TEST)
        self.gl.DepthFunc(func)

    def disable_depth(self):
        self.gl.Disable(G.GL_DEPTH_TEST)

    def enable_depth_write(self):
        self.gl.DepthMask(1)

    def disable_depth_write(self):
        self.gl.DepthMask(0)

    def invalidate_depth_target(self):
        att = np.array([G.GL_DEPTH_ATTACHMENT], dtype=np.uint32)
        self.gl.InvalidateFramebuffer(G.GL_DRAW_FRAMEBUFFER, 1, att)

    def read_pixels_rgba8(self, x, y, w, h):
        out = np.empty((h, w, 4), dtype=np.uint8)
        self.gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, 0)
        self.gl.ReadPixels(x, y, w, h, G.GL_RGBA, G.GL_UNSIGNED_BYTE, out)
        return out

    def read_texture(self, tex):
        """Read back a render-target texture (BGRA8 bytes as stored, or R8)."""
        gl = self.gl
        gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, tex.fbo)
        if tex.format == G.GL_R8:
            out = np.empty((tex.height, tex.width), dtype=np.uint8)
            gl.ReadPixels(0, 0, tex.width, tex.height, G.GL_RED, G.GL_UNSIGNED_BYTE, out)
        else:
            out = np.empty((tex.height, tex.width, 4), dtype=np.uint8)
            gl.ReadPixels(0, 0, tex.width, tex.height, G.GL_BGRA, G.GL_UNSIGNED_BYTE, out)
        gl.BindFramebuffer(G.GL_READ_FRAMEBUFFER, 0)
        return out

    def destroy(self):
        self.gl.MakeCurrent(None)
        self.gl.DestroyContext(self.ctx)
