"""Frame model: the *output* of WebRender's CPU frame builder / batcher, i.e. the
input contract of the draw backend.

The Rust frame builder cannot be built here (no cargo), so scene generators
(scenes.py) fill these structures directly.  Everything in this file restates
*data layouts only*:

  data textures, 1024 texels wide   webrender_build/src/lib.rs:19,
                                    renderer/vertex.rs:978-1040
  PrimitiveHeaderF / I              gpu_types.rs:473-492
  TransformData (m, inv_m)          gpu_types.rs:764-768
  RenderTaskData                    render_task.rs (task_rect, [dps, origin])
  GPU cache blocks                  gpu_cache.rs:46,79-89,184-189
  GpuBufferF / GpuBufferI           renderer/gpu_buffer.rs:73-85
  PrimitiveInstanceData encodings   gpu_types.rs:511-528 (glyph), 564-589
                                    (quad), 690-703 (brush)
  quad prim blocks / header         quad.rs:941-1000
  CompositeInstance                 gpu_types.rs:289-311
"""
import numpy as np
from . import glconst as G

TEX_W = 1024  # MAX_VERTEX_TEXTURE_WIDTH

# ps_quad.glsl:60-80 / quad.rs QuadFlags
QF_IS_OPAQUE, QF_APPLY_DEVICE_CLIP, QF_IGNORE_DEVICE_SCALE, QF_USE_AA_SEGMENTS, QF_IS_MASK = 1, 2, 4, 8, 16
PART_CENTER, PART_LEFT, PART_TOP, PART_RIGHT, PART_BOTTOM, PART_ALL = 0, 1, 2, 3, 4, 5
INVALID_QUAD_SEGMENT = 0xFF
INVALID_BRUSH_SEGMENT = 0xFFFF
CLIP_TASK_EMPTY = 0x7FFFFFFF  # render_task.glsl:81
TRANSFORM_NON_AXIS_ALIGNED = 1 << 23  # transform.glsl:25

# brush.glsl:57-69
BRUSH_FLAG_PERSPECTIVE_INTERPOLATION = 1
BRUSH_FLAG_SEGMENT_RELATIVE = 2
BRUSH_FLAG_SEGMENT_REPEAT_X = 4
BRUSH_FLAG_SEGMENT_REPEAT_Y = 8
BRUSH_FLAG_TEXEL_RECT = 512
BRUSH_FLAG_FORCE_AA = 1024
BRUSH_FLAG_NORMALIZED_UVS = 2048


class BlockStore:
    """A growable array of 16-byte texels laid out as a 1024-wide 2-D texture.
    Multi-texel records never straddle a row (gpu_cache.rs:330-420 free lists,
    gpu_buffer.rs:291-336)."""

    def __init__(self, dtype):
        self.dtype = dtype
        self.data = np.zeros((4096, 4), dtype=dtype)
        self.len = 0

    def alloc(self, n):
        col = self.len % TEX_W
        if col + n > TEX_W:
            self.len += TEX_W - col
        addr = self.len
        self.len += n
        while self.len > len(self.data):
            self.data = np.concatenate([self.data, np.zeros_like(self.data)])
        return addr

    def push(self, blocks):
        blocks = np.asarray(blocks, dtype=self.dtype).reshape(-1, 4)
        addr = self.alloc(len(blocks))
        self.data[addr:addr + len(blocks)] = blocks
        return addr

    def rows(self, min_rows=1):
        return max(min_rows, (self.len + TEX_W - 1) // TEX_W)

    def texture_data(self, min_rows=1):
        r = self.rows(min_rows)
        need = r * TEX_W
        while need > len(self.data):
            self.data = np.concatenate([self.data, np.zeros_like(self.data)])
        return np.ascontiguousarray(self.data[:need]).reshape(r, TEX_W, 4)


def identity_transform():
    return np.eye(4, dtype=np.float32)


class TextureRef:
    """A named, persistent backend texture (tile surface, atlas, render-task
    target...).  Resolved to a device.Texture by the Renderer."""

    def __init__(self, name, w, h, fmt=G.GL_RGBA8, filter_=G.GL_LINEAR,
                 render_target=False, with_depth=False, pixels=None, upload_format=None, upload_type=None):
        self.upload_type = upload_type      # GL type of `pixels` (default GL_UNSIGNED_BYTE)
        self.name, self.w, self.h, self.fmt, self.filter = name, w, h, fmt, filter_
        self.render_target, self.with_depth = render_target, with_depth
        self.pixels = pixels  # optional numpy upload (atlases)
        self.upload_format = upload_format


class Step:
    """One DrawElementsInstanced with its GL state."""

    def __init__(self, shader, desc, instances, blend=None, depth="none", textures=None):
        self.shader, self.desc, self.instances = shader, desc, instances
        self.blend = blend          # None (blend off) or Device.BLEND_MODES key
        self.depth = depth          # "opaque" | "alpha" | "none"
        self.textures = textures or {}   # slot -> TextureRef


class Target:
    """A render target and its ordered draw steps."""

    def __init__(self, texture, kind="picture_tile", clear_color=None, clear_depth=False,
                 clear_rect=None):
        self.texture = texture
        self.kind = kind            # picture_tile | color | alpha | texture_cache
        self.clear_color = clear_color
        self.clear_depth = clear_depth
        self.clear_rect = clear_rect
        self.opaque = []            # Steps, already in draw (front-to-back) order
        self.alpha = []             # Steps, back-to-front
        self.steps = []             # generic ordered steps (non-tile targets)


class CompositeTile:
    """One CompositeInstance.  Plain picture-cache tiles (no colour, whole
    texture) go through the FAST_PATH program; tiles / external surfaces with a
    colour, a uv sub-rect (texels) or a flip through "composite TEXTURE_2D"
    (composite.rs:1090-1160, renderer/mod.rs:3260-3334)."""

    def __init__(self, texture, rect, clip_rect=None, opaque=True, color=None, uv_rect=None,
                 flip=(0.0, 0.0), yuv=None):
        """yuv: an external YUV surface (composite.rs ExternalSurfaceDependency::Yuv -> "composite TEXTURE_2D,YUV"):
        dict(planes=[TextureRef per plane], uv_rects=[texel rect per plane], color_space, format, depth); `texture` = planes[0]"""
        self.yuv = yuv
        self.texture, self.rect = texture, rect
        self.clip_rect = clip_rect or rect
        self.opaque, self.color = opaque, color
        self.uv_rect, self.flip = uv_rect, flip

    @property
    def fast(self):
        return self.yuv is None and self.color is None and self.uv_rect is None and self.flip == (0.0, 0.0)


class Frame:
    def __init__(self, width, height, clear_color=(1.0, 1.0, 1.0, 1.0)):
        self.width, self.height, self.clear_color = width, height, clear_color
        self.prim_headers_f = BlockStore(np.float32)
        self.prim_headers_i = BlockStore(np.int32)
        self.transforms = BlockStore(np.float32)
        self.render_tasks = BlockStore(np.float32)
        self.gpu_cache = BlockStore(np.float32)
        self.gpu_buffer_f = BlockStore(np.float32)
        self.gpu_buffer_i = BlockStore(np.int32)
        self.passes = []            # list of lists of Target
        self.composite_tiles = []
        self.static_textures = []   # atlases to upload once
        self.readback = []          # TextureRefs the tests read back besides the window
        self.add_transform(identity_transform(), identity_transform())  # id 0 = IDENTITY
        self._n_headers = 0

    # ---- data-texture writers -------------------------------------------
    def add_transform(self, m, inv_m, axis_aligned=True):
        """TransformData: m then inv_m, column-major rows of 4 (gpu_types.rs:764)."""
        blocks = np.concatenate([np.asarray(m, np.float32).reshape(4, 4),
                                 np.asarray(inv_m, np.float32).reshape(4, 4)])
        addr = self.transforms.push(blocks)
        tid = addr // 8
        return tid if axis_aligned else (tid | TRANSFORM_NON_AXIS_ALIGNED)

    def add_render_task(self, task_rect, device_pixel_scale=1.0, content_origin=(0.0, 0.0)):
        x0, y0, x1, y1 = task_rect
        addr = self.render_tasks.push([[x0, y0, x1, y1],
                                       [device_pixel_scale, content_origin[0], content_origin[1], 0.0]])
        return addr // 2

    def add_prim_header(self, local_rect, local_clip_rect, z, specific_prim_address,
                        transform_id, picture_task_address, user_data=(0, 0, 0, 0)):
        fa = self.prim_headers_f.push([list(local_rect), list(local_clip_rect)])
        ia = self.prim_headers_i.push([[z, specific_prim_address, transform_id, picture_task_address],
                                       list(user_data)])
        assert fa == ia and fa % 2 == 0
        return fa // 2

    # ---- instance encoders ------------------------------------------------
    def quad_instance(self, bounds, clip, color, z_id, task_address, transform_id=0,
                      quad_flags=QF_APPLY_DEVICE_CLIP, edge_flags=0, part=PART_ALL,
                      segment=INVALID_QUAD_SEGMENT, uv_rect=(0, 0, 0, 0),
                      scale_offset=(1.0, 1.0, 0.0, 0.0), segments=(), pattern_input=(0, 0)):
        """quad.rs:941-1000 + gpu_types.rs:564-589.  `color` is premultiplied."""
        blocks = [list(bounds), list(clip), list(uv_rect), list(scale_offset), list(color)]
        for rect, uv in segments:
            blocks += [list(rect), list(uv)]
        addr_f = self.gpu_buffer_f.push(blocks)
        addr_i = self.gpu_buffer_i.push([[transform_id, z_id, int(pattern_input[0]), int(pattern_input[1])]])
        return [addr_i, addr_f,
                (quad_flags << 24) | (edge_flags << 16) | (part << 8) | segment,
                task_address]

    def brush_instance(self, prim_header, clip_task=CLIP_TASK_EMPTY,
                       segment=INVALID_BRUSH_SEGMENT, brush_flags=0, edge_flags=0,
                       resource_address=0, brush_kind=0):
        """gpu_types.rs:690-703 BrushInstance -> PrimitiveInstanceData."""
        flags = (brush_flags & 0xFFF) | ((edge_flags & 0xF) << 12)
        z = (segment & 0xFFFF) | (flags << 16)
        w = (resource_address & 0xFFFFFF) | (brush_kind << 24)
        return [prim_header, clip_task, np.int32(np.uint32(z & 0xFFFFFFFF)),
                np.int32(np.uint32(w & 0xFFFFFFFF))]

    @staticmethod
    def glyph_instance(prim_header, glyph_index, resource_address, clip_task=CLIP_TASK_EMPTY,
                       subpx_dir=0, color_mode=0):
        """gpu_types.rs:511-528 GlyphInstance::build."""
        return [prim_header, clip_task,
                (subpx_dir << 24) | (color_mode << 16) | glyph_index, resource_address]

    def add_text_run(self, color_premul, glyph_points):
        """prim_store/text_run.rs:107-132: [premultiplied colour] then the glyph
        points, two per block (an odd tail block keeps the previous .zw)."""
        pts = np.asarray(glyph_points, np.float32).reshape(-1, 2)
        blocks = [list(color_premul)]
        blk = [0.0, 0.0, 0.0, 0.0]
        for i, (x, y) in enumerate(pts):
            if (i & 1) == 0:
                blk[0], blk[1] = x, y
            else:
                blk[2], blk[3] = x, y
                blocks.append(list(blk))
        if len(pts) & 1:
            blocks.append(list(blk))
        return self.gpu_cache.push(blocks)

    def add_glyph_resource(self, uv_rect, offset, scale=1.0):
        """texture_cache.rs:157-168 ImageSource blocks of a cached glyph:
        [uv_rect texels], [glyph.left, -glyph.top, scale, 0] (resource_cache.rs
        glyph user_data)."""
        return self.gpu_cache.push([list(uv_rect), [offset[0], offset[1], scale, 0.0]])

    @staticmethod
    def composite_instance(rect, clip_rect, color=(1, 1, 1, 1), uv_rect=(0, 0, 1, 1),
                           uv_type=0, flip=(0.0, 0.0), yuv=None):
        """gpu_types.rs:289-311 CompositeInstance (120 bytes).  yuv: CompositeInstance::new_yuv (:338-365): params = [_, colour
        space, format, channel bit depth], one texel uv rect per plane."""
        inst = np.zeros(30, dtype=np.float32)
        inst[0:4], inst[4:8], inst[8:12] = rect, clip_rect, color
        inst[12:16] = [0.0, float(uv_type), 0.0, 0.0]
        inst[16:20] = inst[20:24] = inst[24:28] = uv_rect
        if yuv is not None:
            inst[12:16] = [0.0, float(yuv["color_space"]), float(yuv["format"]), float(yuv["depth"])]
            for k in range(3):
                inst[16 + 4 * k:20 + 4 * k] = yuv["uv_rects"][min(k, len(yuv["uv_rects"]) - 1)]
        inst[28:30] = flip
        return inst
