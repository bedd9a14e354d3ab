/* wrhip.h -- C ABI of libwrhip, the MI355X (gfx950) draw backend for WebRender.
 *
 * Drop-in boundary: these are exactly the 99 `extern "C"` functions that
 * WebRender's software-GL shim binds in the reference
 * (swgl/src/swgl_fns.rs:23-322 of servo/webrender @ 2024-12-20, cited per
 * function below as "fns:<line>"), with the same names, argument meaning and
 * error behaviour (sticky GL error read by GetError, only GL_OUT_OF_MEMORY is
 * ever raised -- swgl/src/gl.cc:1125-1134).  `impl Gl for Context`
 * (swgl_fns.rs:500-2489) links against these unchanged; see INTEGRATION.md.
 *
 * Plain C types only: no torch / HIP types cross this boundary.  All pointers
 * are host pointers.  Calls are made from one thread after MakeCurrent().
 * Rendering is deferred and executed by HIP kernels on the context's stream;
 * Finish(), ReadPixels(), GetColorBuffer(flush=1) and MapBuffer-style host
 * reads synchronise as the GL contract requires.
 */
#ifndef WRHIP_H
#define WRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int8_t GLbyte;
typedef uint8_t GLubyte;
typedef int16_t GLshort;
typedef uint16_t GLushort;
typedef int32_t GLint;
typedef uint32_t GLuint;
typedef int64_t GLint64;
typedef uint64_t GLuint64;
typedef float GLfloat;
typedef double GLdouble;
typedef uint32_t GLenum;
typedef uint8_t GLboolean;
typedef uint32_t GLbitfield;
typedef int32_t GLsizei;
typedef size_t GLsizeiptr;
typedef intptr_t GLintptr;
typedef void GLvoid;
typedef char GLchar;

/* Opaque handles (swgl_fns.rs:17-20, gl.cc:747). */
typedef struct WrhipContext WrhipContext;
typedef struct LockedTexture LockedTexture;

/* YuvRangedColorSpace, composite.h (passed through to CompositeYUV). */
typedef int32_t YuvRangedColorSpace;

/* ---- state ------------------------------------------------------------ */
void ActiveTexture(GLenum texture);                                 /* fns:24  gl.cc:1376 */
void BindTexture(GLenum target, GLuint texture);                    /* fns:25  gl.cc:1590 */
void BindBuffer(GLenum target, GLuint buffer);                      /* fns:26  gl.cc:1594 */
void BindVertexArray(GLuint vao);                                   /* fns:27  gl.cc:1583 */
void BindFramebuffer(GLenum target, GLuint fb);                     /* fns:28  gl.cc:1598 */
void BindRenderbuffer(GLenum target, GLuint rb);                    /* fns:29  gl.cc:1608 */
void BlendFunc(GLenum srgb, GLenum drgb, GLenum sa, GLenum da);     /* fns:30  gl.cc:1325 */
void BlendColor(GLfloat r, GLfloat g, GLfloat b, GLfloat a);        /* fns:31  gl.cc:1336 */
void BlendEquation(GLenum mode);                                    /* fns:32  gl.cc:1341 */
void Enable(GLenum cap);                                            /* fns:33  gl.cc:1097 */
void Disable(GLenum cap);                                           /* fns:34  gl.cc:1111 */
/* ---- queries ---------------------------------------------------------- */
void GenQueries(GLsizei n, GLuint* result);                         /* fns:35  gl.cc:1383 */
void BeginQuery(GLenum target, GLuint id);                          /* fns:36  gl.cc:1540 */
void EndQuery(GLenum target);                                       /* fns:37  gl.cc:1556 */
void GetQueryObjectui64v(GLuint id, GLenum pname, GLuint64* params);/* fns:38  gl.cc:1571 */
/* ---- object creation -------------------------------------------------- */
void GenBuffers(int32_t n, GLuint* result);                         /* fns:39  gl.cc:1397 */
void GenTextures(int32_t n, GLuint* result);                        /* fns:40  gl.cc:1858 */
void GenFramebuffers(int32_t n, GLuint* result);                    /* fns:41  gl.cc:1896 */
void GenRenderbuffers(int32_t n, GLuint* result);                   /* fns:42  gl.cc:1873 */
/* ---- buffers ---------------------------------------------------------- */
void BufferData(GLenum target, GLsizeiptr size, const GLvoid* data,
                GLenum usage);                                      /* fns:43  gl.cc:2007 */
void BufferSubData(GLenum target, GLintptr offset, GLsizeiptr size,
                   const GLvoid* data);                             /* fns:44  gl.cc:2021 */
void* MapBuffer(GLenum target, GLbitfield access);                  /* fns:45  gl.cc:2030 */
void* MapBufferRange(GLenum target, GLintptr offset, GLsizeiptr length,
                     GLbitfield access);                            /* fns:46  gl.cc:2035 */
GLboolean UnmapBuffer(GLenum target);                               /* fns:52  gl.cc:2044 */
/* ---- textures / framebuffers ----------------------------------------- */
void TexStorage2D(GLenum target, GLint levels, GLenum internal_format,
                  GLsizei width, GLsizei height);                   /* fns:53  gl.cc:1732 */
void FramebufferTexture2D(GLenum target, GLenum attachment, GLenum textarget,
                          GLuint texture, GLint level);             /* fns:60  gl.cc:2070 */
GLenum CheckFramebufferStatus(GLenum target);                       /* fns:67  gl.cc:2362 */
void InvalidateFramebuffer(GLenum target, GLsizei num_attachments,
                           const GLenum* attachments);              /* fns:68  gl.cc:2534 */
void TexImage2D(GLenum target, GLint level, GLint internal_format,
                GLsizei width, GLsizei height, GLint border, GLenum format,
                GLenum ty, const void* data);                       /* fns:69  gl.cc:1818 */
void TexSubImage2D(GLenum target, GLint level, GLint xoffset, GLint yoffset,
                   GLsizei width, GLsizei height, GLenum format, GLenum ty,
                   const void* data);                               /* fns:80  gl.cc:1791 */
void GenerateMipmap(GLenum target);                                 /* fns:91  gl.cc:1830 */
/* ---- programs / vertex arrays ----------------------------------------- */
GLint GetUniformLocation(GLuint program, const GLchar* name);       /* fns:92  gl.cc:1507 */
void BindAttribLocation(GLuint program, GLuint index,
                        const GLchar* name);                        /* fns:93  gl.cc:1489 */
GLint GetAttribLocation(GLuint program, const GLchar* name);        /* fns:94  gl.cc:1498 */
void GenVertexArrays(int32_t n, GLuint* result);                    /* fns:95  gl.cc:1412 */
void VertexAttribPointer(GLuint index, GLint size, GLenum type_,
                         GLboolean normalized, GLsizei stride,
                         const GLvoid* offset);                     /* fns:96  gl.cc:1929 */
void VertexAttribIPointer(GLuint index, GLint size, GLenum type_,
                          GLsizei stride, const GLvoid* offset);    /* fns:104 gl.cc:1949 */
GLuint CreateShader(GLenum shader_type);                            /* fns:111 gl.cc:1425 */
void AttachShader(GLuint program, GLuint shader);                   /* fns:112 gl.cc:1439 */
GLuint CreateProgram(void);                                         /* fns:113 gl.cc:1455 */
void Uniform1i(GLint location, GLint v0);                           /* fns:114 gl.cc:2049 */
void Uniform4fv(GLint location, GLsizei count, const GLfloat* value);/* fns:115 gl.cc:2055 */
void UniformMatrix4fv(GLint location, GLsizei count, GLboolean transpose,
                      const GLfloat* value);                        /* fns:116 gl.cc:2061 */
/* ---- the hot path ----------------------------------------------------- */
void DrawElementsInstanced(GLenum mode, GLsizei count, GLenum type_,
                           GLintptr indices, GLsizei instancecount);/* fns:122 gl.cc:2702 */
void EnableVertexAttribArray(GLuint index);                         /* fns:129 gl.cc:1969 */
void VertexAttribDivisor(GLuint index, GLuint divisor);             /* fns:130 gl.cc:1996 */
void LinkProgram(GLuint program);                                   /* fns:131 gl.cc:1471 */
GLint GetLinkStatus(GLuint program);                                /* fns:132 gl.cc:1482 */
void UseProgram(GLuint program);                                    /* fns:133 gl.cc:1082 */
void SetViewport(GLint x, GLint y, GLsizei width, GLsizei height);  /* fns:134 gl.cc:1093 */
void FramebufferRenderbuffer(GLenum target, GLenum attachment,
                             GLenum renderbuffertarget,
                             GLuint renderbuffer);                  /* fns:135 gl.cc:2085 */
void RenderbufferStorage(GLenum target, GLenum internalformat, GLsizei width,
                         GLsizei height);                           /* fns:141 gl.cc:1910 */
void DepthMask(GLboolean flag);                                     /* fns:142 gl.cc:1350 */
void DepthFunc(GLenum func);                                        /* fns:143 gl.cc:1352 */
void SetScissor(GLint x, GLint y, GLsizei width, GLsizei height);   /* fns:144 gl.cc:1363 */
void ClearColor(GLfloat r, GLfloat g, GLfloat b, GLfloat a);        /* fns:145 gl.cc:1367 */
void ClearDepth(GLdouble depth);                                    /* fns:146 gl.cc:1374 */
void Clear(GLbitfield mask);                                        /* fns:147 gl.cc:2498 */
void ClearTexSubImage(GLenum target, GLint level, GLint xoffset, GLint yoffset,
                      GLint zoffset, GLsizei width, GLsizei height,
                      GLsizei depth, GLenum format, GLenum ty,
                      const void* data);                            /* fns:148 gl.cc:2370 */
void ClearTexImage(GLenum target, GLint level, GLenum format, GLenum ty,
                   const void* data);                               /* fns:161 gl.cc:2490 */
void ClearColorRect(GLuint fbo, GLint xoffset, GLint yoffset, GLsizei width,
                    GLsizei height, GLfloat r, GLfloat g, GLfloat b,
                    GLfloat a);                                     /* fns:162 gl.cc:2520 */
void PixelStorei(GLenum name, GLint param);                         /* fns:173 gl.cc:1612 */
void ReadPixels(GLint x, GLint y, GLsizei width, GLsizei height, GLenum format,
                GLenum ty, void* data);                             /* fns:174 gl.cc:2556 */
void Finish(void);                                                  /* fns:183 gl.cc:2802 */
void ShaderSourceByName(GLuint shader, const GLchar* name);         /* fns:184 gl.cc:1431 */
void TexParameteri(GLenum target, GLenum pname, GLint param);       /* fns:185 gl.cc:1854 */
void CopyImageSubData(GLuint src_name, GLenum src_target, GLint src_level,
                      GLint src_x, GLint src_y, GLint src_z, GLuint dst_name,
                      GLenum dst_target, GLint dst_level, GLint dst_x,
                      GLint dst_y, GLint dst_z, GLsizei src_width,
                      GLsizei src_height, GLsizei src_depth);       /* fns:186 gl.cc:2608 */
void CopyTexSubImage2D(GLenum target, GLint level, GLint xoffset, GLint yoffset,
                       GLint x, GLint y, GLsizei width,
                       GLsizei height);                             /* fns:203 gl.cc:2650 */
void BlitFramebuffer(GLint src_x0, GLint src_y0, GLint src_x1, GLint src_y1,
                     GLint dst_x0, GLint dst_y0, GLint dst_x1, GLint dst_y1,
                     GLbitfield mask, GLenum filter);               /* fns:213 composite.h:434 */
void GetIntegerv(GLenum pname, GLint* params);                      /* fns:225 gl.cc:1151 */
void GetBooleanv(GLenum pname, GLboolean* params);                  /* fns:226 gl.cc:1197 */
const char* GetString(GLenum name);                                 /* fns:227 gl.cc:1209 */
const char* GetStringi(GLenum name, GLuint index);                  /* fns:228 gl.cc:1226 */
GLenum GetError(void);                                              /* fns:229 gl.cc:1126 */
/* ---- swgl extensions -------------------------------------------------- */
void InitDefaultFramebuffer(int32_t x, int32_t y, int32_t width, int32_t height,
                            int32_t stride, void* buf);             /* fns:230 gl.cc:2302 */
void* GetColorBuffer(GLuint fbo, GLboolean flush, int32_t* width,
                     int32_t* height, int32_t* stride);             /* fns:238 gl.cc:2322 */
void ResolveFramebuffer(GLuint fbo);                                /* fns:245 gl.cc:2345 */
void SetTextureBuffer(GLuint tex, GLenum internal_format, GLsizei width,
                      GLsizei height, GLsizei stride, void* buf,
                      GLsizei min_width, GLsizei min_height);       /* fns:246 gl.cc:2354 */
void SetTextureParameter(GLuint tex, GLenum pname, GLint param);    /* fns:256 gl.cc:1834 */
void DeleteTexture(GLuint n);                                       /* fns:257 gl.cc:1865 */
void DeleteRenderbuffer(GLuint n);                                  /* fns:258 gl.cc:1890 */
void DeleteFramebuffer(GLuint n);                                   /* fns:259 gl.cc:1903 */
void DeleteBuffer(GLuint n);                                        /* fns:260 gl.cc:1404 */
void DeleteVertexArray(GLuint n);                                   /* fns:261 gl.cc:1419 */
void DeleteQuery(GLuint n);                                         /* fns:262 gl.cc:1390 */
void DeleteShader(GLuint shader);                                   /* fns:263 gl.cc:1451 */
void DeleteProgram(GLuint program);                                 /* fns:264 gl.cc:1460 */
LockedTexture* LockFramebuffer(GLuint fbo);                         /* fns:265 composite.h:485 */
LockedTexture* LockTexture(GLuint tex);                             /* fns:266 composite.h:497 */
void LockResource(LockedTexture* resource);                         /* fns:267 composite.h:510 */
void UnlockResource(LockedTexture* resource);                       /* fns:268 composite.h:518 */
void* GetResourceBuffer(LockedTexture* resource, int32_t* width,
                        int32_t* height, int32_t* stride);          /* fns:269 composite.h:540 */
void Composite(LockedTexture* locked_dst, LockedTexture* locked_src, GLint src_x,
               GLint src_y, GLsizei src_width, GLsizei src_height, GLint dst_x,
               GLint dst_y, GLsizei dst_width, GLsizei dst_height,
               GLboolean opaque, GLboolean flip_x, GLboolean flip_y,
               GLenum filter, GLint clip_x, GLint clip_y, GLsizei clip_width,
               GLsizei clip_height);                                /* fns:275 composite.h:560 */
void CompositeYUV(LockedTexture* locked_dst, LockedTexture* locked_y,
                  LockedTexture* locked_u, LockedTexture* locked_v,
                  YuvRangedColorSpace color_space, GLuint color_depth,
                  GLint src_x, GLint src_y, GLsizei src_width,
                  GLsizei src_height, GLint dst_x, GLint dst_y,
                  GLsizei dst_width, GLsizei dst_height, GLboolean flip_x,
                  GLboolean flip_y, GLint clip_x, GLint clip_y,
                  GLsizei clip_width, GLsizei clip_height);         /* fns:295 composite.h:1330 */
WrhipContext* CreateContext(void);                                  /* fns:317 gl.cc:2816 */
void ReferenceContext(WrhipContext* ctx);                           /* fns:318 gl.cc:2818 */
void DestroyContext(WrhipContext* ctx);                             /* fns:319 gl.cc:2825 */
void MakeCurrent(WrhipContext* ctx);                                /* fns:320 gl.cc:2808 */
size_t ReportMemory(WrhipContext* ctx,
                    size_t (*size_of_op)(const void* ptr));         /* fns:321 gl.cc:2840 */

/* ---- libwrhip additions (not part of the reference ABI) --------------- *
 * Introspection / measurement hooks used by bench.py and the tests.  None of
 * them is needed by the Rust side.                                        */

/* Last-flush statistics: kernel launches, raster-kernel GPU nanoseconds
 * (hipEvent on the context stream), algorithmic bytes of the raster launch. */
typedef struct WrhipStats {
  uint64_t flushes;            /* flush_all() calls that launched work      */
  uint64_t kernel_launches;    /* HIP kernel launches since reset           */
  uint64_t raster_launches;    /* launches of the tile raster kernel        */
  uint64_t raster_ns;          /* summed GPU time of raster launches (events) */
  uint64_t raster_algo_bytes;  /* summed algorithmic bytes      */
  uint64_t raster_pixels;      /* destination pixels owned by those launches */
  uint64_t prims;              /* instances rasterised                      */
  uint64_t h2d_bytes;          /* bytes uploaded host->HBM                  */
  uint64_t d2h_bytes;          /* bytes read back HBM->host                 */
  /* host time spent inside the library (std::chrono, nanoseconds): where the CPU side of a frame goes */
  uint64_t host_record_ns;     /* DrawElementsInstanced: state snapshot + instance bytes             */
  uint64_t host_upload_ns;     /* TexSubImage2D / TexImage2D / BufferData / BufferSubData: staging of uploaded bytes */
  uint64_t host_flush_ns;      /* flush: descriptor arena, staging copy, kernel launches (no waiting) */
  uint64_t host_wait_ns;       /* Finish / ReadPixels / queries: blocked on the stream               */
  uint64_t row_launches;       /* launches of the row kernels (wr_span_rows_kernel / wr_tile_rows_kernel); included in raster_launches */
} WrhipStats;
void WrhipGetStats(WrhipStats* out);
void WrhipResetStats(void);
/* Enable hipEvent timing of every kernel launch (one event pair and a sync per launch: for measurement runs only).
 * 1: every flush launches its own kernels at once (what a Finish per frame gives); 2: the launches stay where throughput
 * mode issues them -- raster launches held back to the next flush, the first of them fused with that flush's setup stage. */
void WrhipSetProfiling(int enabled);
/* Per-kernel-variant totals collected while profiling is on: kind 0 = upload scatter, 1 = setup stage, 3 = mask rows (wr_mask_rows_kernel), 2 = raster
 * kernel wr_raster_kernel<fmt, depth, 4, feat>, 4 = a chained run of thin R8 levels, 5 = wr_raster_dense_kernel, 6 / 7 = kinds 2 / 5
 * fused with the next flush's setup stage (wr_setup_raster[_dense]_kernel: setup bytes and workgroups added), 8 = wr_setup_rows_kernel
 * (3 fused likewise), 9 = wr_span_rows_kernel (the cs_blur / cs_scale targets of a level, a wave per target row piece), 10 = wr_tile_rows_kernel
 * (picture targets of a few large gradient / image prims, likewise), 11 = wr_setup_tile_rows_kernel (10 fused with the next flush's setup stage), 12 = a thin launch of the raster kernel (13: the same with the next flush's setup stage in front, wr_setup_raster_thin_kernel) (wr_raster_kernel<fmt, false, 1, feat>:
 * small levels, several workgroups per bin).  algo_bytes: the launch's algorithmic bytes:
 * raster launches count every destination pixel they own once (twice when the target's old content is loaded) plus
 * the source texels their draws can sample; the setup stage counts instance + descriptor + record bytes.  Returns the
 * number of entries written (<= max). */
typedef struct WrhipKernelStat {
  int32_t kind, fmt, depth, feat;
  uint64_t launches, ns, algo_bytes, workgroups;
} WrhipKernelStat;
int32_t WrhipGetKernelStats(WrhipKernelStat* out, int32_t max);
/* Restrict rasterisation to tile-rows owned by `rank` of `world` (multi-GPU
 * sharding by render-target strips). world<=1 disables. */
void WrhipSetShard(int rank, int world);
/* Restrict rasterisation of render target `tex` to pixel rows [y0,y1) (rows
 * outside are neither computed nor stored); y0==y1 removes the restriction;
 * y1<y0, or a range outside the texture, means the target is not this
 * process's at all: draws and clears into it are dropped when recorded (no
 * instance snapshot, staging, upload or setup work).  Used by the multi-GPU
 * harness to give each rank a screen-space strip. */
void WrhipSetTargetRows(GLuint tex, int32_t y0, int32_t y1);
/* Device pointer + geometry of a texture's HBM storage (for RCCL gather). */
void* WrhipGetTextureDevicePtr(GLuint tex, int32_t* width, int32_t* height,
                               int32_t* stride);
/* Texture id of an FBO's colour attachment (fbo 0 = default framebuffer). */
GLuint WrhipGetFramebufferTexture(GLuint fbo);
/* Name of the HIP device the context runs on, or NULL if none. */
const char* WrhipDeviceName(void);
/* Submit everything recorded so far to the context's HIP stream (pending draws, queued uploads, the
 * held-back composite level) WITHOUT waiting for it: after the call the stream holds all the work
 * whose results a consumer enqueued behind it on the same stream will see.  The multi-GPU harness
 * orders its framebuffer-strip copy and the RCCL all-gather this way instead of a Finish per frame. */
void WrhipFlush(void);
/* ... the same, leaving this flush's own raster launches held back for the next flush (returns 1 if they are): see wrhip.hip */
int WrhipFlushHeld(void);
/* The context's hipStream_t (NULL in the host simulation), e.g. for torch.cuda.ExternalStream. */
void* WrhipGetStream(void);

#ifdef __cplusplus
}
#endif
#endif /* WRHIP_H */
