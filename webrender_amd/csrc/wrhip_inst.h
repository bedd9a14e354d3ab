// The raster kernel instantiations, spread over translation units so that their device compiles run in parallel (one hipcc
// process spends minutes on the largest variants; `make -j` builds wrhip_inst.hip once per group).  X(KERNEL, SIGNATURE,
// FMT, DEPTH, R, FEAT): wrhip.hip declares every one of them `extern template` and launches them, wrhip_inst.hip defines the
// group WR_INST_GROUP names.  FEAT values: WrFeat sets (47 = TEX | R8TEX | GENERIC | BLUR | SHADE, 7 = TEX | R8TEX | GENERIC,
// 5 = TEX | GENERIC, 12 = GENERIC | BLUR, 28 = GENERIC | BLUR | CLIP).
#pragma once
#define WR_SIG_PLAIN (const WrTargetDesc*, int, const WrDrawDesc*, const WrPrim*, const WrRec*, const WrAux*, const float*, unsigned long long*, int)
#define WR_SIG_FUSED (WrSetupArgs, int, const WrTargetDesc*, int, const WrDrawDesc*, const WrPrim*, const WrRec*, const WrAux*, const float*, unsigned long long*, int)
#define WR_INST_1(X) X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, true, 4, 47)
#define WR_INST_2(X) X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, false, 4, 47)
#define WR_INST_3(X)                                                                                                          \
  X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, true, 4, 7) X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, false, 4, 7)    \
  X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, true, 4, 5) X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, false, 4, 5)    \
  X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, true, 4, 0) X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, false, 4, 0)
#define WR_INST_4(X)                                                                                                                      \
  X(wr_setup_raster_kernel, WR_SIG_FUSED, WR_FMT_RGBA8, true, 4, 7) X(wr_setup_raster_kernel, WR_SIG_FUSED, WR_FMT_RGBA8, false, 4, 7)    \
  X(wr_setup_raster_kernel, WR_SIG_FUSED, WR_FMT_RGBA8, true, 4, 5) X(wr_setup_raster_kernel, WR_SIG_FUSED, WR_FMT_RGBA8, false, 4, 5)    \
  X(wr_setup_raster_kernel, WR_SIG_FUSED, WR_FMT_RGBA8, true, 4, 0) X(wr_setup_raster_kernel, WR_SIG_FUSED, WR_FMT_RGBA8, false, 4, 0)
#define WR_INST_5(X)                                                                                                                            \
  X(wr_raster_dense_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, true, 4, 7) X(wr_raster_dense_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, false, 4, 7)          \
  X(wr_setup_raster_dense_kernel, WR_SIG_FUSED, WR_FMT_RGBA8, true, 4, 7) X(wr_setup_raster_dense_kernel, WR_SIG_FUSED, WR_FMT_RGBA8, false, 4, 7)
#define WR_INST_6(X)                                                                                                  \
  X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_R8, false, 4, 0) X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_R8, false, 4, 12) \
  X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_R8, false, 4, 28)
// (thin colour launches: picture blur / down-scale chains into RGBA8 targets of a few bins, 256-thread workgroups, four per bin)
#define WR_INST_7(X) X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, false, 1, 5) X(wr_raster_kernel, WR_SIG_PLAIN, WR_FMT_RGBA8, false, 1, 47)
// (... and the thin colour launch that carries the next flush's setup stage)
#define WR_INST_8(X) X(wr_setup_raster_thin_kernel, WR_SIG_FUSED, WR_FMT_RGBA8, false, 1, 5)
#define WR_INST_ALL(X) WR_INST_1(X) WR_INST_2(X) WR_INST_3(X) WR_INST_4(X) WR_INST_5(X) WR_INST_6(X) WR_INST_7(X) WR_INST_8(X)
#define WR_INST_GROUPS 8
