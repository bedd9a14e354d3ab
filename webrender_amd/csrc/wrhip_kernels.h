// wrhip_kernels.h -- gfx950 kernels of the WebRender draw backend (see wrhip_types.h for the data flow).
//
// A flush is a setup stage and a few raster launches:
//   setup stage        one thread per instance (wr_setup_body): the vertex stage of the bound WebRender shader + swgl's draw_quad
//                      setup (swgl/src/rasterize.h:1549-1633) in strict fp32 (compiled with -ffp-contract=off: the op order and
//                      rounding of the reference's gcc build), side records, ballot binning; normally carried by the first
//                      workgroups of the PREVIOUS flush's longest raster launch (wr_setup_raster_kernel & co.)
//   bin raster         one 256-thread workgroup per 64x64 bin (wr_raster_body): each lane owns 4x4 pixels in registers, prims are
//                      applied in submission order with swgl's integer blend math (swgl/src/blend.h:416-735), pixels are read at
//                      most once and written once; rect-only bins fold the whole blend chain per cell (wr_raster_cells)
//   row kernels        a wave per row (piece): cs_clip_* prims into the mask-row store, cs_blur / cs_scale targets, picture
//                      targets of a few large prims
//
// No MFMA: this is integer pixel work bound by HBM traffic / instruction issue.
#pragma once
#include "wrhip_types.h"

// The kernels, by stage (one translation unit each way: the parts are not self-contained headers, they follow one another)
#include "wrhip_k_setup.h"      // vertex stages, draw_quad setup, binning: the setup stage
#include "wrhip_k_pixels.h"     // span / main() evaluators of every program
#include "wrhip_k_rows.h"       // mask rows, span rows, tile rows
#include "wrhip_k_raster.h"     // bin raster and the kernel entry points
