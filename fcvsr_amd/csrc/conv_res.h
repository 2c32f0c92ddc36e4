// LDS-resident-weight 3x3 MFMA convolution (conv_res.hip): argument block shared with the dispatcher in conv_mfma.hip.
#pragma once
#include "common.h"

namespace fcvsr {

struct ResGroup {
  View src, res[2], dst;
  float* gc_partial;      // ContextBlock partials [B][tiles_y*tiles_x*4][cout+2] (one per wave pair of rows), or nullptr
  int B, H, W;
  int tiles_x, tiles_y;   // 8 x 32 pixel tiles
  int tile_begin;
};

struct ResArgs {
  int n_groups;
  ResGroup g[3];
  int total_tiles;
  int cin, cout, cout_pad, cin_pad;
  const uint16_t* w;      // [tap][cout_pad][cin_pad], 16-bit, cin contiguous (pack_conv_weight_mfma)
  const float* bias;
  int act;
  float slope;
  const float* slope_ptr;
  float rs[2];
  int n_res, res16;
  int ps;                 // PixelShuffle(2) store (cin 64, cout % 256 == 0, rows packed sub-pixel-major): cout block nb is sub-pixel nb / (cout/256)
  const void* zeros;      // >= 16 zero bytes: source of the halo pixels that lie outside the image
  int variant;            // 1: conv3_res_kernel (round 2), 3: conv3_res3_kernel (three wave groups: copy / multiply / store) where it applies
  int dbg;                // profiling ablations (FCVSR_RES_DBG): 1 skip staging, 2 skip MFMA, 4 skip stores
  unsigned long long* stamps;   // diagnostic (FCVSR_RES_STAMPS=1): s_memtime stamps of workgroup 0, [wave 8][phase 64][slot 8]; else nullptr
};

// returns hipSuccess, or hipErrorInvalidValue when the shape is not one the kernel is built for
hipError_t launch_conv3_res(const ResArgs& a, bool bf16, bool dst16, hipStream_t st);
bool conv3_res_supports(int cin, int cout);
// 8 x 32 tiles of one problem
inline int conv3_res_tiles(int B, int H, int W) { return B * ((H + 7) / 8) * ((W + 31) / 32); }

}  // namespace fcvsr
