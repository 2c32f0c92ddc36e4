// Fused end of the up-sampler of GShiftNet_S (reference CVSR_freq.py:2605-2607, :2642-2645):
//     out += conv_last0( PReLU( PixelShuffle2( upconv2(u1) ) ) )          upconv2: 1x1, 64 -> 256; conv_last0: 3x3, 64 -> 1
// The 64-channel tensor at the output resolution (B x 4H x 4W x 64: the largest activation of the whole path, 118 MB per
// clip in 16-bit) is never written.  One workgroup produces an 8 x 32 tile of output pixels:
//   1. GEMM 1 (v_mfma 32x32x16): the 6 x 18 pixels of u1 under the tile's 10 x 34 halo times the 256 x 64 weight block.  Wave
//      w owns sub-pixel w (rows w*64 .. w*64+63 of the sub-pixel-major packing), so PixelShuffle is just the LDS address
//      the result is written to; bias, PReLU, zero padding outside the image and the rounding to the MFMA dtype (what
//      the stand-alone layer would have stored) happen on the way.
//   2. GEMM 2 (v_mfma 16x16x32): conv_last0 as "taps are output columns": P[pixel][tap] = sum_c u2[pixel][c] * w[tap][c] for
//      the 340 halo pixels - 44 small MFMAs instead of 576 multiply-adds per output pixel on the vector ALU.
//   3. out[y][x] += bias + sum_tap P[y+dy][x+dx][tap]   (out holds the bilinear x4 base skip).
// HBM traffic per output pixel: 14 bytes of u1 (with halo) + 8 bytes of out, instead of 128 written + ~170 read.
#include <stdlib.h>
#include "common.h"
#include "mfma_util.h"

namespace fcvsr {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int kTfTH = 8, kTfTW = 32;                       // output tile
constexpr int kTfHH = kTfTH + 2, kTfHW = kTfTW + 2;        // halo of the 3x3 convolution (output resolution)
constexpr int kTfNHP = kTfHH * kTfHW;                      // 340
constexpr int kTfUH = kTfTH / 2 + 2, kTfUW = kTfTW / 2 + 2;   // 6 x 18 pixels of u1
constexpr int kTfNU = kTfUH * kTfUW;                       // 108
constexpr int kTfRow = 64 + 8;                             // halfwords per halo pixel in LDS (padded: conflict-free fragments)
constexpr int kTfPRow = 9;                                 // floats per pixel of the tap table (odd stride)
constexpr int kTfNPT = (kTfNHP + 15) / 16;                 // 22 pixel tiles of GEMM 2

struct TailArgs {
  View u1;                 // (B, H2, W2, 64), 16-bit
  const uint16_t* w2;      // [256][64] upconv2, rows sub-pixel-major ((2i+j)*64 + c), MFMA dtype
  const float* b2;         // [256] bias in the same row order (may be null)
  const float* slope;      // PReLU slope (one shared scalar, :2609)
  const uint16_t* wl;      // [16][64] conv_last0: row = tap (ky*3+kx), rows 9..15 zero
  const float* bl;         // conv_last0 bias (1 value, may be null)
  View out;                // (B, 2*H2, 2*W2, 1) f32, read-modify-write
  int B, H2, W2, tiles_x, tiles_y;
  int ntiles;              // B * tiles_x * tiles_y, split into contiguous runs over the launched workgroups
};

template <bool BF16>
__device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {
  if (BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <bool BF16>
__global__ __launch_bounds__(256, 3) void tail_fused_kernel(TailArgs a) {
  // 48 KiB: the tap table P of step 2 overwrites the u2 tile it was computed from (3 workgroups per CU)
  __shared__ __align__(16) uint16_t u2_s[kTfNHP * kTfRow];
  float* p_s = reinterpret_cast<float*>(u2_s);
  static_assert(kTfNPT * 16 * kTfPRow * 4 <= kTfNHP * kTfRow * 2, "tap table must fit into the u2 tile");
  __shared__ __align__(16) float b2_s[256];
  // Tile runs: a workgroup (three per CU) walks a contiguous run of tiles with its wave's 64 x 64 block of upconv2 weights in
  // registers and the bias in LDS for the whole run.  One tile per workgroup re-read the 32 KB weight block from L2 for every
  // 256 output pixels: 128 bytes per output pixel, six times the 22 bytes the pixel itself moves.
  int t_begin, t_end;
  {                                                        // contiguous runs of tiles per XCD (halo rows of u1 hit its L2)
    const int g = blockIdx.x, nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = g & 7, loc = g >> 3;
    const int ci = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    t_begin = (int)((long long)ci * a.ntiles / nwg);
    t_end = (int)((long long)(ci + 1) * a.ntiles / nwg);
  }
  uint4 wf[2][4], wl0, wl1;
  {
    const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave0 = tid0 >> 6, r0 = lane0 & 31, h0 = lane0 >> 5;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint16_t* wp = a.w2 + ((wave0 * 2 + q) * 32 + r0) * 64 + h0 * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wf[q][kk] = *reinterpret_cast<const uint4*>(wp + kk * 16);
    }
    wl0 = *reinterpret_cast<const uint4*>(a.wl + (lane0 & 15) * 64 + (lane0 >> 4) * 8);
    wl1 = *reinterpret_cast<const uint4*>(a.wl + (lane0 & 15) * 64 + (lane0 >> 4) * 8 + 32);
    b2_s[tid0] = a.b2 ? a.b2[tid0] : 0.f;
  }
  const float slope = a.slope[0];
  const float blast = a.bl ? a.bl[0] : 0.f;
  const int per_img = a.tiles_x * a.tiles_y;
  const int HH = 2 * a.H2, WW = 2 * a.W2;
  __syncthreads();
#pragma unroll 1
  for (int t = t_begin; t < t_end; ++t) {
  // lane-derived offsets are recomputed per tile: hoisted out of the loop they cost registers the tile body has no room for
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  const int b = t / per_img;
  const int t2 = t - b * per_img;
  const int Y0 = (t2 / a.tiles_x) * kTfTH, X0 = (t2 % a.tiles_x) * kTfTW;
  // this thread's output pixel: its current value (the base skip) is requested now and consumed at the very end
  const int oy = Y0 + (tid >> 5), ox = X0 + (tid & 31);
  const bool olive = oy < HH && ox < WW;
  float* op = a.out.p + (long long)b * a.out.sb + (long long)(olive ? oy : 0) * a.out.sy + (long long)(olive ? ox : 0) * a.out.sx;
  const float base = *op;

  // ---- GEMM 1: u2 = PReLU(W2 . u1 + b2), wave = sub-pixel -----------------------------------------------------------------
  {
    const int r = lane & 31, h = lane >> 5;
    const uint16_t* ub = reinterpret_cast<const uint16_t*>(a.u1.p) + (long long)b * a.u1.sb + h * 8;
    uint4 kf[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      int p = nt * 32 + r;
      p = p < kTfNU ? p : kTfNU - 1;
      const int uy = p / kTfUW, ux = p - uy * kTfUW;
      int gy = (Y0 >> 1) - 1 + uy, gx = (X0 >> 1) - 1 + ux;
      gy = gy < 0 ? 0 : (gy > a.H2 - 1 ? a.H2 - 1 : gy);    // clamped pixels only feed halo positions outside the image,
      gx = gx < 0 ? 0 : (gx > a.W2 - 1 ? a.W2 - 1 : gx);    // which are stored as zeros below
      const uint16_t* up = ub + (long long)gy * a.u1.sy + (long long)gx * a.u1.sx;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) kf[nt][kk] = *reinterpret_cast<const uint4*>(up + kk * 16);
    }
    const int sy = wave >> 1, sx = wave & 1;                // this wave's sub-pixel
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float4 bq[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bq[g] = *reinterpret_cast<const float4*>(b2_s + wave * 64 + q * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        f32x16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = mfma<BF16>(wf[q][kk], kf[nt][kk], acc);
        const int p = nt * 32 + r;
        const int uy = p / kTfUW, ux = p - uy * kTfUW;
        const int hr = 2 * uy + sy - 1, hc = 2 * ux + sx - 1;   // position inside the 10 x 34 halo
        const int Y = Y0 - 1 + hr, X = X0 - 1 + hc;
        const bool keep = p < kTfNU && hr >= 0 && hr < kTfHH && hc >= 0 && hc < kTfHW;
        const bool inside = Y >= 0 && Y < HH && X >= 0 && X < WW;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = q * 32 + 8 * g + 4 * h;             // acc[4g..4g+3] = channels c..c+3 of sub-pixel `wave`
          float4 v = make_float4(acc[4 * g] + bq[g].x, acc[4 * g + 1] + bq[g].y, acc[4 * g + 2] + bq[g].z, acc[4 * g + 3] + bq[g].w);
          // PReLU as max(x,0) + slope*min(x,0): the same values as the select form, in packed-f32 instructions
          v.x = fmaf(slope, fminf(v.x, 0.f), fmaxf(v.x, 0.f)); v.y = fmaf(slope, fminf(v.y, 0.f), fmaxf(v.y, 0.f));
          v.z = fmaf(slope, fminf(v.z, 0.f), fmaxf(v.z, 0.f)); v.w = fmaf(slope, fminf(v.w, 0.f), fmaxf(v.w, 0.f));
          uint2 pk = cvt4<BF16>(v);
          if (!inside) pk = make_uint2(0u, 0u);              // zero padding of conv_last0
          if (keep) *reinterpret_cast<uint2*>(u2_s + (hr * kTfHW + hc) * kTfRow + c) = pk;
        }
      }
    }
  }
  __syncthreads();

  // ---- GEMM 2: P[pixel][tap] = u2[pixel][:] . wl[tap][:] ---------------------------------------------------------------------
  {
    const int r16 = lane & 15, g4 = lane >> 4;
    f32x4_t pacc[(kTfNPT + 3) / 4];
#pragma unroll
    for (int j = 0; j < (kTfNPT + 3) / 4; ++j) {
      const int pt = wave + 4 * j;
      int px = pt * 16 + r16;
      px = px < kTfNHP ? px : kTfNHP - 1;
      const uint4 a0 = *reinterpret_cast<const uint4*>(u2_s + px * kTfRow + g4 * 8);
      const uint4 a1 = *reinterpret_cast<const uint4*>(u2_s + px * kTfRow + g4 * 8 + 32);
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      acc = mfma16<BF16>(a0, wl0, acc);
      pacc[j] = mfma16<BF16>(a1, wl1, acc);
    }
    __syncthreads();                                       // every wave has read its u2 pixels: the tile may be overwritten
    // D[row][col]: col = lane & 15 (tap), row = 4 * (lane >> 4) + i (pixel within the 16-pixel tile)
#pragma unroll
    for (int j = 0; j < (kTfNPT + 3) / 4; ++j) {
      const int pt = wave + 4 * j;
      if (pt < kTfNPT && r16 < 9) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p_s[(pt * 16 + 4 * g4 + i) * kTfPRow + r16] = pacc[j][i];
      }
    }
  }
  __syncthreads();

  // ---- out += bias + sum over the 9 taps ---------------------------------------------------------------------------------------
  {
    const int ty = tid >> 5, tx = tid & 31;
    if (olive) {
      float s = base + blast;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) s += p_s[((ty + dy) * kTfHW + tx + dx) * kTfPRow + dy * 3 + dx];
      *op = s;
    }
  }
  __syncthreads();                                         // the tap table is read: the next tile's u2 overwrites it
  }
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_tail_fused(const fcvsr_view* u1, const void* w2, const float* b2, const float* slope, const void* wl,
                                const float* bl, int B, int H2, int W2, const fcvsr_view* out, void* stream) {
  FCVSR_CHECK_ARG(u1 && u1->ptr && w2 && slope && wl && out && out->ptr, "null argument");
  FCVSR_CHECK_ARG((u1->dtype == FCVSR_BF16 || u1->dtype == FCVSR_F16) && u1->c == 64 && u1->sc == 1 &&
                      ((uintptr_t)u1->ptr % 16) == 0 && u1->sx % 8 == 0 && u1->sy % 8 == 0 && u1->sb % 8 == 0,
                  "u1: 64 contiguous 16-bit channels, 16-byte aligned");
  FCVSR_CHECK_ARG(out->dtype == FCVSR_F32 && out->c == 1, "out: one f32 channel");
  FCVSR_CHECK_ARG(((uintptr_t)w2 % 16) == 0 && ((uintptr_t)wl % 16) == 0 && (b2 == nullptr || ((uintptr_t)b2 % 16) == 0),
                  "weights / bias must be 16-byte aligned");
  FCVSR_CHECK_ARG(B > 0 && H2 > 0 && W2 > 0, "bad sizes");
  TailArgs a;
  a.u1 = to_view(*u1); a.w2 = (const uint16_t*)w2; a.b2 = b2; a.slope = slope; a.wl = (const uint16_t*)wl; a.bl = bl;
  a.out = to_view(*out); a.B = B; a.H2 = H2; a.W2 = W2;
  a.tiles_x = cdiv(2 * W2, kTfTW);
  a.tiles_y = cdiv(2 * H2, kTfTH);
  a.ntiles = B * a.tiles_x * a.tiles_y;
  static int wgs[64] = {0};                                // three workgroups per CU (48 KB of LDS, 168 registers each)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!wgs[dev]) {
    hipDeviceProp_t prop;
    wgs[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? 3 * prop.multiProcessorCount : 768;
  }
  static const int tpw = getenv("FCVSR_TAIL_TPW") ? atoi(getenv("FCVSR_TAIL_TPW")) : 0;   // tiles per workgroup (experiments)
  int nwg = tpw > 0 ? cdiv(a.ntiles, tpw) : wgs[dev];
  nwg = nwg < 8 ? 8 : nwg / 8 * 8;
  dim3 grid(nwg < a.ntiles ? nwg : a.ntiles);
  hipStream_t st = (hipStream_t)stream;
  if (u1->dtype == FCVSR_BF16) hipLaunchKernelGGL(tail_fused_kernel<true>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(tail_fused_kernel<false>, grid, dim3(256), 0, st, a);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
