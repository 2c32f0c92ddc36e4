// Last layer of the up-sampler as its own kernel (reference CVSR_freq.py:2607 / :2683 conv_last0, mmedit fcvsr.py: 3x3, 64 -> 1 or 3
// channels at the OUTPUT resolution, + the bilinear x4 base the result already holds):   out += bias + conv3x3(u2)
// for the models whose up-convs are 3x3 (GShiftNet, the RGB twins), where fcvsr_tail_fused does not apply.  The generic MFMA kernel
// pads the 1..3 output channels to a 32-wide N tile (1.6 ms per 16 clips, 97 % of the matrix work wasted); here the layer is what it
// is - a memory-bound pass over u2 (128 bytes per output pixel):
//   * the 10 x 34 halo tile of an 8 x 32 output tile goes to LDS once (144-byte padded pixel rows);
//   * "taps are output columns": P[pixel][tap*C + c] = sum_k u2[pixel][k] * w[c][k][tap] is a (340 x 64) x (64 x 9C) GEMM on
//     v_mfma_f32_16x16x32 (2 MFMAs per 16 pixels and 16 columns) instead of 576 C multiply-adds per pixel on the vector ALU;
//   * out[c][y][x] += bias[c] + sum_tap P[(y+dy, x+dx)][tap*C + c]  - 9 LDS reads per output value.
#include "common.h"
#include "mfma_util.h"

namespace fcvsr {

typedef __attribute__((ext_vector_type(4))) float f32x4v_t;

constexpr int kClTH = 8, kClTW = 32, kClHH = kClTH + 2, kClHW = kClTW + 2, kClNHP = kClHH * kClHW;   // 340 halo pixels
constexpr int kClRow = 64 + 8;                               // halfwords per halo pixel in LDS
constexpr int kClNPT = (kClNHP + 15) / 16;                   // 22 pixel groups of 16

struct ClArgs {
  View u;                  // (B, H, W, 64) 16-bit, dense
  const uint16_t* w;       // [16 * NT][64]: row = tap * C + c (tap = ky*3 + kx), zero rows past 9 C; MFMA dtype
  const float* bias;       // C floats or null
  View out;                // (B, H, W, C) f32 view of the NCHW result, read-modify-write
  int B, H, W, C, tiles_x, tiles_y;
};

template <bool BF16>
__device__ __forceinline__ f32x4v_t mfma16cl(uint4 a, uint4 b, f32x4v_t c) {
  if (BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// NT = column tiles of 16 (C = 1: 9 columns -> NT = 1; C = 3: 27 columns -> NT = 2)
template <bool BF16, int NT>
__global__ __launch_bounds__(256, 3) void conv_last_kernel(ClArgs a) {
  constexpr int PROW = 16 * NT + 1;                          // floats per pixel of the tap table (odd stride)
  __shared__ __align__(16) uint16_t u_s[kClNHP * kClRow];    // 48,960 bytes; the tap table overwrites it
  float* p_s = reinterpret_cast<float*>(u_s);
  static_assert(kClNPT * 16 * (16 * 2 + 1) * 4 <= kClNHP * kClRow * 2, "tap table must fit into the halo tile");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = t & 7, loc = t >> 3;
    t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  const int per_img = a.tiles_x * a.tiles_y;
  const int b = t / per_img, t2 = t - b * per_img;
  const int Y0 = (t2 / a.tiles_x) * kClTH, X0 = (t2 % a.tiles_x) * kClTW;
  // ---- halo tile -> LDS: 8 lanes x 16 bytes per pixel, zeros outside the image (the zero padding of the layer) ----------------
  {
    const int q = tid & 7, p0 = tid >> 3;
    const uint16_t* ub = reinterpret_cast<const uint16_t*>(a.u.p) + (long long)b * a.u.sb + q * 8;
    // all 11 loads of a lane first (clamped, unconditional), then the LDS writes: as a rolled loop with predicated loads the
    // tile arrived in 11 round trips (2.75 TB/s for a kernel that only streams u2)
    constexpr int NI = (kClNHP + 31) / 32;
    uint4 v[NI];
    bool ok[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int hp = p0 + 32 * i;
      hp = hp < kClNHP ? hp : kClNHP - 1;
      const int hy = hp / kClHW, hx = hp - hy * kClHW;
      const int Y = Y0 - 1 + hy, X = X0 - 1 + hx;
      ok[i] = (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W;
      const int cy = Y < 0 ? 0 : (Y > a.H - 1 ? a.H - 1 : Y), cx = X < 0 ? 0 : (X > a.W - 1 ? a.W - 1 : X);
      v[i] = *reinterpret_cast<const uint4*>(ub + (long long)cy * a.u.sy + (long long)cx * a.u.sx);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int hp = p0 + 32 * i;
      if (hp < kClNHP) *reinterpret_cast<uint4*>(u_s + hp * kClRow + q * 8) = ok[i] ? v[i] : make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();
  // ---- P[pixel][column] ---------------------------------------------------------------------------------------------------------------
  {
    const int r16 = lane & 15, g4 = lane >> 4;
    uint4 w0[NT], w1[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      w0[n] = *reinterpret_cast<const uint4*>(a.w + (n * 16 + r16) * 64 + g4 * 8);
      w1[n] = *reinterpret_cast<const uint4*>(a.w + (n * 16 + r16) * 64 + g4 * 8 + 32);
    }
    constexpr int NJ = (kClNPT + 3) / 4;
    f32x4v_t pacc[NJ][NT];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int pt = wave + 4 * j;
      int px = pt * 16 + r16;
      px = px < kClNHP ? px : kClNHP - 1;
      const uint4 a0 = *reinterpret_cast<const uint4*>(u_s + px * kClRow + g4 * 8);
      const uint4 a1 = *reinterpret_cast<const uint4*>(u_s + px * kClRow + g4 * 8 + 32);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x4v_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma16cl<BF16>(a0, w0[n], acc);
        pacc[j][n] = mfma16cl<BF16>(a1, w1[n], acc);
      }
    }
    __syncthreads();                                         // every wave has read its pixels: the tile may be overwritten
    // D[row][col]: col = lane & 15 (column of the tile), row = 4 * (lane >> 4) + i (pixel within the group)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int pt = wave + 4 * j;
      if (pt < kClNPT) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 4; ++i) p_s[(pt * 16 + 4 * g4 + i) * PROW + n * 16 + r16] = pacc[j][n][i];
      }
    }
  }
  __syncthreads();
  // ---- out += bias + sum over the 9 taps -------------------------------------------------------------------------------------------------
  {
    const int ty = tid >> 5, tx = tid & 31;
    const int oy = Y0 + ty, ox = X0 + tx;
    if (oy < a.H && ox < a.W) {
      float* op = a.out.p + (long long)b * a.out.sb + (long long)oy * a.out.sy + (long long)ox * a.out.sx;
      for (int c = 0; c < a.C; ++c) {
        float s = op[(long long)c * a.out.sc] + (a.bias ? a.bias[c] : 0.f);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) s += p_s[((ty + dy) * kClHW + tx + dx) * PROW + (dy * 3 + dx) * a.C + c];
        op[(long long)c * a.out.sc] = s;
      }
    }
  }
}

}  // namespace fcvsr

using namespace fcvsr;

// out (B,H,W,C view, f32, any strides) += bias + conv3x3(u) with u (B,H,W,64) dense 16-bit; w: [16 or 32][64] rows tap*C + c in
// u's dtype (rows past 9C zero); C in 1..3.
extern "C" int fcvsr_conv_last(const fcvsr_view* u, const void* w, const float* bias, int B, int H, int W, int C, const fcvsr_view* out,
                               void* stream) {
  FCVSR_CHECK_ARG(u && u->ptr && w && out && out->ptr, "null argument");
  FCVSR_CHECK_ARG((u->dtype == FCVSR_BF16 || u->dtype == FCVSR_F16) && u->c == 64 && u->sc == 1 && ((uintptr_t)u->ptr % 16) == 0 &&
                      u->sx % 8 == 0 && u->sy % 8 == 0 && u->sb % 8 == 0, "u: 64 contiguous 16-bit channels, 16-byte aligned");
  FCVSR_CHECK_ARG(out->dtype == FCVSR_F32 && C >= 1 && C <= 3 && out->c == C, "out: 1..3 f32 channels");
  FCVSR_CHECK_ARG(((uintptr_t)w % 16) == 0 && B > 0 && H > 0 && W > 0, "bad arguments");
  ClArgs a;
  a.u = to_view(*u); a.w = (const uint16_t*)w; a.bias = bias; a.out = to_view(*out);
  a.B = B; a.H = H; a.W = W; a.C = C;
  a.tiles_x = cdiv(W, kClTW); a.tiles_y = cdiv(H, kClTH);
  const dim3 grid(B * a.tiles_x * a.tiles_y);
  hipStream_t st = (hipStream_t)stream;
  const bool bf = u->dtype == FCVSR_BF16;
  if (C == 1) {
    if (bf) hipLaunchKernelGGL((conv_last_kernel<true, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_last_kernel<false, 1>), grid, dim3(256), 0, st, a);
  } else {
    if (bf) hipLaunchKernelGGL((conv_last_kernel<true, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_last_kernel<false, 2>), grid, dim3(256), 0, st, a);
  }
  FCVSR_LAUNCH_CHECK();
  return 0;
}
