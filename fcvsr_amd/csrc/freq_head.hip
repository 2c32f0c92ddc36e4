// Fused narrow 1x1 stacks of MGAAbk on the spectrum grid (reference CVSR_freq.py:1379-1385 convcorr, :1392-1396 convcrt,
// applied at :1472-1488):   out(4 ch, f32) = W_last . relu( [W_mid . relu]( W_0 . x ) ),   x: 128 channels, hidden width 64.
//   NHID = 2: convcorr away from the CorrBlock strip (input: the bf16 offset spectra, 128 -> 64 -> 64 -> 4)
//   NHID = 1: convcrt (input: the f32 spectrum of the centre group, 128 -> 64 -> 4)
// Same scheme as freq_mlp3_kernel: 128 pixels per workgroup, weights are the MFMA A operand (rows = couts), the wave's 32
// pixels the B operand, hidden activations stay in LDS rows private to the wave.  The 4 output channels are rows 0..3 of
// the last 32-row tile, i.e. acc[0..3] of lanes 0..31: one 16-byte store per pixel, no transposition.
#include "common.h"
#include "mfma_util.h"

namespace fcvsr {

constexpr int kFhPix = 128, kFhLD = 64 + 8;

struct FreqHeadArgs {
  const void* x;            // (npix, 128): bf16 (dense rows of sx halfwords) or f32 (rows sx floats apart)
  long long sx;
  float* out;               // (npix, 4) f32 dense
  int npix;
  const uint16_t* w0;       // [>=64][128] bf16, cin contiguous
  const uint16_t* w1;       // [>=64][64]   (NHID = 2 only)
  const uint16_t* wl;       // [>=32][64]: rows 0..3 live
};

template <bool SRC_F32, int NHID>
__global__ __launch_bounds__(256, 4) void freq_head_kernel(FreqHeadArgs a) {
  __shared__ __align__(16) uint16_t A_s[kFhPix * kFhLD];      // input chunk, later the hidden tile
  __shared__ __align__(16) uint16_t B_s[64 * kFhLD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int flat0 = blockIdx.x * kFhPix;
  const int npix = a.npix;
  f32x16_t acc[2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nf][i] = 0.f;
  };
  // weight chunk [ROWS couts][64 cin] -> B_s (ROWS = 64: 512 pieces, 2 per thread; ROWS = 32: 1 per thread)
  auto stage_w = [&](const uint16_t* w, int cin_pad, int c0, int rows) {
    uint4 v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = (tid >> 3) + 32 * u;
      v[u] = make_uint4(0, 0, 0, 0);
      if (row < rows) v[u] = *reinterpret_cast<const uint4*>(w + (long long)row * cin_pad + c0 + (tid & 7) * 8);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<uint4*>(B_s + ((tid >> 3) + 32 * u) * kFhLD + (tid & 7) * 8) = v[u];
  };
  auto mma_chunk = [&](const uint16_t* prow, int ntile) {
    const uint16_t* wrow = B_s + r * kFhLD + h * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint4 pf = *reinterpret_cast<const uint4*>(prow + kk * 16);
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
        if (nf < ntile) acc[nf] = mfma<true>(*reinterpret_cast<const uint4*>(wrow + nf * 32 * kFhLD + kk * 16), pf, acc[nf]);
    }
  };
  auto store_hidden = [&]() {                            // relu, bf16, row of pixel r: couts nf*32 + 8g + 4h + (0..3)
    uint16_t* trow = A_s + (wave * 32 + r) * kFhLD;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = make_float4(fmaxf(acc[nf][4 * g], 0.f), fmaxf(acc[nf][4 * g + 1], 0.f), fmaxf(acc[nf][4 * g + 2], 0.f),
                                     fmaxf(acc[nf][4 * g + 3], 0.f));
        *reinterpret_cast<uint2*>(trow + nf * 32 + 8 * g + 4 * h) = cvt4<true>(v);
      }
  };

  // ---- layer 0: 128 -> 64, relu (two 64-channel chunks) ------------------------------------------------------------------
  zero_acc();
  for (int c0 = 0; c0 < 128; c0 += 64) {
    __syncthreads();
    if (SRC_F32) {
      const int q = tid & 15, p0 = tid >> 4;
      const float* sb = reinterpret_cast<const float*>(a.x) + c0 + q * 4;
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int pix = flat0 + p0 + i * 16;                   // clamped, not predicated (predicated loads were issued one
        pix = pix < npix ? pix : npix - 1;               // round trip after the other); rows past the end are never stored
        v[i] = *reinterpret_cast<const float4*>(sb + (long long)pix * a.sx);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<uint2*>(A_s + (p0 + i * 16) * kFhLD + q * 4) = cvt4<true>(v[i]);
    } else {
      const int q = tid & 7, p0 = tid >> 3;
      const uint16_t* sb = reinterpret_cast<const uint16_t*>(a.x) + c0 + q * 8;
      uint4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int pix = flat0 + p0 + i * 32;
        pix = pix < npix ? pix : npix - 1;
        v[i] = *reinterpret_cast<const uint4*>(sb + (long long)pix * a.sx);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(A_s + (p0 + i * 32) * kFhLD + q * 8) = v[i];
    }
    stage_w(a.w0, 128, c0, 64);
    __syncthreads();
    mma_chunk(A_s + (wave * 32 + r) * kFhLD + h * 8, 2);
  }
  __syncthreads();                                        // every wave is done with the staged input: rows become hidden rows
  store_hidden();

  // ---- optional middle layer 64 -> 64, relu ----------------------------------------------------------------------------------
  if (NHID == 2) {
    zero_acc();
    __syncthreads();
    stage_w(a.w1, 64, 0, 64);
    __syncthreads();
    mma_chunk(A_s + (wave * 32 + r) * kFhLD + h * 8, 2);
    __builtin_amdgcn_wave_barrier();                      // the wave's rows are private: overwrite in place
    store_hidden();
  }

  // ---- last layer 64 -> 4 (one 32-row tile, rows 0..3 live) ----------------------------------------------------------------
  zero_acc();
  __syncthreads();
  stage_w(a.wl, 64, 0, 32);
  __syncthreads();
  mma_chunk(A_s + (wave * 32 + r) * kFhLD + h * 8, 1);
  const int pix = flat0 + wave * 32 + r;
  if (h == 0 && pix < npix)
    *reinterpret_cast<float4*>(a.out + (long long)pix * 4) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_freq_head(const void* x, int x_dtype, int64_t x_pix_stride, int64_t npix, const void* w0, const void* w_mid,
                               const void* w_last, float* out, void* stream) {
  FCVSR_CHECK_ARG(x && w0 && w_last && out, "null argument");
  FCVSR_CHECK_ARG(x_dtype == FCVSR_F32 || x_dtype == FCVSR_BF16, "x: f32 or bf16");
  FCVSR_CHECK_ARG(npix > 0 && npix < (1ll << 30) && x_pix_stride >= 128 && x_pix_stride % 8 == 0, "bad sizes / strides");
  FCVSR_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w0 % 16) == 0 && ((uintptr_t)w_last % 16) == 0 &&
                      ((uintptr_t)out % 16) == 0 && (w_mid == nullptr || ((uintptr_t)w_mid % 16) == 0), "16-byte alignment");
  FreqHeadArgs a;
  a.x = x; a.sx = x_pix_stride; a.out = out; a.npix = (int)npix;
  a.w0 = (const uint16_t*)w0; a.w1 = (const uint16_t*)w_mid; a.wl = (const uint16_t*)w_last;
  dim3 grid(cdiv(npix, kFhPix));
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == FCVSR_F32) {
    if (w_mid) hipLaunchKernelGGL((freq_head_kernel<true, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((freq_head_kernel<true, 1>), grid, dim3(256), 0, st, a);
  } else {
    if (w_mid) hipLaunchKernelGGL((freq_head_kernel<false, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((freq_head_kernel<false, 1>), grid, dim3(256), 0, st, a);
  }
  FCVSR_LAUNCH_CHECK();
  return 0;
}
