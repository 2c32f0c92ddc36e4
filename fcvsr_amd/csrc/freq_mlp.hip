// Fused offset-spectrum MLP of MGAAbk (reference CVSR_freq.py:1371-1377 convfuse, applied at :1472-1474):
//     off = (xa - xb) + W4 . relu( W2 . relu( W0 . [xa | xb] ) )          1x1 convolutions, no bias, C = 2*n_feats = 128
// for both alignment directions (xa = x1_f or x3_f, xb = x2_f) in one launch.  As three stand-alone 1x1 launches the stack
// moves 3.3 KB per spectrum pixel (two f32 spectra in, two 16-bit hidden tensors out and back in, the spectra again for
// the residual, the result out); here a workgroup keeps its 128 pixels on chip from the first load to the final store:
// 1 KB in (f32 spectra, converted to bf16 while staging) + 256 B out.
//   * layer 0: K = 256 in four 64-channel chunks staged global -> LDS like conv1_lean_kernel; weights are the A operand
//     (rows = couts) and pixels the B operand, so a lane ends up with 4 consecutive couts of ONE pixel per accumulator
//     group and the hidden activations go to LDS as 8-byte stores (T_s[pixel][128], rows private to the wave that owns
//     the 32 pixels);
//   * layers 2 and 4 read their pixel operand straight from T_s (no staging), only the 128 x 64 weight chunks are staged;
//   * the last epilogue transposes through LDS (the dead T_s rows) so that the f32 residual loads and the 16-bit stores
//     are contiguous per pixel.
// Frequency-domain layers always multiply in bf16 (unnormalised spectra exceed the f16 range), f32 accumulate.
#include "common.h"
#include "mfma_util.h"

namespace fcvsr {

constexpr int kFmC = 128;                 // channels of a packed spectrum (imag | real of n_feats = 64)
constexpr int kFmPix = 128;               // pixels per workgroup (32 per wave)
constexpr int kFmLD = 64 + 8;             // halfwords per row of a staged 64-channel chunk
constexpr int kFmTD = kFmC + 8;           // halfwords per pixel row of the hidden tile
constexpr int kFmTBytes = kFmPix * kFmTD * 2;             // 34816
constexpr int kFmBBytes = kFmC * kFmLD * 2;               // 18432
constexpr int kFmLds = kFmTBytes + kFmBBytes;             // A_s (18432) aliases the head of T_s

struct FreqMlpArgs {
  const float* xa[2];      // per direction: (npix, >=128) f32, pixel stride sx (floats)
  const float* xb[2];
  uint16_t* dst[2];        // per direction: (npix, 128) bf16, pixel stride dsx (halfwords)
  long long sx, dsx;
  int npix, n_groups, tiles_per_group;
  const uint16_t* w0;      // [128][256] bf16 (cin contiguous)
  const uint16_t* w2;      // [128][128]
  const uint16_t* w4;      // [128][128]
};

__global__ __launch_bounds__(256, 3) void freq_mlp3_kernel(FreqMlpArgs a) {
  extern __shared__ __align__(16) unsigned char lds[];
  uint16_t* T_s = reinterpret_cast<uint16_t*>(lds);
  uint16_t* A_s = T_s;                                              // layer 0 only
  uint16_t* B_s = reinterpret_cast<uint16_t*>(lds + kFmTBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int gi = blockIdx.x / a.tiles_per_group;
  const int flat0 = (blockIdx.x - gi * a.tiles_per_group) * kFmPix;
  const float* xa = a.xa[gi];
  const float* xb = a.xb[gi];
  const int npix = a.npix;

  f32x16_t acc[4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nf][i] = 0.f;
  };
  // weights chunk [128 couts][64 cin] -> B_s: 1024 16-byte pieces, 4 per thread
  auto stage_w = [&](const uint16_t* w, int cin_pad, int c0) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = (tid >> 3) + 32 * u;
      v[u] = *reinterpret_cast<const uint4*>(w + (long long)row * cin_pad + c0 + (tid & 7) * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4*>(B_s + ((tid >> 3) + 32 * u) * kFmLD + (tid & 7) * 8) = v[u];
  };
  // D[cout][pixel]: weights are the first operand (rows), the wave's 32 pixels the second (columns)
  auto mma_chunk = [&](const uint16_t* prow) {          // prow: this lane's pixel row at the chunk's first channel (+ h*8)
    const uint16_t* wrow = B_s + r * kFmLD + h * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint4 pf = *reinterpret_cast<const uint4*>(prow + kk * 16);
      uint4 wf[4];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) wf[nf] = *reinterpret_cast<const uint4*>(wrow + nf * 32 * kFmLD + kk * 16);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) acc[nf] = mfma<true>(wf[nf], pf, acc[nf]);
    }
  };
  // relu + round to bf16, hidden tile row of pixel r: acc[nf][4g..4g+3] = couts nf*32 + 8g + 4h + (0..3)
  auto store_hidden = [&]() {
    uint16_t* trow = T_s + (wave * 32 + r) * kFmTD;
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = make_float4(fmaxf(acc[nf][4 * g], 0.f), fmaxf(acc[nf][4 * g + 1], 0.f), fmaxf(acc[nf][4 * g + 2], 0.f),
                                     fmaxf(acc[nf][4 * g + 3], 0.f));
        *reinterpret_cast<uint2*>(trow + nf * 32 + 8 * g + 4 * h) = cvt4<true>(v);
      }
  };

  // ---- layer 0: [xa | xb] (256 f32 channels) -> 128, relu ------------------------------------------------------------------
  zero_acc();
  {
    const int q = tid & 15, p0 = tid >> 4;                // 16 lanes x 4 channels per pixel, 16 pixels per iteration
    for (int c0 = 0; c0 < 2 * kFmC; c0 += 64) {
      const float* sb = (c0 < kFmC ? xa : xb) + (c0 & (kFmC - 1)) + q * 4;
      __syncthreads();
      {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int pix = flat0 + p0 + i * 16;
          v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (pix < npix) v[i] = *reinterpret_cast<const float4*>(sb + (long long)pix * a.sx);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint2*>(A_s + (p0 + i * 16) * kFmLD + q * 4) = cvt4<true>(v[i]);
      }
      stage_w(a.w0, 2 * kFmC, c0);
      __syncthreads();
      mma_chunk(A_s + (wave * 32 + r) * kFmLD + h * 8);
    }
  }
  __syncthreads();                                        // A_s (aliased by T_s) is dead from here on
  store_hidden();

  // ---- layers 2 and 4: 128 -> 128; the pixel operand is the hidden tile itself -------------------------------------------
#pragma unroll 1
  for (int layer = 0; layer < 2; ++layer) {
    const uint16_t* w = layer == 0 ? a.w2 : a.w4;
    zero_acc();
    for (int c0 = 0; c0 < kFmC; c0 += 64) {
      __syncthreads();                                    // previous chunk's B_s reads are done (and T_s rows are written)
      stage_w(w, kFmC, c0);
      __syncthreads();
      mma_chunk(T_s + (wave * 32 + r) * kFmTD + c0 + h * 8);
    }
    if (layer == 0) {
      __builtin_amdgcn_wave_barrier();                    // rows are private to the wave: no workgroup barrier needed
      store_hidden();
    }
  }

  // ---- epilogue: off = acc + xa - xb, 16-bit store; transposed through the wave's (dead) hidden rows, 64 couts at a time ----
  __builtin_amdgcn_wave_barrier();
  float* E_s = reinterpret_cast<float*>(lds + wave * (32 * kFmTD * 2));   // 32 x 68 floats = 8704 bytes = the wave's T_s rows
  constexpr int EROW = 64 + 4;
  const int co = lane & 7, psub = lane >> 3;              // 8 lanes x 8 couts per pixel, 8 pixels per pass
  uint16_t* dp = a.dst[gi];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int nf2 = 0; nf2 < 2; ++nf2)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nf = half * 2 + nf2;
        *reinterpret_cast<float4*>(E_s + r * EROW + nf2 * 32 + 8 * g + 4 * h) =
            make_float4(acc[nf][4 * g], acc[nf][4 * g + 1], acc[nf][4 * g + 2], acc[nf][4 * g + 3]);
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pl = j * 8 + psub;
      const int pix = flat0 + wave * 32 + pl;
      if (pix < npix) {
        const int n = half * 64 + co * 8;
        const float* es = E_s + pl * EROW + co * 8;
        const float4 e0 = *reinterpret_cast<const float4*>(es), e1 = *reinterpret_cast<const float4*>(es + 4);
        const float* pa = xa + (long long)pix * a.sx + n;
        const float* pb = xb + (long long)pix * a.sx + n;
        const float4 a0 = *reinterpret_cast<const float4*>(pa), a1 = *reinterpret_cast<const float4*>(pa + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(pb), b1 = *reinterpret_cast<const float4*>(pb + 4);
        // same order as the stand-alone layer: ((acc + 1*xa) + (-1)*xb)
        const float4 x0 = make_float4(fmaf(-1.f, b0.x, fmaf(1.f, a0.x, e0.x)), fmaf(-1.f, b0.y, fmaf(1.f, a0.y, e0.y)),
                                      fmaf(-1.f, b0.z, fmaf(1.f, a0.z, e0.z)), fmaf(-1.f, b0.w, fmaf(1.f, a0.w, e0.w)));
        const float4 x1 = make_float4(fmaf(-1.f, b1.x, fmaf(1.f, a1.x, e1.x)), fmaf(-1.f, b1.y, fmaf(1.f, a1.y, e1.y)),
                                      fmaf(-1.f, b1.z, fmaf(1.f, a1.z, e1.z)), fmaf(-1.f, b1.w, fmaf(1.f, a1.w, e1.w)));
        const uint2 lo = cvt4<true>(x0), hi = cvt4<true>(x1);
        *reinterpret_cast<uint4*>(dp + (long long)pix * a.dsx + n) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_freq_mlp3(const float* const* xa, const float* const* xb, int n_dirs, int64_t src_pix_stride, int64_t npix,
                               const void* w0, const void* w2, const void* w4, void* const* dst, int64_t dst_pix_stride,
                               void* stream) {
  FCVSR_CHECK_ARG(xa && xb && dst && w0 && w2 && w4 && (n_dirs == 1 || n_dirs == 2), "null argument / 1..2 directions");
  FCVSR_CHECK_ARG(npix > 0 && npix < (1ll << 30) && src_pix_stride >= kFmC && src_pix_stride % 4 == 0 &&
                      dst_pix_stride >= kFmC && dst_pix_stride % 8 == 0, "bad sizes / strides");
  FCVSR_CHECK_ARG(((uintptr_t)w0 % 16) == 0 && ((uintptr_t)w2 % 16) == 0 && ((uintptr_t)w4 % 16) == 0, "weights 16-byte aligned");
  FreqMlpArgs a;
  for (int d = 0; d < 2; ++d) {
    const int s = d < n_dirs ? d : 0;
    FCVSR_CHECK_ARG(xa[s] && xb[s] && dst[s] && ((uintptr_t)xa[s] % 16) == 0 && ((uintptr_t)xb[s] % 16) == 0 &&
                        ((uintptr_t)dst[s] % 16) == 0, "spectra / destination must be 16-byte aligned");
    a.xa[d] = xa[s]; a.xb[d] = xb[s]; a.dst[d] = (uint16_t*)dst[s];
  }
  a.sx = src_pix_stride; a.dsx = dst_pix_stride; a.npix = (int)npix; a.n_groups = n_dirs;
  a.tiles_per_group = cdiv(npix, kFmPix);
  a.w0 = (const uint16_t*)w0; a.w2 = (const uint16_t*)w2; a.w4 = (const uint16_t*)w4;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)freq_mlp3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFmLds);
    if (e != hipSuccess) {
      set_error("fcvsr_freq_mlp3: %s", hipGetErrorString(e));
      return (int)e;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(freq_mlp3_kernel, dim3(a.tiles_per_group * n_dirs), dim3(256), kFmLds, (hipStream_t)stream, a);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
