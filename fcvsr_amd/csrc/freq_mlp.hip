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
      if (kk & 1) __builtin_amdgcn_sched_barrier(0);      // at most two 16-deep steps of fragments in registers
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

  // ---- eight units: layer 0 in four 64-channel chunks (pixel chunk + weight chunk staged), layers 2 and 4 in two chunks each
  // (weight chunk only; the pixel operand is the hidden tile).  Every global load of unit u + 1 is issued before the MFMAs
  // of unit u, and the residual spectra of an epilogue half go out in one batch: a
  // workgroup used to pay ~20 memory round trips one after the other (4 pixel chunks, 8 weight chunks, 8 residual groups: 48 us
  // per 128 pixels with three workgroups per CU); now the prologue's and the two residual batches' are the exposed ones.
  const int q = tid & 15, p0 = tid >> 4;                  // pixel staging: 16 lanes x 4 channels per pixel, 16 pixels per iteration
  float4 xv[8];
  uint4 wv[4];
  // addresses = wave-uniform 64-bit base + 32-bit lane byte offset (the host checks npix * stride < 2^32 bytes): the 64-bit
  // per-pixel offsets of the eight staged pixels and the four residual pixels cost ~40 registers and pushed the prefetched
  // weight chunk into scratch
  auto gld4 = [](const void* base, unsigned off) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off); };
  auto gldq = [](const void* base, unsigned off) { return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + off); };
  const unsigned sxb = (unsigned)a.sx * 4u;
  auto load_x = [&](int c0) {
    const float* sb = (c0 < kFmC ? xa : xb) + (c0 & (kFmC - 1));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int pix = flat0 + p0 + i * 16;
      pix = pix < npix ? pix : npix - 1;                  // clamped, not predicated: rows past the end are never stored
      xv[i] = gld4(sb, (unsigned)pix * sxb + q * 16);
    }
  };
  auto load_w = [&](const uint16_t* w, int cin_pad, int c0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = (tid >> 3) + 32 * u;
      wv[u] = gldq(w + c0, (unsigned)(row * cin_pad + (tid & 7) * 8) * 2u);
    }
  };
  auto put_x = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<uint2*>(A_s + (p0 + i * 16) * kFmLD + q * 4) = cvt4<true>(xv[i]);
  };
  auto put_w = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4*>(B_s + ((tid >> 3) + 32 * u) * kFmLD + (tid & 7) * 8) = wv[u];
  };
  // epilogue residuals: lane (co, psub) owns 8 couts of pixels j*8 + psub, j = 0..3, per 64-cout half
  const int co = lane & 7, psub = lane >> 3;
  float4 ra[4][2], rb[4][2];
  auto load_res = [&](int half) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int pix = flat0 + wave * 32 + j * 8 + psub;
      pix = pix < npix ? pix : npix - 1;
      const unsigned off = (unsigned)pix * sxb + co * 32;
      ra[j][0] = gld4(xa + half * 64, off); ra[j][1] = gld4(xa + half * 64 + 4, off);
      rb[j][0] = gld4(xb + half * 64, off); rb[j][1] = gld4(xb + half * 64 + 4, off);
    }
  };

  // ---- layer 0: [xa | xb] (256 f32 channels) -> 128, relu ------------------------------------------------------------------
  zero_acc();
  load_x(0);
  load_w(a.w0, 2 * kFmC, 0);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    __syncthreads();                                      // the previous chunk's fragment reads are done
    put_x();
    put_w();
    if (u < 3) { load_x((u + 1) * 64); load_w(a.w0, 2 * kFmC, (u + 1) * 64); }
    else load_w(a.w2, kFmC, 0);
    __syncthreads();
    mma_chunk(A_s + (wave * 32 + r) * kFmLD + h * 8);
  }
  __syncthreads();                                        // A_s (aliased by T_s) is dead from here on
  store_hidden();

  // ---- layers 2 and 4: 128 -> 128; the pixel operand is the hidden tile itself -------------------------------------------
#pragma unroll
  for (int u = 0; u < 4; ++u) {                           // u = 0, 1: layer 2; u = 2, 3: layer 4
    if ((u & 1) == 0) zero_acc();
    __syncthreads();                                      // previous chunk's B_s reads are done (and T_s rows are written)
    put_w();
    if (u == 0) load_w(a.w2, kFmC, 64);
    else if (u == 1) load_w(a.w4, kFmC, 0);
    else if (u == 2) load_w(a.w4, kFmC, 64);
    __syncthreads();
    mma_chunk(T_s + (wave * 32 + r) * kFmTD + (u & 1) * 64 + h * 8);
    if (u == 1) {
      __builtin_amdgcn_wave_barrier();                    // rows are private to the wave: no workgroup barrier needed
      store_hidden();
    }
  }

  // ---- epilogue: off = acc + xa - xb, 16-bit store; transposed through the wave's (dead) hidden rows, 64 couts at a time ----
  __builtin_amdgcn_wave_barrier();
  float* E_s = reinterpret_cast<float*>(lds + wave * (32 * kFmTD * 2));   // 32 x 68 floats = 8704 bytes = the wave's T_s rows
  constexpr int EROW = 64 + 4;
  uint16_t* dp = a.dst[gi];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __builtin_amdgcn_sched_barrier(0);                    // one half's residuals at a time (both at once spill)
    load_res(half);                                       // all 16 loads of the half in one batch, behind the transposition
#pragma unroll
    for (int nf2 = 0; nf2 < 2; ++nf2)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nf = half * 2 + nf2;
        *reinterpret_cast<float4*>(E_s + r * EROW + nf2 * 32 + 8 * g + 4 * h) =
            make_float4(acc[nf][4 * g], acc[nf][4 * g + 1], acc[nf][4 * g + 2], acc[nf][4 * g + 3]);
      }
    __builtin_amdgcn_wave_barrier();
    uint4 outv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pl = j * 8 + psub;
      const float* es = E_s + pl * EROW + co * 8;
      const float4 e0 = *reinterpret_cast<const float4*>(es), e1 = *reinterpret_cast<const float4*>(es + 4);
      const float4 a0 = ra[j][0], a1 = ra[j][1], b0 = rb[j][0], b1 = rb[j][1];
      // same order as the stand-alone layer: ((acc + 1*xa) + (-1)*xb)
      const float4 x0 = make_float4(fmaf(-1.f, b0.x, fmaf(1.f, a0.x, e0.x)), fmaf(-1.f, b0.y, fmaf(1.f, a0.y, e0.y)),
                                    fmaf(-1.f, b0.z, fmaf(1.f, a0.z, e0.z)), fmaf(-1.f, b0.w, fmaf(1.f, a0.w, e0.w)));
      const float4 x1 = make_float4(fmaf(-1.f, b1.x, fmaf(1.f, a1.x, e1.x)), fmaf(-1.f, b1.y, fmaf(1.f, a1.y, e1.y)),
                                    fmaf(-1.f, b1.z, fmaf(1.f, a1.z, e1.z)), fmaf(-1.f, b1.w, fmaf(1.f, a1.w, e1.w)));
      const uint2 lo = cvt4<true>(x0), hi = cvt4<true>(x1);
      outv[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pix = flat0 + wave * 32 + j * 8 + psub;
      if (pix < npix)
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(dp + half * 64) + ((unsigned)pix * (unsigned)a.dsx + co * 8) * 2u) = outv[j];
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
  FCVSR_CHECK_ARG(npix * src_pix_stride * 4 < (1ll << 32) && npix * dst_pix_stride * 2 < (1ll << 32), "tensor too large for 32-bit offsets");
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
  static DevOnce attr;
  {
    hipError_t e = once_per_device(attr, [&] {
      return hipFuncSetAttribute((const void*)freq_mlp3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFmLds);
    });
    if (e != hipSuccess) {
      set_error("fcvsr_freq_mlp3: %s", hipGetErrorString(e));
      return (int)e;
    }
  }
  hipLaunchKernelGGL(freq_mlp3_kernel, dim3(a.tiles_per_group * n_dirs), dim3(256), kFmLds, (hipStream_t)stream, a);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
