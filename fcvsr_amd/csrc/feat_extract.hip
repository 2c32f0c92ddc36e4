// feat_extract of GShiftNet(_S): Conv2d(7, 7*n, 3, 1, 1) on the stacked LR frames (reference CVSR_freq.py:2589, applied at
// :2614-2621).  With 7 input channels the whole 3x3 patch is only 63 values, so the layer is ONE K = 64 GEMM step per
// (32 pixels x 32 couts) instead of nine 16-deep tap steps with a barrier pair each: the generic implicit-GEMM kernel spent
// its time on staging and barriers (18 MFMAs per 18 barriers and a full epilogue per 64-cout block, 7 blocks).
//   * a workgroup = 128 consecutive pixels: the im2col tile A[pixel][k = tap*Cin + c] (f16, 8-bit pixels are exact) is
//     built once in LDS straight from the planar NCHW frames (coalesced 4-byte loads, zero padding by predicate);
//   * each wave keeps the fragments of its 32 pixels in 16 VGPRs and walks the 7 blocks of 64 output channels: weights
//     (A operand, rows = couts, [cout][64] f16) come straight from L2, 8 MFMAs per block, accumulators transposed through a
//     wave-private LDS tile so that every store is 16 bytes in a 128-byte run;
//   * the output channel blocks go to up to three destination tensors (the frame groups f1 | f2 | f3 of the reference's
//     channel split, :2617-2619), in the storage dtype of the activations.
#include <type_traits>
#include "common.h"
#include "mfma_util.h"

namespace fcvsr {

constexpr int kFePix = 128;
constexpr int kFeK = 64;
constexpr int kFeLD = kFeK + 8;                    // halfwords per im2col row
constexpr int kFeERow = 64 + 4;                    // floats per pixel row of the epilogue transpose
constexpr int kFeMaxBlk = 16;

struct FeatArgs {
  View x;                   // (B, H, W, Cin) logical view of the planar frames, f32
  int B, H, W, cin;
  const uint16_t* w;        // [n_blk*64][64] f16: k = tap*cin + c (zero beyond 9*cin)
  const float* bias;        // [n_blk*64] or null
  int n_blk;                // 64-cout blocks
  void* dst[kFeMaxBlk];     // per block: destination base (16-bit), pixel stride (elements) and channel offset
  long long dsx[kFeMaxBlk];
  int dch[kFeMaxBlk];
};

template <bool DSTBF, int CIN>
__global__ __launch_bounds__(256, 3) void feat_extract_kernel(FeatArgs a) {
  extern __shared__ __align__(16) unsigned char lds[];
  uint16_t* A_s = reinterpret_cast<uint16_t*>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  float* E_s = reinterpret_cast<float*>(lds + kFePix * kFeLD * 2) + wave * (32 * kFeERow);
  const long long npix = (long long)a.B * a.H * a.W;
  const long long flat0 = (long long)blockIdx.x * kFePix;

  // ---- im2col: thread = (pixel, half of k).  k, tap and channel are compile-time inside each half; the 32 loads of a thread
  // are unconditional (clamped address, value zeroed when the tap is outside the image) so that they are issued as one batch.
  {
    const int pl = tid & (kFePix - 1), half = tid >> 7;   // half is wave-uniform
    const long long p = flat0 + pl;
    const bool pok = p < npix;
    const long long pc = pok ? p : npix - 1;
    const int xx = (int)(pc % a.W);
    const int yy = (int)((pc / a.W) % a.H);
    const int b = (int)(pc / ((long long)a.W * a.H));
    const float* xb = a.x.p + (long long)b * a.x.sb;
    const int sy = (int)a.x.sy, sx = (int)a.x.sx, sc = (int)a.x.sc;   // one image spans < 2^31 elements (host check)
    float f[32];
    auto gather = [&](auto koff) {
      constexpr int K0 = decltype(koff)::value;
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int k = K0 + j;
        const int tap = k / CIN, c = k - tap * CIN;      // compile-time after unrolling
        const int iy = yy + tap / 3 - 1, ix = xx + tap % 3 - 1;
        const bool in = pok && k < 9 * CIN && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const int iyc = iy < 0 ? 0 : (iy > a.H - 1 ? a.H - 1 : iy), ixc = ix < 0 ? 0 : (ix > a.W - 1 ? a.W - 1 : ix);
        const float v = xb[iyc * sy + ixc * sx + (k < 9 * CIN ? c : 0) * sc];
        f[j] = in ? v : 0.f;
      }
    };
    if (half == 0) gather(std::integral_constant<int, 0>{});
    else gather(std::integral_constant<int, 32>{});
    uint4* d = reinterpret_cast<uint4*>(A_s + pl * kFeLD + half * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      typedef __attribute__((ext_vector_type(8))) _Float16 h8;
      const h8 pk = {(_Float16)f[8 * q], (_Float16)f[8 * q + 1], (_Float16)f[8 * q + 2], (_Float16)f[8 * q + 3],
                     (_Float16)f[8 * q + 4], (_Float16)f[8 * q + 5], (_Float16)f[8 * q + 6], (_Float16)f[8 * q + 7]};
      d[q] = __builtin_bit_cast(uint4, pk);
    }
  }
  __syncthreads();

  // ---- the wave's 32 pixels as the B operand (columns), kept in registers for all cout blocks ---------------------------------
  uint4 pf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) pf[kk] = *reinterpret_cast<const uint4*>(A_s + (wave * 32 + r) * kFeLD + kk * 16 + h * 8);

  uint4 wf[2][4];
#define FCVSR_FE_LOADW(BLK)                                                                                   \
  _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)             \
      wf[nf][kk] = *reinterpret_cast<const uint4*>(a.w + ((long long)(BLK) * 64 + nf * 32 + r) * kFeK + kk * 16 + h * 8)
  FCVSR_FE_LOADW(0);
  const int co = lane & 7, psub = lane >> 3;           // epilogue: 8 lanes x 8 couts per pixel, 8 pixels per pass
  for (int blk = 0; blk < a.n_blk; ++blk) {
    f32x16_t acc[2];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nf][i] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc[nf] = mfma<false>(wf[nf][kk], pf[kk], acc[nf]);
    }
    if (blk + 1 < a.n_blk) FCVSR_FE_LOADW(blk + 1);     // next block's weights under this block's epilogue
    // acc[nf][4g..4g+3] = couts nf*32 + 8g + 4h + (0..3) of pixel r
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(E_s + r * kFeERow + nf * 32 + 8 * g + 4 * h) =
            make_float4(acc[nf][4 * g], acc[nf][4 * g + 1], acc[nf][4 * g + 2], acc[nf][4 * g + 3]);
    __builtin_amdgcn_wave_barrier();
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (a.bias) {
      b0 = *reinterpret_cast<const float4*>(a.bias + blk * 64 + co * 8);
      b1 = *reinterpret_cast<const float4*>(a.bias + blk * 64 + co * 8 + 4);
    }
    uint16_t* dp = reinterpret_cast<uint16_t*>(a.dst[blk]) + a.dch[blk] + co * 8;
    const long long dsx = a.dsx[blk];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pl = j * 8 + psub;
      const long long p = flat0 + wave * 32 + pl;
      if (p < npix) {
        const float* es = E_s + pl * kFeERow + co * 8;
        const float4 e0 = *reinterpret_cast<const float4*>(es), e1 = *reinterpret_cast<const float4*>(es + 4);
        const float4 x0 = make_float4(e0.x + b0.x, e0.y + b0.y, e0.z + b0.z, e0.w + b0.w);
        const float4 x1 = make_float4(e1.x + b1.x, e1.y + b1.y, e1.z + b1.z, e1.w + b1.w);
        const uint2 lo = cvt4<DSTBF>(x0), hi = cvt4<DSTBF>(x1);
        *reinterpret_cast<uint4*>(dp + p * dsx) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#undef FCVSR_FE_LOADW
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_feat_extract(const fcvsr_view* x, int B, int H, int W, const void* w, const float* bias, int n_blk,
                                  void* const* dst, const int64_t* dst_pix_stride, const int32_t* dst_ch_off, int dst_dtype,
                                  void* stream) {
  FCVSR_CHECK_ARG(x && x->ptr && w && dst && dst_pix_stride && dst_ch_off, "null argument");
  FCVSR_CHECK_ARG(x->dtype == FCVSR_F32 && x->c == 7, "x: f32, 7 channels (the Y models' frame stack)");
  FCVSR_CHECK_ARG((long long)H * x->sy < (1ll << 31) && 7ll * x->sc < (1ll << 31), "image too large for 32-bit offsets");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && n_blk >= 1 && n_blk <= kFeMaxBlk, "bad sizes");
  FCVSR_CHECK_ARG(dst_dtype == FCVSR_BF16 || dst_dtype == FCVSR_F16, "16-bit destinations");
  FCVSR_CHECK_ARG(((uintptr_t)w % 16) == 0 && (bias == nullptr || ((uintptr_t)bias % 16) == 0), "weights / bias 16-byte aligned");
  FeatArgs a;
  a.x = to_view(*x); a.B = B; a.H = H; a.W = W; a.cin = x->c;
  a.w = (const uint16_t*)w; a.bias = bias; a.n_blk = n_blk;

  for (int i = 0; i < kFeMaxBlk; ++i) {
    const int s = i < n_blk ? i : 0;
    FCVSR_CHECK_ARG(dst[s] && ((uintptr_t)dst[s] % 16) == 0 && dst_pix_stride[s] % 8 == 0 && dst_ch_off[s] % 8 == 0 &&
                        dst_ch_off[s] >= 0, "destinations: 16-byte aligned, strides / offsets multiples of 8");
    a.dst[i] = dst[s]; a.dsx[i] = dst_pix_stride[s]; a.dch[i] = dst_ch_off[s];
  }
  const size_t lds = (size_t)kFePix * kFeLD * 2 + 4ull * 32 * kFeERow * 4;
  static DevOnce attr;
  {
    hipError_t e = once_per_device(attr, [&] {
      hipError_t e1 = hipFuncSetAttribute((const void*)feat_extract_kernel<true, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e1 == hipSuccess)
        e1 = hipFuncSetAttribute((const void*)feat_extract_kernel<false, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      return e1;
    });
    if (e != hipSuccess) {
      set_error("fcvsr_feat_extract: %s", hipGetErrorString(e));
      return (int)e;
    }
  }
  const long long npix = (long long)B * H * W;
  dim3 grid(cdiv(npix, kFePix));
  hipStream_t st = (hipStream_t)stream;
  if (dst_dtype == FCVSR_BF16) hipLaunchKernelGGL((feat_extract_kernel<true, 7>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((feat_extract_kernel<false, 7>), grid, dim3(256), lds, st, a);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
