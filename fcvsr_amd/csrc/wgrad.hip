// Weight gradient of a 2-D convolution (the backward half of every nn.Conv2d on the path: reference
// CVSR_train/train_LD_freqCVSR_S_22.py:250 `loss.backward()` through CVSR_freq.py's convolutions).
//
//   dW[co][ci][ky][kx] = sum_{b,oy,ox} gy[b,oy,ox,co] * x[b, oy*stride - pad + ky, ox*stride - pad + kx, ci]
//
// Exact f32 and bit-reproducible: the pixel list is cut into slabs, a workgroup accumulates one (slab, tap, 16-cout x 64-cin
// block) in registers in a fixed order, and a second launch adds the slabs in order (no float atomics).  This is the
// exact-parity path (the reference trains in f32); the 16-bit matrix-core variant below is the throughput path.
//
// Layout: x and gy are channel-contiguous (NHWC) f32 views; the result is written in the reference's parameter layout
// (cout, cin, kh, kw) so it can be returned as the .grad of the nn.Conv2d weight as is.
#include <stdlib.h>
#include "common.h"

namespace fcvsr {

struct WgradArgs {
  View x, gy;
  int B, H, W, Ho, Wo, kh, kw, stride, pad, cin, cout;
  long long npix;          // B * Ho * Wo
  int n_slabs;
  long long slab_pix;      // pixels per slab
  float* partial;          // [n_slabs][kh*kw][cin][cout]
  float* dw;               // [cout][cin][kh][kw]
  int accumulate;          // 1: dw += the sum (the gradient buffer was zeroed at the start of the pass), 0: dw = the sum
  // bias gradient fused into the matrix-core form 2 (fcvsr_wgrad_set_bias_out): per-slab column sums of gy [n_slabs][cout] -> dbias
  float* dbp;
  float* dbias;
  int db_accumulate;
};

// Per-thread switch of the three weight-gradient entry points between "dw = sum" and "dw += sum": the training step keeps every
// parameter gradient in one flat, pre-zeroed buffer and lets the reduction add straight into it, which removes autograd's
// AccumulateGrad addition per parameter and pass (fcvsr_wgrad_set_accumulate; fcvsr_amd/train/ops.py).
static thread_local int g_wgrad_accumulate = 0;
// One-shot request consumed by the next fcvsr_conv2d_wgrad_mfma / _groups call of this thread: also produce dL/dbias = column sums of gy
// (f32, fixed order) - the kernel has every gy tile in registers anyway, the stand-alone column-sum launches (two per layer) go away.
static thread_local float* g_wgrad_bias_out = nullptr;
static thread_local int g_wgrad_bias_accumulate = 0;

constexpr int kWgCo = 16, kWgCi = 64, kWgPx = 32;

__global__ __launch_bounds__(256) void wgrad_partial_kernel(WgradArgs a) {
  __shared__ float gy_s[kWgPx][kWgCo];
  __shared__ __align__(16) float x_s[kWgPx][kWgCi];
  const int tid = threadIdx.x;
  const int slab = blockIdx.x, tap = blockIdx.y;
  const int nci = (a.cin + kWgCi - 1) / kWgCi;
  const int co0 = (blockIdx.z / nci) * kWgCo, ci0 = (blockIdx.z % nci) * kWgCi;
  const int ky = tap / a.kw, kx = tap % a.kw;
  const int tco = tid & 15, tci = tid >> 4;                 // 16 couts x 16 groups of 4 cins
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const long long p0 = (long long)slab * a.slab_pix;
  long long p1 = p0 + a.slab_pix;
  if (p1 > a.npix) p1 = a.npix;
  __shared__ long long goff_s[kWgPx], xoff_s[kWgPx];       // element offsets of the 32 pixels' gy / shifted x records (-1: none)
  for (long long pc = p0; pc < p1; pc += kWgPx) {
    __syncthreads();
    // pixel -> (b, oy, ox) once per PIXEL (32 lanes, 32-bit divisions), not once per staged element: the 64-bit div / mod
    // triple per element was most of this kernel's time (3 ms for conv_last0's and feat_extract's weight gradients)
    if (tid < kWgPx) {
      const long long p = pc + tid;
      long long go = -1, xo = -1;
      if (p < p1) {
        const unsigned pp = (unsigned)p;                   // host checks npix < 2^31
        const unsigned t = pp / (unsigned)a.Wo;
        const int ox = (int)(pp - t * (unsigned)a.Wo);
        const int b = (int)(t / (unsigned)a.Ho);
        const int oy = (int)(t - (unsigned)b * (unsigned)a.Ho);
        go = (long long)b * a.gy.sb + (long long)oy * a.gy.sy + (long long)ox * a.gy.sx;
        const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) xo = (long long)b * a.x.sb + (long long)iy * a.x.sy + (long long)ix * a.x.sx;
      }
      goff_s[tid] = go;
      xoff_s[tid] = xo;
    }
    __syncthreads();
    // stage gy[pc .. pc+32)[co0 .. co0+16) and the tap-shifted x[..][ci0 .. ci0+64) (zeros outside the image / past the slab)
    for (int i = tid; i < kWgPx * kWgCo; i += 256) {
      const int q = i >> 4, c = i & 15;
      const long long go = goff_s[q];
      const bool ok = go >= 0 && co0 + c < a.cout;
      const float v = a.gy.p[(ok ? go : 0) + (ok ? co0 + c : 0)];
      gy_s[q][c] = ok ? v : 0.f;
    }
    for (int i = tid; i < kWgPx * kWgCi; i += 256) {
      const int q = i >> 6, c = i & 63;
      const long long xo = xoff_s[q];
      const bool ok = xo >= 0 && ci0 + c < a.cin;
      const float v = a.x.p[(ok ? xo : 0) + (ok ? ci0 + c : 0)];
      x_s[q][c] = ok ? v : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int q = 0; q < kWgPx; ++q) {
      const float g = gy_s[q][tco];
      const float4 xv = *reinterpret_cast<const float4*>(&x_s[q][tci * 4]);
      acc[0] = fmaf(g, xv.x, acc[0]);
      acc[1] = fmaf(g, xv.y, acc[1]);
      acc[2] = fmaf(g, xv.z, acc[2]);
      acc[3] = fmaf(g, xv.w, acc[3]);
    }
  }
  const int co = co0 + tco;
  if (co < a.cout) {
    float* pp = a.partial + (((long long)slab * (a.kh * a.kw) + tap) * a.cin) * a.cout;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ci = ci0 + tci * 4 + e;
      if (ci < a.cin) pp[(long long)ci * a.cout + co] = acc[e];
    }
  }
}

// dw[co][ci][tap] = sum over slabs, in slab order
// Threads walk the PARTIAL layout ([tap][ci][co], co fastest): every slab read is coalesced and only the n final writes are
// scattered (with threads in dw order each lane read its own 64-byte sector per slab: 38 us for a 64 x 64 x 9 layer, 341
// launches per training step).  Same summation order (slab 0, 1, ...) as before: bit-identical results.
// 4 -> 4 channel layers (the multi-scale ConvBlk of MGAAbk, reference CVSR_freq.py:1284-1316: 1x1, 3x3 and 5x5 convolutions of 4-channel
// spectra): the block kernel above spends 16 x 64 lanes on 4 x 4 channels (130 us per layer, 2.1 ms per training step).  Here a lane
// is a PIXEL: wave w owns the taps t = w, w + 4, ... (at most 7 of 25), a lane multiplies its pixel's 4 gradient channels with the
// 4 input channels of each of its taps (16 products per tap, f32, pixels of a slab in order), and the 64 lane sums are folded by a
// fixed butterfly at the end.  Same partial layout and ordered final reduction as the block kernel: exact f32, bit-reproducible.
template <int K>
__global__ __launch_bounds__(256) void wgrad_c4_kernel(WgradArgs a) {
  constexpr int NT = (K * K + 3) / 4, PADK = K / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slab = blockIdx.x;
  float acc[NT][16];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const long long p0 = (long long)slab * a.slab_pix;
  long long p1 = p0 + a.slab_pix;
  if (p1 > a.npix) p1 = a.npix;
  for (long long pc = p0; pc < p1; pc += 64) {
    const long long p = pc + lane;
    if (p < p1) {
      const unsigned pp = (unsigned)p;
      const unsigned t1 = pp / (unsigned)a.W;
      const int ox = (int)(pp - t1 * (unsigned)a.W);
      const int b = (int)(t1 / (unsigned)a.H);
      const int oy = (int)(t1 - (unsigned)b * (unsigned)a.H);
      const float4 g = *reinterpret_cast<const float4*>(a.gy.p + (long long)b * a.gy.sb + (long long)oy * a.gy.sy + (long long)ox * a.gy.sx);
      const float* xb = a.x.p + (long long)b * a.x.sb;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tap = wave + 4 * t;
        if (tap < K * K) {
          const int ky = tap / K, kx = tap - ky * K;
          const int iy = oy + ky - PADK, ix = ox + kx - PADK;
          if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            const float4 xv = *reinterpret_cast<const float4*>(xb + (long long)iy * a.x.sy + (long long)ix * a.x.sx);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
              for (int co = 0; co < 4; ++co) acc[t][ci * 4 + co] = fmaf(xs[ci], gs[co], acc[t][ci * 4 + co]);
          }
        }
      }
    }
  }
  float* pp = a.partial + (long long)slab * (K * K) * 16;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tap = wave + 4 * t;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = acc[t][i];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if (lane == 0 && tap < K * K) pp[tap * 16 + i] = v;    // [tap][ci][co]
    }
  }
}

// Sum of the slabs' partials in a FIXED order (bit-reproducible, no atomics): a workgroup owns 64 consecutive outputs; wave g adds
// slabs g, g + 4, g + 8, ... in order with eight loads in flight, the four wave sums are added ((s0 + s1) + s2) + s3.
// (Round 3: the one-thread-per-output loop over up to 256 slabs, four loads in flight, took 14-20 us per layer - 2.6 ms per step.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradArgs a) {
  const long long nw = (long long)a.cout * a.cin * a.kh * a.kw;
  const long long nbw = (nw + 63) / 64;                   // blocks past these sum the bias partials [n_slabs][cout]
  const bool isb = (long long)blockIdx.x >= nbw;
  const long long n = isb ? a.cout : nw;
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long j = ((long long)blockIdx.x - (isb ? nbw : 0)) * 64 + o;
  __shared__ float part[4][64];
  float s = 0.f;
  if (j < n) {
    const float* p = (isb ? a.dbp : a.partial) + j;
    int sl = g;
    for (; sl + 28 < a.n_slabs; sl += 32) {                // slabs sl, sl + 4, ..., sl + 28
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(long long)(sl + 4 * u) * n];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sl < a.n_slabs; sl += 4) s += p[(long long)sl * n];
  }
  part[g][o] = s;
  __syncthreads();
  if (g == 0 && j < n && isb) {
    const float t = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
    a.dbias[j] = a.db_accumulate ? a.dbias[j] + t : t;
  } else if (g == 0 && j < n) {
    const float t = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
    const int taps = a.kh * a.kw;
    const int co = (int)(j % a.cout);
    const int ci = (int)((j / a.cout) % a.cin);
    const int tap = (int)(j / ((long long)a.cout * a.cin));
    float* d = a.dw + ((long long)co * a.cin + ci) * taps + tap;
    *d = a.accumulate ? *d + t : t;
  }
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" long long fcvsr_conv2d_wgrad_scratch_elems(int B, int Ho, int Wo, int cin, int cout, int kh, int kw) {
  const long long npix = (long long)B * Ho * Wo;
  long long n_slabs = (npix + 511) / 512;
  if (n_slabs > 96) n_slabs = 96;
  if (n_slabs < 1) n_slabs = 1;
  return n_slabs * kh * kw * (long long)cin * cout;
}

extern "C" int fcvsr_conv2d_wgrad(const fcvsr_view* x, const fcvsr_view* gy, int B, int H, int W, int kh, int kw, int stride, int pad,
                                  float* dw, float* scratch, long long scratch_elems, void* stream) {
  FCVSR_CHECK_ARG(x && gy && dw && scratch, "null argument");
  FCVSR_CHECK_ARG(x->dtype == FCVSR_F32 && gy->dtype == FCVSR_F32 && x->sc == 1 && gy->sc == 1 && x->ptr && gy->ptr,
                  "x and gy must be channel-contiguous f32 views");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && kh >= 1 && kw >= 1 && stride >= 1 && pad >= 0, "bad geometry");
  FCVSR_CHECK_ARG((long long)B * H * W < (1ll << 31), "too many pixels for 32-bit pixel indices");
  WgradArgs a;
  a.x = to_view(*x);
  a.gy = to_view(*gy);
  a.B = B; a.H = H; a.W = W; a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad;
  a.Ho = (H + 2 * pad - kh) / stride + 1;
  a.Wo = (W + 2 * pad - kw) / stride + 1;
  FCVSR_CHECK_ARG(a.Ho > 0 && a.Wo > 0, "empty output");
  a.cin = x->c; a.cout = gy->c;
  a.npix = (long long)B * a.Ho * a.Wo;
  const long long need = fcvsr_conv2d_wgrad_scratch_elems(B, a.Ho, a.Wo, a.cin, a.cout, kh, kw);
  FCVSR_CHECK_ARG(scratch_elems >= need, "scratch too small (fcvsr_conv2d_wgrad_scratch_elems)");
  a.n_slabs = (int)(need / ((long long)kh * kw * a.cin * a.cout));
  a.slab_pix = (a.npix + a.n_slabs - 1) / a.n_slabs;
  a.slab_pix = (a.slab_pix + kWgPx - 1) / kWgPx * kWgPx;
  a.partial = scratch;
  a.dw = dw; a.accumulate = g_wgrad_accumulate;
  a.dbp = nullptr; a.dbias = nullptr; a.db_accumulate = 0;
  hipStream_t st = (hipStream_t)stream;
  const int nco = (a.cout + kWgCo - 1) / kWgCo, nci = (a.cin + kWgCi - 1) / kWgCi;
  FCVSR_CHECK_ARG(kh * kw <= 65535 && (long long)nco * nci <= 65535, "grid too large");
  const bool c4 = a.cin == 4 && a.cout == 4 && stride == 1 && kh == kw && (kh == 1 || kh == 3 || kh == 5) && pad == kh / 2 &&
                  ((uintptr_t)x->ptr % 16) == 0 && ((uintptr_t)gy->ptr % 16) == 0 && x->sx % 4 == 0 && x->sy % 4 == 0 && x->sb % 4 == 0 &&
                  gy->sx % 4 == 0 && gy->sy % 4 == 0 && gy->sb % 4 == 0;
  if (c4 && kh == 1) hipLaunchKernelGGL(wgrad_c4_kernel<1>, dim3(a.n_slabs), dim3(256), 0, st, a);
  else if (c4 && kh == 3) hipLaunchKernelGGL(wgrad_c4_kernel<3>, dim3(a.n_slabs), dim3(256), 0, st, a);
  else if (c4) hipLaunchKernelGGL(wgrad_c4_kernel<5>, dim3(a.n_slabs), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(wgrad_partial_kernel, dim3(a.n_slabs, kh * kw, nco * nci), dim3(256), 0, st, a);
  FCVSR_LAUNCH_CHECK();
  const long long n = (long long)a.cout * a.cin * kh * kw;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, a);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================================================
// Matrix-core weight gradient (16-bit operands, f32 accumulate): the throughput path of the training step for the 3x3 / 1x1
// stride-1 layers with multiples of 64 channels (SCNetbk bodies, conv_KP, F, conv3, recorb0: ~95 % of the weight-gradient FLOPs).
//
//   dW[tap][co][ci] = sum_p gy[p][co] * x[p + tap][ci]   =  per tap a (64 x K) x (K x 64) GEMM with K = pixels
//
// The reduction runs over PIXELS, but NHWC keeps channels contiguous, so both MFMA operands (8 consecutive k per lane) need the
// transposed image.  A workgroup transposes while staging: f32 -> bf16, ds_write_b16 into [channel][row][x] LDS images whose
// channel pitch is padded by 16 bytes (conflict-free ds_read_b128 over 32 channels).  A tap's horizontal shift would misalign the
// 16-byte fragment reads, so the input tile is stored three times, pre-shifted by kx - 1; vertical shifts are row offsets.
//   * workgroup = 256 threads = 4 waves = the 2 x 2 (cout, cin) fragment pairs of a 64 x 64 block, all 9 taps each: 9 f32
//     accumulator fragments per wave (144 VGPRs) that persist over the workgroup's whole slab of 4 x 32 pixel tiles;
//   * per 16-pixel k-step: 1 + 9 ds_read_b128 feed 9 MFMAs;
//   * partial sums per slab go to the same scratch layout as the exact kernel and are added in slab order by
//     wgrad_reduce_kernel: deterministic, no atomics.
#include "mfma_util.h"

namespace fcvsr {

constexpr int kGTY = 4, kGTX = 32;
constexpr int kGyPitch = kGTY * kGTX * 2 + 16;          // bytes per cout row of the gy^T image (4 rows x 32 px bf16 + pad)

template <int KS>
__global__ __launch_bounds__(256, 1) void wgrad_mfma_kernel(WgradArgs a, int tiles_x, int tiles_y, int tiles_per_slab) {
  constexpr int PAD = KS / 2, HY = kGTY + 2 * PAD, NKX = KS;
  constexpr int kXPitch = HY * kGTX * 2 + 16;           // bytes per cin row of one pre-shifted x^T image
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char* gy_s = lds;                             // [64 co][kGyPitch]
  unsigned char* x_s = lds + 64 * kGyPitch;              // [NKX][64 ci][kXPitch]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int mf = wave >> 1, nf = wave & 1;               // this wave's (cout, cin) fragment pair
  const int nci = a.cin / 64;
  const int co0 = (blockIdx.y / nci) * 64, ci0 = (blockIdx.y % nci) * 64;
  const int slab = blockIdx.x;
  const int total_tiles = a.B * tiles_x * tiles_y;
  const int t_begin = slab * tiles_per_slab;
  int t_end = t_begin + tiles_per_slab;
  if (t_end > total_tiles) t_end = total_tiles;

  f32x16_t acc[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  for (int tile = t_begin; tile < t_end; ++tile) {
    const int b = tile / (tiles_x * tiles_y);
    const int t2 = tile - b * tiles_x * tiles_y;
    const int ty0 = (t2 / tiles_x) * kGTY, tx0 = (t2 % tiles_x) * kGTX;
    __syncthreads();                                     // the previous tile's images are no longer read
    // ---- gy tile -> gy^T image: thread = (pixel, 4-cout quad), 16 quads per pixel, 16 pixels per pass -----------------------
    {
      const int q = tid & 15, p0 = tid >> 4;
      for (int p = p0; p < kGTY * kGTX; p += 16) {
        const int y = p >> 5, x = p & 31;
        const int gyy = ty0 + y, gxx = tx0 + x;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gyy < a.Ho && gxx < a.Wo)
          v = *reinterpret_cast<const float4*>(a.gy.p + (long long)b * a.gy.sb + (long long)gyy * a.gy.sy + (long long)gxx * a.gy.sx + co0 + q * 4);
        const uint2 pk = cvt4<true>(v);
        unsigned char* d = gy_s + (q * 4) * kGyPitch + (y * kGTX + x) * 2;
        *reinterpret_cast<uint16_t*>(d) = (uint16_t)(pk.x & 0xffff);
        *reinterpret_cast<uint16_t*>(d + kGyPitch) = (uint16_t)(pk.x >> 16);
        *reinterpret_cast<uint16_t*>(d + 2 * kGyPitch) = (uint16_t)(pk.y & 0xffff);
        *reinterpret_cast<uint16_t*>(d + 3 * kGyPitch) = (uint16_t)(pk.y >> 16);
      }
    }
    // ---- x halo tile -> NKX pre-shifted x^T images: image kx holds x[.., tx0 + xx + kx - PAD] at column xx ---------------------
    {
      constexpr int HXW = kGTX + 2 * PAD;                // halo width
      const int q = tid & 15, p0 = tid >> 4;
      for (int p = p0; p < HY * HXW; p += 16) {
        const int hy = p / HXW, hx = p - hy * HXW;
        const int iy = ty0 + hy - PAD, ix = tx0 + hx - PAD;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
          v = *reinterpret_cast<const float4*>(a.x.p + (long long)b * a.x.sb + (long long)iy * a.x.sy + (long long)ix * a.x.sx + ci0 + q * 4);
        const uint2 pk = cvt4<true>(v);
        const uint16_t e0 = (uint16_t)(pk.x & 0xffff), e1 = (uint16_t)(pk.x >> 16), e2 = (uint16_t)(pk.y & 0xffff), e3 = (uint16_t)(pk.y >> 16);
#pragma unroll
        for (int kx = 0; kx < NKX; ++kx) {
          const int xx = hx - kx;                        // column of this halo pixel in image kx
          if (xx >= 0 && xx < kGTX) {
            unsigned char* d = x_s + ((kx * 64 + q * 4) * kXPitch) + (hy * kGTX + xx) * 2;
            *reinterpret_cast<uint16_t*>(d) = e0;
            *reinterpret_cast<uint16_t*>(d + kXPitch) = e1;
            *reinterpret_cast<uint16_t*>(d + 2 * kXPitch) = e2;
            *reinterpret_cast<uint16_t*>(d + 3 * kXPitch) = e3;
          }
        }
      }
    }
    __syncthreads();
    // ---- 8 k-steps of 16 pixels (row y, half s): A = gy^T[co][k], B = x^T[kx][ci][row y + ky][k] ---------------------------------
    const unsigned char* ga = gy_s + (mf * 32 + r) * kGyPitch + h * 16;
    const unsigned char* xa = x_s + (nf * 32 + r) * kXPitch + h * 16;
#pragma unroll
    for (int ks = 0; ks < kGTY * 2; ++ks) {
      const int y = ks >> 1, s = ks & 1;
      const uint4 af = *reinterpret_cast<const uint4*>(ga + (y * kGTX + s * 16) * 2);
#pragma unroll
      for (int t = 0; t < KS * KS; ++t) {
        const int ky = t / KS, kx = t - ky * KS;
        const uint4 bf = *reinterpret_cast<const uint4*>(xa + kx * 64 * kXPitch + ((y + ky) * kGTX + s * 16) * 2);
        acc[t] = mfma<true>(af, bf, acc[t]);
      }
    }
  }
  // ---- partial[slab][tap][ci][co]: lane (r = ci, h) holds couts (i&3) + 8 (i>>2) + 4h of its fragment ---------------------------
  float* pp = a.partial + ((long long)slab * (KS * KS)) * a.cin * a.cout;
#pragma unroll
  for (int t = 0; t < KS * KS; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co0 + mf * 32 + 8 * g + 4 * h, ci = ci0 + nf * 32 + r;
      *reinterpret_cast<float4*>(pp + ((long long)t * a.cin + ci) * a.cout + co) =
          make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Second form (round 3): the same products without the transposing stores.  The workgroup stages its tiles PIXEL-major, as they
// lie in memory - 8 floats -> one 16-byte ds_write_b128 of bf16 per lane (the first form issues four 2-byte ds_write_b16 per
// 4 channels, three times over for the pre-shifted x images: ~12 us of staging per 4 x 32 tile) - and lets the LDS do the
// transpose on the way out: ds_read_b64_tr_b16 hands lane i of a 16-lane group channel i of 4 consecutive pixel records, which is
// exactly the "8 consecutive k of one row" the MFMA operands want when k runs over pixels.  A tap's shift is a whole number of
// 128-byte pixel records, so ONE x image serves all nine taps, and a fragment of x row yy is read once for the (up to three)
// taps that use it: 88 transposed 8-byte reads per wave and tile feed its 72 MFMAs (first form: 80 ds_read_b128).
// Records are 8 chunks of 16 bytes; chunk c of pixel record P sits in slot c ^ (4 * ((P >> 1) & 1)): the 4 rows x 2 blocks of a
// 32-lane half then cover all 64 banks once, for any start pixel.  Same MFMA instruction, same k order per tap: results are
// bit-identical to the first form.
typedef __attribute__((ext_vector_type(4))) short wg_s4_t;

__device__ __forceinline__ uint2 lds_tr16(const unsigned char* p) {
  const wg_s4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}

template <int KS>
__global__ __launch_bounds__(256, 2) void wgrad_tr_kernel(WgradArgs a, int tiles_x, int tiles_y, int tiles_per_slab) {
  constexpr int PAD = KS / 2, HY = kGTY + 2 * PAD, HXW = kGTX + 2 * PAD;
  constexpr int NGY = kGTY * kGTX, NX = HY * HXW;
  __shared__ __align__(16) unsigned char gy_s[NGY * 128];
  __shared__ __align__(16) unsigned char x_s[NX * 128];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mf = wave >> 1, nf = wave & 1;               // this wave's (cout, cin) fragment pair
  const int nci = a.cin / 64;
  const int co0 = (blockIdx.y / nci) * 64, ci0 = (blockIdx.y % nci) * 64;
  const int slab = blockIdx.x;
  const int total_tiles = a.B * tiles_x * tiles_y;
  const int t_begin = slab * tiles_per_slab;
  int t_end = t_begin + tiles_per_slab;
  if (t_end > total_tiles) t_end = total_tiles;
  // transposed-read addresses: 16-lane group g4 = (k half kg, channel half mh); lane 4q + p supplies row q, channels 4p .. 4p+3
  const int g4 = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3, kg = g4 >> 1, mh = g4 & 1;
  const int cA = mf * 4 + 2 * mh + (p4 >> 1), cB = nf * 4 + 2 * mh + (p4 >> 1);
  // gy: k-steps start at multiples of 16 pixels, so the slot XOR depends on q alone
  const unsigned addrA = (unsigned)((8 * kg + q) * 128 + 16 * (cA ^ (((q >> 1) & 1) << 2)) + 8 * (p4 & 1));
  unsigned addrB[4];                                      // x: the start pixel's residue mod 4 (tap shift, row pitch 34) picks the variant
#pragma unroll
  for (int b4 = 0; b4 < 4; ++b4)
    addrB[b4] = (unsigned)((8 * kg + q) * 128 + 16 * (cB ^ ((((b4 + q) >> 1) & 1) << 2)) + 8 * (p4 & 1));

  f32x16_t acc[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int c8 = tid & 7, ps = tid >> 3;                  // staging role: 8-channel chunk, pixel slot (32 per pass)
  const bool do_bias = a.dbp != nullptr && ci0 == 0;      // uniform: the cin-block-0 workgroup of every cout block sums gy's columns
  float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int b = tile / (tiles_x * tiles_y);
    const int t2 = tile - b * tiles_x * tiles_y;
    const int ty0 = (t2 / tiles_x) * kGTY, tx0 = (t2 % tiles_x) * kGTX;
    // ---- staging: gy (4 passes of 32 pixels) and x (7 passes for KS = 3, 4 for KS = 1), two float4 per lane and pass; x in two
    // batches so that at most 7 passes (56 registers) are in flight next to the 144 accumulator registers -----------------------
    constexpr int PGY = NGY / 32, PX = (NX + 31) / 32, PXA = PX > 4 ? 3 : PX;
    auto load_x = [&](const int i, float4* v) {
      const int P = i * 32 + ps;
      const int hy = P / HXW, hx = P - hy * HXW;
      const int iy = ty0 + hy - PAD, ix = tx0 + hx - PAD;
      v[0] = v[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (P < NX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
        const float* src = a.x.p + (long long)b * a.x.sb + (long long)iy * a.x.sy + (long long)ix * a.x.sx + ci0 + c8 * 8;
        v[0] = *reinterpret_cast<const float4*>(src);
        v[1] = *reinterpret_cast<const float4*>(src + 4);
      }
    };
    auto store_rec = [&](unsigned char* img, const int P, const float4* v) {
      const uint2 lo = cvt4<true>(v[0]), hi = cvt4<true>(v[1]);
      *reinterpret_cast<uint4*>(img + P * 128 + 16 * (c8 ^ (((P >> 1) & 1) << 2))) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    {
      float4 gv[PGY][2], xv[PXA][2];
#pragma unroll
      for (int i = 0; i < PGY; ++i) {
        const int P = i * 32 + ps, y = P >> 5, x = P & 31;
        const int gyy = ty0 + y, gxx = tx0 + x;
        gv[i][0] = gv[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gyy < a.Ho && gxx < a.Wo) {
          const float* src = a.gy.p + (long long)b * a.gy.sb + (long long)gyy * a.gy.sy + (long long)gxx * a.gy.sx + co0 + c8 * 8;
          gv[i][0] = *reinterpret_cast<const float4*>(src);
          gv[i][1] = *reinterpret_cast<const float4*>(src + 4);
        }
      }
#pragma unroll
      for (int i = 0; i < PXA; ++i) load_x(i, xv[i]);
      __syncthreads();                                   // the previous tile's images are no longer read
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < PGY; ++i) {
          bs[0] += gv[i][0].x; bs[1] += gv[i][0].y; bs[2] += gv[i][0].z; bs[3] += gv[i][0].w;
          bs[4] += gv[i][1].x; bs[5] += gv[i][1].y; bs[6] += gv[i][1].z; bs[7] += gv[i][1].w;
        }
      }
#pragma unroll
      for (int i = 0; i < PGY; ++i) store_rec(gy_s, i * 32 + ps, gv[i]);
#pragma unroll
      for (int i = 0; i < PXA; ++i) store_rec(x_s, i * 32 + ps, xv[i]);
    }
    if (PX > PXA) {
      float4 xv[PX - PXA > 0 ? PX - PXA : 1][2];
#pragma unroll
      for (int i = PXA; i < PX; ++i) load_x(i, xv[i - PXA]);
#pragma unroll
      for (int i = PXA; i < PX; ++i)
        if (i * 32 + ps < NX) store_rec(x_s, i * 32 + ps, xv[i - PXA]);
    }
    __syncthreads();
    // ---- A fragments of the tile's 8 k-steps (row y, half s), kept for the whole tile --------------------------------------
    uint4 af[kGTY][2];
#pragma unroll
    for (int y = 0; y < kGTY; ++y)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const unsigned char* pa = gy_s + addrA + (y * kGTX + s2 * 16) * 128;
        const uint2 lo = lds_tr16(pa), hi = lds_tr16(pa + 4 * 128);
        af[y][s2] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    // ---- x row yy, half s, shift kx: one fragment for the taps (ky = yy - y, kx) with 0 <= y < 4 ---------------------------------
#pragma unroll
    for (int yy = 0; yy < HY; ++yy)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int P0 = yy * HXW + s2 * 16 + kx;          // start pixel of the k-step in the x image (compile-time)
          const unsigned char* pb = x_s + addrB[P0 & 3] + P0 * 128;
          const uint2 lo = lds_tr16(pb), hi = lds_tr16(pb + 4 * 128);
          const uint4 bf = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
          for (int ky = 0; ky < KS; ++ky) {
            const int y = yy - ky;
            if (y >= 0 && y < kGTY) acc[ky * KS + kx] = mfma<true>(af[y][s2], bf, acc[ky * KS + kx]);
          }
        }
  }
  // ---- partial[slab][tap][ci][co]: lane (r = ci, h) holds couts (i&3) + 8 (i>>2) + 4h of its fragment ---------------------------
  const int r = lane & 31, h = lane >> 5;
  float* pp = a.partial + ((long long)slab * (KS * KS)) * a.cin * a.cout;
#pragma unroll
  for (int t = 0; t < KS * KS; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co0 + mf * 32 + 8 * g + 4 * h, ci = ci0 + nf * 32 + r;
      *reinterpret_cast<float4*>(pp + ((long long)t * a.cin + ci) * a.cout + co) =
          make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
    }
  if (do_bias) {                                         // 32 pixel slots x 64 channels -> 64 column sums of this slab, slots in order
    __syncthreads();
    float* red = reinterpret_cast<float*>(gy_s);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) red[ps * 64 + c8 * 8 + jj] = bs[jj];
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
      for (int rr = 0; rr < 32; ++rr) t += red[rr * 64 + tid];
      a.dbp[(long long)slab * a.cout + co0 + tid] = t;
    }
  }
}

}  // namespace fcvsr

static int wgrad_mfma_slabs(int B, int Ho, int Wo, int cin, int cout) {
  const int tiles = B * ((Ho + 3) / 4) * ((Wo + 31) / 32);
  int n = 256 / ((cin / 64) * (cout / 64));             // at most one workgroup per CU
  // Every slab writes (and the reduction reads) a full kh*kw*cin*cout f32 partial - 147 KB for a 64 -> 64 3x3 layer - so on the small
  // problems of a training step (4 clips of 128 x 128: 512 tiles) one slab per CU moves more partial bytes than the layer has
  // activations.  Measured all the same (round 3, 4 x 7 x 128 x 128 step, FCVSR_WGRAD_MIN_TILES = 1 / 4 / 8 / 16 / 32 tiles per slab: 43.1 /
  // 47.4 / 56.0 / 74.8 / 112.5 ms per step): a tile costs a workgroup ~12 us of staging (the f32 -> bf16 transposes through LDS), so
  // the default stays one slab per CU; the knob remains for larger batches.
  static int min_tiles = -1;
  if (min_tiles < 0) { const char* e = getenv("FCVSR_WGRAD_MIN_TILES"); min_tiles = e ? atoi(e) : 1; if (min_tiles < 1) min_tiles = 1; }
  const int cap = (tiles + min_tiles - 1) / min_tiles;
  if (n > cap) n = cap;
  if (n > tiles) n = tiles;
  if (n < 1) n = 1;
  return n;
}

// FCVSR_WGRAD_FORM: 2 (default) = wgrad_tr_kernel (pixel-major images, transposing LDS reads), 1 = wgrad_mfma_kernel (transposing stores)
static int launch_wgrad_mfma(const WgradArgs& a, dim3 grid, int kh, int tiles_x, int tiles_y, int per_slab, hipStream_t st) {
  static const int form = getenv("FCVSR_WGRAD_FORM") ? atoi(getenv("FCVSR_WGRAD_FORM")) : 2;
  if (form == 2) {
    if (kh == 3) hipLaunchKernelGGL(wgrad_tr_kernel<3>, grid, dim3(256), 0, st, a, tiles_x, tiles_y, per_slab);
    else hipLaunchKernelGGL(wgrad_tr_kernel<1>, grid, dim3(256), 0, st, a, tiles_x, tiles_y, per_slab);
    return 0;
  }
  if (kh == 3) {
    const size_t ldsb = 64 * kGyPitch + 3 * 64 * ((kGTY + 2) * kGTX * 2 + 16);
    static DevOnce attr3;
    hipError_t e = once_per_device(attr3, [&] {
      return hipFuncSetAttribute((const void*)wgrad_mfma_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    });
    if (e != hipSuccess) { set_error("fcvsr_conv2d_wgrad_mfma: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(wgrad_mfma_kernel<3>, grid, dim3(256), ldsb, st, a, tiles_x, tiles_y, per_slab);
  } else {
    const size_t ldsb = 64 * kGyPitch + 64 * (kGTY * kGTX * 2 + 16);
    hipLaunchKernelGGL(wgrad_mfma_kernel<1>, grid, dim3(256), ldsb, st, a, tiles_x, tiles_y, per_slab);
  }
  return 0;
}

extern "C" long long fcvsr_conv2d_wgrad_mfma_scratch_elems(int B, int Ho, int Wo, int cin, int cout, int kh, int kw) {
  return (long long)wgrad_mfma_slabs(B, Ho, Wo, cin, cout) * ((long long)kh * kw * cin * cout + cout);      // + the bias partials
}

/* One-shot: the next fcvsr_conv2d_wgrad_mfma / fcvsr_conv2d_wgrad_mfma_groups call of this thread also writes (accumulate = 0) or adds
 * (1) dL/dbias = sum over pixels of gy into dbias[cout].  Returns 1 when the build's weight-gradient form supports it (form 2), else 0
 * (the request is then ignored: use fcvsr_colsum). */
extern "C" int fcvsr_wgrad_set_bias_out(float* dbias, int accumulate) {
  static const int form = getenv("FCVSR_WGRAD_FORM") ? atoi(getenv("FCVSR_WGRAD_FORM")) : 2;
  if (form != 2) return 0;
  g_wgrad_bias_out = dbias; g_wgrad_bias_accumulate = accumulate ? 1 : 0;
  return 1;
}

extern "C" int fcvsr_conv2d_wgrad_mfma_eligible(int cin, int cout, int kh, int kw, int stride, int pad) {
  return (kh == kw && (kh == 1 || kh == 3) && stride == 1 && pad == kh / 2 && cin % 64 == 0 && cout % 64 == 0) ? 1 : 0;
}

// Same contract as fcvsr_conv2d_wgrad (f32 NHWC x / gy views, f32 (cout,cin,kh,kw) result), products in bf16 on the matrix cores.
extern "C" int fcvsr_conv2d_wgrad_mfma(const fcvsr_view* x, const fcvsr_view* gy, int B, int H, int W, int kh, int kw, int stride, int pad,
                                       float* dw, float* scratch, long long scratch_elems, void* stream) {
  FCVSR_CHECK_ARG(x && gy && dw && scratch, "null argument");
  FCVSR_CHECK_ARG(x->dtype == FCVSR_F32 && gy->dtype == FCVSR_F32 && x->sc == 1 && gy->sc == 1 && x->ptr && gy->ptr,
                  "x and gy must be channel-contiguous f32 views");
  FCVSR_CHECK_ARG(fcvsr_conv2d_wgrad_mfma_eligible(x->c, gy->c, kh, kw, stride, pad), "layer not eligible for the matrix-core weight gradient");
  FCVSR_CHECK_ARG(x->sx % 4 == 0 && x->sy % 4 == 0 && x->sb % 4 == 0 && gy->sx % 4 == 0 && gy->sy % 4 == 0 && gy->sb % 4 == 0 &&
                      ((uintptr_t)x->ptr % 16) == 0 && ((uintptr_t)gy->ptr % 16) == 0, "views must be 16-byte aligned");
  WgradArgs a;
  a.x = to_view(*x); a.gy = to_view(*gy);
  a.B = B; a.H = H; a.W = W; a.kh = kh; a.kw = kw; a.stride = 1; a.pad = pad;
  a.Ho = H; a.Wo = W; a.cin = x->c; a.cout = gy->c;
  a.npix = (long long)B * H * W;
  a.n_slabs = wgrad_mfma_slabs(B, H, W, a.cin, a.cout);
  float* const bias_out = g_wgrad_bias_out;
  g_wgrad_bias_out = nullptr;                            // one-shot
  FCVSR_CHECK_ARG(scratch_elems >= (long long)a.n_slabs * ((long long)kh * kw * a.cin * a.cout + a.cout), "scratch too small (fcvsr_conv2d_wgrad_mfma_scratch_elems)");
  a.slab_pix = 0;
  a.partial = scratch; a.dw = dw; a.accumulate = g_wgrad_accumulate;
  a.dbp = bias_out ? scratch + (long long)a.n_slabs * kh * kw * a.cin * a.cout : nullptr;
  a.dbias = bias_out; a.db_accumulate = g_wgrad_bias_accumulate;
  const int tiles_x = cdiv(W, kGTX), tiles_y = cdiv(H, kGTY);
  const int total = B * tiles_x * tiles_y;
  const int per_slab = (total + a.n_slabs - 1) / a.n_slabs;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(a.n_slabs, (a.cin / 64) * (a.cout / 64));
  { const int e = launch_wgrad_mfma(a, grid, kh, tiles_x, tiles_y, per_slab, st); if (e) return e; }
  FCVSR_LAUNCH_CHECK();
  const long long n = (long long)a.cout * a.cin * kh * kw;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64 + (a.dbp ? (a.cout + 63) / 64 : 0))), dim3(256), 0, st, a);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

// The same weight gradient summed over up to 3 problems that share the weight (the pyramid levels of a BlockRCB layer, reference
// CVSR_freq.py:766-777: one nn.Conv2d applied to every level): one matrix-core launch per problem into consecutive slab ranges of
// ONE scratch buffer, then ONE ordered reduction over all slabs (levels in order): no per-level gradient tensors, no additions.
extern "C" long long fcvsr_conv2d_wgrad_mfma_groups_scratch_elems(const int* B, const int* H, const int* W, int n_groups, int cin, int cout,
                                                                  int kh, int kw) {
  long long slabs = 0;
  for (int g = 0; g < n_groups; ++g) slabs += wgrad_mfma_slabs(B[g], H[g], W[g], cin, cout);
  return slabs * ((long long)kh * kw * cin * cout + cout);
}

extern "C" int fcvsr_conv2d_wgrad_mfma_groups(const fcvsr_view* xs, const fcvsr_view* gys, const int* B, const int* H, const int* W, int n_groups,
                                              int kh, int kw, int pad, float* dw, float* scratch, long long scratch_elems, void* stream) {
  FCVSR_CHECK_ARG(xs && gys && B && H && W && dw && scratch, "null argument");
  FCVSR_CHECK_ARG(n_groups >= 1 && n_groups <= 3, "1..3 problems");
  const int cin = xs[0].c, cout = gys[0].c;
  FCVSR_CHECK_ARG(fcvsr_conv2d_wgrad_mfma_eligible(cin, cout, kh, kw, 1, pad), "layer not eligible for the matrix-core weight gradient");
  const long long per = (long long)kh * kw * cin * cout;
  hipStream_t st = (hipStream_t)stream;
  int slab0 = 0, slabs_total = 0;
  for (int g = 0; g < n_groups; ++g) slabs_total += wgrad_mfma_slabs(B[g], H[g], W[g], cin, cout);
  float* const bias_out = g_wgrad_bias_out;
  g_wgrad_bias_out = nullptr;                            // one-shot
  FCVSR_CHECK_ARG(scratch_elems >= (long long)slabs_total * (per + cout), "scratch too small (fcvsr_conv2d_wgrad_mfma_groups_scratch_elems)");
  float* const dbp_all = bias_out ? scratch + (long long)slabs_total * per : nullptr;
  WgradArgs a;
  for (int g = 0; g < n_groups; ++g) {
    const fcvsr_view* x = xs + g, *gy = gys + g;
    FCVSR_CHECK_ARG(x->dtype == FCVSR_F32 && gy->dtype == FCVSR_F32 && x->sc == 1 && gy->sc == 1 && x->ptr && gy->ptr && x->c == cin && gy->c == cout,
                    "x and gy must be channel-contiguous f32 views of the same channel counts");
    FCVSR_CHECK_ARG(x->sx % 4 == 0 && x->sy % 4 == 0 && x->sb % 4 == 0 && gy->sx % 4 == 0 && gy->sy % 4 == 0 && gy->sb % 4 == 0 &&
                        ((uintptr_t)x->ptr % 16) == 0 && ((uintptr_t)gy->ptr % 16) == 0, "views must be 16-byte aligned");
    a.x = to_view(*x); a.gy = to_view(*gy);
    a.B = B[g]; a.H = H[g]; a.W = W[g]; a.kh = kh; a.kw = kw; a.stride = 1; a.pad = pad;
    a.Ho = H[g]; a.Wo = W[g]; a.cin = cin; a.cout = cout;
    a.npix = (long long)B[g] * H[g] * W[g];
    a.n_slabs = wgrad_mfma_slabs(B[g], H[g], W[g], cin, cout);
    FCVSR_CHECK_ARG(scratch_elems >= (long long)(slab0 + a.n_slabs) * per, "scratch too small (fcvsr_conv2d_wgrad_mfma_groups_scratch_elems)");
    a.slab_pix = 0;
    a.partial = scratch + (long long)slab0 * per; a.dw = dw; a.accumulate = g_wgrad_accumulate;
    a.dbp = dbp_all ? dbp_all + (long long)slab0 * cout : nullptr; a.dbias = bias_out; a.db_accumulate = g_wgrad_bias_accumulate;
    const int tiles_x = cdiv(W[g], kGTX), tiles_y = cdiv(H[g], kGTY);
    const int total = B[g] * tiles_x * tiles_y;
    const int per_slab = (total + a.n_slabs - 1) / a.n_slabs;
    const dim3 grid(a.n_slabs, (cin / 64) * (cout / 64));
    { const int e = launch_wgrad_mfma(a, grid, kh, tiles_x, tiles_y, per_slab, st); if (e) return e; }
    FCVSR_LAUNCH_CHECK();
    slab0 += a.n_slabs;
  }
  a.n_slabs = slab0;
  a.partial = scratch;
  a.dbp = dbp_all;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per + 63) / 64 + (dbp_all ? (cout + 63) / 64 : 0))), dim3(256), 0, st, a);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" void fcvsr_wgrad_set_accumulate(int on) { g_wgrad_accumulate = on ? 1 : 0; }
extern "C" int fcvsr_wgrad_get_accumulate(void) { return g_wgrad_accumulate; }
