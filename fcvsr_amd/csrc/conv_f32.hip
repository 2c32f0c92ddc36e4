// Exact-f32 convolution on the matrix cores: v_mfma_f32_32x32x2_f32 takes f32 operands, accumulates in f32 and is bit-equal to
// an fmaf chain (MI355X guide, "FP32-input MFMA"), at the f32 vector peak (157 TFLOP/s) instead of the ~10 TFLOP/s the direct
// VALU kernel (conv_direct.hip) reaches.  This is the arithmetic of `precision="f32"` - the mode that meets the north star's
// 1e-4 max-abs gate - for the layers that carry the FLOPs: 3x3 / 1x1, stride 1, one dense NHWC f32 source whose channel count is
// a multiple of 32, f32 NHWC destination (reference CVSR_freq.py:705-803 RCB / BlockRCB / SCGroupbk bodies, :1409-1416
// conv_KP / F, :1430 conv3, :2608 recorb0, the cross-scale 1x1 convolutions).  Everything else stays on the direct kernel.
//
//   * workgroup = 256 threads = 4 waves -> 4 x 32 output pixels (1x1: 128 flat pixels) x 64 output channels;
//   * operand roles swapped like conv_res.hip: weights are the MFMA A operand (rows = couts), pixels the B operand, so a lane
//     ends up with 4 consecutive output channels of one pixel per accumulator quad = one 16-byte f32 store, no LDS transpose;
//   * input channels go through LDS in chunks of 32: the (4+2) x (32+2) halo tile once per chunk, the 64 x 32 weight block once
//     per tap; rows padded to 36 floats (144 bytes) make every ds_read_b128 conflict-free;
//   * a lane reads 4 consecutive k of its row at once: K-steps are grouped in eights, lane half h takes k0 + 4h .. k0 + 4h + 3
//     (the same permutation on both operands, so every product meets its partner);
//   * 32 MFMAs of 64 cycles per wave between barriers: the matrix pipe, not staging, sets the pace.
#include "common.h"

namespace fcvsr {

typedef __attribute__((ext_vector_type(16))) float f32x16v_t;

struct F32Args {
  View src, res[2], dst;
  int B, H, W;
  int ks;                  // 1 or 3
  int cin, cout, cout_pad; // weights [ks*ks][cout_pad][cin] f32
  const float* w;
  const float* bias;
  int act;
  float slope;
  const float* slope_ptr;
  int n_res;
  float rs[2];
  int tiles_x, tiles_y;    // 3x3: 4 x 32 pixel tiles per image; 1x1: tiles_x = flat tiles of 128 pixels, tiles_y = 1
  int n_nblk;              // cout blocks of 32 * NF
  int ps;                  // PixelShuffle(2) store: couts are packed sub-pixel-major (row (2i+j)*(cout/4) + c <- channel 4c+2i+j), dst is (B,2H,2W,cout/4)
  int cq4;                 // cout / 4
  int generic;             // scalar epilogue: any cout (conv_last0: 1 or 3), any dst / res strides
};

constexpr int kFCK = 32, kFLD = kFCK + 4;           // channel chunk, padded LDS row (floats)

template <int KS, int NF>
__global__ __launch_bounds__(256, 2) void conv_f32mfma_kernel(F32Args a) {
  constexpr int PAD = KS / 2;
  constexpr int HH = 4 + 2 * PAD, HWD = 32 + 2 * PAD, NHP = KS == 3 ? HH * HWD : 128;
  extern __shared__ __align__(16) float ldsf[];
  float* X_s = ldsf;                                 // [NHP][kFLD]
  float* W_s = ldsf + NHP * kFLD;                    // [32 * NF][kFLD]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nb = blockIdx.x % a.n_nblk;
  const int tl = blockIdx.x / a.n_nblk;
  const int n0 = nb * 32 * NF;
  int b = 0, ty0 = 0, tx0 = 0;
  long long flat0 = 0;
  const long long npix = (long long)a.B * a.H * a.W;
  if (KS == 3) {
    const int per_img = a.tiles_x * a.tiles_y;
    b = tl / per_img;
    const int t2 = tl - b * per_img;
    ty0 = (t2 / a.tiles_x) * 4;
    tx0 = (t2 % a.tiles_x) * 32;
  } else {
    flat0 = (long long)tl * 128;
  }

  f32x16v_t acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nf][i] = 0.f;

  const long long wtap = (long long)a.cout_pad * a.cin;
  for (int c0 = 0; c0 < a.cin; c0 += kFCK) {
    __syncthreads();                                 // the previous chunk's tiles are no longer read
    // ---- stage the pixel tile of channels [c0, c0 + 32): 8 lanes x float4 per pixel ---------------------------------------
    {
      const int q = tid & 7, p0 = tid >> 3;          // 32 pixels per pass
      for (int hp = p0; hp < NHP; hp += 32) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KS == 3) {
          const int hy = hp / HWD, hx = hp - hy * HWD;
          const int iy = ty0 + hy - PAD, ix = tx0 + hx - PAD;
          if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
            v = *reinterpret_cast<const float4*>(a.src.p + (long long)b * a.src.sb + (long long)iy * a.src.sy + (long long)ix * a.src.sx + c0 + q * 4);
        } else {
          const long long p = flat0 + hp;
          if (p < npix) v = *reinterpret_cast<const float4*>(a.src.p + p * a.src.sx + c0 + q * 4);
        }
        *reinterpret_cast<float4*>(X_s + hp * kFLD + q * 4) = v;
      }
    }
    for (int tap = 0; tap < KS * KS; ++tap) {
      if (tap > 0) __syncthreads();                  // the previous tap's weights are no longer read
      // ---- this tap's 64 couts x 32 cins: 8 lanes x float4 per cout, 32 couts per pass --------------------------------------
      {
        const int q = tid & 7, co = tid >> 3;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(a.w + tap * wtap + (long long)(n0 + co + 32 * i) * a.cin + c0 + q * 4);
          *reinterpret_cast<float4*>(W_s + (co + 32 * i) * kFLD + q * 4) = v;
        }
      }
      __syncthreads();
      const int ky = tap / KS, kx = tap - ky * KS;
      const float* xrow = X_s + (KS == 3 ? ((wave + ky) * HWD + r + kx) : (wave * 32 + r)) * kFLD + 4 * h;
      const float* wrow = W_s + r * kFLD + 4 * h;
#pragma unroll
      for (int k8 = 0; k8 < kFCK / 8; ++k8) {
        const float4 xv = *reinterpret_cast<const float4*>(xrow + k8 * 8);
        const float4 w0 = *reinterpret_cast<const float4*>(wrow + k8 * 8);
        float4 w1 = w0;
        if (NF > 1) w1 = *reinterpret_cast<const float4*>(wrow + 32 * kFLD + k8 * 8);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, xv.x, acc[0], 0, 0, 0);
        if (NF > 1) acc[NF - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.x, xv.x, acc[NF - 1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, xv.y, acc[0], 0, 0, 0);
        if (NF > 1) acc[NF - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.y, xv.y, acc[NF - 1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, xv.z, acc[0], 0, 0, 0);
        if (NF > 1) acc[NF - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.z, xv.z, acc[NF - 1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, xv.w, acc[0], 0, 0, 0);
        if (NF > 1) acc[NF - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.w, xv.w, acc[NF - 1], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane (r, h) holds, for pixel r of tile row `wave`, couts n0 + 32 nf + 8 g + 4 h + [0,4) in registers 4g..4g+3 ----
  float ns = 1.f;
  if (a.act == FCVSR_ACT_RELU) ns = 0.f;
  else if (a.act == FCVSR_ACT_LEAKY) ns = a.slope;
  else if (a.act == FCVSR_ACT_PRELU) ns = *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<uintptr_t>(a.slope_ptr));
  bool ok;
  int py, px, bb;
  if (KS == 3) {
    py = ty0 + wave; px = tx0 + r; bb = b;
    ok = py < a.H && px < a.W;
  } else {
    const long long p = flat0 + wave * 32 + r;
    ok = p < npix;
    px = (int)(p % a.W);
    py = (int)((p / a.W) % a.H);
    bb = (int)(p / ((long long)a.W * a.H));
  }
  if (!ok) return;
  const long long dpix = (long long)bb * a.dst.sb + (long long)py * a.dst.sy + (long long)px * a.dst.sx;
  const long long r0pix = (long long)bb * a.res[0].sb + (long long)py * a.res[0].sy + (long long)px * a.res[0].sx;
  const long long r1pix = (long long)bb * a.res[1].sb + (long long)py * a.res[1].sy + (long long)px * a.res[1].sx;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + 32 * nf + 8 * g + 4 * h;
      if (n >= a.cout) continue;
      float x[4] = {acc[nf][4 * g], acc[nf][4 * g + 1], acc[nf][4 * g + 2], acc[nf][4 * g + 3]};
      if (a.generic) {
        // skinny / strided layers (conv_last0: 1 or 3 output channels into the NCHW result): element by element
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e < a.cout) {
            float v = x[e] + (a.bias ? a.bias[n + e] : 0.f);
            v = v >= 0.f ? v : v * ns;
            if (a.n_res > 0) v = fmaf(a.rs[0], a.res[0].p[r0pix + (long long)(n + e) * a.res[0].sc], v);
            if (a.n_res > 1) v = fmaf(a.rs[1], a.res[1].p[r1pix + (long long)(n + e) * a.res[1].sc], v);
            a.dst.p[dpix + (long long)(n + e) * a.dst.sc] = v;
          }
        }
        continue;
      }
      if (a.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n);
        x[0] += b4.x; x[1] += b4.y; x[2] += b4.z; x[3] += b4.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = x[e] >= 0.f ? x[e] : x[e] * ns;       // ns = 0 (ReLU), slope, or 1 (none): no branch on `act`
      if (a.ps) {
        // packed row n = sp * (cout/4) + c: 4 consecutive rows = channels c..c+3 of sub-pixel sp -> one store at (2y+i, 2x+j)
        const int sp = n / a.cq4, c = n - sp * a.cq4;
        const long long o = (long long)bb * a.dst.sb + (long long)(2 * py + (sp >> 1)) * a.dst.sy + (long long)(2 * px + (sp & 1)) * a.dst.sx + c;
        *reinterpret_cast<float4*>(a.dst.p + o) = make_float4(x[0], x[1], x[2], x[3]);
        continue;
      }
      if (a.n_res > 0) {
        const float4 t = *reinterpret_cast<const float4*>(a.res[0].p + r0pix + n);
        x[0] = fmaf(a.rs[0], t.x, x[0]); x[1] = fmaf(a.rs[0], t.y, x[1]); x[2] = fmaf(a.rs[0], t.z, x[2]); x[3] = fmaf(a.rs[0], t.w, x[3]);
      }
      if (a.n_res > 1) {
        const float4 t = *reinterpret_cast<const float4*>(a.res[1].p + r1pix + n);
        x[0] = fmaf(a.rs[1], t.x, x[0]); x[1] = fmaf(a.rs[1], t.y, x[1]); x[2] = fmaf(a.rs[1], t.z, x[2]); x[3] = fmaf(a.rs[1], t.w, x[3]);
      }
      *reinterpret_cast<float4*>(a.dst.p + dpix + n) = make_float4(x[0], x[1], x[2], x[3]);
    }
}

static bool dense16(const fcvsr_view& v, int cmul) {
  return v.ptr && v.dtype == FCVSR_F32 && v.sc == 1 && v.c % cmul == 0 && v.sx % 4 == 0 && v.sy % 4 == 0 && v.sb % 4 == 0 &&
         ((uintptr_t)v.ptr % 16) == 0;
}

}  // namespace fcvsr

using namespace fcvsr;

// 1 when fcvsr_conv2d_f32mfma takes this layer (the callers route everything else to fcvsr_conv2d)
static int f32_mode(const fcvsr_conv_desc* d) {     // 0 no, 1 vector epilogue, 2 pixel-shuffle epilogue, 3 generic (scalar) epilogue
  if (!d || d->n_src != 1 || d->kh != d->kw || (d->kh != 1 && d->kh != 3) || d->stride != 1 || d->pad != d->kh / 2) return 0;
  if (d->gc_wmask || d->cout <= 0 || d->cout_pad % 64 || d->cout_pad < d->cout || !dense16(d->src[0], 32)) return 0;
  if (d->kh == 1 && !(d->src[0].sy == d->src[0].sx * d->W && d->src[0].sb == d->src[0].sy * d->H)) return 0;
  if (d->pixel_shuffle) {
    if (d->cout % 16 || d->n_res != 0 || !dense16(d->dst, 4) || d->dst.c != d->cout / 4) return 0;
    if (d->bias && ((uintptr_t)d->bias % 16)) return 0;
    return 2;
  }
  bool vec = d->cout % 4 == 0 && dense16(d->dst, 4) && d->dst.c == d->cout && !(d->bias && ((uintptr_t)d->bias % 16));
  for (int i = 0; i < d->n_res && vec; ++i) vec = dense16(d->res[i], 4);
  if (vec) return 1;
  if (d->cout > 32 || !d->dst.ptr || d->dst.dtype != FCVSR_F32) return 0;        // generic path: skinny layers only
  for (int i = 0; i < d->n_res; ++i)
    if (!d->res[i].ptr || d->res[i].dtype != FCVSR_F32) return 0;
  return 3;
}

extern "C" int fcvsr_conv2d_f32mfma_eligible(const fcvsr_conv_desc* d) { return f32_mode(d) != 0; }

// weight: f32 [kh*kw][cout_pad][cin] (cout_pad a multiple of 64, zero rows beyond cout; rows in sub-pixel-major order when
// pixel_shuffle is set, like fcvsr_conv2d_mfma); all other fields as fcvsr_conv2d
extern "C" int fcvsr_conv2d_f32mfma(const fcvsr_conv_desc* d, void* stream) {
  const int mode = f32_mode(d);
  FCVSR_CHECK_ARG(mode != 0, "layer not eligible (see fcvsr_conv2d_f32mfma_eligible)");
  FCVSR_CHECK_ARG(d->weight != nullptr && d->B > 0 && d->H > 0 && d->W > 0, "bad descriptor");
  FCVSR_CHECK_ARG(!(d->act == FCVSR_ACT_PRELU) || d->slope_ptr != nullptr, "PReLU needs slope_ptr");
  F32Args a;
  a.src = to_view(d->src[0]);
  a.dst = to_view(d->dst);
  a.res[0] = d->n_res > 0 ? to_view(d->res[0]) : a.dst;
  a.res[1] = d->n_res > 1 ? to_view(d->res[1]) : a.dst;
  a.B = d->B; a.H = d->H; a.W = d->W; a.ks = d->kh;
  a.cin = d->src[0].c; a.cout = d->cout; a.cout_pad = d->cout_pad;
  a.w = (const float*)d->weight; a.bias = d->bias; a.act = d->act; a.slope = d->slope; a.slope_ptr = d->slope_ptr;
  a.n_res = d->n_res; a.rs[0] = d->res_scale[0]; a.rs[1] = d->res_scale[1];
  a.ps = mode == 2; a.cq4 = d->cout / 4; a.generic = mode == 3;
  const int nf = d->cout > 32 ? 2 : 1;
  a.n_nblk = (d->cout + 32 * nf - 1) / (32 * nf);
  long long tiles;
  if (d->kh == 3) {
    a.tiles_x = cdiv(d->W, 32); a.tiles_y = cdiv(d->H, 4);
    tiles = (long long)d->B * a.tiles_x * a.tiles_y;
  } else {
    a.tiles_x = cdiv((long long)d->B * d->H * d->W, 128); a.tiles_y = 1;
    tiles = a.tiles_x;
  }
  FCVSR_CHECK_ARG(tiles * a.n_nblk < (1ll << 31), "grid too large");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(tiles * a.n_nblk));
  const size_t pix = d->kh == 3 ? 6 * 34 : 128;
  const size_t lds = (pix * kFLD + (size_t)32 * nf * kFLD) * sizeof(float);
  if (d->kh == 3) {
    if (nf == 2) hipLaunchKernelGGL((conv_f32mfma_kernel<3, 2>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((conv_f32mfma_kernel<3, 1>), grid, dim3(256), lds, st, a);
  } else {
    if (nf == 2) hipLaunchKernelGGL((conv_f32mfma_kernel<1, 2>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((conv_f32mfma_kernel<1, 1>), grid, dim3(256), lds, st, a);
  }
  FCVSR_LAUNCH_CHECK();
  return 0;
}
