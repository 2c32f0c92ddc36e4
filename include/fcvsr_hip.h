/*
 * fcvsr_hip.h - C ABI of libfcvsr_hip.so: the MI355X (gfx950) hot path of FCVSR's per-frame forward.
 *
 * The reference (QZ1-boy/FCVSR) has no FFI/plugin boundary: its hot path is the Python method
 * GShiftNet_S.forward / GShiftNet.forward (CVSR_train/arch/CVSR_freq.py:2611-2646, :2688-2756) built from
 * stock torch ops.  The drop-in seam is therefore the Python nn.Module (fcvsr_amd/arch/CVSR_freq.py);
 * *below* that seam every arithmetic step is one of the entry points declared here.  Each entry point cites
 * the reference function(s) it replaces.  INTEGRATION.md shows the ctypes binding a reference maintainer adds.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers (HBM) unless named host_*;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue, never synchronise;
 *  - every function returns 0 on success, a negative FCVSR_E_* code on bad arguments, or a positive
 *    hipError_t when a launch fails; fcvsr_last_error() returns a human-readable message (thread-local);
 *  - activations are described by `fcvsr_view`: a strided 4-D (b,y,x,c) window into a buffer, so NCHW boundary
 *    tensors, NHWC internal tensors, channel slices and channel concatenations need no copies.
 *    Internal layout is NHWC (channels innermost) so that a pixel's channels are one contiguous HBM segment.
 */
#ifndef FCVSR_HIP_H
#define FCVSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCVSR_ABI_VERSION 1

enum { FCVSR_E_ARG = -1, FCVSR_E_UNSUPPORTED = -2, FCVSR_E_NOGPU = -3 };
enum { FCVSR_F32 = 0, FCVSR_BF16 = 1, FCVSR_F16 = 2 };
enum { FCVSR_ACT_NONE = 0, FCVSR_ACT_RELU = 1, FCVSR_ACT_LEAKY = 2, FCVSR_ACT_PRELU = 3 };

/* strided (b,y,x,c) window; strides in ELEMENTS of `dtype`; ptr already points at element (0,0,0,0) */
typedef struct fcvsr_view {
  void*   ptr;
  int64_t sb, sy, sx, sc;
  int32_t c;      /* number of channels in this window */
  int32_t dtype;  /* FCVSR_F32 | FCVSR_BF16 | FCVSR_F16 */
} fcvsr_view;

/* One 2-D convolution with fused epilogue.  Replaces nn.Conv2d call sites of the path
 * (CVSR_freq.py:2589 feat_extract, :1371-1396 convfuse/convcorr/convcrt, :344-357 ConvBlk convs, :1409-1416
 * conv_KP/F, :1430 conv3, :2594-2609 rconcat/upconv/conv_last0, :705-803 RCB/BlockRCB/SCGroupbk convs) together with
 * the elementwise ops the reference launches around them (bias, ReLU/LeakyReLU/PReLU, residual adds,
 * torch.cat of inputs, nn.PixelShuffle(2) of the output).
 *   dst = PS?( act(conv(cat(src[0..n_src)), W) + bias) + sum_i res_scale[i]*res[i] )
 */
typedef struct fcvsr_conv_desc {
  int32_t     n_src;          /* 1..3 inputs concatenated along channels (torch.cat(dim=1)) */
  fcvsr_view  src[3];
  int32_t     B, H, W;        /* input spatial size */
  int32_t     kh, kw, stride, pad;
  int32_t     cout;
  const void* weight;         /* packed by fcvsr_pack_conv_weight layout: [kh*kw][cin_total][cout_pad] (f32) */
  int32_t     cout_pad;       /* row length of packed weight (multiple of 16) */
  const float* bias;          /* cout floats or NULL */
  int32_t     act;            /* FCVSR_ACT_* */
  float       slope;          /* LEAKY slope */
  const float* slope_ptr;     /* PRELU: device pointer to the scalar slope (nn.PReLU(), :2590) */
  int32_t     n_res;          /* 0..2 residual inputs at OUTPUT (pre-shuffle) resolution with cout channels */
  fcvsr_view  res[2];
  float       res_scale[2];
  fcvsr_view  dst;            /* output window; if pixel_shuffle: (2*Ho, 2*Wo) spatial, cout/4 channels */
  int32_t     pixel_shuffle;  /* 0/1: out[c,2h+i,2w+j] = conv[4c+2i+j,h,w] (:2633-2642) */
  /* fcvsr_conv2d_mfma only: ContextBlock fusion (:657-701).  When both are non-NULL the epilogue also emits, per workgroup,
   * online-softmax partials of the layer output r: sum_p exp(l_p - m) r_p[c], m = max l_p, sum exp, with l_p = <r_p, gc_wmask>.
   * gc_partial: [B][ceil(H/4)*ceil(W/32)][cout+2] floats (one per 4x32 tile); combine with fcvsr_gc_finish. */
  const float* gc_wmask;
  float*       gc_partial;
} fcvsr_conv_desc;

const char* fcvsr_last_error(void);
int  fcvsr_abi_version(void);
/* number of HIP devices visible (0 when there is no GPU); never throws */
int  fcvsr_device_count(void);

/* direct (VALU, f32) convolution: any kernel size / channel count; used for skinny layers and as the exact-f32 path */
int fcvsr_conv2d(const fcvsr_conv_desc* d, void* stream);

/* Matrix-core (MFMA) implicit-GEMM convolution: 1x1 or 3x3, stride 1, "same" padding, f32 activations in HBM converted
 * to `mma_dtype` (FCVSR_BF16 | FCVSR_F16) while staging through LDS, f32 accumulate and f32 epilogue.
 * `descs[0..n_groups)` (1..3) are problems that share weights/epilogue but have their own tensors and sizes (the three
 * pyramid levels of BlockRCB, CVSR_freq.py:766-777) and run in ONE launch.
 * weight: 16-bit [kh*kw][cout_pad][cin_pad], cout_pad = ceil128(cout), cin_pad = ceil64(cin), zero padded.
 * src views may be f32 or already `mma_dtype` (all alike); dst may be f32 or `mma_dtype` (16-bit storage for tensors whose
 * only consumers are further MFMA convolutions: bit-identical results, half the HBM bytes); res views are f32.
 * With pixel_shuffle the weight/bias rows must be ordered sub-pixel-major: row (2*i+j)*(cout/4)+c holds original output
 * channel 4*c+2*i+j, so that one lane's 4 consecutive couts land in one pixel of the shuffled output. */
int fcvsr_conv2d_mfma(const fcvsr_conv_desc* descs, int n_groups, int mma_dtype, void* stream);
/* Name (template instance) of the kernel the calling thread's last fcvsr_conv2d_mfma call launched, e.g.
 * "conv3_res_kernel<true, true, 1>": measurement aid for bench.py's per-kernel roofline, not part of the data path. */
const char* fcvsr_last_conv_kernel(void);
/* fcvsr_conv2d on the matrix cores with f32 operands (v_mfma_f32_32x32x2_f32: exact f32, bit-equal to an fmaf chain): the
 * arithmetic of the exact-f32 mode for 3x3 / 1x1 stride-1 layers with one dense NHWC f32 source of a multiple of 32 channels,
 * no pixel shuffle.  weight: f32 [kh*kw][cout_pad][cin], cout_pad a multiple of 64 (zero rows past cout); every other field as
 * in fcvsr_conv2d.  fcvsr_conv2d_f32mfma_eligible returns 1 when the descriptor qualifies. */
int fcvsr_conv2d_f32mfma_eligible(const fcvsr_conv_desc* d);
int fcvsr_conv2d_f32mfma(const fcvsr_conv_desc* d, void* stream);
/* Backward of the convolutions (reference: `loss.backward()` through nn.Conv2d, CVSR_train/train_LD_freqCVSR_S_22.py:250).
 * The input gradient of a stride-1 "same" convolution is itself such a convolution of the output gradient with the transposed,
 * tap-flipped weight, so it goes through fcvsr_conv2d / fcvsr_conv2d_mfma; the weight gradient is this entry point:
 *   dw[co][ci][ky][kx] = sum_{b,oy,ox} gy[b,oy,ox,co] * x[b, oy*stride - pad + ky, ox*stride - pad + kx, ci]
 * x: (B,H,W,cin), gy: (B,Ho,Wo,cout), both channel-contiguous f32 views; dw: f32 (cout,cin,kh,kw) = nn.Conv2d.weight layout;
 * scratch: >= fcvsr_conv2d_wgrad_scratch_elems(...) floats.  Exact f32, fixed summation order (bit-reproducible). */
long long fcvsr_conv2d_wgrad_scratch_elems(int B, int Ho, int Wo, int cin, int cout, int kh, int kw);
int fcvsr_conv2d_wgrad(const fcvsr_view* x, const fcvsr_view* gy, int B, int H, int W, int kh, int kw, int stride, int pad,
                       float* dw, float* scratch, long long scratch_elems, void* stream);
/* The same weight gradient with bf16 products on the matrix cores (f32 accumulation, deterministic slab order): 3x3 / 1x1,
 * stride 1, cin and cout multiples of 64 (fcvsr_conv2d_wgrad_mfma_eligible); same arguments and result layout. */
int fcvsr_conv2d_wgrad_mfma_eligible(int cin, int cout, int kh, int kw, int stride, int pad);
long long fcvsr_conv2d_wgrad_mfma_scratch_elems(int B, int Ho, int Wo, int cin, int cout, int kh, int kw);
int fcvsr_conv2d_wgrad_mfma(const fcvsr_view* x, const fcvsr_view* gy, int B, int H, int W, int kh, int kw, int stride, int pad,
                            float* dw, float* scratch, long long scratch_elems, void* stream);
/* Training-path helpers (reference: the per-layer work of `loss.backward()` + `optimizer.step()`, train_LD_freqCVSR_S_22.py:244-251).
 * fcvsr_pack_weight_mfma: nn.Conv2d.weight (cout,cin,kh,kw) f32 -> the 16-bit operand layout of fcvsr_conv2d_mfma,
 *   [kh*kw][rows_pad][cols_pad] zero padded; transposed = 0: rows = cout, cols = cin (forward); transposed = 1: rows = cin,
 *   cols = cout, taps flipped (the input-gradient convolution).  One launch (the weights change every optimizer step).
 * fcvsr_act_bwd: out = g * (y > 0 ? 1 : slope), y = the activation's output (LeakyReLU / ReLU backward), n % 4 == 0, out may alias g.
 * fcvsr_colsum: out[c] = sum over the npix rows of a dense (npix, C) f32 matrix (bias gradient), deterministic two-stage. */
int fcvsr_pack_weight_mfma(const float* w, int cout, int cin, int kh, int kw, void* dst, int rows_pad, int cols_pad, int dtype,
                           int transposed, void* stream);
/* The same packing for n_items weights in ONE launch (a training pass re-packs ~200 weights).  tab: device memory, 9 x int64 per item =
 * {source pointer (f32 contiguous (cout,cin,kh,kw)), destination pointer, cout, cin, kh*kw, rows_pad, cols_pad, transposed, first block};
 * item i owns ceil(kh*kw*rows_pad*cols_pad / fcvsr_pack_weights_multi_block_elems()) consecutive blocks; total_blocks = their sum. */
int fcvsr_pack_weights_multi_block_elems(void);
int fcvsr_pack_weights_mfma_multi(const long long* tab, int n_items, int total_blocks, int dtype, void* stream);
int fcvsr_act_bwd(const float* g, const float* y, float* out, float slope, long long n, void* stream);
long long fcvsr_colsum_scratch_elems(long long npix, int C);
int fcvsr_colsum(const float* x, long long npix, int C, float* out, float* scratch, long long scratch_elems, void* stream);
/* RCB tail with the ContextBlock under training (CVSR_freq.py:657-701 inside RCB.forward :705-725), dense (B, HW, 64) f32:
 *   out = LeakyReLU_slope(r + add(r)) + z,  add = W2 . LeakyReLU_slope(W1 . ctx),  ctx = sum_p softmax_p(wmask . r[p]) r[p].
 * forward: three launches; stats (B x fcvsr_rcbt_stat_elems() floats) is what the backward needs besides r;
 *   scratch >= B * fcvsr_rcbt_nblk(HW) * 66 floats.
 * backward: g = dL/dout -> gr = dL/dr (dL/dz = g), dwmask[64], dw1[64x64], dw2[64x64] (row-major like the 1x1 conv weights);
 *   scratch >= B * nblk * 64 + B * 65 + 4 + 2 * B * 4096 floats.  Two-stage fixed-order reductions (bit-reproducible). */
int fcvsr_rcbt_nblk(int HW);
int fcvsr_rcbt_stat_elems(void);
int fcvsr_rcbt_forward(const float* r, const float* z, const float* wmask, const float* w1, const float* w2, float slope, int B, int HW,
                       int C, float* out, float* stats, float* scratch, long long scratch_elems, void* stream);
int fcvsr_rcbt_backward(const float* r, const float* g, const float* wmask, const float* w1, const float* w2, const float* stats,
                        float slope, int B, int HW, int C, float* gr, float* dwmask, float* dw1, float* dw2, float* scratch,
                        long long scratch_elems, int accumulate /* 1: dwmask / dw1 / dw2 += */, void* stream);

/* One DivEnh band (i >= 1) of MultiFreq_Refinment with its running sums, forward and backward (training path; reference
 * CVSR_freq.py:2104-2133 applied at :2201-2254 with CALayer :1812-1828):  t = f - Sf + 0.2 So, e1 = (0.2 a t + b) f, e2 = (0.2 a So + b) f,
 * So' = So + e1 CA(e1) + e2 CA(e2), Sf' = Sf + f.  Tensors dense (B, HW, C) f32, C in {32, 64}; w1 (C/16, C), w2 (C, C/16).
 * forward scratch >= B * nblk * 2C floats; backward scratch >= B * nblk * 2C + 2BC + 2BC(C/16) floats; stats = B * stat_elems(C) floats. */
int fcvsr_divenh_band_nblk(int HW);
int fcvsr_divenh_band_stat_elems(int C);
int fcvsr_divenh_band_forward(const float* f, const float* sf, const float* so, const float* a, const float* b, const float* w1,
                              const float* w2, int B, int HW, int C, float* sf_out, float* so_out, float* stats, float* scratch,
                              long long scratch_elems, void* stream);
int fcvsr_divenh_band_backward(const float* f, const float* sf, const float* so, const float* a, const float* b, const float* w1,
                               const float* w2, const float* stats, const float* gsf, const float* gso, int B, int HW, int C, float* gf,
                               float* gsf_out, float* gso_out, float* ga, float* gb, float* dw1, float* dw2, float* scratch,
                               long long scratch_elems, int accumulate, void* stream);
/* backward of fcvsr_corr_lookup: g = dL/dcorr on the first x_count columns; gx1 / gx2 dense (B,H,Wf,pix_stride), ZEROED by the caller */
int fcvsr_corr_lookup_bwd(const float* x1f, const float* x2f, int64_t pix_stride, int B, int H, int Wf, int C, int radius, int x_count,
                          const fcvsr_view* g, float* gx1_zeroed, float* gx2_zeroed, void* stream);
long long fcvsr_colsum_groups_scratch_elems(const long long* npix, int n_groups, int C);
int fcvsr_colsum_groups(const float* const* xs, const long long* npix, int n_groups, int C, float* out, float* scratch,
                        long long scratch_elems, void* stream);   /* column sums of 1..3 matrices added together, one ordered second stage */
/* PReLU with one shared slope (nn.PReLU(), CVSR_freq.py:2590 / ConvBlk :349), slope in device memory:
 *   fcvsr_prelu_fwd: y = x > 0 ? x : slope[0] * x;   fcvsr_prelu_bwd: gx and gslope[0] (two-stage sum; scratch >= 2048 floats). */
int fcvsr_prelu_fwd(const float* x, const float* slope, float* y, long long n, void* stream);
int fcvsr_prelu_bwd(const float* g, const float* x, const float* slope, float* gx, float* gslope, float* scratch, long long n, void* stream);
/* Weight gradient of a 3x3 "same" convolution with ONE output channel (conv_last0, :2607): x dense (B,H,W,C) f32, gy dense (B,H,W) f32,
 * dw (1,C,3,3); C in {16, 32, 64}; x is read once; fixed summation order. */
long long fcvsr_wgrad_cout1_scratch_elems(int B, int H, int C);
int fcvsr_wgrad_cout1(const float* x, const float* gy, int B, int H, int W, int C, float* dw, float* scratch, long long scratch_elems, void* stream);
/* Backward of one IAC iteration under training (CVSR_freq.py:1230-1250; forward = fcvsr_warp, fcvsr_sac_v, fcvsr_sac_h, which leave
 * s = flow_warp(prev, off) and v = SAC_v(s) in memory).  All tensors f32 NHWC; gy, yout (the iteration's output), v, s, gfin, gv, prev,
 * gprev dense (B,H,W,C); k1 / gk: views of the iteration's 3*C kernel channels inside the predictor output / its gradient.
 *   fcvsr_iac_bwd_sac : gfin (+)= gy * lrelu'(yout);  gv = transposed horizontal pass;  gk (+)= both passes' kernel gradients
 *   fcvsr_iac_bwd_warp: gs = transposed vertical pass of gv;  gprev += bilinear scatter of gs (float atomics: gprev must be zeroed);
 *                       goff (B,H,W,2) = d/d(off) of the bilinear sample.  C in {32, 64}. */
int fcvsr_iac_bwd_sac(const float* gy, const float* yout, const float* v, const float* s, const fcvsr_view* k1, float slope, int B, int H,
                      int W, int C, float* gfin, int fin_accumulate, float* gv, const fcvsr_view* gk, int k_accumulate, void* stream);
int fcvsr_iac_bwd_warp(const float* gv, const fcvsr_view* k1, const float* prev, const fcvsr_view* off, int B, int H, int W, int C,
                       float* gprev_zeroed, float* goff, void* stream);
/* One-shot request consumed by the next fcvsr_conv2d_wgrad_mfma / fcvsr_conv2d_wgrad_mfma_groups call of the calling thread: also write
 * (accumulate = 0) or add (1) dL/dbias[cout] = sum over pixels of gy (nn.Conv2d bias gradient; f32, fixed summation order) - the kernel
 * has the gy tiles in registers, the separate fcvsr_colsum launches go away.  Returns 1 if the weight-gradient form in use supports it
 * (default form), 0 otherwise (request ignored). */
int fcvsr_wgrad_set_bias_out(float* dbias, int accumulate);

/* fcvsr_conv2d_wgrad_mfma summed over 1..3 problems that share the weight (the pyramid levels of a BlockRCB layer): one launch per
 * problem into consecutive slab ranges of one scratch buffer and ONE ordered reduction (no per-level gradient tensors). */
long long fcvsr_conv2d_wgrad_mfma_groups_scratch_elems(const int* B, const int* H, const int* W, int n_groups, int cin, int cout, int kh,
                                                       int kw);
int fcvsr_conv2d_wgrad_mfma_groups(const fcvsr_view* xs, const fcvsr_view* gys, const int* B, const int* H, const int* W, int n_groups, int kh,
                                   int kw, int pad, float* dw, float* scratch, long long scratch_elems, void* stream);
/* Adjoints of the two resamplings inside fcvsr_xscale (BlockRCB cross-scale sum under training), f32 NHWC:
 *   fcvsr_up2_adjoint:   g (B,2H,2W,C) -> (B,H,W,C), transposed x2 bilinear up-sampling (align_corners = False, clamped);
 *   fcvsr_pool2_adjoint: g (B,H,W,C) -> (B,2H,2W,C), transposed 2x2 mean. */
int fcvsr_up2_adjoint(const float* g, float* out, int B, int H, int W, int C, void* stream);
int fcvsr_pool2_adjoint(const float* g, float* out, int B, int H, int W, int C, void* stream);
/* Per-thread switches: while on, fcvsr_conv2d_wgrad / _mfma / _mfma_groups (dw) and fcvsr_colsum / _groups / fcvsr_wgrad_cout1 (out, dw)
 * ADD their result to the destination instead of overwriting it.  The training step keeps every parameter gradient in one flat,
 * pre-zeroed buffer and lets the reductions add straight into it (no AccumulateGrad addition per parameter and pass). */
void fcvsr_wgrad_set_accumulate(int on);
int fcvsr_wgrad_get_accumulate(void);
void fcvsr_colsum_set_accumulate(int on);
/* Diagnostic (FCVSR_RES_STAMPS=1 in the environment): copies the in-kernel cycle stamps the last resident-weight convolution
 * launch recorded for one workgroup, [wave 8][phase 64][slot 8] uint64, to host memory.  Not part of the data path. */
int fcvsr_debug_res_stamps(void* host_out, size_t bytes);

/* ---- frequency transforms: torch.fft.rfft2 / irfft2 (norm='backward') of NHWC channel groups -------------
 * Spectrum layout: buffer [B][H][Wf][pix_stride] with Wf=W/2+1; channel c of the group has its imaginary part at
 * channel im_off+c and its real part at re_off+c (the reference packs [imag, real], CVSR_freq.py:1456-1465).
 * fcvsr_rfft2 : real src (B,H,W,n; f32 or 16-bit storage) -> f32 spectrum  (replaces :1452-1465 and the fftn of :2082-2084)
 * fcvsr_irfft2: spectrum (optionally multiplied by a real (H,Wf) mask per group of `mask_every` channels...)
 *               -> real dst (B,H,W,n), scaled 1/(H*W)     (replaces :1497-1505 and the ifftn(...).real of :2085-2090)
 *   work: scratch buffer of the same size as the spectrum region used (B*H*Wf*2n floats), or NULL to run the
 *   column pass in place (destroys the spectrum).  mask: (H,Wf) floats or NULL.
 */
int fcvsr_rfft2(const fcvsr_view* src, int B, int H, int W, int n,
                float* spec, int64_t pix_stride, int im_off, int re_off, void* stream);
int fcvsr_irfft2(const float* spec, int64_t pix_stride, int im_off, int re_off, int B, int H, int W, int n,
                 const float* mask, float* work, const fcvsr_view* dst, void* stream);
/* Band split of MultiFreq_Refinment (reference :2082-2090) in one call: dst[m] = irfft2(spec * masks[m]) for m < n_bands.
 * masks: n_bands contiguous (H,Wf) real masks; work: n_bands * B*H*Wf*pix_stride floats; dst: n_bands f32 views (B,H,W,>=n).
 * When H has a two-stage factorisation the spectrum columns are read once for all bands (results identical to n_bands
 * calls of fcvsr_irfft2, which is also the fallback). */
int fcvsr_irfft2_bands(const float* spec, int64_t pix_stride, int im_off, int re_off, int B, int H, int W, int n,
                       const float* masks, int n_bands, float* work, const fcvsr_view* dst, void* stream);

/* feat_extract (:2589, Conv2d(Cin, n_blk*64, 3, 1, 1), Cin = 7: 9*Cin <= 64) as one K = 64 GEMM step per output tile.
 * x: (B,H,W,Cin) f32 view of the planar frames; w: [n_blk*64][64] f16, column k = tap*Cin + c (zero beyond 9*Cin);
 * output block i (64 channels) goes to dst[i] (16-bit, dtype dst_dtype) at channel offset dst_ch_off[i], pixels
 * dst_pix_stride[i] elements apart (flat pixel index (b*H + y)*W + x). */
int fcvsr_feat_extract(const fcvsr_view* x, int B, int H, int W, const void* w, const float* bias, int n_blk,
                       void* const* dst, const int64_t* dst_pix_stride, const int32_t* dst_ch_off, int dst_dtype,
                       void* stream);

/* ---- MGAAbk pieces (CVSR_freq.py:1365-1547) ---------------------------------------------------------------- */
/* CorrBlock lookup on the integer grid (:1279-1337): x1f,x2f NHWC (B,H,Wf,C) with pixel stride
 * pix_stride (floats); dst (B,H,x_count,>=81); channels beyond (2r+1)^2 are zero-filled.  Only the first x_count
 * columns are produced (x_count = Wf for the whole map): the lookup is identically zero for x > radius+1, because the
 * reference samples a (C/2 x 2)-pixel image (column index x+i-r must be 0 or 1), so callers may evaluate a strip only */
int fcvsr_corr_lookup(const float* x1f, const float* x2f, int64_t pix_stride, int B, int H, int Wf, int C, int radius,
                      int x_count, const fcvsr_view* dst, void* stream);
/* per-(b,c) sums over (y,x) of a view, deterministic two-stage: out[b][c] (f32).  scratch: B*nblk*C floats */
int fcvsr_channel_sum(const fcvsr_view* src, int B, int H, int W, float* out, float* scratch, int64_t scratch_elems,
                      void* stream);
/* CALayer gate (:1812-1828): gate[b][c] = sigmoid(W2 relu(W1 (sum[b][:]*inv_hw)));  W1 (cr x c), W2 (c x cr) row-major */
int fcvsr_ca_gate(const float* sum, float inv_hw, const float* w1, const float* w2, int B, int c, int cr,
                  float* gate, void* stream);
/* ConvBlk tail + similarity + complex packing (:344-357 `CA(out)+out`, :1495-1498):
 *   o = (u*gate + u) * sim;  u,sim (Bn,H,Wf,4) with u's batch = dir*B+b and sim's batch = b;
 *   writes re (o[0:2]) to spec channel re_off + 2*g + j and im (o[2:4]) to im_off + 2*g + j, g = dir*A + i */
int fcvsr_convblk_tail(const float* u, const float* gate, const float* sim, int B, int ndir, int H, int Wf,
                       float* spec, int64_t pix_stride, int re_off, int im_off, int g_stride, int g0, void* stream);
/* One whole ConvBlk head (:344-357 applied at :1494-1498) in two launches:
 *   u = conv2(PReLU(conv1(x)))  (k x k, 4 -> 4, no bias; w1 / w2 in the direct packing [k*k][4][16] f32, one PReLU slope),
 *   then what fcvsr_convblk_tail does, with the CALayer gate (4 -> 4 -> 4, ca_w1 / ca_w2 row-major, no bias) evaluated from
 *   per-tile channel sums inside the second launch.  x, u_scratch: (ndir*B, H, Wf, 4) f32 dense; sim: (B, H, Wf, 4);
 *   partial_scratch: 4 * ndir*B * ceil(H/16)*ceil(Wf/16) floats. */
int fcvsr_convblk(const float* x, const float* w1, const float* w2, const float* prelu_slope, int ksize, const float* ca_w1,
                  const float* ca_w2, const float* sim, int B, int ndir, int H, int Wf, float* u_scratch,
                  float* partial_scratch, int64_t partial_elems, float* spec, int64_t pix_stride, int re_off, int im_off,
                  int g_stride, int g0, void* stream);
/* flow_warp (:1188-1227): bilinear, zeros padding, sample at (x+off[0], y+off[1]) */
int fcvsr_warp(const fcvsr_view* src, const fcvsr_view* off, int B, int H, int W, const fcvsr_view* dst, void* stream);
/* SAC (:1253-1276) vertical pass: v = sum_t s[clamp(y+t-1)] * k1[c*3+t] */
int fcvsr_sac_v(const fcvsr_view* s, const fcvsr_view* k1, int B, int H, int W, const fcvsr_view* dst, void* stream);
/* SAC horizontal pass (kernel1 again, :1273) + IAC residual and LeakyReLU(0.1) (:1243-1248) */
int fcvsr_sac_h(const fcvsr_view* v, const fcvsr_view* k1, const fcvsr_view* feat_in, float slope,
                int B, int H, int W, const fcvsr_view* dst, void* stream);

/* One fused IAC iteration (:1230-1250): dst = LeakyReLU_slope(SAC_h(SAC_v(flow_warp(prev, off), k1), k1) + feat_in).
 * k1: the 3*C kernel1 channels of this iteration, f32 or 16-bit; prev / feat_in / dst: f32 or 16-bit (all alike);
 * off f32.  C % 32 == 0. */
int fcvsr_iac_step(const fcvsr_view* prev, const fcvsr_view* off, const fcvsr_view* k1, const fcvsr_view* feat_in,
                   float slope, int B, int H, int W, const fcvsr_view* dst, void* stream);
/* The same iteration for BOTH alignment directions in one launch: prev, off, feat_in and dst point at arrays of 2 views
 * (forward, backward); the two directions share k1 (:1524-1545), which is then read once.  C % 64 == 0. */
int fcvsr_iac_step2(const fcvsr_view* prev, const fcvsr_view* off, const fcvsr_view* k1, const fcvsr_view* feat_in,
                    float slope, int B, int H, int W, const fcvsr_view* dst, void* stream);
/* fcvsr_iac_step2 with the last kernel-predictor layer (F[1], a 1x1 convolution, :1416) folded in: the 3*C adaptive
 * kernel channels of the iteration are computed per tile on the matrix cores from k0 (the 64-channel, 16-bit input of
 * F[1]) and never stored.  wk: this iteration's 192 x 64 weight rows in k0's dtype (cin contiguous, i.e. a row block of
 * the MFMA packing of F[1]); kbias: their 192 f32 biases.  C == 64. */
int fcvsr_iac_step2_fused(const fcvsr_view* prev, const fcvsr_view* off, const fcvsr_view* k0, const void* wk,
                          const float* kbias, const fcvsr_view* feat_in, float slope, int B, int H, int W,
                          const fcvsr_view* dst, void* stream);

/* The whole convfuse stack (:1371-1377, applied at :1472-1474) for up to two alignment directions in one launch:
 *   dst[d] = (xa[d] - xb[d]) + W4 . relu(W2 . relu(W0 . [xa[d] | xb[d]]))     1x1, no bias, 128 channels (n_feats = 64)
 * xa[d], xb[d]: f32 spectra, npix pixels of 128 contiguous channels, src_pix_stride floats apart; dst[d]: bf16, 128 channels,
 * dst_pix_stride halfwords apart; w0 [128][256], w2 / w4 [128][128] bf16 with cin contiguous (the MFMA packing of the three
 * layers).  The two hidden tensors stay on chip. */
int fcvsr_freq_mlp3(const float* const* xa, const float* const* xb, int n_dirs, int64_t src_pix_stride, int64_t npix,
                    const void* w0, const void* w2, const void* w4, void* const* dst, int64_t dst_pix_stride, void* stream);

/* The narrow 1x1 stacks on the spectrum grid as one launch each: out(npix,4) f32 = W_last . relu([W_mid . relu](W_0 . x)),
 * x (npix, 128 channels, x_pix_stride elements apart) f32 or bf16, hidden width 64, bf16 operands / f32 accumulate.
 * w0 [>=64][128], w_mid [>=64][64] or NULL, w_last [>=32][64] (rows 0..3 live) bf16 with cin contiguous (MFMA packing).
 * convcrt (:1392-1396): w_mid = NULL on the centre spectrum; convcorr (:1379-1385) away from the CorrBlock strip: the
 * offset spectra with the first 128 input columns of convcorr.0. */
int fcvsr_freq_head(const void* x, int x_dtype, int64_t x_pix_stride, int64_t npix, const void* w0, const void* w_mid,
                    const void* w_last, float* out, void* stream);

/* ---- MultiFreq_Refinment pieces (CVSR_freq.py:2104-2133, :2201-2254) ---------------------------------------- */
/* DivEnh expressions, i==0 (first=1): t=f-mean_f; e1=0.2*a*t*f+b*f.  i>0: t=f-s_f+0.2*s_o; e1 as above;
 * e2=0.2*a*s_o*f+b*f.   mode 0: write per-(b,c) sums of e1,e2 to sums[2][B][C] (two-stage, deterministic);
 * mode 1: o=e1*g1(+e2*g2); s_f+=f; s_o+=o (in place). All tensors dense NHWC (B,H,W,C). */
int fcvsr_divenh(int mode, int first, const float* f, float* s_f, float* s_o, const float* a, const float* b,
                 const float* mean_f_sum, float inv_hw, const float* g1, const float* g2,
                 float* sums, float* scratch, int64_t scratch_elems, int B, int H, int W, int C, void* stream);
/* mode 1 of fcvsr_divenh for band i fused with the reduction the next step needs (s_f, s_o are touched once):
 *   f_next != NULL: sums[2][B][C] = per-(b,c) sums of e1, e2 of band i+1 (= mode 0 of the next block, bit-identical);
 *   f_next == NULL: sums[0][B][C] = per-(b,c) sums of the updated s_o (input of the final CALayer), sums[1] = 0. */
int fcvsr_divenh_apply_next(int first, const float* f, float* s_f, float* s_o, const float* a, const float* b,
                            const float* mean_f_sum, float inv_hw, const float* g1, const float* g2,
                            const float* f_next, const float* a_next, const float* b_next, float* sums,
                            float* scratch, int64_t scratch_elems, int B, int H, int W, int C, void* stream);
/* out = z*gate[b][c] + x   (final CALayer of MFFR, :2229-2230); x read as x_dtype, out stored as out_dtype */
int fcvsr_scale_add(const float* z, const float* gate, const void* x, int x_dtype, void* out, int out_dtype, int B, int H,
                    int W, int C, void* stream);

/* ---- SCNetbk pieces (CVSR_freq.py:657-822) ------------------------------------------------------------------- */
/* ContextBlock (:657-701): add[b][c] = W2 lrelu0.2(W1 ctx), ctx = sum_p r[p]*softmax_p(r[p].wmask)
 *   scratch: B*nblk*(C+2) floats */
int fcvsr_gc_context(const float* r, const float* wmask, const float* w1, const float* w2, int B, int H, int W, int C,
                     float* add, float* scratch, int64_t scratch_elems, void* stream);
/* second half of fcvsr_gc_context for partials produced by the fused conv epilogue (or any producer of the same layout) */
int fcvsr_gc_finish(const float* partial, int nparts, const float* w1, const float* w2, int B, int C, float* add,
                    void* stream);
/* RCB tail (:722-725): out = lrelu0.2(r + add[b][c]) + z; r is f32, z and out are f32 or 16-bit (io_dtype: trunk16 mode) */
int fcvsr_gc_apply(const float* r, const float* add, const void* z, void* out, int io_dtype, float slope,
                   int B, int H, int W, int C, void* stream);
/* BlockRCB cross-scale sum (:766-777): out = x + r_scale*r + avgpool2(dn) + bilinear_up2(up); dn/up may be NULL.
 * dn is (B,2H,2W,C), up is (B,H/2,W/2,C); all tensors f32 or all 16-bit (io_dtype) */
int fcvsr_xscale(const void* x, const void* r, float r_scale, const void* dn, const void* up, void* out, int io_dtype,
                 int B, int H, int W, int C, void* stream);

/* The three calls above for all pyramid levels of a BlockRCB in ONE launch each (the small levels are launch-latency bound).
 * fcvsr_gc_apply_levels can also emit pool = avgpool2(out) (the bilinear x0.5 of Interpolate, :623-632): the cross-scale
 * 1x1 convolution commutes with that average, so the caller convolves the pooled tensor (a quarter of the pixels) and
 * fcvsr_xscale_levels adds the result as dn at the level's own resolution (dn_pooled = 1). */
typedef struct {
  const float* partial;   /* [B][nparts][C+2] */
  float*       add;       /* [B][C] */
  int32_t      nparts;
} fcvsr_gc_finish_level;
int fcvsr_gc_finish_levels(const fcvsr_gc_finish_level* lv, int n_levels, const float* w1, const float* w2, int B, int C,
                           void* stream);
typedef struct {
  const void*  r;         /* (B,H,W,C), r_dtype: f32 or io_dtype */
  const float* add;       /* [B][C] */
  const void*  z;         /* (B,H,W,C) io_dtype */
  void*        out;       /* (B,H,W,C) io_dtype */
  void*        pool;      /* (B,H/2,W/2,C) io_dtype or NULL (needs even H, W) */
  int32_t      B, H, W;
} fcvsr_gc_apply_level;
int fcvsr_gc_apply_levels(const fcvsr_gc_apply_level* lv, int n_levels, int io_dtype, int r_dtype, float slope, int C,
                          void* stream);
typedef struct {
  const void* x;          /* (B,H,W,C) */
  const void* r;
  const void* dn;         /* NULL, or (B,H,W,C) when dn_pooled, else (B,2H,2W,C) */
  const void* up;         /* NULL or (B,H/2,W/2,C) */
  void*       out;
  float       r_scale;
  int32_t     dn_pooled;
  int32_t     B, H, W;
} fcvsr_xscale_level;
int fcvsr_xscale_levels(const fcvsr_xscale_level* lv, int n_levels, int io_dtype, int C, void* stream);

/* Full-resolution level of BlockRCB's second half in one pass (reference CVSR_freq.py:722-725 + :766-777, level 0; 16-bit
 * storage modes):  R = lrelu(r + add[b], slope) + z  is formed in registers (rounded to the storage type, as the two-kernel
 * sequence fcvsr_gc_apply_levels -> fcvsr_xscale_levels stores it: results are bit-identical to that sequence),
 *   pool = avg_pool2x2(R)            [B, H/2, W/2, C]   (input of down.0, which commutes with the pooling)
 *   out  = x + r_scale * R + bilinear_x2(up)            (up = up.0(R of level 1), [B, H/2, W/2, C], align_corners=False)
 * x, r, z, up, out, pool are NHWC in io_dtype (FCVSR_BF16 / FCVSR_F16), add is f32 [B, C]; C % 8 == 0, H and W even. */
int fcvsr_rcb_level0(const void* x, const void* r, const float* add, const void* z, const void* up, void* out, void* pool,
                     float slope, float r_scale, int io_dtype, int B, int H, int W, int C, void* stream);

/* ---- tail ------------------------------------------------------------------------------------------------------ */
/* nn.PixelShuffle(2) of a dense NHWC tensor (B,H,W,C) -> (B,2H,2W,C/4) (:2634-2635) */
/* ContextBlock softmax-pool partials (:657-701) from a STORED 16-bit r, all pyramid levels in one launch: one [C+2] record per
 * 4 x 32 pixel tile (sum_p exp(l_p - m) r_p[c], m = max l_p, sum exp; l_p = <r_p, wmask>), the layout fcvsr_conv2d_mfma's fused
 * epilogue writes - fcvsr_gc_finish_levels consumes either.  r: dense (B,H,W,64) in r_dtype (BF16 / F16);
 * partial: [B][ceil(H/4)*ceil(W/32)][66] floats. */
typedef struct {
  const void* r;
  float*      partial;
  int32_t     B, H, W;
} fcvsr_gc_partial_level;
int fcvsr_gc_partial_levels(const fcvsr_gc_partial_level* lv, int n_levels, int r_dtype, const float* wmask, int C, void* stream);

int fcvsr_pixel_shuffle(const float* src, float* dst, int B, int H, int W, int C, void* stream);
/* F.interpolate(scale_factor=4, bilinear, align_corners=False) (:2644): src view (B,H,W,c) -> dst view (B,4H,4W,c) */
int fcvsr_bilinear_up4(const fcvsr_view* src, int B, int H, int W, const fcvsr_view* dst, void* stream);

/* Fused end of the S-model up-sampler (:2605-2607, :2642-2645):
 *   out += conv_last0( PReLU( PixelShuffle2( upconv2(u1) ) ) ),   upconv2 1x1 64->256, conv_last0 3x3 64->1.
 * u1 (B,H2,W2,64) 16-bit; w2 [256][64] in u1's dtype, rows sub-pixel-major (row (2i+j)*64+c = original channel 4c+2i+j),
 * b2 its 256 biases in the same order (or NULL); slope: the shared PReLU scalar; wl [16][64] in u1's dtype, row = tap
 * ky*3+kx of conv_last0 (rows 9..15 zero); bl: its bias (or NULL); out (B,2*H2,2*W2,1) f32 is read-modify-written (it
 * holds the bilinear base skip).  The (B,2*H2,2*W2,64) intermediate is never stored. */
int fcvsr_tail_fused(const fcvsr_view* u1, const void* w2, const float* b2, const float* slope, const void* wl,
                     const float* bl, int B, int H2, int W2, const fcvsr_view* out, void* stream);
/* conv_last0 alone (:2607 / :2683; the RGB twins' 3-channel variant): out += bias + conv3x3(u), u (B,H,W,64) dense 16-bit at the
 * OUTPUT resolution, out a (B,H,W,C) f32 view (any strides: the NCHW result), C = 1..3.  w: [16 (C = 1) or 32][64] in u's dtype,
 * row tap*C + c (tap = ky*3 + kx), zero rows past 9C.  Used where fcvsr_tail_fused does not apply (3x3 up-convs). */
int fcvsr_conv_last(const fcvsr_view* u, const void* w, const float* bias, int B, int H, int W, int C, const fcvsr_view* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FCVSR_HIP_H */
