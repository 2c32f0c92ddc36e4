// Batched 2-D real FFT / inverse for NHWC channel groups on gfx950, any length (mixed radix, any prime factor).
//
// Design (MI355X-first): with channels innermost, BOTH passes of the 2-D transform vectorise over channels:
//   * a lane owns one (element, channel) pair; L consecutive lanes are L consecutive channels of the same pixel, so every
//     HBM access of either pass is an L*4-byte contiguous segment and every LDS access is conflict-free along channels;
//   * the whole 1-D transform of L channels lives in LDS (two ping-pong complex buffers, Stockham autosort), one
//     workgroup per (image row | spectrum column, channel chunk); no transposes through HBM;
//   * real rows use the two-for-one trick: channels c and c+L ride as the real and imaginary part of one complex FFT.
// A Stockham stage of radix r computes y[o] = sum_q x[j+q*N/r] * W_N^(q*(k+p*Ns)*N/(Ns*r)) directly (r MACs per output),
// which supports any radix without register arrays; total cost N*sum(r_i) complex MACs per transform.
// Twiddles are generated per workgroup in f64 (sincospi) and rounded once to f32, like pocketfft's table.
//
// Replaces torch.fft.rfft2 / irfft2 (CVSR_freq.py:1452-1454, :1499, :1504) and the per-channel
// fftn/fftshift/mask/ifftshift/ifftn(.real) loop of Split_freq (:2082-2090, via the symmetrised half-spectrum mask).
#include <stdlib.h>
#include <mutex>
#include "common.h"

namespace fcvsr {

struct FftPlan {
  int N;
  int nfac;
  int fac[20];
  unsigned magic[20];   // ceil(2^32 / Ns) of each stage (Ns = product of the previous radices): j / Ns == umulhi(j, magic)
};

static bool make_plan(int N, FftPlan* p) {
  p->N = N;
  p->nfac = 0;
  int n = N;
  // power-of-two radices first (their Ns stay powers of two), then 3, 5, then any other prime (generic butterfly)
  // compound radices 8 (= 4 x 2) and 9 (= 3 x 3) are register butterflies too: one LDS round trip instead of two
  while (n % 8 == 0) { p->fac[p->nfac++] = 8; n /= 8; }
  while (n % 4 == 0) { p->fac[p->nfac++] = 4; n /= 4; }
  while (n % 2 == 0) { p->fac[p->nfac++] = 2; n /= 2; }
  while (n % 9 == 0) { p->fac[p->nfac++] = 9; n /= 9; }
  for (int f = 3; f <= n; f += 2) {
    while (n % f == 0) {
      if (p->nfac >= 20) return false;
      p->fac[p->nfac++] = f;
      n /= f;
    }
  }
  if (N == 1) { p->fac[0] = 1; p->nfac = 1; }
  if (n != 1 || N >= 65536) return false;
  unsigned long long Ns = 1;
  for (int s = 0; s < p->nfac; ++s) {
    p->magic[s] = Ns == 1 ? 0u : (unsigned)(((1ull << 32) + Ns - 1) / Ns);   // exact for j < 2^16
    Ns *= p->fac[s];
  }
  return true;
}

// LDS layout (floats): re0[N*L] im0[N*L] re1[N*L] im1[N*L] tw[2*N]
__device__ __forceinline__ void make_twiddles(float* tw, int N, bool inverse) {
  for (int m = threadIdx.x; m < N; m += blockDim.x) {
    double s, c;
    sincospi(2.0 * (double)m / (double)N, &s, &c);
    tw[2 * m] = (float)c;
    tw[2 * m + 1] = inverse ? (float)s : (float)(-s);
  }
}

struct Cx { float r, i; };
__device__ __forceinline__ Cx cadd(Cx a, Cx b) { return {a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ Cx csub(Cx a, Cx b) { return {a.r - b.r, a.i - b.i}; }
__device__ __forceinline__ Cx cmul(Cx a, Cx w) { return {fmaf(a.r, w.r, -a.i * w.i), fmaf(a.r, w.i, a.i * w.r)}; }
// multiply by -i (forward) or +i (inverse)
template <bool INV> __device__ __forceinline__ Cx rot90(Cx a) { return INV ? Cx{-a.i, a.r} : Cx{a.i, -a.r}; }

template <bool INV>
__device__ __forceinline__ void dft3(Cx a, Cx b, Cx c, Cx& o0, Cx& o1, Cx& o2) {
  const Cx t1 = cadd(b, c);
  const Cx m = {a.r - 0.5f * t1.r, a.i - 0.5f * t1.i};
  const Cx d0 = csub(b, c);
  const Cx sd = rot90<INV>(Cx{0.86602540378443865f * d0.r, 0.86602540378443865f * d0.i});
  o0 = cadd(a, t1); o1 = cadd(m, sd); o2 = csub(m, sd);
}
template <bool INV>
__device__ __forceinline__ void dft4(Cx v0, Cx v1, Cx v2, Cx v3, Cx* o) {
  const Cx a = cadd(v0, v2), b = csub(v0, v2), c = cadd(v1, v3), d = rot90<INV>(csub(v1, v3));
  o[0] = cadd(a, c); o[2] = csub(a, c); o[1] = cadd(b, d); o[3] = csub(b, d);
}

// One Stockham stage with a register butterfly of compile-time radix R: thread = (butterfly j, channel lane l).
// y[(jhi*Ns*R + k + p*Ns)] = sum_q x[j + q*M] * W_N^(q*k*mult) * W_R^(p*q),  k = j % Ns, jhi = j / Ns, M = N/R.
template <int R, bool INV>
__device__ __forceinline__ void stage_radix(const float* xr, const float* xi, float* yr, float* yi, const float* tw, int N,
                                            int Ns, unsigned magic, int L, int logL) {
  const int M = N / R;
  const int mult = M / Ns;                       // N / (Ns*R)
  for (int t = threadIdx.x; t < M * L; t += blockDim.x) {
    const int l = t & (L - 1);
    const int j = t >> logL;
    const int jhi = Ns == 1 ? j : (int)__umulhi((unsigned)j, magic);
    const int k = j - jhi * Ns;
    Cx v[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int idx = (j + q * M) * L + l;
      v[q] = {xr[idx], xi[idx]};
      if (q > 0) {
        const int e = q * k * mult;                // < N
        v[q] = cmul(v[q], Cx{tw[2 * e], tw[2 * e + 1]});
      }
    }
    Cx u[R];
    if (R == 2) {
      u[0] = cadd(v[0], v[1]); u[1] = csub(v[0], v[1]);
    } else if (R == 4) {
      const Cx a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = rot90<INV>(csub(v[1], v[3]));
      u[0] = cadd(a, c); u[2] = csub(a, c); u[1] = cadd(b, d); u[3] = csub(b, d);
    } else if (R == 3) {
      const Cx t1 = cadd(v[1], v[2]);
      const Cx m = {v[0].r - 0.5f * t1.r, v[0].i - 0.5f * t1.i};
      const Cx d0 = csub(v[1], v[2]);
      const Cx sd = rot90<INV>(Cx{0.86602540378443865f * d0.r, 0.86602540378443865f * d0.i});
      u[0] = cadd(v[0], t1); u[1] = cadd(m, sd); u[2] = csub(m, sd);
    } else if (R == 5) {
      const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
      const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
      const Cx t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
      u[0] = {v[0].r + t1.r + t2.r, v[0].i + t1.i + t2.i};
      const Cx m1 = {v[0].r + c1 * t1.r + c2 * t2.r, v[0].i + c1 * t1.i + c2 * t2.i};
      const Cx m2 = {v[0].r + c2 * t1.r + c1 * t2.r, v[0].i + c2 * t1.i + c1 * t2.i};
      const Cx r1 = rot90<INV>(Cx{s1 * t3.r + s2 * t4.r, s1 * t3.i + s2 * t4.i});
      const Cx r2 = rot90<INV>(Cx{s2 * t3.r - s1 * t4.r, s2 * t3.i - s1 * t4.i});
      u[1] = cadd(m1, r1); u[4] = csub(m1, r1); u[2] = cadd(m2, r2); u[3] = csub(m2, r2);
    } else if (R == 8) {
      // radix 8 = DFT4 of the even and of the odd inputs, then W8^k on the odd half (forward W8 = e^{-i pi/4})
      Cx e[4], o[4];
      dft4<INV>(v[0], v[2], v[4], v[6], e);
      dft4<INV>(v[1], v[3], v[5], v[7], o);
      const float h = 0.70710678118654752f;
      const Cx o1 = INV ? Cx{h * (o[1].r - o[1].i), h * (o[1].r + o[1].i)} : Cx{h * (o[1].r + o[1].i), h * (o[1].i - o[1].r)};
      const Cx o2 = rot90<INV>(o[2]);
      const Cx o3 = INV ? Cx{h * (-o[3].r - o[3].i), h * (o[3].r - o[3].i)} : Cx{h * (o[3].i - o[3].r), h * (-o[3].r - o[3].i)};
      u[0] = cadd(e[0], o[0]); u[4] = csub(e[0], o[0]);
      u[1] = cadd(e[1], o1);   u[5] = csub(e[1], o1);
      u[2] = cadd(e[2], o2);   u[6] = csub(e[2], o2);
      u[3] = cadd(e[3], o3);   u[7] = csub(e[3], o3);
    } else if (R == 9) {
      // radix 9 = 3 x 3: A[n2][k1] = DFT3 over n1 of v[3 n1 + n2]; times W9^(n2 k1); X[k1 + 3 k2] = DFT3 over n2
      Cx A[3][3];
#pragma unroll
      for (int n2 = 0; n2 < 3; ++n2) dft3<INV>(v[n2], v[3 + n2], v[6 + n2], A[n2][0], A[n2][1], A[n2][2]);
      const float c1 = 0.76604444311897804f, s1 = 0.64278760968653933f;      // 2 pi / 9
      const float c2 = 0.17364817766693035f, s2 = 0.98480775301220806f;      // 4 pi / 9
      const float c4 = -0.93969262078590838f, s4 = 0.34202014332566873f;     // 8 pi / 9
      const Cx w1 = {c1, INV ? s1 : -s1}, w2 = {c2, INV ? s2 : -s2}, w4 = {c4, INV ? s4 : -s4};
      A[1][1] = cmul(A[1][1], w1); A[1][2] = cmul(A[1][2], w2);
      A[2][1] = cmul(A[2][1], w2); A[2][2] = cmul(A[2][2], w4);
#pragma unroll
      for (int k1 = 0; k1 < 3; ++k1) dft3<INV>(A[0][k1], A[1][k1], A[2][k1], u[k1], u[k1 + 3], u[k1 + 6]);
    }
    const int o0 = (jhi * Ns * R + k) * L + l;
#pragma unroll
    for (int pq = 0; pq < R; ++pq) {
      yr[o0 + pq * Ns * L] = u[pq].r;
      yi[o0 + pq * Ns * L] = u[pq].i;
    }
  }
}

// Any other (prime) radix: one thread per OUTPUT element, r MACs each, inputs re-read from LDS (no register arrays).
__device__ __forceinline__ void stage_generic(const float* xr, const float* xi, float* yr, float* yi, const float* tw, int N,
                                              int Ns, int r, int L, int logL) {
  const int M = N / r;
  const int span = Ns * r;
  const int mult = N / span;
  for (int t = threadIdx.x; t < N * L; t += blockDim.x) {
    const int l = t & (L - 1);
    const int o = t >> logL;
    const int k = o % Ns;
    const int pq = (o / Ns) % r;
    const int jhi = o / span;
    const int j = jhi * Ns + k;
    const int step = k + pq * Ns;
    int e = 0;
    float ar = 0.f, ai = 0.f;
    for (int q = 0; q < r; ++q) {
      const int idx = (j + q * M) * L + l;
      const float vr = xr[idx], vi = xi[idx];
      const float wr = tw[2 * e * mult], wi = tw[2 * e * mult + 1];
      ar = fmaf(vr, wr, ar); ar = fmaf(-vi, wi, ar);
      ai = fmaf(vr, wi, ai); ai = fmaf(vi, wr, ai);
      e += step;
      if (e >= span) e -= span;
    }
    yr[o * L + l] = ar;
    yi[o * L + l] = ai;
  }
}

// runs all stages; returns 0 if the result is in buffer 0, 1 if in buffer 1.  L must be a power of two.
template <bool INV>
__device__ __forceinline__ int run_stages(float* lds, const FftPlan& plan, int L) {
  const int N = plan.N;
  const int NL = N * L;
  const int logL = 31 - __clz(L);
  const float* tw = lds + 4 * NL;
  int cur = 0;
  int Ns = 1;
  for (int s = 0; s < plan.nfac; ++s) {
    const int r = plan.fac[s];
    const float* xr = lds + cur * 2 * NL;
    const float* xi = xr + NL;
    float* yr = lds + (cur ^ 1) * 2 * NL;
    float* yi = yr + NL;
    const unsigned mg = plan.magic[s];
    if (r == 8) stage_radix<8, INV>(xr, xi, yr, yi, tw, N, Ns, mg, L, logL);
    else if (r == 9) stage_radix<9, INV>(xr, xi, yr, yi, tw, N, Ns, mg, L, logL);
    else if (r == 4) stage_radix<4, INV>(xr, xi, yr, yi, tw, N, Ns, mg, L, logL);
    else if (r == 2) stage_radix<2, INV>(xr, xi, yr, yi, tw, N, Ns, mg, L, logL);
    else if (r == 3) stage_radix<3, INV>(xr, xi, yr, yi, tw, N, Ns, mg, L, logL);
    else if (r == 5) stage_radix<5, INV>(xr, xi, yr, yi, tw, N, Ns, mg, L, logL);
    else if (r > 1) stage_generic(xr, xi, yr, yi, tw, N, Ns, r, L, logL);
    if (r > 1) {
      __syncthreads();
      cur ^= 1;
    }
    Ns *= r;
  }
  return cur;
}

// Workgroup id -> (line, chunk) such that the channel chunks of one image line (which read / write the same 128-byte memory
// lines) run on ONE XCD: ids go round-robin to the 8 XCDs, so inside a group of 8 lines x nchunk workgroups id % 8 selects
// the line and id / 8 the chunk.  The host rounds the line count (gridDim.y) up to a multiple of 8.
__device__ __forceinline__ void xcd_line_chunk(int* line, int* chunk) {
  const int nchunk = gridDim.x;
  const int lin = blockIdx.y * nchunk + blockIdx.x;
  const int grp = lin / (8 * nchunk), rem = lin - grp * (8 * nchunk);
  *line = grp * 8 + (rem & 7);
  *chunk = rem >> 3;
}

// ---- forward rows: real (B,H,W,n) -> half spectrum, two channels per complex lane -------------------------------
__device__ __forceinline__ float ld_act(const float* base, long long idx, int dt) {
  if (dt == FCVSR_F32) return base[idx];
  const unsigned short u = reinterpret_cast<const unsigned short*>(base)[idx];
  if (dt == FCVSR_BF16) return __uint_as_float((unsigned)u << 16);
  return (float)__builtin_bit_cast(_Float16, u);
}

__device__ __forceinline__ float4 ld_act4(const float* base, long long idx, int dt) {
  if (dt == FCVSR_F32) return *reinterpret_cast<const float4*>(base + idx);
  const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
  if (dt == FCVSR_BF16)
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 a = __builtin_bit_cast(h2, v.x), b = __builtin_bit_cast(h2, v.y);
  return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
}

// `vec`: every channel of the workgroup's chunk exists and all offsets are 4-channel aligned -> 16-byte accesses
__global__ __launch_bounds__(512) void rfft_rows_kernel(View src, int src_dt, int n, int H, int W, int L, float* spec,
                                                        long long ps, int im_off, int re_off, FftPlan plan, int vec, int nbatch) {
  extern __shared__ __align__(16) float lds[];
  const int NL = W * L;
  int row, chunk;
  xcd_line_chunk(&row, &chunk);
  const int b = row / H, y = row % H;
  if (b >= nbatch) return;                              // padding lines of the rounded-up grid
  const int c0 = chunk * 2 * L;
  const int Wf = W / 2 + 1;
  make_twiddles(lds + 4 * NL, W, false);
  const long long sp = (long long)b * src.sb + (long long)y * src.sy;
  const int logL = 31 - __clz(L);
  if (vec) {
    const int G4 = L >> 1;                         // 4-channel groups per pixel (2L channels)
    const int logG = logL - 1;
    for (int t = threadIdx.x; t < W * G4; t += blockDim.x) {
      const int g = t & (G4 - 1), x = t >> logG;
      const int j0 = g * 4;                        // channel offset inside the chunk: [0,L) -> real lane, [L,2L) -> imag lane
      const float4 v = ld_act4(src.p, sp + (long long)x * src.sx + c0 + j0, src_dt);
      float* d = (j0 < L) ? (lds + x * L + j0) : (lds + NL + x * L + (j0 - L));
      *reinterpret_cast<float4*>(d) = v;
    }
  } else {
    for (int t = threadIdx.x; t < NL; t += blockDim.x) {
      const int l = t & (L - 1), x = t >> logL;
      const int ca = c0 + l, cb = c0 + L + l;
      const long long px = sp + (long long)x * src.sx;
      lds[t] = ca < n ? ld_act(src.p, px + (long long)ca * src.sc, src_dt) : 0.f;
      lds[NL + t] = cb < n ? ld_act(src.p, px + (long long)cb * src.sc, src_dt) : 0.f;
    }
  }
  __syncthreads();
  const int cur = run_stages<false>(lds, plan, L);
  const float* zr = lds + (cur ? 2 * NL : 0);
  const float* zi = zr + NL;
  float* op = spec + ((long long)(b * H + y) * Wf) * ps;
  if (vec) {
    const int Q4 = L >> 2, logQ = logL - 2;
    for (int t = threadIdx.x; t < Wf * Q4; t += blockDim.x) {
      const int q = t & (Q4 - 1), k = t >> logQ;
      const int kn = (W - k) % W;
      const float4 kr = *reinterpret_cast<const float4*>(zr + k * L + q * 4), ki = *reinterpret_cast<const float4*>(zi + k * L + q * 4);
      const float4 nr = *reinterpret_cast<const float4*>(zr + kn * L + q * 4), ni = *reinterpret_cast<const float4*>(zi + kn * L + q * 4);
      float* o = op + (long long)k * ps + c0 + q * 4;
      *reinterpret_cast<float4*>(o + re_off) = make_float4(0.5f * (kr.x + nr.x), 0.5f * (kr.y + nr.y), 0.5f * (kr.z + nr.z), 0.5f * (kr.w + nr.w));
      *reinterpret_cast<float4*>(o + im_off) = make_float4(0.5f * (ki.x - ni.x), 0.5f * (ki.y - ni.y), 0.5f * (ki.z - ni.z), 0.5f * (ki.w - ni.w));
      *reinterpret_cast<float4*>(o + re_off + L) = make_float4(0.5f * (ki.x + ni.x), 0.5f * (ki.y + ni.y), 0.5f * (ki.z + ni.z), 0.5f * (ki.w + ni.w));
      *reinterpret_cast<float4*>(o + im_off + L) = make_float4(0.5f * (nr.x - kr.x), 0.5f * (nr.y - kr.y), 0.5f * (nr.z - kr.z), 0.5f * (nr.w - kr.w));
    }
    return;
  }
  for (int t = threadIdx.x; t < Wf * L; t += blockDim.x) {
    const int l = t & (L - 1), k = t >> (31 - __clz(L));
    const int kn = (W - k) % W;
    const float kr = zr[k * L + l], ki = zi[k * L + l];
    const float nr = zr[kn * L + l], ni = zi[kn * L + l];
    const int ca = c0 + l, cb = c0 + L + l;
    float* o = op + (long long)k * ps;
    if (ca < n) { o[re_off + ca] = 0.5f * (kr + nr); o[im_off + ca] = 0.5f * (ki - ni); }
    if (cb < n) { o[re_off + cb] = 0.5f * (ki + ni); o[im_off + cb] = 0.5f * (nr - kr); }
  }
}

// ---- columns: complex length-H transform of spectrum columns, forward or inverse, optional real mask -----------
__global__ __launch_bounds__(512) void fft_cols_kernel(const float* in, float* out, long long ps, int im_off, int re_off,
                                                       int n, int H, int Wf, int L, int inverse, const float* mask,
                                                       FftPlan plan, int vec, int nbatch) {
  extern __shared__ __align__(16) float lds[];
  const int NL = H * L;
  int col, chunk;
  xcd_line_chunk(&col, &chunk);
  const int b = col / Wf, kx = col % Wf;
  if (b >= nbatch) return;
  const int c0 = chunk * L;
  make_twiddles(lds + 4 * NL, H, inverse != 0);
  const int logL = 31 - __clz(L);
  if (vec) {
    const int Q4 = L >> 2, logQ = logL - 2;
    for (int t = threadIdx.x; t < H * Q4; t += blockDim.x) {
      const int q = t & (Q4 - 1), y = t >> logQ;
      const float* px = in + ((long long)(b * H + y) * Wf + kx) * ps + c0 + q * 4;
      const float m = mask ? mask[y * Wf + kx] : 1.f;
      const float4 vr = *reinterpret_cast<const float4*>(px + re_off), vi = *reinterpret_cast<const float4*>(px + im_off);
      *reinterpret_cast<float4*>(lds + y * L + q * 4) = make_float4(vr.x * m, vr.y * m, vr.z * m, vr.w * m);
      *reinterpret_cast<float4*>(lds + NL + y * L + q * 4) = make_float4(vi.x * m, vi.y * m, vi.z * m, vi.w * m);
    }
  } else {
    for (int t = threadIdx.x; t < NL; t += blockDim.x) {
      const int l = t & (L - 1), y = t >> logL;
      const int c = c0 + l;
      const float* px = in + ((long long)(b * H + y) * Wf + kx) * ps;
      const float m = mask ? mask[y * Wf + kx] : 1.f;
      lds[t] = c < n ? px[re_off + c] * m : 0.f;
      lds[NL + t] = c < n ? px[im_off + c] * m : 0.f;
    }
  }
  __syncthreads();
  const int cur = inverse ? run_stages<true>(lds, plan, L) : run_stages<false>(lds, plan, L);
  const float* zr = lds + (cur ? 2 * NL : 0);
  const float* zi = zr + NL;
  if (vec) {
    const int Q4 = L >> 2, logQ = logL - 2;
    for (int t = threadIdx.x; t < H * Q4; t += blockDim.x) {
      const int q = t & (Q4 - 1), y = t >> logQ;
      float* px = out + ((long long)(b * H + y) * Wf + kx) * ps + c0 + q * 4;
      *reinterpret_cast<float4*>(px + re_off) = *reinterpret_cast<const float4*>(zr + y * L + q * 4);
      *reinterpret_cast<float4*>(px + im_off) = *reinterpret_cast<const float4*>(zi + y * L + q * 4);
    }
    return;
  }
  for (int t = threadIdx.x; t < NL; t += blockDim.x) {
    const int l = t & (L - 1), y = t >> (31 - __clz(L));
    const int c = c0 + l;
    if (c < n) {
      float* px = out + ((long long)(b * H + y) * Wf + kx) * ps;
      px[re_off + c] = zr[t];
      px[im_off + c] = zi[t];
    }
  }
}

// ---- inverse rows: half spectrum -> real (c2r semantics: imag of DC / Nyquist ignored), two channels per lane ----
__global__ __launch_bounds__(512) void irfft_rows_kernel(const float* spec, long long ps, int im_off, int re_off, int n,
                                                         int H, int W, int L, View dst, float scale, FftPlan plan, int vec,
                                                         int nbatch) {
  extern __shared__ __align__(16) float lds[];
  const int NL = W * L;
  int row, chunk;
  xcd_line_chunk(&row, &chunk);
  const int b = row / H, y = row % H;
  if (b >= nbatch) return;
  const int c0 = chunk * 2 * L;
  const int Wf = W / 2 + 1;
  make_twiddles(lds + 4 * NL, W, true);
  const float* ip = spec + ((long long)(b * H + y) * Wf) * ps;
  const int logL = 31 - __clz(L);
  if (vec) {
    const int Q4 = L >> 2, logQ = logL - 2;
    for (int t = threadIdx.x; t < W * Q4; t += blockDim.x) {
      const int q = t & (Q4 - 1), k = t >> logQ;
      const int kk = (k <= W / 2) ? k : W - k;
      const float sgn = (k > W / 2) ? -1.f : 1.f;
      const float keep = ((kk == 0) || ((W % 2 == 0) && kk == W / 2)) ? 0.f : sgn;   // imag parts: dropped at DC/Nyquist
      const float* px = ip + (long long)kk * ps + c0 + q * 4;
      const float4 ar = *reinterpret_cast<const float4*>(px + re_off), ai = *reinterpret_cast<const float4*>(px + im_off);
      const float4 br = *reinterpret_cast<const float4*>(px + re_off + L), bi = *reinterpret_cast<const float4*>(px + im_off + L);
      *reinterpret_cast<float4*>(lds + k * L + q * 4) =
          make_float4(ar.x - keep * bi.x, ar.y - keep * bi.y, ar.z - keep * bi.z, ar.w - keep * bi.w);
      *reinterpret_cast<float4*>(lds + NL + k * L + q * 4) =
          make_float4(keep * ai.x + br.x, keep * ai.y + br.y, keep * ai.z + br.z, keep * ai.w + br.w);
    }
  } else
  for (int t = threadIdx.x; t < NL; t += blockDim.x) {
    const int l = t & (L - 1), k = t >> logL;
    const int kk = (k <= W / 2) ? k : W - k;
    const bool cj = k > W / 2;
    const bool real_only = (kk == 0) || ((W % 2 == 0) && kk == W / 2);
    const int ca = c0 + l, cb = c0 + L + l;
    const float* px = ip + (long long)kk * ps;
    float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f;
    if (ca < n) { ar = px[re_off + ca]; ai = px[im_off + ca]; }
    if (cb < n) { br = px[re_off + cb]; bi = px[im_off + cb]; }
    if (real_only) { ai = 0.f; bi = 0.f; }
    if (cj) { ai = -ai; bi = -bi; }
    lds[t] = ar - bi;        // Z = Xa + i*Xb
    lds[NL + t] = ai + br;
  }
  __syncthreads();
  const int cur = run_stages<true>(lds, plan, L);
  const float* zr = lds + (cur ? 2 * NL : 0);
  const float* zi = zr + NL;
  float* op = dst.p + (long long)b * dst.sb + (long long)y * dst.sy;
  if (vec) {
    const int Q4 = L >> 2, logQ = logL - 2;
    for (int t = threadIdx.x; t < W * Q4; t += blockDim.x) {
      const int q = t & (Q4 - 1), x = t >> logQ;
      float* px = op + (long long)x * dst.sx + c0 + q * 4;
      const float4 a4 = *reinterpret_cast<const float4*>(zr + x * L + q * 4), b4 = *reinterpret_cast<const float4*>(zi + x * L + q * 4);
      *reinterpret_cast<float4*>(px) = make_float4(a4.x * scale, a4.y * scale, a4.z * scale, a4.w * scale);
      *reinterpret_cast<float4*>(px + L) = make_float4(b4.x * scale, b4.y * scale, b4.z * scale, b4.w * scale);
    }
    return;
  }
  for (int t = threadIdx.x; t < NL; t += blockDim.x) {
    const int l = t & (L - 1), x = t >> logL;
    const int ca = c0 + l, cb = c0 + L + l;
    float* px = op + (long long)x * dst.sx;
    if (ca < n) px[(long long)ca * dst.sc] = zr[t] * scale;
    if (cb < n) px[(long long)cb * dst.sc] = zi[t] * scale;
  }
}

// =====================================================================================================================
// Two-stage path.  N = R1 * R2 with both factors done as register butterflies (8 ... 20 points each, themselves Ra x Rb
// Cooley-Tukey products of the 2/3/4/5-point kernels with compile-time twiddles): the FIRST stage reads its inputs straight
// from HBM, the LAST stage writes its outputs straight to HBM, and the transform crosses LDS exactly once, through ONE
// complex buffer and one barrier (the multi-stage Stockham path above copies in, ping-pongs between two buffers once per
// stage and copies out: five LDS round trips and barriers for 180 = 4 x 5 x 9, four for 320 = 8 x 8 x 5).  Half the LDS per
// line (three 480-thread workgroups per CU for the 180-point columns), a third of the LDS instructions, no per-workgroup f64
// twiddle generation (a per-length table is built once on the host).  Lengths without such a factorisation, odd channel
// counts and strided channels stay on the path above.
template <int R> struct RootTab;
template <> struct RootTab<8> {
  static constexpr float c[8] = {1.000000000e+00f, 7.071067812e-01f, 6.123233996e-17f, -7.071067812e-01f, -1.000000000e+00f, -7.071067812e-01f, -1.836970199e-16f, 7.071067812e-01f};
  static constexpr float s[8] = {0.000000000e+00f, 7.071067812e-01f, 1.000000000e+00f, 7.071067812e-01f, 1.224646799e-16f, -7.071067812e-01f, -1.000000000e+00f, -7.071067812e-01f};
};
template <> struct RootTab<9> {
  static constexpr float c[9] = {1.000000000e+00f, 7.660444431e-01f, 1.736481777e-01f, -5.000000000e-01f, -9.396926208e-01f, -9.396926208e-01f, -5.000000000e-01f, 1.736481777e-01f, 7.660444431e-01f};
  static constexpr float s[9] = {0.000000000e+00f, 6.427876097e-01f, 9.848077530e-01f, 8.660254038e-01f, 3.420201433e-01f, -3.420201433e-01f, -8.660254038e-01f, -9.848077530e-01f, -6.427876097e-01f};
};
template <> struct RootTab<10> {
  static constexpr float c[10] = {1.000000000e+00f, 8.090169944e-01f, 3.090169944e-01f, -3.090169944e-01f, -8.090169944e-01f, -1.000000000e+00f, -8.090169944e-01f, -3.090169944e-01f, 3.090169944e-01f, 8.090169944e-01f};
  static constexpr float s[10] = {0.000000000e+00f, 5.877852523e-01f, 9.510565163e-01f, 9.510565163e-01f, 5.877852523e-01f, 1.224646799e-16f, -5.877852523e-01f, -9.510565163e-01f, -9.510565163e-01f, -5.877852523e-01f};
};
template <> struct RootTab<12> {
  static constexpr float c[12] = {1.000000000e+00f, 8.660254038e-01f, 5.000000000e-01f, 6.123233996e-17f, -5.000000000e-01f, -8.660254038e-01f, -1.000000000e+00f, -8.660254038e-01f, -5.000000000e-01f, -1.836970199e-16f, 5.000000000e-01f, 8.660254038e-01f};
  static constexpr float s[12] = {0.000000000e+00f, 5.000000000e-01f, 8.660254038e-01f, 1.000000000e+00f, 8.660254038e-01f, 5.000000000e-01f, 1.224646799e-16f, -5.000000000e-01f, -8.660254038e-01f, -1.000000000e+00f, -8.660254038e-01f, -5.000000000e-01f};
};
template <> struct RootTab<15> {
  static constexpr float c[15] = {1.000000000e+00f, 9.135454576e-01f, 6.691306064e-01f, 3.090169944e-01f, -1.045284633e-01f, -5.000000000e-01f, -8.090169944e-01f, -9.781476007e-01f, -9.781476007e-01f, -8.090169944e-01f, -5.000000000e-01f, -1.045284633e-01f, 3.090169944e-01f, 6.691306064e-01f, 9.135454576e-01f};
  static constexpr float s[15] = {0.000000000e+00f, 4.067366431e-01f, 7.431448255e-01f, 9.510565163e-01f, 9.945218954e-01f, 8.660254038e-01f, 5.877852523e-01f, 2.079116908e-01f, -2.079116908e-01f, -5.877852523e-01f, -8.660254038e-01f, -9.945218954e-01f, -9.510565163e-01f, -7.431448255e-01f, -4.067366431e-01f};
};
template <> struct RootTab<16> {
  static constexpr float c[16] = {1.000000000e+00f, 9.238795325e-01f, 7.071067812e-01f, 3.826834324e-01f, 6.123233996e-17f, -3.826834324e-01f, -7.071067812e-01f, -9.238795325e-01f, -1.000000000e+00f, -9.238795325e-01f, -7.071067812e-01f, -3.826834324e-01f, -1.836970199e-16f, 3.826834324e-01f, 7.071067812e-01f, 9.238795325e-01f};
  static constexpr float s[16] = {0.000000000e+00f, 3.826834324e-01f, 7.071067812e-01f, 9.238795325e-01f, 1.000000000e+00f, 9.238795325e-01f, 7.071067812e-01f, 3.826834324e-01f, 1.224646799e-16f, -3.826834324e-01f, -7.071067812e-01f, -9.238795325e-01f, -1.000000000e+00f, -9.238795325e-01f, -7.071067812e-01f, -3.826834324e-01f};
};
template <> struct RootTab<20> {
  static constexpr float c[20] = {1.000000000e+00f, 9.510565163e-01f, 8.090169944e-01f, 5.877852523e-01f, 3.090169944e-01f, 6.123233996e-17f, -3.090169944e-01f, -5.877852523e-01f, -8.090169944e-01f, -9.510565163e-01f, -1.000000000e+00f, -9.510565163e-01f, -8.090169944e-01f, -5.877852523e-01f, -3.090169944e-01f, -1.836970199e-16f, 3.090169944e-01f, 5.877852523e-01f, 8.090169944e-01f, 9.510565163e-01f};
  static constexpr float s[20] = {0.000000000e+00f, 3.090169944e-01f, 5.877852523e-01f, 8.090169944e-01f, 9.510565163e-01f, 1.000000000e+00f, 9.510565163e-01f, 8.090169944e-01f, 5.877852523e-01f, 3.090169944e-01f, 1.224646799e-16f, -3.090169944e-01f, -5.877852523e-01f, -8.090169944e-01f, -9.510565163e-01f, -1.000000000e+00f, -9.510565163e-01f, -8.090169944e-01f, -5.877852523e-01f, -3.090169944e-01f};
};

template <int R, bool INV>
__device__ __forceinline__ Cx mul_root(Cx a, int m) {           // a * W_R^m (forward W = e^{-2 pi i / R}); m is a constant after unrolling
  m %= R;
  if (m == 0) return a;
  if (4 * m == R) return rot90<INV>(a);
  if (2 * m == R) return Cx{-a.r, -a.i};
  if (4 * m == 3 * R) return rot90<!INV>(a);
  return cmul(a, Cx{RootTab<R>::c[m], INV ? RootTab<R>::s[m] : -RootTab<R>::s[m]});
}

template <int R, bool INV> struct Dft;                            // in place, natural order in and out
template <bool INV> struct Dft<2, INV> {
  static __device__ __forceinline__ void run(Cx* v) { const Cx a = v[0], b = v[1]; v[0] = cadd(a, b); v[1] = csub(a, b); }
};
template <bool INV> struct Dft<3, INV> {
  static __device__ __forceinline__ void run(Cx* v) { dft3<INV>(v[0], v[1], v[2], v[0], v[1], v[2]); }
};
template <bool INV> struct Dft<4, INV> {
  static __device__ __forceinline__ void run(Cx* v) { Cx o[4]; dft4<INV>(v[0], v[1], v[2], v[3], o); v[0] = o[0]; v[1] = o[1]; v[2] = o[2]; v[3] = o[3]; }
};
template <bool INV> struct Dft<5, INV> {
  static __device__ __forceinline__ void run(Cx* v) {
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const Cx t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const Cx u0 = {v[0].r + t1.r + t2.r, v[0].i + t1.i + t2.i};
    const Cx m1 = {v[0].r + c1 * t1.r + c2 * t2.r, v[0].i + c1 * t1.i + c2 * t2.i};
    const Cx m2 = {v[0].r + c2 * t1.r + c1 * t2.r, v[0].i + c2 * t1.i + c1 * t2.i};
    const Cx r1 = rot90<INV>(Cx{s1 * t3.r + s2 * t4.r, s1 * t3.i + s2 * t4.i});
    const Cx r2 = rot90<INV>(Cx{s2 * t3.r - s1 * t4.r, s2 * t3.i - s1 * t4.i});
    v[0] = u0; v[1] = cadd(m1, r1); v[4] = csub(m1, r1); v[2] = cadd(m2, r2); v[3] = csub(m2, r2);
  }
};
// R = Ra * Rb:  A[n2][k1] = DFT_Ra over n1 of v[Rb n1 + n2];  A[n2][k1] *= W_R^(n2 k1);  X[k1 + Ra k2] = DFT_Rb over n2
template <int Ra, int Rb, bool INV>
__device__ __forceinline__ void dft_ct(Cx* v) {
  constexpr int R = Ra * Rb;
  Cx A[Rb][Ra];
#pragma unroll
  for (int n2 = 0; n2 < Rb; ++n2) {
#pragma unroll
    for (int n1 = 0; n1 < Ra; ++n1) A[n2][n1] = v[Rb * n1 + n2];
    Dft<Ra, INV>::run(A[n2]);
#pragma unroll
    for (int k1 = 1; k1 < Ra; ++k1) A[n2][k1] = mul_root<R, INV>(A[n2][k1], n2 * k1);
  }
#pragma unroll
  for (int k1 = 0; k1 < Ra; ++k1) {
    Cx t[Rb];
#pragma unroll
    for (int n2 = 0; n2 < Rb; ++n2) t[n2] = A[n2][k1];
    Dft<Rb, INV>::run(t);
#pragma unroll
    for (int k2 = 0; k2 < Rb; ++k2) v[k1 + Ra * k2] = t[k2];
  }
}
template <bool INV> struct Dft<8, INV> { static __device__ __forceinline__ void run(Cx* v) { dft_ct<2, 4, INV>(v); } };
template <bool INV> struct Dft<9, INV> { static __device__ __forceinline__ void run(Cx* v) { dft_ct<3, 3, INV>(v); } };
template <bool INV> struct Dft<10, INV> { static __device__ __forceinline__ void run(Cx* v) { dft_ct<2, 5, INV>(v); } };
template <bool INV> struct Dft<12, INV> { static __device__ __forceinline__ void run(Cx* v) { dft_ct<3, 4, INV>(v); } };
template <bool INV> struct Dft<15, INV> { static __device__ __forceinline__ void run(Cx* v) { dft_ct<3, 5, INV>(v); } };
template <bool INV> struct Dft<16, INV> { static __device__ __forceinline__ void run(Cx* v) { dft_ct<4, 4, INV>(v); } };
template <bool INV> struct Dft<20, INV> { static __device__ __forceinline__ void run(Cx* v) { dft_ct<4, 5, INV>(v); } };

// second stage of a lane: inputs y[j + q R1] from LDS times W_N^(q j), DFT_R2 in registers
template <int R1, int R2, bool INV>
__device__ __forceinline__ void stage2_load(const float* zr, const float* zi, const float2* tw, int j, int l, int L, Cx* v) {
#pragma unroll
  for (int q = 0; q < R2; ++q) {
    const int idx = (j + q * R1) * L + l;
    v[q] = Cx{zr[idx], zi[idx]};
    if (q > 0) {
      const float2 w = tw[q * j];                      // forward table (cos, -sin); same address across the channel lanes
      v[q] = cmul(v[q], Cx{w.x, INV ? -w.y : w.y});
    }
  }
  Dft<R2, INV>::run(v);
}

__device__ __forceinline__ void load_twiddles(float2* tw, const float2* tab, int N) {
  for (int m = threadIdx.x; m < N; m += blockDim.x) tw[m] = tab[m];
}

// ---- columns, two stages: lane = (butterfly j, channel l) -------------------------------------------------------------------
// MASK is a template parameter and the waves-per-SIMD range is pinned: with a run-time mask test every (mask, re, im) load
// triple sat in its own block, and hipcc, free to chase a higher occupancy than the LDS footprint allows anyway, kept the
// loads of the row kernels in groups of four with a full wait between groups.
template <int R1, int R2, bool INV, bool MASK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 6)))
void fft_cols2_kernel(const float* in, float* out, long long ps, int im_off, int re_off, int n,
                                                        int Wf, int logL, const float* mask, const float2* twtab, int nbatch) {
  constexpr int N = R1 * R2;
  extern __shared__ __align__(16) float lds[];
  const int L = 1 << logL;
  float* zr = lds;
  float* zi = lds + N * L;
  float2* tw = reinterpret_cast<float2*>(lds + 2 * N * L);
  int col, chunk;
  xcd_line_chunk(&col, &chunk);
  const int b = col / Wf, kx = col - b * Wf;
  if (b >= nbatch) return;
  load_twiddles(tw, twtab, N);
  const int l = threadIdx.x & (L - 1), j = threadIdx.x >> logL;
  const int c = chunk * L + l;
  const bool live = c < n;
  const long long col0 = ((long long)b * N * Wf + kx) * ps + (live ? c : 0);   // dead lanes load channel 0 and discard it
  const long long rstride = (long long)Wf * ps;
  if (j < R2) {
    Cx v[R1];
    float mk[R1];
#pragma unroll
    for (int q = 0; q < R1; ++q) {
      const int y = j + q * R2;
      const float* px = in + col0 + (long long)y * rstride;
      mk[q] = MASK ? mask[y * Wf + kx] : 1.f;
      v[q] = Cx{px[re_off], px[im_off]};
    }
#pragma unroll
    for (int q = 0; q < R1; ++q) v[q] = live ? (MASK ? Cx{v[q].r * mk[q], v[q].i * mk[q]} : v[q]) : Cx{0.f, 0.f};
    Dft<R1, INV>::run(v);
#pragma unroll
    for (int p = 0; p < R1; ++p) {
      const int idx = (j * R1 + p) * L + l;
      zr[idx] = v[p].r;
      zi[idx] = v[p].i;
    }
  }
  __syncthreads();
  if (j < R1) {
    Cx v[R2];
    stage2_load<R1, R2, INV>(zr, zi, tw, j, l, L, v);
    if (live) {
#pragma unroll
      for (int p = 0; p < R2; ++p) {
        float* px = out + col0 + (long long)(j + p * R1) * rstride;
        px[re_off] = v[p].r;
        px[im_off] = v[p].i;
      }
    }
  }
}

// ---- inverse columns for several real masks at once (band split of MultiFreq_Refinment, reference :2082-2090): the spectrum
// column is read ONCE, then masked, transformed and written once per band (NM bands: (1 + NM) x 237 MB instead of 2 NM x 237 MB
// at 16 x 180 x 320 x 64).  Same arithmetic per band as fft_cols2_kernel<R1, R2, true, true>: bit-identical results.
template <int R1, int R2>
__global__ __launch_bounds__(512) void fft_cols2_bands_kernel(const float* in, float* out, long long out_band_stride, long long ps,
                                                              int im_off, int re_off, int n, int Wf, int logL, const float* masks,
                                                              int nm, const float2* twtab, int nbatch) {
  constexpr int N = R1 * R2;
  extern __shared__ __align__(16) float lds[];
  const int L = 1 << logL;
  float* zr = lds;
  float* zi = lds + N * L;
  float2* tw = reinterpret_cast<float2*>(lds + 2 * N * L);
  int col, chunk;
  xcd_line_chunk(&col, &chunk);
  const int b = col / Wf, kx = col - b * Wf;
  if (b >= nbatch) return;
  load_twiddles(tw, twtab, N);
  const int l = threadIdx.x & (L - 1), j = threadIdx.x >> logL;
  const int c = chunk * L + l;
  const bool live = c < n;
  const long long col0 = ((long long)b * N * Wf + kx) * ps + (live ? c : 0);
  const long long rstride = (long long)Wf * ps;
  Cx raw[R1];
  if (j < R2) {
#pragma unroll
    for (int q = 0; q < R1; ++q) {
      const float* px = in + col0 + (long long)(j + q * R2) * rstride;
      raw[q] = Cx{px[re_off], px[im_off]};
    }
  }
  for (int m = 0; m < nm; ++m) {
    if (j < R2) {
      const float* mk = masks + (long long)m * N * Wf + kx;
      float mv[R1];
#pragma unroll
      for (int q = 0; q < R1; ++q) mv[q] = mk[(j + q * R2) * Wf];
      Cx v[R1];
#pragma unroll
      for (int q = 0; q < R1; ++q) v[q] = live ? Cx{raw[q].r * mv[q], raw[q].i * mv[q]} : Cx{0.f, 0.f};
      Dft<R1, true>::run(v);
#pragma unroll
      for (int p = 0; p < R1; ++p) {
        const int idx = (j * R1 + p) * L + l;
        zr[idx] = v[p].r;
        zi[idx] = v[p].i;
      }
    }
    __syncthreads();
    if (j < R1) {
      Cx v[R2];
      stage2_load<R1, R2, true>(zr, zi, tw, j, l, L, v);
      if (live) {
        float* ob = out + (long long)m * out_band_stride + col0;
#pragma unroll
        for (int p = 0; p < R2; ++p) {
          float* px = ob + (long long)(j + p * R1) * rstride;
          px[re_off] = v[p].r;
          px[im_off] = v[p].i;
        }
      }
    }
    __syncthreads();                                             // the next band's first stage overwrites z
  }
}

// a pair of adjacent channels of one pixel (the real / imaginary part of a two-for-one lane)
template <int DT>
__device__ __forceinline__ Cx ld_pair(const float* base, long long idx) {
  if (DT == FCVSR_F32) { const float2 v = *reinterpret_cast<const float2*>(base + idx); return Cx{v.x, v.y}; }
  const unsigned v = *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(base) + idx);
  if (DT == FCVSR_BF16) return Cx{__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)};
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 h = __builtin_bit_cast(h2, v);
  return Cx{(float)h[0], (float)h[1]};
}

// ---- forward rows, two stages + untangle: complex lane l carries channels c0 + 2l (real part) and c0 + 2l + 1 (imaginary) ----
// The source dtype is a template parameter: with a run-time switch every load sat in its own branch and the 16 loads of a
// lane were issued one round trip after the other (131 us of a 170 us pass was load wait).
template <int R1, int R2, int SDT>
__global__ __launch_bounds__(512, 4) void rfft_rows2_kernel(View src, int n, int H, int logL, float* spec, long long ps,
                                                         int im_off, int re_off, const float2* twtab, int nbatch) {
  constexpr int N = R1 * R2, Wf = N / 2 + 1;
  extern __shared__ __align__(16) float lds[];
  const int L = 1 << logL;
  float* zr = lds;
  float* zi = lds + N * L;
  float2* tw = reinterpret_cast<float2*>(lds + 2 * N * L);
  int grp, chunk;
  xcd_line_chunk(&grp, &chunk);
  const int row0 = grp, nrows = nbatch * H;
  if (row0 >= nrows) return;
  load_twiddles(tw, twtab, N);
  const int l = threadIdx.x & (L - 1), j = threadIdx.x >> logL;
  const int ca = chunk * 2 * L + 2 * l;
  const bool live = ca < n && j < R2;                             // n is even on this path
  // addresses = wave-uniform 64-bit base (line, chunk, q) + one 32-bit lane offset (host checks W * sx < 2^31)
  // loads are unconditional (dead lanes read the row's first pair and discard it): predicated, every 16-bit load + unpack
  // became its own basic block and the 16 loads of a lane were issued one round trip after the other
  const int lane_off = live ? j * (int)src.sx + 2 * l : 0;
  auto issue = [&](const int row, Cx* v) {
    const int b = row / H, y = row - b * H;
    const long long sp = (long long)b * src.sb + (long long)y * src.sy + chunk * 2 * L;
#pragma unroll
    for (int q = 0; q < R1; ++q) {
      const float* bq = SDT == FCVSR_F32 ? src.p + (sp + (long long)q * R2 * src.sx)
                                         : reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(src.p) + (sp + (long long)q * R2 * src.sx));
      const Cx ld = ld_pair<SDT>(bq, lane_off);
      v[q] = live ? ld : Cx{0.f, 0.f};
    }
  };
  {
    const int row = row0;
    if (j < R2) {
      Cx v[R1];
      issue(row, v);
      Dft<R1, false>::run(v);
#pragma unroll
      for (int p = 0; p < R1; ++p) {
        const int idx = (j * R1 + p) * L + l;
        zr[idx] = v[p].r;
        zi[idx] = v[p].i;
      }
    }
    __syncthreads();
    if (j < R1) {                                                  // in place: a lane writes exactly the positions it read
      Cx w[R2];
      stage2_load<R1, R2, false>(zr, zi, tw, j, l, L, w);
#pragma unroll
      for (int p = 0; p < R2; ++p) {
        const int idx = (j + p * R1) * L + l;
        zr[idx] = w[p].r;
        zi[idx] = w[p].i;
      }
    }
    __syncthreads();
    // Z = FFT(xa + i xb):  Xa[k] = (Z[k] + conj Z[N-k]) / 2,  Xb[k] = (Z[k] - conj Z[N-k]) / (2i)
    float* op = spec + ((long long)row * Wf) * ps;
    for (int t = threadIdx.x; t < Wf * L; t += blockDim.x) {
      const int l2 = t & (L - 1), k = t >> logL;
      const int kn = k ? N - k : 0;
      const int c2 = chunk * 2 * L + 2 * l2;
      if (c2 < n) {
        const float kr = zr[k * L + l2], ki = zi[k * L + l2], nr = zr[kn * L + l2], ni = zi[kn * L + l2];
        float* o = op + (long long)k * ps + c2;
        *reinterpret_cast<float2*>(o + re_off) = make_float2(0.5f * (kr + nr), 0.5f * (ki + ni));
        *reinterpret_cast<float2*>(o + im_off) = make_float2(0.5f * (ki - ni), 0.5f * (nr - kr));
      }
    }
  }
}

// ---- inverse rows, two stages: Hermitian extension and two-for-one packing on load, real pairs stored from registers --------
template <int R1, int R2>
__global__ __launch_bounds__(512, 4) void irfft_rows2_kernel(const float* spec, long long ps, int im_off, int re_off, int n, int H,
                                                          int logL, View dst, float scale, const float2* twtab, int nbatch) {
  constexpr int N = R1 * R2, Wf = N / 2 + 1;
  extern __shared__ __align__(16) float lds[];
  const int L = 1 << logL;
  float* zr = lds;
  float* zi = lds + N * L;
  float2* tw = reinterpret_cast<float2*>(lds + 2 * N * L);
  int grp, chunk;
  xcd_line_chunk(&grp, &chunk);
  const int row0 = grp, nrows = nbatch * H;
  if (row0 >= nrows) return;
  load_twiddles(tw, twtab, N);
  const int l = threadIdx.x & (L - 1), j = threadIdx.x >> logL;
  const int ca = chunk * 2 * L + 2 * l;
  const bool live = ca < n;
  auto issue = [&](const int row, Cx* v) {                         // Hermitian extension + two-for-one packing: Z = Xa + i Xb
    const float* ip = spec + ((long long)row * Wf) * ps + chunk * 2 * L;      // wave-uniform; lane part below is 32-bit
    const bool on = live && j < R2;
    float2 re[R1], im[R1];                                         // all 2 R1 loads first, then the packing arithmetic
#pragma unroll
    for (int q = 0; q < R1; ++q) {
      const int k = j + q * R2;
      const int kk = (k <= N / 2) ? k : N - k;
      const int lo = on ? kk * (int)ps + 2 * l : 0;               // unconditional loads (see rfft_rows2_kernel)
      re[q] = *reinterpret_cast<const float2*>(ip + re_off + lo);   // (Re Xa, Re Xb)
      im[q] = *reinterpret_cast<const float2*>(ip + im_off + lo);   // (Im Xa, Im Xb)
    }
#pragma unroll
    for (int q = 0; q < R1; ++q) {
      const int k = j + q * R2;
      const int kk = (k <= N / 2) ? k : N - k;
      const float sgn = (k > N / 2) ? -1.f : 1.f;
      const float keep = ((kk == 0) || ((N % 2 == 0) && kk == N / 2)) ? 0.f : sgn;   // imag parts: dropped at DC / Nyquist
      v[q] = on ? Cx{re[q].x - keep * im[q].y, keep * im[q].x + re[q].y} : Cx{0.f, 0.f};
    }
  };
  {
    const int row = row0;
    if (j < R2) {
      Cx v[R1];
      issue(row, v);
      Dft<R1, true>::run(v);
#pragma unroll
      for (int p = 0; p < R1; ++p) {
        const int idx = (j * R1 + p) * L + l;
        zr[idx] = v[p].r;
        zi[idx] = v[p].i;
      }
    }
    __syncthreads();
    if (j < R1) {
      Cx w[R2];
      stage2_load<R1, R2, true>(zr, zi, tw, j, l, L, w);
      if (live) {
        const int b = row / H, y = row - b * H;
        float* op = dst.p + (long long)b * dst.sb + (long long)y * dst.sy + chunk * 2 * L;
        const int lo = j * (int)dst.sx + 2 * l;
#pragma unroll
        for (int p = 0; p < R2; ++p)
          *reinterpret_cast<float2*>(op + (long long)p * R1 * dst.sx + lo) = make_float2(w[p].r * scale, w[p].i * scale);
      }
    }
  }
}

// Factorisations of the two-stage path (first factor = first stage).  Listed lengths only; everything else uses the plan.
struct TwoStage { int N, R1, R2; };
static const TwoStage kTwoStage[] = {{64, 8, 8},    {72, 8, 9},    {80, 8, 10},   {96, 8, 12},   {128, 8, 16},  {144, 12, 12},
                                     {160, 10, 16}, {180, 12, 15}, {192, 12, 16}, {240, 15, 16}, {256, 16, 16}, {320, 16, 20}};
static const TwoStage* two_stage(int N) {
  static const bool off = getenv("FCVSR_FFT_2S") && atoi(getenv("FCVSR_FFT_2S")) == 0;
  if (off) return nullptr;
  for (const TwoStage& t : kTwoStage)
    if (t.N == N) return &t;
  return nullptr;
}

// per-device table of W_N^m = (cos, -sin)(2 pi m / N), f64 on the host, rounded once (like pocketfft's table)
static const float2* twiddle_table(int N, hipStream_t st) {
  struct Entry { int N; float2* p; };
  static Entry tabs[64][16];
  static int ntab[64] = {0};
  static std::mutex mu;                                    // first use may come from several host threads (streamed harness)
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("fft: bad device index"); return nullptr; }
  for (int i = 0; i < ntab[dev]; ++i)
    if (tabs[dev][i].N == N) return tabs[dev][i].p;
  if (ntab[dev] >= 16) { set_error("fft: too many twiddle tables"); return nullptr; }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
    set_error("fft: first transform of length %d on this device inside a stream capture (run it once eagerly first)", N);
    return nullptr;
  }
  float2* host = (float2*)malloc(sizeof(float2) * N);
  if (!host) { set_error("fft: out of host memory"); return nullptr; }
  for (int m = 0; m < N; ++m) {
    const double a = 2.0 * 3.14159265358979323846 * (double)m / (double)N;
    host[m] = make_float2((float)cos(a), (float)(-sin(a)));
  }
  float2* d = nullptr;
  hipError_t e = hipMalloc(&d, sizeof(float2) * N);
  if (e == hipSuccess) e = hipMemcpy(d, host, sizeof(float2) * N, hipMemcpyHostToDevice);
  free(host);
  if (e != hipSuccess) { set_error("fft: twiddle table: %s", hipGetErrorString(e)); return nullptr; }
  tabs[dev][ntab[dev]].N = N;
  tabs[dev][ntab[dev]].p = d;
  ++ntab[dev];
  return d;
}

// lanes per workgroup of the two-stage kernels: a power of two with max(R1, R2) * L <= 512 threads and one complex buffer
// (8 N L bytes) within ~48 KB, so that three workgroups share a CU
static int pick_lanes2(const TwoStage& ts, int lanes_needed) {
  const int rmax = ts.R1 > ts.R2 ? ts.R1 : ts.R2;
  int L = 32;
  while (L > 1 && (rmax * L > 512 || 8ll * ts.N * L > 48 * 1024)) L >>= 1;
  while (L > 1 && L / 2 >= lanes_needed) L >>= 1;
  return L;
}
static int ilog2(int v) { int r = 0; while ((1 << r) < v) ++r; return r; }

#define FCVSR_FFT2_DISPATCH(ts, CALL)                                                                        \
  do {                                                                                                       \
    switch ((ts).N) {                                                                                        \
      case 64: CALL(8, 8); break;     case 72: CALL(8, 9); break;     case 80: CALL(8, 10); break;           \
      case 96: CALL(8, 12); break;    case 128: CALL(8, 16); break;   case 144: CALL(12, 12); break;         \
      case 160: CALL(10, 16); break;  case 180: CALL(12, 15); break;  case 192: CALL(12, 16); break;         \
      case 240: CALL(15, 16); break;  case 256: CALL(16, 16); break;  case 320: CALL(16, 20); break;         \
    }                                                                                                        \
  } while (0)
// Column passes: the register butterflies of 15 x 16 / 16 x 20 points (and, with the band masks, 12 x 16 / 16 x 16 too) do not fit the
// 128-register budget of the three-workgroups-per-CU column kernels - hipcc spilled them to scratch - so those lengths are not
// built as two-stage COLUMN kernels: column transforms of 240 / 320 rows (192 / 256 with masks) take the multi-stage path.
#define FCVSR_FFT2_DISPATCH_COLS(ts, CALL)                                                                   \
  do {                                                                                                       \
    switch ((ts).N) {                                                                                        \
      case 64: CALL(8, 8); break;     case 72: CALL(8, 9); break;     case 80: CALL(8, 10); break;           \
      case 96: CALL(8, 12); break;    case 128: CALL(8, 16); break;   case 144: CALL(12, 12); break;         \
      case 160: CALL(10, 16); break;  case 180: CALL(12, 15); break;  case 192: CALL(12, 16); break;         \
      case 256: CALL(16, 16); break;                                                                         \
    }                                                                                                        \
  } while (0)
#define FCVSR_FFT2_DISPATCH_BANDS(ts, CALL)                                                                  \
  do {                                                                                                       \
    switch ((ts).N) {                                                                                        \
      case 64: CALL(8, 8); break;     case 72: CALL(8, 9); break;     case 80: CALL(8, 10); break;           \
      case 96: CALL(8, 12); break;    case 128: CALL(8, 16); break;   case 144: CALL(12, 12); break;         \
      case 160: CALL(10, 16); break;  case 180: CALL(12, 15); break;                                         \
    }                                                                                                        \
  } while (0)
static const TwoStage* two_stage_cols(int N) { return (N == 240 || N == 320) ? nullptr : two_stage(N); }
static const TwoStage* two_stage_bands(int N) { return (N == 192 || N == 240 || N == 256 || N == 320) ? nullptr : two_stage(N); }

static int launch_cols2(const TwoStage& ts, const float* in, float* out, long long ps, int im_off, int re_off, int n, int Wf,
                        int B, bool inverse, const float* mask, hipStream_t st) {
  const float2* tw = twiddle_table(ts.N, st);
  if (!tw) return FCVSR_E_ARG;
  const int L = pick_lanes2(ts, n), logL = ilog2(L);
  const int rmax = ts.R1 > ts.R2 ? ts.R1 : ts.R2;
  const dim3 grid(cdiv(n, L), (B * Wf + 7) / 8 * 8), block((rmax * L + 63) / 64 * 64);
  const size_t lds = 8ull * ts.N * L + 8ull * ts.N;
#define FCVSR_COLS2(A_, B_)                                                                                                 \
  if (inverse && mask) hipLaunchKernelGGL((fft_cols2_kernel<A_, B_, true, true>), grid, block, lds, st, in, out, ps, im_off, \
                                          re_off, n, Wf, logL, mask, tw, B);                                                 \
  else if (inverse) hipLaunchKernelGGL((fft_cols2_kernel<A_, B_, true, false>), grid, block, lds, st, in, out, ps, im_off,   \
                                       re_off, n, Wf, logL, mask, tw, B);                                                    \
  else hipLaunchKernelGGL((fft_cols2_kernel<A_, B_, false, false>), grid, block, lds, st, in, out, ps, im_off, re_off, n,   \
                          Wf, logL, mask, tw, B)
  FCVSR_FFT2_DISPATCH_COLS(ts, FCVSR_COLS2);
#undef FCVSR_COLS2
  return 0;
}

static int launch_cols2_bands(const TwoStage& ts, const float* in, float* out, long long out_band_stride, long long ps, int im_off,
                             int re_off, int n, int Wf, int B, const float* masks, int nm, hipStream_t st) {
  const float2* tw = twiddle_table(ts.N, st);
  if (!tw) return FCVSR_E_ARG;
  const int L = pick_lanes2(ts, n), logL = ilog2(L);
  const int rmax = ts.R1 > ts.R2 ? ts.R1 : ts.R2;
  const dim3 grid(cdiv(n, L), (B * Wf + 7) / 8 * 8), block((rmax * L + 63) / 64 * 64);
  const size_t lds = 8ull * ts.N * L + 8ull * ts.N;
#define FCVSR_COLS2B(A_, B_) \
  hipLaunchKernelGGL((fft_cols2_bands_kernel<A_, B_>), grid, block, lds, st, in, out, out_band_stride, ps, im_off, re_off, n, Wf, logL, masks, nm, tw, B)
  FCVSR_FFT2_DISPATCH_BANDS(ts, FCVSR_COLS2B);
#undef FCVSR_COLS2B
  return 0;
}

static int launch_rfft_rows2(const TwoStage& ts, const fcvsr_view* src, int n, int B, int H, float* spec, long long ps, int im_off,
                             int re_off, hipStream_t st) {
  const float2* tw = twiddle_table(ts.N, st);
  if (!tw) return FCVSR_E_ARG;
  static const int envL = getenv("FCVSR_FFT_ROWL") ? atoi(getenv("FCVSR_FFT_ROWL")) : 0;
  int L = pick_lanes2(ts, n / 2);
  if (envL > 0 && envL < L) L = envL;
  const int logL = ilog2(L);
  const int rmax = ts.R1 > ts.R2 ? ts.R1 : ts.R2;
  const dim3 grid(cdiv(n, 2 * L), (B * H + 7) / 8 * 8), block((rmax * L + 63) / 64 * 64);
  const size_t lds = 8ull * ts.N * L + 8ull * ts.N;
#define FCVSR_ROWS2(A_, B_)                                                                                                      \
  if (src->dtype == FCVSR_F32)                                                                                                   \
    hipLaunchKernelGGL((rfft_rows2_kernel<A_, B_, FCVSR_F32>), grid, block, lds, st, to_view(*src), n, H, logL, spec, ps, im_off, \
                       re_off, tw, B);                                                                                           \
  else if (src->dtype == FCVSR_BF16)                                                                                             \
    hipLaunchKernelGGL((rfft_rows2_kernel<A_, B_, FCVSR_BF16>), grid, block, lds, st, to_view(*src), n, H, logL, spec, ps,       \
                       im_off, re_off, tw, B);                                                                                   \
  else                                                                                                                           \
    hipLaunchKernelGGL((rfft_rows2_kernel<A_, B_, FCVSR_F16>), grid, block, lds, st, to_view(*src), n, H, logL, spec, ps, im_off, \
                       re_off, tw, B)
  FCVSR_FFT2_DISPATCH(ts, FCVSR_ROWS2);
#undef FCVSR_ROWS2
  return 0;
}

static int launch_irfft_rows2(const TwoStage& ts, const float* spec, long long ps, int im_off, int re_off, int n, int B, int H,
                              const fcvsr_view* dst, float scale, hipStream_t st) {
  const float2* tw = twiddle_table(ts.N, st);
  if (!tw) return FCVSR_E_ARG;
  static const int envL = getenv("FCVSR_FFT_ROWL") ? atoi(getenv("FCVSR_FFT_ROWL")) : 0;
  int L = pick_lanes2(ts, n / 2);
  if (envL > 0 && envL < L) L = envL;
  const int logL = ilog2(L);
  const int rmax = ts.R1 > ts.R2 ? ts.R1 : ts.R2;
  const dim3 grid(cdiv(n, 2 * L), (B * H + 7) / 8 * 8), block((rmax * L + 63) / 64 * 64);
  const size_t lds = 8ull * ts.N * L + 8ull * ts.N;
#define FCVSR_IROWS2(A_, B_) \
  hipLaunchKernelGGL((irfft_rows2_kernel<A_, B_>), grid, block, lds, st, spec, ps, im_off, re_off, n, H, logL, to_view(*dst), scale, tw, B)
  FCVSR_FFT2_DISPATCH(ts, FCVSR_IROWS2);
#undef FCVSR_IROWS2
  return 0;
}

// channel pairs of a view are loaded / stored as one 4-byte (16-bit) or 8-byte (f32) access
static bool pair_ok(const fcvsr_view* v) {
  const int al = v->dtype == FCVSR_F32 ? 8 : 4;
  return v->sc == 1 && v->sx % 2 == 0 && v->sy % 2 == 0 && v->sb % 2 == 0 && ((uintptr_t)v->ptr % al) == 0;
}

static long long lds_budget() {
  static long long b = -1;
  if (b < 0) {
    const char* e = getenv("FCVSR_FFT_LDS_KB");
    b = (e ? atoll(e) : 64) * 1024;   // measured: 48-64 KiB (2-3 workgroups per CU) is 1.5x faster than 1 fat workgroup
  }
  return b;
}

static int pick_lanes(int N, int n_lanes_needed) {
  // LDS bytes = 16*N*L + 8*N within the budget; L power of two in [1,32]
  int L = 32;
  while (L > 1 && (16ll * N * L + 8ll * N) > lds_budget()) L >>= 1;
  while (L > 1 && L / 2 >= n_lanes_needed) L >>= 1;
  return L;
}

template <class K>
static hipError_t allow_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_rfft2(const fcvsr_view* src, int B, int H, int W, int n, float* spec, int64_t pix_stride,
                           int im_off, int re_off, void* stream) {
  FCVSR_CHECK_ARG(src && src->ptr && spec, "null pointer");
  FCVSR_CHECK_ARG(src->dtype == FCVSR_F32 || src->dtype == FCVSR_BF16 || src->dtype == FCVSR_F16, "bad src dtype");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 1 && n > 0 && n <= src->c, "bad sizes");
  FftPlan pw, ph;
  FCVSR_CHECK_ARG(make_plan(W, &pw) && make_plan(H, &ph), "length has too many factors");
  const int Wf = W / 2 + 1;
  hipStream_t st = (hipStream_t)stream;
  const bool spec_ok = pix_stride % 4 == 0 && im_off % 4 == 0 && re_off % 4 == 0 && ((uintptr_t)spec % 16) == 0;
  const bool spec_pair = pix_stride % 2 == 0 && im_off % 2 == 0 && re_off % 2 == 0 && ((uintptr_t)spec % 8) == 0;
  const TwoStage* tsw = two_stage(W);
  const TwoStage* tsh = two_stage_cols(H);
  if (tsw && n % 2 == 0 && pair_ok(src) && spec_pair) {
    const int rc = launch_rfft_rows2(*tsw, src, n, B, H, spec, (long long)pix_stride, im_off, re_off, st);
    if (rc) return rc;
    FCVSR_LAUNCH_CHECK();
  } else {
    const int L = pick_lanes(W, (n + 1) / 2);
    FCVSR_CHECK_ARG(16ll * W * L + 8ll * W <= 160 * 1024, "row too long for LDS");
    const size_t lds = 16ull * W * L + 8ull * W;
    (void)allow_lds(rfft_rows_kernel, lds);
    dim3 grid(cdiv(n, 2 * L), (B * H + 7) / 8 * 8);    // rows rounded up: the kernel remaps ids in groups of 8 rows
    const int sal = src->dtype == FCVSR_F32 ? 16 : 8;
    const int vec = (L >= 4 && n % (2 * L) == 0 && src->sc == 1 && src->sx % 4 == 0 && src->sy % 4 == 0 && src->sb % 4 == 0 &&
                     ((uintptr_t)src->ptr % sal) == 0 && spec_ok) ? 1 : 0;
    hipLaunchKernelGGL(rfft_rows_kernel, grid, dim3(512), lds, st, to_view(*src), (int)src->dtype, n, H, W, L, spec,
                       (long long)pix_stride, im_off, re_off, pw, vec, B);
    FCVSR_LAUNCH_CHECK();
  }
  if (tsh) {
    const int rc = launch_cols2(*tsh, spec, spec, (long long)pix_stride, im_off, re_off, n, Wf, B, false, nullptr, st);
    if (rc) return rc;
    FCVSR_LAUNCH_CHECK();
  } else {
    const int L = pick_lanes(H, n);
    FCVSR_CHECK_ARG(16ll * H * L + 8ll * H <= 160 * 1024, "column too long for LDS");
    const size_t lds = 16ull * H * L + 8ull * H;
    (void)allow_lds(fft_cols_kernel, lds);
    dim3 grid(cdiv(n, L), (B * Wf + 7) / 8 * 8);
    const int vec = (L >= 4 && n % L == 0 && spec_ok) ? 1 : 0;
    hipLaunchKernelGGL(fft_cols_kernel, grid, dim3(512), lds, st, (const float*)spec, spec, (long long)pix_stride, im_off,
                       re_off, n, H, Wf, L, 0, (const float*)nullptr, ph, vec, B);
    FCVSR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int fcvsr_irfft2(const float* spec, int64_t pix_stride, int im_off, int re_off, int B, int H, int W, int n,
                            const float* mask, float* work, const fcvsr_view* dst, void* stream) {
  FCVSR_CHECK_ARG(spec && dst && dst->ptr, "null pointer");
  FCVSR_CHECK_ARG(dst->dtype == FCVSR_F32, "f32 only");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 1 && n > 0 && n <= dst->c, "bad sizes");
  FftPlan pw, ph;
  FCVSR_CHECK_ARG(make_plan(W, &pw) && make_plan(H, &ph), "length has too many factors");
  const int Wf = W / 2 + 1;
  hipStream_t st = (hipStream_t)stream;
  float* mid = work ? work : const_cast<float*>(spec);
  const bool spec_ok = pix_stride % 4 == 0 && im_off % 4 == 0 && re_off % 4 == 0 && ((uintptr_t)spec % 16) == 0;
  const bool mid_pair = pix_stride % 2 == 0 && im_off % 2 == 0 && re_off % 2 == 0 && ((uintptr_t)mid % 8) == 0;
  const TwoStage* tsw = two_stage(W);
  const TwoStage* tsh = two_stage_cols(H);
  if (tsh) {
    const int rc = launch_cols2(*tsh, spec, mid, (long long)pix_stride, im_off, re_off, n, Wf, B, true, mask, st);
    if (rc) return rc;
    FCVSR_LAUNCH_CHECK();
  } else {
    const int L = pick_lanes(H, n);
    FCVSR_CHECK_ARG(16ll * H * L + 8ll * H <= 160 * 1024, "column too long for LDS");
    const size_t lds = 16ull * H * L + 8ull * H;
    (void)allow_lds(fft_cols_kernel, lds);
    dim3 grid(cdiv(n, L), (B * Wf + 7) / 8 * 8);
    const int vec = (L >= 4 && n % L == 0 && spec_ok && ((uintptr_t)mid % 16) == 0) ? 1 : 0;
    hipLaunchKernelGGL(fft_cols_kernel, grid, dim3(512), lds, st, spec, mid, (long long)pix_stride, im_off, re_off, n, H,
                       Wf, L, 1, mask, ph, vec, B);
    FCVSR_LAUNCH_CHECK();
  }
  if (tsw && n % 2 == 0 && pair_ok(dst) && mid_pair) {
    const int rc = launch_irfft_rows2(*tsw, mid, (long long)pix_stride, im_off, re_off, n, B, H, dst,
                                      1.0f / ((float)H * (float)W), st);
    if (rc) return rc;
    FCVSR_LAUNCH_CHECK();
  } else {
    const int L = pick_lanes(W, (n + 1) / 2);
    FCVSR_CHECK_ARG(16ll * W * L + 8ll * W <= 160 * 1024, "row too long for LDS");
    const size_t lds = 16ull * W * L + 8ull * W;
    (void)allow_lds(irfft_rows_kernel, lds);
    dim3 grid(cdiv(n, 2 * L), (B * H + 7) / 8 * 8);
    const int vec = (L >= 4 && n % (2 * L) == 0 && spec_ok && ((uintptr_t)mid % 16) == 0 && dst->sc == 1 && dst->sx % 4 == 0 &&
                     dst->sy % 4 == 0 && dst->sb % 4 == 0 && ((uintptr_t)dst->ptr % 16) == 0) ? 1 : 0;
    hipLaunchKernelGGL(irfft_rows_kernel, grid, dim3(512), lds, st, (const float*)mid, (long long)pix_stride, im_off,
                       re_off, n, H, W, L, to_view(*dst), 1.0f / ((float)H * (float)W), pw, vec, B);
    FCVSR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int fcvsr_irfft2_bands(const float* spec, int64_t pix_stride, int im_off, int re_off, int B, int H, int W, int n,
                                  const float* masks, int n_bands, float* work, const fcvsr_view* dst, void* stream) {
  FCVSR_CHECK_ARG(spec && masks && work && dst && n_bands >= 1 && n_bands <= 16, "null argument / 1..16 bands");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 1 && n > 0, "bad sizes");
  const int Wf = W / 2 + 1;
  const long long band = (long long)B * H * Wf * pix_stride;           // floats per band of `work`
  hipStream_t st = (hipStream_t)stream;
  const TwoStage* tsh = two_stage_bands(H);
  static const bool off = getenv("FCVSR_FFT_BANDS") && atoi(getenv("FCVSR_FFT_BANDS")) == 0;
  if (!tsh || off) {                                                    // no fused column pass for this height: band by band
    for (int m = 0; m < n_bands; ++m) {
      const int rc = fcvsr_irfft2(spec, pix_stride, im_off, re_off, B, H, W, n, masks + (long long)m * H * Wf, work, &dst[m], stream);
      if (rc) return rc;
    }
    return 0;
  }
  for (int m = 0; m < n_bands; ++m)
    FCVSR_CHECK_ARG(dst[m].ptr && dst[m].dtype == FCVSR_F32 && n <= dst[m].c, "dst: f32 views with >= n channels");
  {
    const int rc = launch_cols2_bands(*tsh, spec, work, band, (long long)pix_stride, im_off, re_off, n, Wf, B, masks, n_bands, st);
    if (rc) return rc;
    FCVSR_LAUNCH_CHECK();
  }
  // row passes: the inverse column pass of fcvsr_irfft2 is skipped by handing it each band's transformed columns... the row
  // kernels are launched directly here (same dispatch as fcvsr_irfft2)
  FftPlan pw;
  FCVSR_CHECK_ARG(make_plan(W, &pw), "length has too many factors");
  const TwoStage* tsw = two_stage(W);
  for (int m = 0; m < n_bands; ++m) {
    const float* mid = work + (long long)m * band;
    const bool mid_pair = pix_stride % 2 == 0 && im_off % 2 == 0 && re_off % 2 == 0 && ((uintptr_t)mid % 8) == 0;
    const bool spec_ok = pix_stride % 4 == 0 && im_off % 4 == 0 && re_off % 4 == 0 && ((uintptr_t)mid % 16) == 0;
    if (tsw && n % 2 == 0 && pair_ok(&dst[m]) && mid_pair) {
      const int rc = launch_irfft_rows2(*tsw, mid, (long long)pix_stride, im_off, re_off, n, B, H, &dst[m], 1.0f / ((float)H * (float)W), st);
      if (rc) return rc;
      FCVSR_LAUNCH_CHECK();
    } else {
      const int L = pick_lanes(W, (n + 1) / 2);
      FCVSR_CHECK_ARG(16ll * W * L + 8ll * W <= 160 * 1024, "row too long for LDS");
      const size_t lds = 16ull * W * L + 8ull * W;
      (void)allow_lds(irfft_rows_kernel, lds);
      dim3 grid(cdiv(n, 2 * L), (B * H + 7) / 8 * 8);
      const int vec = (L >= 4 && n % (2 * L) == 0 && spec_ok && dst[m].sc == 1 && dst[m].sx % 4 == 0 && dst[m].sy % 4 == 0 &&
                       dst[m].sb % 4 == 0 && ((uintptr_t)dst[m].ptr % 16) == 0) ? 1 : 0;
      hipLaunchKernelGGL(irfft_rows_kernel, grid, dim3(512), lds, st, mid, (long long)pix_stride, im_off, re_off, n, H, W, L,
                         to_view(dst[m]), 1.0f / ((float)H * (float)W), pw, vec, B);
      FCVSR_LAUNCH_CHECK();
    }
  }
  return 0;
}
