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
  {
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
  {
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
  {
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
  {
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
