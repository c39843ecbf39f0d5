// Run-time channel count, 9 <= M <= 32: the kernels of the wide-channel path (assx_widem.hip) whose register / lane-group
// layouts are tied to a compile-time M <= 8, restated for any M.  The reference is generic in M (src/bss/ilrma.py:61-62,
// src/bss/iva.py:39-59); arrays with more than 8 channels are rare, so this path is FUNCTIONAL, not tuned: every kernel
// is the simplest deterministic form (matrices in LDS, one wave or one workgroup per bin), the arithmetic contract (floors,
// pivot rule, condition guard, Gauss-Seidel order) is the M <= 8 path's.  Everything that is already generic there (the
// P / R maps, the matrix-core source model, the fixed-order reductions, power_cov_kernel) is shared.
#pragma once
#include "assx_small_linalg.hpp"
#include "assx_stream.hpp"
#include "assx_widem.hpp"

namespace assx {
namespace widem {

// ------------------------------------------------------------------------------------------------------------------
// y = W x per (f, t), run-time M; W_f in (dynamic) LDS.  Writes Y (optionally scaled) and / or |y|^2.
// ------------------------------------------------------------------------------------------------------------------
template <typename R, int MC>  // MC: compile-time bound of the channel count (x of a frame stays in registers)
__global__ void __launch_bounds__(256) demix_map_rt_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W,
                                                          const Cx<R>* __restrict__ scale, Cx<R>* __restrict__ Y,
                                                          R* __restrict__ P, int F, int T, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rt[];
  Cx<R>* w = reinterpret_cast<Cx<R>*>(smem_rt);
  const int f = blockIdx.y, b = blockIdx.z;
  for (int i = threadIdx.x; i < M * M; i += blockDim.x) w[i] = W[((size_t)b * F + f) * ((size_t)M * M) + i];
  __syncthreads();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const size_t FT = (size_t)F * T, base = (size_t)b * M * FT + (size_t)f * T + t;
  Cx<R> x[MC];
#pragma unroll
  for (int m = 0; m < MC; ++m) x[m] = m < M ? X[base + (size_t)m * FT] : cmake<R>(0, 0);
  for (int n = 0; n < M; ++n) {
    Cx<R> s = cmake<R>(0, 0);
#pragma unroll
    for (int m = 0; m < MC; ++m)
      if (m < M) cfma(s, w[n * M + m], x[m]);  // m ascending, as the compile-time kernel
    if (P) P[base + (size_t)n * FT] = cabs2(s);
    if (Y) {
      if (scale) s = cmul(s, scale[((size_t)b * M + n) * F + f]);
      Y[base + (size_t)n * FT] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Weighted covariance of NC sources of a bin per workgroup: a thread owns ONE pair (i <= j) of the Hermitian matrix and
// the sums of that pair for the workgroup's sources: per frame it forms x_i conj(x_j) once and fans it into the NC sums
// with the frame's reciprocal weights.  Frames arrive in tiles of
// RT_TILE, staged FRAME-MAJOR in LDS (at a given frame every thread reads two of the M samples of one short row:
// broadcasts, no bank conflicts) next to the tile's weights [frame][source] (wave-uniform reads).  blockIdx.y = (source
// group, chunk of 256 pairs).  NC trades re-staging the tile per source group against workgroups in flight and the
// weight reads per frame (measured: profiles/r03_manychan_bench.txt).
// U[b,n,f] = (1/T) sum_t x x^H / max(r_n, eps); r_kind as cov_bin_kernel (0 none, 1 (N,T), 2 (N,F,T)).
// ------------------------------------------------------------------------------------------------------------------
constexpr int RT_TILE = 64;

template <typename R, int NC>
__global__ void __launch_bounds__(256) cov_rt_kernel(const Cx<R>* __restrict__ X, const R* __restrict__ r, int r_kind,
                                                    int N, R eps, Cx<R>* __restrict__ U, int F, int T, R inv_T, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rt[];
  Cx<R>* xt = reinterpret_cast<Cx<R>*>(smem_rt);            // [RT_TILE][M]
  R* wt = reinterpret_cast<R*>(xt + (size_t)RT_TILE * M);   // [RT_TILE][N]
  const int f = blockIdx.x, b = blockIdx.z;
  const int NP = M * (M + 1) / 2;
  const int pchunks = (NP + (int)blockDim.x - 1) / (int)blockDim.x;
  const int sg = (int)blockIdx.y / pchunks, pc = (int)blockIdx.y - sg * pchunks;  // source group, pair chunk
  const int n0 = sg * NC, ns = min(NC, N - n0);                                   // this workgroup's sources n0 .. n0+ns-1
  const size_t FT = (size_t)F * T;
  const Cx<R>* xb = X + (size_t)b * M * FT + (size_t)f * T;
  int pi = -1, pj = -1;
  {
    int p = pc * (int)blockDim.x + (int)threadIdx.x;
    if (p < NP) {  // p -> (i, j >= i), rows ascending
      int i = 0;
      while (p >= M - i) {
        p -= M - i;
        ++i;
      }
      pi = i;
      pj = i + p;
    }
  }
  R ar[NC], ai[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n) ar[n] = ai[n] = 0;
  for (int t0 = 0; t0 < T; t0 += RT_TILE) {
    __syncthreads();  // the previous tile is no longer read
    for (int e = threadIdx.x; e < M * RT_TILE; e += blockDim.x) {
      const int m = e / RT_TILE, tt = e - m * RT_TILE;  // consecutive threads: consecutive frames of one channel (coalesced)
      xt[(size_t)tt * M + m] = (t0 + tt < T) ? xb[(size_t)m * FT + t0 + tt] : cmake<R>(0, 0);
    }
    for (int e = threadIdx.x; e < ns * RT_TILE; e += blockDim.x) {
      const int q = e / RT_TILE, n = n0 + q, tt = e - q * RT_TILE, t = t0 + tt;
      R wv = 0;
      if (t < T) {
        if (r_kind == 0) wv = (R)1;
        else if (r_kind == 1) wv = fast_rcp(floor_eps<R>(r[((size_t)b * N + n) * T + t], eps));
        else wv = fast_rcp(floor_eps<R>(r[((size_t)b * N + n) * FT + (size_t)f * T + t], eps));
      }
      wt[(size_t)tt * NC + q] = wv;
    }
    __syncthreads();
    if (pi >= 0) {
      for (int tt = 0; tt < RT_TILE; ++tt) {  // frames in ascending order
        const Cx<R> xi = xt[(size_t)tt * M + pi], xj = xt[(size_t)tt * M + pj];
        const R pr = fma(xi.x, xj.x, xi.y * xj.y), pim = fma(xi.y, xj.x, -(xi.x * xj.y));  // x_i conj(x_j)
        const R* wrow = wt + (size_t)tt * NC;
#pragma unroll
        for (int q = 0; q < NC; ++q)
          if (q < ns) {
            ar[q] = fma(wrow[q], pr, ar[q]);
            ai[q] = fma(wrow[q], pim, ai[q]);
          }
      }
    }
  }
  if (pi >= 0) {
#pragma unroll
    for (int q = 0; q < NC; ++q)
      if (q < ns) {
        Cx<R>* un = U + (((size_t)b * N + n0 + q) * F + f) * ((size_t)M * M);
        const R re = ar[q] * inv_T, im = pi == pj ? (R)0 : ai[q] * inv_T;
        un[pi * M + pj] = cmake<R>(re, im);
        if (pi != pj) un[pj * M + pi] = cmake<R>(re, -im);
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// One wave per bin, matrices (float64) in LDS.  Helpers: every lane of the wave calls them; `wave_sync_lds()` orders the
// wave's LDS traffic between phases.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_sum_d(double v) { return wave_allreduce_sum<double>(v); }

// In-place inverse of A (M x M, row-major, LDS) by Gauss-Jordan with partial row pivoting, LAPACK's pivot rule
// (|re| + |im|, first maximum) -- gj_inverse_rt of assx_generic.hip with the row operations spread over the wave.
// piv: M ints in LDS.  Returns false on an exactly zero pivot.
__device__ inline bool wave_gj_inverse(Cd* A, int* piv, int M, int lane) {
  bool ok = true;
  for (int c = 0; c < M; ++c) {
    // pivot search over rows c..M-1: first row with the largest |re| + |im|
    double best = -1.0;
    int p = c;
    for (int r = c + lane; r < M; r += WAVE) {
      const double v = cabs1(A[r * M + c]);
      if (v > best) {
        best = v;
        p = r;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ob = __shfl_xor(best, off, WAVE);
      const int op = __shfl_xor(p, off, WAVE);
      if (ob > best || (ob == best && op < p)) {
        best = ob;
        p = op;
      }
    }
    if (lane == 0) piv[c] = p;
    if (!(best > 0.0)) ok = false;
    if (p != c)
      for (int j = lane; j < M; j += WAVE) cswap(A[c * M + j], A[p * M + j]);
    wave_sync_lds();
    const Cd pv = A[c * M + c];
    // eliminate with the UNSCALED pivot row and the LU multiplier (an exactly dependent row cancels to exact zeros)
    for (int e = lane; e < M * M; e += WAVE) {
      const int r = e / M, j = e - r * M;
      if (r == c || j == c) continue;
      const Cd fct = cdiv(A[r * M + c], pv);
      const Cd a = A[c * M + j];
      Cd v = A[e];
      v.x = v.x - (fct.x * a.x - fct.y * a.y);
      v.y = v.y - (fct.x * a.y + fct.y * a.x);
      A[e] = v;
    }
    wave_sync_lds();
    for (int r = lane; r < M; r += WAVE) {
      if (r == c) continue;
      const Cd fct = cdiv(A[r * M + c], pv);
      A[r * M + c] = cmake<double>(-fct.x, -fct.y);
    }
    wave_sync_lds();
    const Cd ipv = cdiv(cmake<double>(1.0, 0.0), pv);
    for (int j = lane; j < M; j += WAVE) A[c * M + j] = (j == c) ? ipv : cmul(A[c * M + j], ipv);
    wave_sync_lds();
  }
  for (int c = M - 1; c >= 0; --c) {
    const int p = piv[c];
    if (p != c)
      for (int i = lane; i < M; i += WAVE) cswap(A[i * M + c], A[i * M + p]);
    wave_sync_lds();
  }
  return ok;
}

// spectral norm of A (LDS) by squaring G = A^H A / tr: the arithmetic of group_spectral_norm (assx_group_linalg.hpp);
// G0, G1, G2: M x M scratch in LDS
__device__ inline double wave_spectral_norm(const Cd* A, Cd* G0, Cd* G1, Cd* G2, int M, int lane) {
  for (int e = lane; e < M * M; e += WAVE) {
    const int i = e / M, j = e - i * M;
    Cd g = cmake<double>(0.0, 0.0);
    for (int k = 0; k < M; ++k) cfma(g, cconj(A[k * M + i]), A[k * M + j]);
    G0[e] = g;
  }
  wave_sync_lds();
  double trp = 0.0;
  for (int i = lane; i < M; i += WAVE) trp += G0[i * M + i].x;
  const double tr = wave_sum_d(trp);
  if (!(tr > 0.0) || !isfinite(tr)) return tr > 0.0 ? tr : 0.0;
  for (int e = lane; e < M * M; e += WAVE) {
    G0[e] = cscale(G0[e], 1.0 / tr);
    G1[e] = G0[e];
  }
  wave_sync_lds();
  Cd* g = G1;
  Cd* h = G2;
  for (int it = 0; it < 24; ++it) {
    for (int e = lane; e < M * M; e += WAVE) {
      const int i = e / M, j = e - i * M;
      Cd s = cmake<double>(0.0, 0.0);
      for (int k = 0; k < M; ++k) cfma(s, g[i * M + k], g[k * M + j]);
      h[e] = s;
    }
    wave_sync_lds();
    double t2p = 0.0;
    for (int i = lane; i < M; i += WAVE) t2p += h[i * M + i].x;
    const double t2 = wave_sum_d(t2p);
    for (int e = lane; e < M * M; e += WAVE) h[e] = cscale(h[e], 1.0 / t2);
    wave_sync_lds();
    Cd* tmp = g;
    g = h;
    h = tmp;
    if (1.0 - t2 < 1e-14) break;  // wave-uniform
  }
  double lp = 0.0;  // Rayleigh quotient tr(G0 Gk), tr(Gk) = 1
  for (int e = lane; e < M * M; e += WAVE) {
    const int i = e / M, j = e - i * M;
    const Cd a = G0[e], bt = g[j * M + i];
    lp += a.x * bt.x - a.y * bt.y;
  }
  return sqrt(wave_sum_d(lp) * tr);
}

// IP sweep (ilrma.py:512-530, iva.py:500-518) of one bin per wave, run-time M: the semantics of ip_group_kernel
// (condition guard with the Frobenius bounds and the exact norms inside the factor-M band, singular / rejected flags,
// den_floor, per-bin power statistic from the plain covariance C).
template <typename R>
__global__ void __launch_bounds__(64) ip_rt_kernel(const Cx<R>* __restrict__ U, Cx<R>* __restrict__ W,
                                                  const Cx<R>* __restrict__ C, double* __restrict__ pw, double thr,
                                                  int32_t* __restrict__ status, int B, int F, double den_floor, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rt[];
  const int MM = M * M, lane = threadIdx.x;
  Cd* Wl = reinterpret_cast<Cd*>(smem_rt);
  Cd* Ul = Wl + MM;
  Cd* Al = Ul + MM;
  Cd* Il = Al + MM;
  Cd* G0 = Il + MM;
  Cd* G1 = G0 + MM;
  Cd* G2 = G1 + MM;
  int* piv = reinterpret_cast<int*>(G2 + MM);
  const int bf = blockIdx.x, b = bf / F, f = bf - b * F, N = M;
  for (int e = lane; e < MM; e += WAVE) {
    const Cx<R> v = W[(size_t)bf * MM + e];
    Wl[e] = cmake<double>((double)v.x, (double)v.y);
  }
  int flags = 0;
  for (int n = 0; n < N; ++n) {
    for (int e = lane; e < MM; e += WAVE) {
      const Cx<R> v = U[(((size_t)b * N + n) * F + f) * MM + e];
      Ul[e] = cmake<double>((double)v.x, (double)v.y);
    }
    wave_sync_lds();
    double nA2p = 0.0;
    for (int e = lane; e < MM; e += WAVE) {  // A = W U_n
      const int i = e / M, j = e - i * M;
      Cd a = cmake<double>(0.0, 0.0);
      for (int k = 0; k < M; ++k) cfma(a, Wl[i * M + k], Ul[k * M + j]);
      Al[e] = a;
      Il[e] = a;
      nA2p += cabs2(a);
    }
    wave_sync_lds();
    const bool nonsing = wave_gj_inverse(Il, piv, M, lane);
    const bool singular = !nonsing;
    double nI2p = 0.0;
    for (int e = lane; e < MM; e += WAVE) nI2p += cabs2(Il[e]);
    const double nA2 = wave_sum_d(nA2p), nI2 = wave_sum_d(nI2p);
    // cond_2(A) < thr: Frobenius bounds, exact spectral norms only inside the factor-M band (group_cond_below)
    double c2 = nA2 * nI2, thr2 = thr * thr, m2 = (double)M * (double)M;
    if (!(c2 > 1e-290 && c2 < 1e290 && thr2 < 1e290)) {
      c2 = sqrt(nA2) * sqrt(nI2);
      thr2 = thr;
      m2 = (double)M;
    }
    const bool amb = !singular && (c2 == c2) && c2 >= thr2 && c2 < thr2 * m2;
    bool ok = !singular && (c2 == c2) && c2 < thr2;
    if (amb) ok = wave_spectral_norm(Al, G0, G1, G2, M, lane) * wave_spectral_norm(Il, G0, G1, G2, M, lane) < thr;
    if (singular) flags |= ASSX_STATUS_SINGULAR;
    else if (!ok) flags |= ASSX_STATUS_COND_REJECT;
    // w = (W U)^{-1} e_n ; den = sqrt(w^H U_n w) ; W[n,:] = conj(w) / den
    double qx = 0.0, qy = 0.0;
    for (int e = lane; e < MM; e += WAVE) {
      const int i = e / M, j = e - i * M;
      const Cd term = cmul(cmul(cconj(Il[i * M + n]), Ul[e]), Il[j * M + n]);
      qx += term.x;
      qy += term.y;
    }
    Cd den = csqrt_fast(cmake<double>(wave_sum_d(qx), wave_sum_d(qy)));
    if (den.x < den_floor) den = cmake<double>(den_floor, 0.0);
    if (ok && !singular)
      for (int j = lane; j < M; j += WAVE) Wl[n * M + j] = cdiv_fast(cconj(Il[j * M + n]), den);
    wave_sync_lds();
  }
  for (int e = lane; e < MM; e += WAVE) W[(size_t)bf * MM + e] = cmake<R>((R)Wl[e].x, (R)Wl[e].y);
  if (pw) {  // per-bin share of mean|y_n|^2 = mean_f w_n^H C_f w_n
    for (int e = lane; e < MM; e += WAVE) {
      const Cx<R> v = C[(size_t)bf * MM + e];
      Ul[e] = cmake<double>((double)v.x, (double)v.y);
    }
    wave_sync_lds();
    for (int n = 0; n < N; ++n) {
      double s = 0.0;
      for (int e = lane; e < MM; e += WAVE) {
        const int i = e / M, j = e - i * M;
        const Cd t1 = cmul(Wl[n * M + i], Ul[e]);
        s += t1.x * Wl[n * M + j].x + t1.y * Wl[n * M + j].y;  // Re(W[n,i] C[i,j] conj(W[n,j]))
      }
      s = wave_sum_d(s);
      if (lane == 0) pw[((size_t)b * N + n) * F + f] = s;
    }
  }
  if (flags && status && lane == 0) atomicOr(&status[b], flags);
}
// ISS sweep (ilrma.py:537-564, iva.py:525-543; round 6) of one bin per wave, run-time M: the arithmetic of iss_group_kernel
// (assx_group_linalg.hpp) -- the rank-one updates  Y <- Y - v_n Y[n]  restated on W with the weighted covariances U_s:
//   t_s = U_s conj(w_n),  q_s = w_s . t_s,  d_s = w_n . t_s (real),  v_s = q_s / d_s (s != n),  v_n = 1 - 1 / sqrt(T d_n),
//   W <- W - v w_n^T  with the OLD row n -- sums, not means, as the reference (the 1 / T of the paper is absent).
// U_s (M x M) is staged in LDS for every (n, s): M^4 multiply-adds and M^2 reads of U per bin.  FUNCTIONAL, not tuned, like
// the rest of this file (5 <= M <= 8 runs iss_group_kernel).  Optionally the per-bin power statistic from C.
template <typename R>
__global__ void __launch_bounds__(64) iss_rt_kernel(const Cx<R>* __restrict__ U, Cx<R>* __restrict__ W,
                                                   const Cx<R>* __restrict__ C, double* __restrict__ pw, double n_frames,
                                                   int B, int F, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rt[];
  const int MM = M * M, lane = threadIdx.x, N = M;
  Cd* Wl = reinterpret_cast<Cd*>(smem_rt);  // M x M
  Cd* Ul = Wl + MM;                         // M x M: U_s (then C)
  Cd* tv = Ul + MM;                         // M: t_s
  Cd* vv = tv + M;                          // M: v
  Cd* wn = vv + M;                          // M: the old row n
  const int bf = blockIdx.x, b = bf / F, f = bf - b * F;
  for (int e = lane; e < MM; e += WAVE) {
    const Cx<R> v = W[(size_t)bf * MM + e];
    Wl[e] = cmake<double>((double)v.x, (double)v.y);
  }
  wave_sync_lds();
  for (int n = 0; n < N; ++n) {
    for (int j = lane; j < M; j += WAVE) wn[j] = Wl[n * M + j];
    wave_sync_lds();
    for (int s = 0; s < N; ++s) {
      for (int e = lane; e < MM; e += WAVE) {
        const Cx<R> v = U[(((size_t)b * N + s) * F + f) * MM + e];
        Ul[e] = cmake<double>((double)v.x, (double)v.y);
      }
      wave_sync_lds();
      for (int i = lane; i < M; i += WAVE) {  // t_s[i] = sum_j U_s[i][j] conj(w_n[j])
        Cd a = cmake<double>(0.0, 0.0);
        for (int j = 0; j < M; ++j) cfma(a, Ul[i * M + j], cconj(wn[j]));
        tv[i] = a;
      }
      wave_sync_lds();
      double qx = 0.0, qy = 0.0, dx = 0.0;
      for (int i = lane; i < M; i += WAVE) {
        const Cd tq = cmul(Wl[s * M + i], tv[i]);
        const Cd td = cmul(wn[i], tv[i]);
        qx += tq.x;
        qy += tq.y;
        dx += td.x;  // Hermitian form: real
      }
      const double q_re = wave_sum_d(qx), q_im = wave_sum_d(qy), d = wave_sum_d(dx);
      if (lane == 0) vv[s] = (s == n) ? cmake<double>(1.0 - 1.0 / sqrt(n_frames * d), 0.0) : cmake<double>(q_re / d, q_im / d);
      wave_sync_lds();
    }
    for (int e = lane; e < MM; e += WAVE) {  // every row uses the OLD row n
      const int i = e / M, j = e - i * M;
      const Cd dlt = cmul(vv[i], wn[j]);
      Wl[e] = cmake<double>(Wl[e].x - dlt.x, Wl[e].y - dlt.y);
    }
    wave_sync_lds();
  }
  for (int e = lane; e < MM; e += WAVE) W[(size_t)bf * MM + e] = cmake<R>((R)Wl[e].x, (R)Wl[e].y);
  if (pw) {  // per-bin share of mean|y_n|^2 = mean_f w_n^H C_f w_n
    for (int e = lane; e < MM; e += WAVE) {
      const Cx<R> v = C[(size_t)bf * MM + e];
      Ul[e] = cmake<double>((double)v.x, (double)v.y);
    }
    wave_sync_lds();
    for (int n = 0; n < N; ++n) {
      double sacc = 0.0;
      for (int e = lane; e < MM; e += WAVE) {
        const int i = e / M, j = e - i * M;
        const Cd t1 = cmul(Wl[n * M + i], Ul[e]);
        sacc += t1.x * Wl[n * M + j].x + t1.y * Wl[n * M + j].y;
      }
      sacc = wave_sum_d(sacc);
      if (lane == 0) pw[((size_t)b * N + n) * F + f] = sacc;
    }
  }
}
inline size_t iss_rt_lds_bytes(int M) { return ((size_t)2 * M * M + 3 * M) * sizeof(Cd); }

// IP2 / pairwise update of rows (pm, pn) (ilrma.py:566-633, iva.py:544-599; round 6) of one bin per wave, run-time M: the
// arithmetic of ip2_group_kernel (assx_group_linalg.hpp) with the matrices in LDS -- P_x = (W U_x)^{-1} [e_pm e_pn] by
// the pivoted Gauss-Jordan inverse above, the condition guard of ip_rt_kernel, V_x = P_x^H U_x P_x (2 x 2), the eigenvectors
// of V_pn^{-1} V_pm in LAPACK zgeev's convention sorted by descending eigenvalue, w_x = conj(P_x v_x / sqrt(v_x^H V_x v_x)).
// Both rows use the OLD W.  The 2 x 2 part is wave-uniform and evaluated by every lane.  FUNCTIONAL, not tuned.
template <typename R>
__global__ void __launch_bounds__(64) ip2_rt_kernel(const Cx<R>* __restrict__ U, Cx<R>* __restrict__ W,
                                                   const Cx<R>* __restrict__ C, double* __restrict__ pw, double thr,
                                                   int32_t* __restrict__ status, int B, int F, int pm, int pn, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rt[];
  const int MM = M * M, lane = threadIdx.x, N = M;
  Cd* Wl = reinterpret_cast<Cd*>(smem_rt);
  Cd* Ul = Wl + MM;
  Cd* Al = Ul + MM;
  Cd* Ix[2] = {Al + MM, Al + 2 * MM};
  Cd* G0 = Al + 3 * MM;
  Cd* G1 = G0 + MM;
  Cd* G2 = G1 + MM;
  int* piv = reinterpret_cast<int*>(G2 + MM);
  const int bf = blockIdx.x, b = bf / F, f = bf - b * F;
  for (int e = lane; e < MM; e += WAVE) {
    const Cx<R> v = W[(size_t)bf * MM + e];
    Wl[e] = cmake<double>((double)v.x, (double)v.y);
  }
  int flags = 0;
  const int col[2] = {pm, pn};
  bool okx[2];
  Cd V[2][2][2];  // V[x][a][b], x = 0 -> source pm, 1 -> source pn
  for (int x = 0; x < 2; ++x) {
    for (int e = lane; e < MM; e += WAVE) {
      const Cx<R> v = U[(((size_t)b * N + col[x]) * F + f) * MM + e];
      Ul[e] = cmake<double>((double)v.x, (double)v.y);
    }
    wave_sync_lds();
    double nA2p = 0.0;
    for (int e = lane; e < MM; e += WAVE) {  // A = W U_x
      const int i = e / M, j = e - i * M;
      Cd a = cmake<double>(0.0, 0.0);
      for (int k = 0; k < M; ++k) cfma(a, Wl[i * M + k], Ul[k * M + j]);
      Al[e] = a;
      Ix[x][e] = a;
      nA2p += cabs2(a);
    }
    wave_sync_lds();
    const bool singular = !wave_gj_inverse(Ix[x], piv, M, lane);
    double nI2p = 0.0;
    for (int e = lane; e < MM; e += WAVE) nI2p += cabs2(Ix[x][e]);
    const double nA2 = wave_sum_d(nA2p), nI2 = wave_sum_d(nI2p);
    double c2 = nA2 * nI2, thr2 = thr * thr, m2 = (double)M * (double)M;
    if (!(c2 > 1e-290 && c2 < 1e290 && thr2 < 1e290)) {
      c2 = sqrt(nA2) * sqrt(nI2);
      thr2 = thr;
      m2 = (double)M;
    }
    const bool amb = !singular && (c2 == c2) && c2 >= thr2 && c2 < thr2 * m2;
    bool ok = !singular && (c2 == c2) && c2 < thr2;
    if (amb) ok = wave_spectral_norm(Al, G0, G1, G2, M, lane) * wave_spectral_norm(Ix[x], G0, G1, G2, M, lane) < thr;
    okx[x] = ok;
    if (singular) flags |= ASSX_STATUS_SINGULAR;  // numpy.linalg.inv raises
    else if (!ok) flags |= ASSX_STATUS_COND_REJECT;
    for (int aa = 0; aa < 2; ++aa)
      for (int bb = 0; bb < 2; ++bb) {
        double sx = 0.0, sy = 0.0;
        for (int e = lane; e < MM; e += WAVE) {
          const int i = e / M, j = e - i * M;
          const Cd term = cmul(cmul(cconj(Ix[x][i * M + col[aa]]), Ul[e]), Ix[x][j * M + col[bb]]);
          sx += term.x;
          sy += term.y;
        }
        V[x][aa][bb] = cmake<double>(wave_sum_d(sx), wave_sum_d(sy));
      }
    wave_sync_lds();
  }
  // ---- wave-uniform 2 x 2 part (the statements of ip2_group_kernel)
  const Cd detn = csub(cmul(V[1][0][0], V[1][1][1]), cmul(V[1][0][1], V[1][1][0]));
  if (detn.x == 0.0 && detn.y == 0.0) flags |= ASSX_STATUS_SINGULAR;
  const Cd idet = cdiv(cmake<double>(1.0, 0.0), detn);
  const Cd ni[2][2] = {{cmul(V[1][1][1], idet), cmul(cmake<double>(-V[1][0][1].x, -V[1][0][1].y), idet)},
                       {cmul(cmake<double>(-V[1][1][0].x, -V[1][1][0].y), idet), cmul(V[1][0][0], idet)}};
  Cd VV[2][2];
  for (int aa = 0; aa < 2; ++aa)
    for (int bb = 0; bb < 2; ++bb) VV[aa][bb] = cadd(cmul(ni[aa][0], V[0][0][bb]), cmul(ni[aa][1], V[0][1][bb]));
  const Cd htr = cscale(cadd(VV[0][0], VV[1][1]), 0.5);
  const Cd det = csub(cmul(VV[0][0], VV[1][1]), cmul(VV[0][1], VV[1][0]));
  const Cd disc = csqrt_principal(csub(cmul(htr, htr), det));
  Cd lam[2] = {cadd(htr, disc), csub(htr, disc)};
  const bool first_big = (lam[0].x > lam[1].x) || (lam[0].x == lam[1].x && lam[0].y >= lam[1].y);
  if (!first_big) {
    const Cd t = lam[0];
    lam[0] = lam[1];
    lam[1] = t;
  }
  for (int x = 0; x < 2; ++x) {  // x = 0: eigenvector of the larger eigenvalue -> row pm; x = 1 -> row pn
    const Cd c1[2] = {VV[0][1], csub(lam[x], VV[0][0])};
    const Cd c2v[2] = {csub(lam[x], VV[1][1]), VV[1][0]};
    const double n1 = cabs2(c1[0]) + cabs2(c1[1]), n2 = cabs2(c2v[0]) + cabs2(c2v[1]);
    Cd v[2] = {n1 >= n2 ? c1[0] : c2v[0], n1 >= n2 ? c1[1] : c2v[1]};
    const double nrm = sqrt(n1 >= n2 ? n1 : n2);
    v[0] = cscale(v[0], 1.0 / nrm);
    v[1] = cscale(v[1], 1.0 / nrm);
    const int kbig = (cabs2(v[1]) > cabs2(v[0])) ? 1 : 0;  // zgeev: the component of largest modulus real
    const double mag = sqrt(cabs2(v[kbig]));
    const Cd rot = cscale(cconj(v[kbig]), 1.0 / mag);
    v[0] = cmul(v[0], rot);
    v[1] = cmul(v[1], rot);
    v[kbig].y = 0.0;
    Cd q = cmake<double>(0.0, 0.0);
    for (int aa = 0; aa < 2; ++aa)
      for (int bb = 0; bb < 2; ++bb) cfma(q, cmul(cconj(v[aa]), V[x][aa][bb]), v[bb]);
    const Cd den = csqrt_principal(q);
    v[0] = cdiv(v[0], den);
    v[1] = cdiv(v[1], den);
    if (okx[x] && !(flags & ASSX_STATUS_SINGULAR))  // w_x[c] = conj(P_x[c][0] v0 + P_x[c][1] v1)
      for (int c = lane; c < M; c += WAVE)
        Wl[col[x] * M + c] = cconj(cadd(cmul(Ix[x][c * M + pm], v[0]), cmul(Ix[x][c * M + pn], v[1])));
  }
  wave_sync_lds();
  for (int e = lane; e < MM; e += WAVE) W[(size_t)bf * MM + e] = cmake<R>((R)Wl[e].x, (R)Wl[e].y);
  if (pw) {
    for (int e = lane; e < MM; e += WAVE) {
      const Cx<R> v = C[(size_t)bf * MM + e];
      Ul[e] = cmake<double>((double)v.x, (double)v.y);
    }
    wave_sync_lds();
    for (int n = 0; n < N; ++n) {
      double sacc = 0.0;
      for (int e = lane; e < MM; e += WAVE) {
        const int i = e / M, j = e - i * M;
        const Cd t1 = cmul(Wl[n * M + i], Ul[e]);
        sacc += t1.x * Wl[n * M + j].x + t1.y * Wl[n * M + j].y;
      }
      sacc = wave_sum_d(sacc);
      if (lane == 0) pw[((size_t)b * N + n) * F + f] = sacc;
    }
  }
  if (flags && status && lane == 0) atomicOr(&status[b], flags);
}
inline size_t ip2_rt_lds_bytes(int M) { return (size_t)8 * M * M * sizeof(Cd) + (size_t)M * sizeof(int); }

inline size_t ip_rt_lds_bytes(int M) { return (size_t)7 * M * M * sizeof(Cd) + (size_t)M * sizeof(int); }

// -2 T log|det W_f| per bin (the loss's last term), run-time M: LU with partial pivoting by one wave in LDS
template <typename R>
__global__ void __launch_bounds__(64) logdet_rt_kernel(const Cx<R>* __restrict__ W, double* __restrict__ lpart, int B, int F,
                                                      int T, int lstride, int offset, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rt[];
  Cd* A = reinterpret_cast<Cd*>(smem_rt);
  const int bf = blockIdx.x, b = bf / F, f = bf - b * F, lane = threadIdx.x, MM = M * M;
  for (int e = lane; e < MM; e += WAVE) {
    const Cx<R> v = W[(size_t)bf * MM + e];
    A[e] = cmake<double>((double)v.x, (double)v.y);
  }
  wave_sync_lds();
  double logabs = 0.0;
  for (int c = 0; c < M; ++c) {
    double best = -1.0;
    int p = c;
    for (int r = c + lane; r < M; r += WAVE) {
      const double v = cabs1(A[r * M + c]);
      if (v > best) {
        best = v;
        p = r;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ob = __shfl_xor(best, off, WAVE);
      const int op = __shfl_xor(p, off, WAVE);
      if (ob > best || (ob == best && op < p)) {
        best = ob;
        p = op;
      }
    }
    if (p != c)
      for (int j = lane; j < M; j += WAVE) cswap(A[c * M + j], A[p * M + j]);
    wave_sync_lds();
    const Cd pv = A[c * M + c];
    logabs += 0.5 * log(cabs2(pv));  // log 0 = -inf for a singular bin, as numpy.linalg.det -> log gives
    for (int e = lane; e < (M - c - 1) * (M - c - 1); e += WAVE) {
      const int r = c + 1 + e / (M - c - 1), j = c + 1 + e % (M - c - 1);
      const Cd fct = cdiv(A[r * M + c], pv);
      const Cd a = A[c * M + j];
      Cd v = A[r * M + j];
      v.x = v.x - (fct.x * a.x - fct.y * a.y);
      v.y = v.y - (fct.x * a.y + fct.y * a.x);
      A[r * M + j] = v;
    }
    wave_sync_lds();
  }
  if (lane == 0) lpart[(size_t)b * lstride + offset + f] = -2.0 * (double)T * logabs;
}

}  // namespace widem
}  // namespace assx
