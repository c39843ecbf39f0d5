// Per-bin M x M complex linear algebra in registers (always float64, whatever the storage dtype:
// the 1e12 condition-number guard of the reference is meaningless in float32).
#pragma once
#include "assx_common.hpp"

namespace assx {

using Cd = Cx<double>;

__device__ __forceinline__ double cabs1(Cd a) { return fabs(a.x) + fabs(a.y); }  // LAPACK izamax metric

__device__ __forceinline__ void cswap(Cd& a, Cd& b) {
  Cd t = a;
  a = b;
  b = t;
}

// In-place inverse by Gauss-Jordan elimination with partial (row) pivoting -- the same pivoting
// rule as the LAPACK zgesv behind numpy.linalg.solve/inv (src/bss/ilrma.py:523,
// src/algorithm/projection_back.py:19).  Fully unrolled: every index is static, so A lives in VGPRs.
// Returns false when an exactly zero pivot is met (LAPACK info > 0 -> numpy raises LinAlgError).
// det (optional) receives det(A).
template <int M>
__device__ __forceinline__ bool gj_inverse(Cd (&A)[M][M], Cd* det_out) {
  int piv[M];
  bool ok = true;
  Cd det = cmake<double>(1.0, 0.0);
#pragma unroll
  for (int c = 0; c < M; ++c) {
    int p = c;
    double best = cabs1(A[c][c]);
#pragma unroll
    for (int r = c + 1; r < M; ++r) {
      double v = cabs1(A[r][c]);
      if (v > best) {
        best = v;
        p = r;
      }
    }
    piv[c] = p;
    if (!(best > 0.0)) ok = false;
#pragma unroll
    for (int r = c + 1; r < M; ++r) {
      if (p == r) {
#pragma unroll
        for (int j = 0; j < M; ++j) cswap(A[c][j], A[r][j]);
      }
    }
    if (p != c) det = cmake<double>(-det.x, -det.y);
    Cd pv = A[c][c];
    det = cmul(det, pv);
    Cd ipv = cdiv(cmake<double>(1.0, 0.0), pv);
    A[c][c] = cmake<double>(1.0, 0.0);
#pragma unroll
    for (int j = 0; j < M; ++j) A[c][j] = cmul(A[c][j], ipv);
#pragma unroll
    for (int r = 0; r < M; ++r) {
      if (r != c) {
        Cd f = A[r][c];
        A[r][c] = cmake<double>(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < M; ++j) {
          A[r][j].x = fma(-f.x, A[c][j].x, A[r][j].x);
          A[r][j].x = fma(f.y, A[c][j].y, A[r][j].x);
          A[r][j].y = fma(-f.x, A[c][j].y, A[r][j].y);
          A[r][j].y = fma(-f.y, A[c][j].x, A[r][j].y);
        }
      }
    }
  }
  // undo the row interchanges as column interchanges, in reverse order
#pragma unroll
  for (int c = M - 1; c >= 0; --c) {
    int p = piv[c];
#pragma unroll
    for (int r = c + 1; r < M; ++r) {
      if (p == r) {
#pragma unroll
        for (int i = 0; i < M; ++i) cswap(A[i][c], A[i][r]);
      }
    }
  }
  if (det_out) *det_out = det;
  return ok;
}

// det(A) by LU with partial pivoting (numpy.linalg.det, src/bss/ilrma.py:675); A is destroyed.
template <int M>
__device__ __forceinline__ Cd lu_det(Cd (&A)[M][M]) {
  Cd det = cmake<double>(1.0, 0.0);
#pragma unroll
  for (int c = 0; c < M; ++c) {
    int p = c;
    double best = cabs1(A[c][c]);
#pragma unroll
    for (int r = c + 1; r < M; ++r) {
      double v = cabs1(A[r][c]);
      if (v > best) {
        best = v;
        p = r;
      }
    }
#pragma unroll
    for (int r = c + 1; r < M; ++r) {
      if (p == r) {
#pragma unroll
        for (int j = 0; j < M; ++j) cswap(A[c][j], A[r][j]);
      }
    }
    if (p != c) det = cmake<double>(-det.x, -det.y);
    Cd pv = A[c][c];
    det = cmul(det, pv);
    if (best > 0.0) {
      Cd ipv = cdiv(cmake<double>(1.0, 0.0), pv);
#pragma unroll
      for (int r = c + 1; r < M; ++r) {
        Cd f = cmul(A[r][c], ipv);
#pragma unroll
        for (int j = c + 1; j < M; ++j) {
          A[r][j].x = fma(-f.x, A[c][j].x, A[r][j].x);
          A[r][j].x = fma(f.y, A[c][j].y, A[r][j].x);
          A[r][j].y = fma(-f.x, A[c][j].y, A[r][j].y);
          A[r][j].y = fma(-f.y, A[c][j].x, A[r][j].y);
        }
      }
    }
  }
  return det;
}

template <int M>
__device__ __forceinline__ double frob2(const Cd (&A)[M][M]) {
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) s += cabs2(A[i][j]);
  return s;
}

// Largest singular value of a (flat, row-major) M x M complex matrix: lambda_max of the Gram matrix
// by repeated squaring with trace normalisation (rare slow path of the cond guard; lives in scratch).
__device__ __noinline__ double spectral_norm_slow(const Cd* A, int M) {
  Cd G[64], H[64], G0[64];
  double tr = 0.0;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < M; ++j) {
      Cd s = cmake<double>(0.0, 0.0);
      for (int k = 0; k < M; ++k) {  // (A^H A)[i][j] = sum_k conj(A[k][i]) A[k][j]
        Cd a = cconj(A[k * M + i]);
        cfma(s, a, A[k * M + j]);
      }
      G[i * M + j] = s;
      if (i == j) tr += s.x;
    }
  if (!(tr > 0.0) || !isfinite(tr)) return tr > 0.0 ? tr : 0.0;
  for (int i = 0; i < M * M; ++i) {
    G[i] = cscale(G[i], 1.0 / tr);
    G0[i] = G[i];
  }
  for (int it = 0; it < 24; ++it) {
    double t2 = 0.0;
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < M; ++j) {
        Cd s = cmake<double>(0.0, 0.0);
        for (int k = 0; k < M; ++k) cfma(s, G[i * M + k], G[k * M + j]);
        H[i * M + j] = s;
        if (i == j) t2 += s.x;
      }
    for (int i = 0; i < M * M; ++i) G[i] = cscale(H[i], 1.0 / t2);
  }
  // Rayleigh quotient tr(G0 Gk) / tr(Gk), tr(Gk) = 1
  double lam = 0.0;
  for (int i = 0; i < M; ++i)
    for (int k = 0; k < M; ++k) lam += G0[i * M + k].x * G[k * M + i].x - G0[i * M + k].y * G[k * M + i].y;
  return sqrt(lam * tr);
}

// cond_2(A) < thr, given A and its computed inverse.  Frobenius bounds settle all but a factor-M band
// around the threshold:  cond_2 <= ||A||_F ||A^-1||_F <= M cond_2.
template <int M>
__device__ __forceinline__ bool cond2_below(double nA2, double nI2, double thr, const Cd (&A0)[M][M],
                                            const Cd (&Ainv)[M][M]) {
  double condF = sqrt(nA2) * sqrt(nI2);
  if (!(condF == condF)) return false;  // NaN < thr is False in numpy
  if (condF < thr) return true;
  if (condF >= thr * (double)M) return false;
  Cd a[M * M], b[M * M];
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) {
      a[i * M + j] = A0[i][j];
      b[i * M + j] = Ainv[i][j];
    }
  return spectral_norm_slow(a, M) * spectral_norm_slow(b, M) < thr;
}

}  // namespace assx
