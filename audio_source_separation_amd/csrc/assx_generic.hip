// Entry points whose channel count is a small run-time number (2 <= M <= 8) and that are NOT on the per-iteration
// path: the least-squares demixing filter of ILRMAbase/IVAbase.compute_demix_filter
// (ref src/bss/ilrma.py:167-173, src/bss/iva.py:119-125).
#include "assx_small_linalg.hpp"

namespace assx {

// In-place inverse of a dense M x M complex matrix held in LDS/scratch (row-major, leading dimension M), Gauss-Jordan
// with partial row pivoting -- LAPACK's zgetrf pivot rule (|re| + |im|), which is what numpy.linalg.inv runs.
// One thread; returns false on an exactly zero pivot (numpy raises LinAlgError("Singular matrix")).
__device__ inline bool gj_inverse_rt(Cd* A, int M) {
  int piv[8];
  bool ok = true;
  for (int c = 0; c < M; ++c) {
    int p = c;
    double best = cabs1(A[c * M + c]);
    for (int r = c + 1; r < M; ++r) {
      double v = cabs1(A[r * M + c]);
      if (v > best) {
        best = v;
        p = r;
      }
    }
    piv[c] = p;
    if (!(best > 0.0)) ok = false;
    if (p != c)
      for (int j = 0; j < M; ++j) cswap(A[c * M + j], A[p * M + j]);
    // eliminate with the UNSCALED pivot row and the factor l = A[r][c] / pivot (the LU multiplier): an exactly
    // dependent row then cancels to exact zeros and the next pivot search reports the singularity, as zgetrf does
    const Cd pv = A[c * M + c];
    for (int r = 0; r < M; ++r) {
      if (r == c) continue;
      const Cd f = cdiv(A[r * M + c], pv);
      for (int j = 0; j < M; ++j) {
        if (j == c) continue;
        Cd a = A[c * M + j], v = A[r * M + j];
        v.x = v.x - (f.x * a.x - f.y * a.y);
        v.y = v.y - (f.x * a.y + f.y * a.x);
        A[r * M + j] = v;
      }
      A[r * M + c] = cmake<double>(-f.x, -f.y);
    }
    const Cd ipv = cdiv(cmake<double>(1.0, 0.0), pv);
    for (int j = 0; j < M; ++j) A[c * M + j] = (j == c) ? ipv : cmul(A[c * M + j], ipv);
  }
  for (int c = M - 1; c >= 0; --c) {
    int p = piv[c];
    if (p != c)
      for (int i = 0; i < M; ++i) cswap(A[i * M + c], A[i * M + p]);
  }
  return ok;
}

// One workgroup (4 waves) per (utterance, bin).  The 2M rows of the stacked matrix [Y; X] are dealt to the waves
// round-robin; a wave accumulates  S[r][j] = sum_t s_r(t) conj(x_j(t))  for its rows over all frames (lanes own
// frames, coalesced 64-frame row segments), in float64 whatever the storage type.  Rows 0..M-1 of S are Y X^H, rows
// M..2M-1 are X X^H.  Then W = (Y X^H) (X X^H)^{-1}.
template <typename R, int M>
__global__ __launch_bounds__(256) void lsq_demix_kernel(const Cx<R>* __restrict__ Y, const Cx<R>* __restrict__ X,
                                                        Cx<R>* __restrict__ W, int32_t* __restrict__ status, int F,
                                                        int T) {
  constexpr int RW = (2 * M + 3) / 4;  // rows per wave
  __shared__ Cd S[2 * M][M];
  const int f = blockIdx.x, b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t plane = (size_t)F * T;
  const Cx<R>* Xb = X + (size_t)b * M * plane + (size_t)f * T;
  const Cx<R>* Yb = Y + (size_t)b * M * plane + (size_t)f * T;

  Cd acc[RW][M];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) acc[i][j] = cmake<double>(0.0, 0.0);

  for (int t = lane; t < T; t += WAVE) {
    Cd x[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
      Cx<R> v = Xb[(size_t)j * plane + t];
      x[j] = cmake<double>((double)v.x, (double)v.y);
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int r = wave + 4 * i;
      if (r < 2 * M) {
        Cd s;
        if (r < M) {
          Cx<R> v = Yb[(size_t)r * plane + t];
          s = cmake<double>((double)v.x, (double)v.y);
        } else {
          s = x[0];
#pragma unroll
          for (int j = 1; j < M; ++j)
            if (r - M == j) s = x[j];
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {  // acc += s conj(x_j)
          // the products are rounded separately (no fma) so that s conj(x_j) is EXACTLY Hermitian-symmetric for
          // rows of X: identical channels then give an exactly singular X X^H, as they do in the reference's zgemm
          acc[i][j].x += __dmul_rn(s.x, x[j].x) + __dmul_rn(s.y, x[j].y);
          acc[i][j].y += __dmul_rn(s.y, x[j].x) - __dmul_rn(s.x, x[j].y);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int r = wave + 4 * i;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      double re = wave_allreduce_sum(acc[i][j].x), im = wave_allreduce_sum(acc[i][j].y);
      if (lane == 0 && r < 2 * M) S[r][j] = cmake<double>(re, im);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bool ok = gj_inverse_rt(&S[M][0], M);
    if (!ok && status) atomicOr(&status[b], ASSX_STATUS_SINGULAR);
  }
  __syncthreads();
  if (threadIdx.x < M * M) {
    const int n = threadIdx.x / M, m = threadIdx.x % M;
    Cd w = cmake<double>(0.0, 0.0);
    for (int k = 0; k < M; ++k) cfma(w, S[n][k], S[M + k][m]);
    W[(((size_t)b * F + f) * M + n) * M + m] = cmake<R>((R)w.x, (R)w.y);
  }
}

template <typename R>
static int launch_lsq(assx_ctx* ctx, int M, const void* Y, const void* X, void* W, int32_t* status, int B, int F, int T,
                      hipStream_t st) {
  dim3 grid(F, B);
  switch (M) {
#define ASSX_LSQ_CASE(MM)                                                                                       \
  case MM:                                                                                                      \
    hipLaunchKernelGGL((lsq_demix_kernel<R, MM>), grid, dim3(256), 0, st, (const Cx<R>*)Y, (const Cx<R>*)X,     \
                       (Cx<R>*)W, status, F, T);                                                                \
    break;
    ASSX_LSQ_CASE(2)
    ASSX_LSQ_CASE(3)
    ASSX_LSQ_CASE(4)
    ASSX_LSQ_CASE(5)
    ASSX_LSQ_CASE(6)
    ASSX_LSQ_CASE(7)
    ASSX_LSQ_CASE(8)
#undef ASSX_LSQ_CASE
    default:
      return fail(ctx, ASSX_E_UNSUPPORTED, "assx_compute_demix_filter: 2 <= M <= 8 required, got %d", M);
  }
  ASSX_LAUNCH_CHECK(ctx, "lsq_demix_kernel");
  return 0;
}

}  // namespace assx

using namespace assx;

extern "C" int assx_compute_demix_filter(assx_ctx* ctx, const void* Y, const void* X, void* W, int32_t* status, int B,
                                         int M, int F, int T, int dtype, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, B >= 1 && F >= 1 && T >= 1, ASSX_E_ARG, "invalid sizes B=%d F=%d T=%d", B, F, T);
  ASSX_REQUIRE(ctx, Y && X && W, ASSX_E_NULL, "assx_compute_demix_filter: NULL array");
  ASSX_REQUIRE(ctx, dtype == ASSX_F32 || dtype == ASSX_F64, ASSX_E_ARG, "bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ASSX_F64) return launch_lsq<double>(ctx, M, Y, X, W, status, B, F, T, st);
  return launch_lsq<float>(ctx, M, Y, X, W, status, B, F, T, st);
}
