// Per-bin least squares (A B^H)(B B^H)^{-1} for a small run-time channel count (2 <= M <= 8), NOT on the per-iteration
// path: the demixing filter of ILRMAbase/IVAbase.compute_demix_filter (ref src/bss/ilrma.py:167-173,
// src/bss/iva.py:119-125) and, for M > 4, the projection-back scale (ref src/algorithm/projection_back.py:13-21).
#include "assx_small_linalg.hpp"
#include "assx_widem.hpp"

namespace assx {

// In-place inverse of a dense M x M complex matrix held in LDS/scratch (row-major, leading dimension M), Gauss-Jordan
// with partial row pivoting -- LAPACK's zgetrf pivot rule (|re| + |im|), which is what numpy.linalg.inv runs.
// One thread; returns false on an exactly zero pivot (numpy raises LinAlgError("Singular matrix")).
__device__ inline bool gj_inverse_rt(Cd* A, int M) {
  int piv[widem::RT_MMAX];
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

// One workgroup (4 waves) per (utterance, bin).  The na + M rows of the stacked matrix [A; B] are dealt to the waves
// round-robin; a wave accumulates  S[r][j] = sum_t s_r(t) conj(b_j(t))  for its rows over all frames (lanes own
// frames, coalesced 64-frame row segments), in float64 whatever the storage type.  Rows 0..na-1 of S are A B^H, the
// rest is B B^H.  Then out = (A B^H) (B B^H)^{-1}.
template <typename R, int M>
__global__ __launch_bounds__(256) void stack_gram_solve_kernel(const Cx<R>* __restrict__ A, size_t a_bstride, int na,
                                                               const Cx<R>* __restrict__ Bm, Cx<R>* __restrict__ out,
                                                               size_t ob, size_t of, size_t oi, size_t oj,
                                                               int32_t* __restrict__ status, int F, int T) {
  constexpr int RW = (8 + M + 3) / 4;  // rows per wave (na <= 8)
  __shared__ Cd S[8 + M][M];
  const int f = blockIdx.x, b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t plane = (size_t)F * T;
  const Cx<R>* Xb = Bm + (size_t)b * M * plane + (size_t)f * T;
  const Cx<R>* Ab = A + (size_t)b * a_bstride + (size_t)f * T;
  const int rows = na + M;

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
      if (r < rows) {
        Cd s;
        if (r < na) {
          Cx<R> v = Ab[(size_t)r * plane + t];
          s = cmake<double>((double)v.x, (double)v.y);
        } else {
          s = x[0];
#pragma unroll
          for (int j = 1; j < M; ++j)
            if (r - na == j) s = x[j];
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {  // acc += s conj(x_j)
          // the products are rounded separately (no fma) so that s conj(x_j) is EXACTLY Hermitian-symmetric for
          // rows of B: identical channels then give an exactly singular B B^H, as they do in the reference's zgemm
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
      if (lane == 0 && r < rows) S[r][j] = cmake<double>(re, im);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bool ok = gj_inverse_rt(&S[na][0], M);
    if (!ok && status) atomicOr(&status[b], ASSX_STATUS_SINGULAR);
  }
  __syncthreads();
  if ((int)threadIdx.x < na * M) {
    const int n = threadIdx.x / M, m = threadIdx.x % M;
    Cd w = cmake<double>(0.0, 0.0);
    for (int k = 0; k < M; ++k) cfma(w, S[n][k], S[na + k][m]);
    out[(size_t)b * ob + (size_t)f * of + (size_t)n * oi + (size_t)m * oj] = cmake<R>((R)w.x, (R)w.y);
  }
}

// The same for a run-time channel count (9 <= M <= 32; assx_widem_rt.hpp's path): S in dynamic LDS, a wave takes the rows
// of [A; B] round-robin and, per row, the columns one after the other (lanes own frames, float64 sums).
template <typename R>
__global__ __launch_bounds__(256) void stack_gram_solve_rt_kernel(const Cx<R>* __restrict__ A, size_t a_bstride, int na,
                                                                  const Cx<R>* __restrict__ Bm, Cx<R>* __restrict__ out,
                                                                  size_t ob, size_t of, size_t oi, size_t oj,
                                                                  int32_t* __restrict__ status, int F, int T, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_sgs[];
  Cd* S = reinterpret_cast<Cd*>(smem_sgs);  // [(na + M)][M]
  const int f = blockIdx.x, b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t plane = (size_t)F * T;
  const Cx<R>* Xb = Bm + (size_t)b * M * plane + (size_t)f * T;
  const Cx<R>* Ab = A + (size_t)b * a_bstride + (size_t)f * T;
  const int rows = na + M;
  for (int r = wave; r < rows; r += 4) {
    const Cx<R>* srow = r < na ? Ab + (size_t)r * plane : Xb + (size_t)(r - na) * plane;
    for (int j = 0; j < M; ++j) {
      double re = 0.0, im = 0.0;
      for (int t = lane; t < T; t += WAVE) {
        const Cx<R> sv = srow[t], xv = Xb[(size_t)j * plane + t];
        const double sx = (double)sv.x, sy = (double)sv.y, xx = (double)xv.x, xy = (double)xv.y;
        re += __dmul_rn(sx, xx) + __dmul_rn(sy, xy);  // separately rounded products: exactly Hermitian for rows of B
        im += __dmul_rn(sy, xx) - __dmul_rn(sx, xy);
      }
      re = wave_allreduce_sum(re);
      im = wave_allreduce_sum(im);
      if (lane == 0) S[r * M + j] = cmake<double>(re, im);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bool ok = gj_inverse_rt(S + (size_t)na * M, M);
    if (!ok && status) atomicOr(&status[b], ASSX_STATUS_SINGULAR);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < na * M; e += blockDim.x) {
    const int n = e / M, m = e % M;
    Cd w = cmake<double>(0.0, 0.0);
    for (int k = 0; k < M; ++k) cfma(w, S[n * M + k], S[(na + k) * M + m]);
    out[(size_t)b * ob + (size_t)f * of + (size_t)n * oi + (size_t)m * oj] = cmake<R>((R)w.x, (R)w.y);
  }
}

template <typename R>
static int launch_sgs(assx_ctx* ctx, const void* A, size_t a_bstride, int na, const void* Bm, int M, void* out, size_t ob,
                      size_t of, size_t oi, size_t oj, int32_t* status, int B, int F, int T, hipStream_t st) {
  dim3 grid(F, B);
  switch (M) {
#define ASSX_SGS_CASE(MM)                                                                                          \
  case MM:                                                                                                         \
    hipLaunchKernelGGL((stack_gram_solve_kernel<R, MM>), grid, dim3(256), 0, st, (const Cx<R>*)A, a_bstride, na,    \
                       (const Cx<R>*)Bm, (Cx<R>*)out, ob, of, oi, oj, status, F, T);                               \
    break;
    ASSX_SGS_CASE(2)
    ASSX_SGS_CASE(3)
    ASSX_SGS_CASE(4)
    ASSX_SGS_CASE(5)
    ASSX_SGS_CASE(6)
    ASSX_SGS_CASE(7)
    ASSX_SGS_CASE(8)
#undef ASSX_SGS_CASE
    default:
      if (M > 8 && M <= widem::RT_MMAX) {
        hipLaunchKernelGGL((stack_gram_solve_rt_kernel<R>), grid, dim3(256), (size_t)(na + M) * M * sizeof(Cd), st,
                           (const Cx<R>*)A, a_bstride, na, (const Cx<R>*)Bm, (Cx<R>*)out, ob, of, oi, oj, status, F, T, M);
        break;
      }
      return fail(ctx, ASSX_E_UNSUPPORTED, "2 <= M <= %d required, got %d", widem::RT_MMAX, M);
  }
  ASSX_LAUNCH_CHECK(ctx, "stack_gram_solve_kernel");
  return 0;
}

int stack_gram_solve(assx_ctx* ctx, const void* A, size_t a_batch_stride, int na, const void* Bm, int M, void* out,
                     size_t ob, size_t of, size_t oi, size_t oj, int32_t* status, int B, int F, int T, int dtype,
                     hipStream_t st) {
  if (na < 1 || na > (M > 8 ? widem::RT_MMAX : 8))
    return fail(ctx, ASSX_E_UNSUPPORTED, "stack_gram_solve: 1 <= na <= %d required, got %d", M > 8 ? widem::RT_MMAX : 8, na);
  if (dtype == ASSX_F64) return launch_sgs<double>(ctx, A, a_batch_stride, na, Bm, M, out, ob, of, oi, oj, status, B, F, T, st);
  if (dtype == ASSX_F32) return launch_sgs<float>(ctx, A, a_batch_stride, na, Bm, M, out, ob, of, oi, oj, status, B, F, T, st);
  return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
}

}  // namespace assx

using namespace assx;

// out[i] = sum_s w[s] * parts[s][i], s ascending (w == NULL: all ones).  The fixed-order combination of per-shard
// partial sums after an all-gather (F-sharded mode): a ring / tree all-reduce may associate differently per rank.
template <typename R>
__global__ void __launch_bounds__(256) ordered_sum_kernel(const R* __restrict__ parts, const double* __restrict__ w,
                                                         R* __restrict__ out, int S, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  R acc = w ? (R)w[0] * parts[i] : parts[i];
  for (int s = 1; s < S; ++s) acc += w ? (R)w[s] * parts[(size_t)s * count + i] : parts[(size_t)s * count + i];
  out[i] = acc;
}

extern "C" int assx_ordered_sum(assx_ctx* ctx, const void* parts, const double* weights, void* out, int S,
                                long long count, int dtype, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, S >= 1 && count >= 1, ASSX_E_ARG, "invalid sizes S=%d count=%lld", S, count);
  ASSX_REQUIRE(ctx, parts && out, ASSX_E_NULL, "assx_ordered_sum: NULL array");
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = (unsigned)(((size_t)count + 255) / 256);
  if (dtype == ASSX_F64)
    hipLaunchKernelGGL((ordered_sum_kernel<double>), dim3(nb), dim3(256), 0, st, (const double*)parts, weights,
                       (double*)out, S, (size_t)count);
  else if (dtype == ASSX_F32)
    hipLaunchKernelGGL((ordered_sum_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)parts, weights,
                       (float*)out, S, (size_t)count);
  else
    return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
  ASSX_LAUNCH_CHECK(ctx, "ordered_sum_kernel");
  return 0;
}

extern "C" int assx_compute_demix_filter(assx_ctx* ctx, const void* Y, const void* X, void* W, int32_t* status, int B,
                                         int M, int F, int T, int dtype, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, B >= 1 && F >= 1 && T >= 1, ASSX_E_ARG, "invalid sizes B=%d F=%d T=%d", B, F, T);
  ASSX_REQUIRE(ctx, Y && X && W, ASSX_E_NULL, "assx_compute_demix_filter: NULL array");
  ASSX_REQUIRE(ctx, M >= 2 && M <= widem::RT_MMAX, ASSX_E_UNSUPPORTED, "assx_compute_demix_filter: 2 <= M <= %d required, got %d",
               widem::RT_MMAX, M);
  const size_t plane = (size_t)F * T;
  return stack_gram_solve(ctx, Y, (size_t)M * plane, M, X, M, W, (size_t)F * M * M, (size_t)M * M, (size_t)M, 1, status, B,
                          F, T, dtype, (hipStream_t)stream);
}
