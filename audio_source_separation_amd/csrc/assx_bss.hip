// AuxIVA / Gauss-ILRMA hot-path kernels for gfx950 (CDNA4, wave64) and their C-ABI entry points.
//
// Data layout (identical to the reference's NumPy arrays, plus a leading utterance axis B):
//   X (B,M,F,T) complex, T fastest -> a wave reads 64 consecutive frames of one (m,f) row: one
//   fully coalesced 1 KiB (c128) / 512 B (c64) request.  Every kernel streams X exactly once and
//   recomputes y = W x in registers; Y is never materialised inside the iteration loop.
//
// Work decomposition (all reductions are two-stage and atomic-free => run-to-run bit-stable):
//   "reduce over t" kernels (covariance, basis update, power, loss, projection-back statistics):
//       one WAVE per (utterance, bin f, t-split); per-lane register accumulators; a butterfly
//       reduce-scatter (wave_reduce_scatter) leaves one total per lane; partials -> small finalize.
//   "reduce over f" kernels (activation update, AuxIVA r_n(t)):
//       lanes own 64 consecutive t; the 4 waves of a workgroup stride over an f-range; cross-wave
//       reduction staged through LDS; partials over f-splits -> small finalize.
//
// Reference citations: see include/assx.h next to each entry point.
#include "assx_common.hpp"
#include "assx_small_linalg.hpp"
#include "assx_stream.hpp"
#include "assx_group_linalg.hpp"
#include "assx_partition.hpp"
#include "assx_cov_wide.hpp"
#include "assx_cov_mfma.hpp"
#include "assx_nmf_internal.hpp"
#include "assx_widem.hpp"

using namespace assx;

namespace {

constexpr int REDUCE_THREADS = 256;

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
template <typename R, int M>
__device__ __forceinline__ void load_filter(const Cx<R>* __restrict__ W, size_t bf, Cx<R> (&w)[M][M]) {
  const Cx<R>* p = W + bf * (M * M);
#pragma unroll
  for (int n = 0; n < M; ++n)
#pragma unroll
    for (int m = 0; m < M; ++m) w[n][m] = p[n * M + m];
}

template <typename R, int M>
__device__ __forceinline__ void demix(const Cx<R> (&w)[M][M], const Cx<R> (&x)[M], Cx<R> (&y)[M]) {
#pragma unroll
  for (int n = 0; n < M; ++n) {
    Cx<R> s = cmake<R>(0, 0);
#pragma unroll
    for (int m = 0; m < M; ++m) cfma(s, w[n][m], x[m]);
    y[n] = s;
  }
}

// ------------------------------------------------------------------------------------------
// (a3) demix:  Y[b,n,f,t] = scale[b,n,f] * sum_m W[b,f,n,m] X[b,m,f,t]
// ------------------------------------------------------------------------------------------
template <typename R, int M>
__global__ void __launch_bounds__(256) demix_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W,
                                                   const Cx<R>* __restrict__ scale, Cx<R>* __restrict__ Y, Dims d) {
  const int f = blockIdx.y, b = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.T) return;
  const size_t FT = (size_t)d.F * d.T;
  Cx<R> w[M][M];
  load_filter<R, M>(W, (size_t)b * d.F + f, w);
  const Cx<R>* xb = X + (size_t)b * M * FT + (size_t)f * d.T + t;
  Cx<R> x[M], y[M];
#pragma unroll
  for (int m = 0; m < M; ++m) x[m] = xb[m * FT];
  demix<R, M>(w, x, y);
  Cx<R>* yb = Y + (size_t)b * M * FT + (size_t)f * d.T + t;
#pragma unroll
  for (int n = 0; n < M; ++n) {
    Cx<R> v = y[n];
    if (scale) v = cmul(v, scale[((size_t)b * M + n) * d.F + f]);
    yb[n * FT] = v;
  }
}

// ------------------------------------------------------------------------------------------
// n_basis > 4 (the reference's default is 10): the source model no longer fits the per-lane registers of the
// streaming kernels, so the two halves of the iteration go through a materialised (B,N,F,T) real array instead:
//   * P = |W x|^2, on which the source-model update IS the batched IS-NMF MM update (ilrma.py:409-430 ==
//     nmf.py:302-327 with target P) -- run on the f64/f32 matrix cores by assx_nmf_update, batch B*N;
//   * R = (T V)^(2/domain), which the covariance kernel then reads as per-bin-per-frame weights (WK_NFT).
// Both maps are one read of their input and one coalesced write.
// ------------------------------------------------------------------------------------------
template <typename R, int M>
__global__ void __launch_bounds__(256) demix_power_map_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W,
                                                             R* __restrict__ P, Dims d) {
  const int f = blockIdx.y, b = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.T) return;
  const size_t FT = (size_t)d.F * d.T;
  Cx<R> w[M][M];
  load_filter<R, M>(W, (size_t)b * d.F + f, w);
  const Cx<R>* xb = X + (size_t)b * M * FT + (size_t)f * d.T + t;
  Cx<R> x[M], y[M];
#pragma unroll
  for (int m = 0; m < M; ++m) x[m] = xb[m * FT];
  demix<R, M>(w, x, y);
  R* pb = P + (size_t)b * M * FT + (size_t)f * d.T + t;
#pragma unroll
  for (int n = 0; n < M; ++n) pb[n * FT] = cabs2(y[n]);
}

// copy back the source models of the sources selected by `mask` (pairwise updates, ilrma.py:432-481)
template <typename R>
__global__ void __launch_bounds__(256) masked_model_copy_kernel(const R* __restrict__ Tsrc, const R* __restrict__ Vsrc,
                                                               R* __restrict__ Tdst, R* __restrict__ Vdst, int B, int N,
                                                               size_t FK, size_t KT, unsigned mask) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nT = (size_t)B * N * FK, nV = (size_t)B * N * KT;
  if (idx < nT) {
    const int n = (idx / FK) % N;
    if ((mask >> n) & 1u) Tdst[idx] = Tsrc[idx];
  } else if (idx < nT + nV) {
    const size_t j = idx - nT;
    const int n = (j / KT) % N;
    if ((mask >> n) & 1u) Vdst[j] = Vsrc[j];
  }
}

// A thread owns one frame of WIDE_FB consecutive bins: the K activations of that frame are loaded once and serve
// all of them (one bin per thread is bound by the K L2 reads per output element, 90 us at K = 10 instead of the
// 30 us the 134 MB write costs); the basis rows are wave-uniform.
constexpr int WIDE_FB = 8;

template <typename R, bool D2>
__global__ void __launch_bounds__(256) source_variance_map_kernel(const R* __restrict__ Tb, const R* __restrict__ V,
                                                                 R* __restrict__ Rv, int F, int T, int K, PowSpec p2d) {
  const int f0 = blockIdx.y * WIDE_FB, bn = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const R* vb = V + (size_t)bn * K * T + t;
  const R* tb = Tb + (size_t)bn * F * K;
  int row[WIDE_FB];
#pragma unroll
  for (int j = 0; j < WIDE_FB; ++j) row[j] = (f0 + j < F ? f0 + j : F - 1) * K;
  R tv[WIDE_FB];
#pragma unroll
  for (int j = 0; j < WIDE_FB; ++j) tv[j] = 0;
  for (int k = 0; k < K; ++k) {
    const R v = vb[(size_t)k * T];
#pragma unroll
    for (int j = 0; j < WIDE_FB; ++j) tv[j] = fma(tb[row[j] + k], v, tv[j]);
  }
#pragma unroll
  for (int j = 0; j < WIDE_FB; ++j)
    if (f0 + j < F)  // floored by the reader (ilrma.py:499-509)
      Rv[((size_t)bn * F + f0 + j) * T + t] = D2 ? tv[j] : powspec<R>(tv[j], p2d);
}

// loss data term sum_{n,f,t} P/R + log R (ilrma.py:672-675) for n_basis > 4, same bin batching; one wave per
// workgroup, one partial per workgroup at lpart[b][blockIdx.y * gridDim.x + blockIdx.x]
template <typename R, int M, bool D2, bool TD = false>
__global__ void __launch_bounds__(64) ilrma_loss_wide_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W,
                                                            const R* __restrict__ Tb, const R* __restrict__ V,
                                                            double* __restrict__ lpart, int lstride, Dims d, R eps,
                                                            PowSpec p2d, R nu = 0,
                                                            R* __restrict__ P_out = nullptr /* (B,N,F,T) |W x|^2 */) {
  const int F = d.F, T = d.T, K = d.K;
  const int f0 = blockIdx.y * WIDE_FB, b = blockIdx.z;
  const int t = blockIdx.x * 64 + threadIdx.x;
  const bool live = t < T;
  const int tc = live ? t : T - 1;
  int row[WIDE_FB];
#pragma unroll
  for (int j = 0; j < WIDE_FB; ++j) row[j] = (f0 + j < F ? f0 + j : F - 1) * K;
  R r[M][WIDE_FB];
#pragma unroll
  for (int n = 0; n < M; ++n) {
    const R* vb = V + ((size_t)b * M + n) * K * T + tc;
    const R* tb = Tb + ((size_t)b * M + n) * F * K;
#pragma unroll
    for (int j = 0; j < WIDE_FB; ++j) r[n][j] = 0;
    for (int k = 0; k < K; ++k) {
      const R v = vb[(size_t)k * T];
#pragma unroll
      for (int j = 0; j < WIDE_FB; ++j) r[n][j] = fma(tb[row[j] + k], v, r[n][j]);
    }
#pragma unroll
    for (int j = 0; j < WIDE_FB; ++j) r[n][j] = floor_eps<R>(D2 ? r[n][j] : powspec<R>(r[n][j], p2d), eps);
  }
  const size_t FT = (size_t)F * T;
  double acc = 0.0, lm = 1.0, tm = 1.0;  // TD: t-ILRMA terms (ilrma.py:1001-1018), see loss_stream_kernel
  int le = 0, te = 0;
#pragma unroll
  for (int j = 0; j < WIDE_FB; ++j) {
    if (f0 + j >= F) break;  // wave-uniform
    Cx<R> w[M][M];
    load_filter<R, M>(W, (size_t)b * F + f0 + j, w);
    const Cx<R>* xb = X + (size_t)b * M * FT + (size_t)(f0 + j) * T + tc;
    Cx<R> x[M], y[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = xb[m * FT];
    demix<R, M>(w, x, y);
    if (P_out && live) {  // the source-model pass that follows needs exactly this map (demix_power_map_kernel)
#pragma unroll
      for (int n = 0; n < M; ++n) P_out[((size_t)b * M + n) * FT + (size_t)(f0 + j) * T + t] = cabs2(y[n]);
    }
    double term = 0.0, rprod = 1.0, tprod = 1.0;
#pragma unroll
    for (int n = 0; n < M; ++n) {
      if (TD) tprod *= fma((double)((R)2 * fast_rcp(nu)), (double)(cabs2(y[n]) * fast_rcp(r[n][j])), 1.0);
      else term += (double)(cabs2(y[n]) * fast_rcp(r[n][j]));
      rprod *= (double)r[n][j];
    }
    if (live) {
      acc += term;
      int e;
      lm = frexp(lm * rprod, &e);
      le += e;
      if (TD) {
        tm = frexp(tm * tprod, &e);
        te += e;
      }
    }
  }
  acc += (double)le * 0.6931471805599453 + log(lm);
  if (TD) acc += (1.0 + 0.5 * (double)nu) * ((double)te * 0.6931471805599453 + log(tm));
  acc = wave_allreduce_sum<double>(acc);
  if (threadIdx.x == 0) lpart[(size_t)b * lstride + (size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
}

// ------------------------------------------------------------------------------------------
// (a2) activation half: lanes own t; 4 waves stride over the f-split; LDS cross-wave reduce
//      part[b][fs][(n*K + k)*2 + s][t]
// ------------------------------------------------------------------------------------------
template <typename R, int NV>
__device__ __forceinline__ void block4_reduce_to_wave0(R (&acc)[NV], R* lds /* [2][NV][64] */) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int wv = threadIdx.x >> 6;
  if (wv >= 2) {
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[((wv - 2) * NV + i) * WAVE + lane] = acc[i];
  }
  __syncthreads();
  if (wv < 2) {
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] += lds[(wv * NV + i) * WAVE + lane];
  }
  __syncthreads();
  if (wv == 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[i * WAVE + lane] = acc[i];
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] += lds[i * WAVE + lane];
  }
}

// ------------------------------------------------------------------------------------------
// generic deterministic reduction: out[g] = scale * sum_l in[g][l]   (one workgroup per g)
// ------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void __launch_bounds__(REDUCE_THREADS) sum_reduce_kernel(const TI* __restrict__ in, TO* __restrict__ out,
                                                                   size_t L, double scale) {
  __shared__ double sm[REDUCE_THREADS];
  const TI* p = in + (size_t)blockIdx.x * L;
  double s = 0.0;
  for (size_t i = threadIdx.x; i < L; i += REDUCE_THREADS) s += (double)p[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = REDUCE_THREADS / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (TO)(sm[0] * scale);
}

// ------------------------------------------------------------------------------------------
// (a6) power statistic, direct pass: part[b][n][ts*F + f] = sum_t |y_n|^2
// ------------------------------------------------------------------------------------------
template <typename R, int M>
__global__ void __launch_bounds__(64) power_partial_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W,
                                                          R* __restrict__ part, Dims d, int TS, int tchunk) {
  constexpr int N = M;
  constexpr int NV = next_pow2_c(N);
  const int f = blockIdx.x / TS, ts = blockIdx.x % TS, b = blockIdx.y;
  const int lane = threadIdx.x;
  const size_t FT = (size_t)d.F * d.T;
  const Cx<R>* xb = X + (size_t)b * M * FT + (size_t)f * d.T;
  Cx<R> w[M][M];
  load_filter<R, M>(W, (size_t)b * d.F + f, w);
  R acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0;
  const int t0 = ts * tchunk, t1 = min(d.T, t0 + tchunk);
  for (int t = t0 + lane; t < t1; t += WAVE) {
    Cx<R> x[M], y[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = xb[m * FT + t];
    demix<R, M>(w, x, y);
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] += cabs2(y[n]);
  }
  R tot = wave_reduce_scatter<R, NV>(acc);
  const int i = scatter_index<NV>();
  if (scatter_leader<NV>() && i < N) part[((size_t)b * N + i) * ((size_t)TS * d.F) + (size_t)ts * d.F + f] = tot;
}

// power from the plain covariance: part[b][n][f] = Re(w_n^H-form) = sum_{m,l} W[n,m] conj(W[n,l]) C[m,l]  (x T outside)
template <typename R, int M>
__global__ void __launch_bounds__(256) power_cov_kernel(const Cx<R>* __restrict__ C, const Cx<R>* __restrict__ W,
                                                       double* __restrict__ part, int B, int F) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * F * M) return;
  const int n = idx % M;
  const int bf = idx / M;
  const int b = bf / F, f = bf % F;
  const Cx<R>* c = C + (size_t)bf * M * M;
  const Cx<R>* w = W + (size_t)bf * M * M + n * M;
  double s = 0.0;
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int l = 0; l < M; ++l) {
      // W[n,m] C[m,l] conj(W[n,l])  (real part; the sum is real because C is Hermitian)
      const double wr = w[m].x, wi = w[m].y, vr = w[l].x, vi = w[l].y, cr = c[m * M + l].x, ci = c[m * M + l].y;
      const double ar = wr * cr - wi * ci, ai = wr * ci + wi * cr;
      s += ar * vr + ai * vi;
    }
  part[((size_t)b * M + n) * F + f] = s;
}

// 'power' normalisation (ilrma.py:304-322)
template <typename R>
__global__ void __launch_bounds__(256) normalize_power_kernel(Cx<R>* __restrict__ W, R* __restrict__ Tb,
                                                             const R* __restrict__ power, int B, int M, int F, int K,
                                                             R eps, PowSpec pd /* a**domain */) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nW = (size_t)B * F * M * M, nT = (size_t)B * M * F * K;
  if (idx < nW) {
    const int n = (idx / M) % M;
    const int b = idx / ((size_t)F * M * M);
    R a = floor_eps<R>(sqrt(power[b * M + n]), eps);
    W[idx] = cmake<R>(W[idx].x / a, W[idx].y / a);
  } else if (idx < nW + nT) {
    const size_t j = idx - nW;
    const int n = (j / ((size_t)F * K)) % M;
    const int b = j / ((size_t)F * K * M);
    R a = floor_eps<R>(sqrt(power[b * M + n]), eps);
    Tb[j] = Tb[j] / powspec<R>(a, pd);
  }
}

// 'power' normalisation with the statistic still split per bin: every workgroup first reduces
// power_bins[b][n][0..F) (double, tiny, L2-resident) for ITS utterance in a fixed order, then rescales its slice
// of W / Tb.  Saves the separate reduction launch.  One utterance per blockIdx.y.
template <typename R>
__global__ void __launch_bounds__(256) normalize_power_bins_kernel(Cx<R>* __restrict__ W, R* __restrict__ Tb,
                                                                  const double* __restrict__ pbins, int M, int F,
                                                                  int K, R eps, PowSpec pd) {
  __shared__ R anorm[32];  // M <= 32 (checked by the entry point)
  const int b = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
  // this thread's element of W / Tb is requested first: its round trip overlaps the reduction instead of following it
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nW = (size_t)F * M * M, nT = (size_t)M * F * K;
  Cx<R>* w = W + (size_t)b * nW + (idx < nW ? idx : 0);
  R* t = Tb + (size_t)b * nT + ((idx >= nW && idx < nW + nT) ? idx - nW : 0);
  const Cx<R> wold = *w;
  const R told = *t;
  for (int n = wv; n < M; n += 4) {  // one wave per source: fixed-order strided sum + butterfly
    const double* p = pbins + ((size_t)b * M + n) * F;
    double s = 0.0;
    constexpr int RC = 8;  // loads of a chunk in flight together (the plain loop was one L2 round trip per 64 bins)
    for (int f0 = lane; f0 < F; f0 += WAVE * RC) {
      double v[RC];
#pragma unroll
      for (int c = 0; c < RC; ++c) v[c] = p[min(f0 + WAVE * c, F - 1)];
#pragma unroll
      for (int c = 0; c < RC; ++c)
        if (f0 + WAVE * c < F) s += v[c];
    }
    s = wave_allreduce_sum<double>(s);
    if (lane == 0) anorm[n] = floor_eps<R>(sqrt((R)(s / (double)F)), eps);
  }
  __syncthreads();
  if (idx < nW) {
    const int n = (idx / M) % M;
    const R a = anorm[n];
    *w = cmake<R>(wold.x / a, wold.y / a);
  } else if (idx < nW + nT) {
    const size_t j = idx - nW;
    const int n = j / ((size_t)F * K);
    *t = told / powspec<R>(anorm[n], pd);
  }
}

// 'projection-back' normalisation (ilrma.py:323-330)
template <typename R>
__global__ void __launch_bounds__(256) normalize_pb_kernel(Cx<R>* __restrict__ W, R* __restrict__ Tb,
                                                          const Cx<R>* __restrict__ scale, int B, int M, int F, int K,
                                                          PowSpec pd) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nW = (size_t)B * F * M * M, nT = (size_t)B * M * F * K;
  if (idx < nW) {
    const int n = (idx / M) % M;
    const int f = (idx / ((size_t)M * M)) % F;
    const int b = idx / ((size_t)F * M * M);
    W[idx] = cmul(W[idx], scale[((size_t)b * M + n) * F + f]);
  } else if (idx < nW + nT) {
    const size_t j = idx - nW;
    const int f = (j / K) % F;
    const int n = (j / ((size_t)F * K)) % M;
    const int b = j / ((size_t)F * K * M);
    const Cx<R> s = scale[((size_t)b * M + n) * F + f];
    const R mag = (R)hypot((double)s.x, (double)s.y);
    Tb[j] = Tb[j] * powspec<R>(mag, pd);
  }
}

// ------------------------------------------------------------------------------------------
// (a7) ILRMA negative log-likelihood partials: part[b][ts*F + f] (double)
// ------------------------------------------------------------------------------------------
// neg2T_logabsdet: assx_group_linalg.hpp

// ------------------------------------------------------------------------------------------
// AuxIVA: s[b,n,t] = sum_f |y_n(f,t)|^2 partials over f-splits: part[b][fs][n][t]
// ------------------------------------------------------------------------------------------
// Round 4: the finalize step (r from the slab sums, loss data term) is folded in behind a "last workgroup done" ticket
// (assx_common.hpp: take_ticket): the FS workgroups of one (utterance, frame block) publish their slabs, the holder of the
// last ticket sums them in the strand order of auxiva_stat_finalize_kernel and writes r -- one launch less per AuxIVA
// iteration (4 -> 3).  tickets == nullptr: slabs only (the separate finalize follows).
template <typename R, int M>
__global__ void __launch_bounds__(256) auxiva_stat_partial_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W,
                                                                 R* part, Dims d, int FS, int fchunk, int* tickets,
                                                                 R* __restrict__ r, double* lpart, double* loss, int kind,
                                                                 R eps, int lstride) {
  constexpr int N = M;
  __shared__ R lds[2 * N * WAVE];
  __shared__ int s_last;
  __shared__ double s_sum[REDUCE_THREADS];
  static_assert(REDUCE_THREADS == 256, "the folded loss sum reproduces sum_reduce_kernel's order with this workgroup");
  const int tb_ = blockIdx.x, fs = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int F = d.F, T = d.T;
  const size_t FT = (size_t)F * T;
  const int t = tb_ * WAVE + lane;
  const bool valid = t < T;
  const int tc = valid ? t : T - 1;
  const int f0 = fs * fchunk, f1 = min(F, f0 + fchunk);
  const Cx<R>* xb = X + (size_t)b * M * FT + tc;
  R acc[N];
#pragma unroll
  for (int n = 0; n < N; ++n) acc[n] = 0;
  for (int f = f0 + wv; f < f1; f += 4) {
    Cx<R> w[M][M];
    load_filter<R, M>(W, (size_t)b * F + f, w);
    Cx<R> x[M], y[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = xb[m * FT + (size_t)f * T];
    demix<R, M>(w, x, y);
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] += cabs2(y[n]);
  }
  block4_reduce_to_wave0<R, N>(acc, lds);
  const bool fold = tickets != nullptr;
  if (wv == 0) {
    if (valid) {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        R* q = part + (((size_t)b * FS + fs) * N + n) * T + t;
        if (fold) st_agent(q, acc[n]);
        else *q = acc[n];
      }
    }
    if (fold) {
      const bool last = take_ticket(tickets + (size_t)b * gridDim.x + tb_, FS);
      if (lane == 0) s_last = last;
    }
  }
  if (!fold) return;
  // With the loss (folded path only): the log-det terms of this workgroup's bins ride along (frame block 0's
  // workgroups write them), and a second ticket per utterance -- taken by everyone who has written loss partials --
  // lets the last writer add them up: the loss needs no launch of its own (logdet_kernel + sum_reduce_kernel before).
  const int nblk = N * (int)gridDim.x;
  int contributions = 0;
  if (loss && tb_ == 0) {
    for (int f = f0 + (int)threadIdx.x; f < f1; f += 256)
      st_agent(lpart + (size_t)b * lstride + nblk + f, neg2T_logabsdet<M, R>(W, (size_t)b * F + f, T));
    contributions = 1;
  }
  __syncthreads();
  if (s_last) {
  // ---- holder of the last ticket: r = sqrt(s) (Laplace, iva.py:490) | s / F (Gauss, iva.py:723) for the N x 64 outputs of
  // this frame block; loss data term per (source, frame block) -> lpart[b][n * blocks + tb]
  const size_t NT = (size_t)N * T;
  for (int n = wv; n < N; n += 4) {
    double term = 0.0;
    if (valid) {
      const R s = slab_sum4(part + (size_t)b * FS * NT + (size_t)n * T + t, NT, FS);
      R rv;
      if (kind == ASSX_IVA_LAPLACE) {
        rv = sqrt(s);
        term = 2.0 * (double)rv;                       // iva.py:615-617
      } else {
        rv = s / (R)F;
        term = (double)F * log((double)floor_eps<R>(rv, eps));  // iva.py:797-800
      }
      r[(size_t)b * NT + (size_t)n * T + t] = rv;
    }
    if (lpart) {
      term = wave_allreduce_sum<double>(term);
      if (lane == 0) st_agent(lpart + (size_t)b * lstride + (size_t)n * gridDim.x + tb_, term);
    }
  }
  contributions += 1;
  }
  if (!loss || contributions == 0) return;
  // every wave's partial stores have been accepted before the workgroup's contribution is counted
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wv == 0) {
    int* t2 = tickets + (size_t)gridDim.z * gridDim.x + b;
    bool last = false;
    for (int c = 0; c < contributions; ++c) last = take_ticket(t2, (int)gridDim.x + FS) || last;
    if (lane == 0) s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- last writer of the utterance: loss[b] = sum of its lstride partials, in sum_reduce_kernel's order
  {
    const double* p = lpart + (size_t)b * lstride;
    double acc2 = 0.0;
    for (int i = threadIdx.x; i < lstride; i += REDUCE_THREADS) acc2 += ld_agent(p + i);
    s_sum[threadIdx.x] = acc2;
    __syncthreads();
    for (int off = REDUCE_THREADS / 2; off >= 1; off >>= 1) {
      if ((int)threadIdx.x < off) s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) loss[b] = s_sum[0];
  }
}

// r = sqrt(s) (Laplace, iva.py:490) | s / F (Gauss, iva.py:723); loss data term per block -> lpart[b][blk]
template <typename R>
__global__ void __launch_bounds__(256) auxiva_stat_finalize_kernel(const R* __restrict__ part, R* __restrict__ r,
                                                                  double* __restrict__ lpart, int N, int F, int T,
                                                                  int FS, int kind, R eps, int lstride, int TBk) {
  // Only launched where the fold above is off (FS == 1 or ASSX_AUX_FOLD=0).  One workgroup per (source, frame block):
  // 64 outputs, 4 threads per output: thread (o, q) sums the slabs fs = q, q+4, ... (independent loads), the four
  // strands are combined in a fixed order -- the order slab_sum4 reproduces.
  __shared__ R strand[4][64];
  const int b = blockIdx.y;
  const size_t NT = (size_t)N * T;
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int n = blockIdx.x / TBk, tb = blockIdx.x % TBk;
  const int t = tb * 64 + o;
  const size_t idx = (size_t)n * T + t;
  R s = 0;
  if (t < T) {
#pragma unroll 4
    for (int fs = q; fs < FS; fs += 4) s += part[((size_t)b * FS + fs) * NT + idx];
  }
  strand[q][o] = s;
  __syncthreads();
  if (q == 0) {
    double term = 0.0;
    if (t < T) {
      s = (strand[0][o] + strand[1][o]) + (strand[2][o] + strand[3][o]);
      R rv;
      if (kind == ASSX_IVA_LAPLACE) {
        rv = sqrt(s);
        term = 2.0 * (double)rv;                       // iva.py:615-617
      } else {
        rv = s / (R)F;
        term = (double)F * log((double)floor_eps<R>(rv, eps));  // iva.py:797-800
      }
      r[(size_t)b * NT + idx] = rv;
    }
    if (lpart) {
      term = wave_allreduce_sum<double>(term);
      if (o == 0) lpart[(size_t)b * lstride + blockIdx.x] = term;
    }
  }
}

template <typename R, int M>
__global__ void __launch_bounds__(64) logdet_kernel(const Cx<R>* __restrict__ W, double* __restrict__ lpart, int B, int F,
                                                   int T, int lstride, int offset) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * F) return;
  const int b = idx / F, f = idx % F;
  lpart[(size_t)b * lstride + offset + f] = neg2T_logabsdet<M, R>(W, (size_t)idx, T);
}

// Completes the loss folded into the basis pass: loss[b] = sum of the data-term partials of the workgroups that
// touched utterance b (found from the partition arithmetic: nothing to zero beforehand) plus the F log-det terms
// logdet_kernel left at lpart[b][ncov ..).  One workgroup per utterance, fixed summation order.
__global__ void __launch_bounds__(REDUCE_THREADS) ilrma_loss_finish_kernel(const double* __restrict__ lpart,
                                                                          double* __restrict__ loss, int F, int ncov,
                                                                          int lstride) {
  __shared__ double sm[REDUCE_THREADS];
  const int b = blockIdx.x;
  double s = 0.0;
  static_assert(REDUCE_THREADS == 256, "strided_sum_256");
  strided_sum_256(lpart + (size_t)b * lstride, ncov, s);
  strided_sum_256(lpart + (size_t)b * lstride + ncov, F, s);
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = REDUCE_THREADS / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[b] = sm[0];
}

// ------------------------------------------------------------------------------------------
// (a8) projection back statistics: per (b,f): G = sum_t y y^H (packed Hermitian, N*N reals) and
//      c[j] = sum_t x_ref conj(y_j) (2N reals).  DEMIX: y = W x on the fly, x_ref = X[ref];
//      otherwise y = Y rows and x_ref = reference row.
// ------------------------------------------------------------------------------------------
template <typename R, int M, bool DEMIX>
__global__ void __launch_bounds__(64) pb_stat_partial_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W,
                                                            const Cx<R>* __restrict__ refsig, int ref,
                                                            R* __restrict__ part, Dims d, int TS, int tchunk) {
  constexpr int N = M;
  constexpr int HM = N * N;
  constexpr int NS = HM + 2 * N;
  constexpr int NV = next_pow2_c(NS);
  const int f = blockIdx.x / TS, ts = blockIdx.x % TS, b = blockIdx.y;
  const int lane = threadIdx.x;
  const size_t FT = (size_t)d.F * d.T;
  const Cx<R>* xb = X + (size_t)b * M * FT + (size_t)f * d.T;
  Cx<R> w[M][M];
  if (DEMIX) load_filter<R, M>(W, (size_t)b * d.F + f, w);
  R acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0;
  const int t0 = ts * tchunk, t1 = min(d.T, t0 + tchunk);
  for (int t = t0 + lane; t < t1; t += WAVE) {
    Cx<R> x[M], y[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = xb[m * FT + t];
    Cx<R> xr;
    if (DEMIX) {
      demix<R, M>(w, x, y);
      xr = xb[(size_t)ref * FT + t];
    } else {
#pragma unroll
      for (int n = 0; n < N; ++n) y[n] = x[n];
      xr = refsig[(size_t)b * FT + (size_t)f * d.T + t];
    }
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] += cabs2(y[n]);
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
      for (int l = m + 1; l < N; ++l) {
        Cx<R> q = cmulc(y[m], y[l]);
        acc[herm_pair_base<N>(m, l)] += q.x;
        acc[herm_pair_base<N>(m, l) + 1] += q.y;
      }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      Cx<R> q = cmulc(xr, y[j]);
      acc[HM + 2 * j] += q.x;
      acc[HM + 2 * j + 1] += q.y;
    }
  }
  R tot = wave_reduce_scatter<R, NV>(acc);
  const int i = scatter_index<NV>();
  if (scatter_leader<NV>() && i < NS) part[(((size_t)b * TS + ts) * d.F + f) * NS + i] = tot;
}

// scale[b,n,f] = (c G^{-1})[n]   (projection_back.py:18-21); one lane per (b,f), float64
template <typename R, int M>
__global__ void __launch_bounds__(64) pb_solve_kernel(const R* __restrict__ part, Cx<R>* __restrict__ scale,
                                                     int32_t* __restrict__ status, int B, int F, int TS) {
  constexpr int N = M;
  constexpr int HM = N * N;
  constexpr int NS = HM + 2 * N;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * F) return;
  const int b = idx / F, f = idx % F;
  double s[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) s[i] = 0.0;
  for (int ts = 0; ts < TS; ++ts) {
    const R* p = part + (((size_t)b * TS + ts) * F + f) * NS;
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] += (double)p[i];
  }
  Cd G[N][N];
#pragma unroll
  for (int m = 0; m < N; ++m) {
    G[m][m] = cmake<double>(s[m], 0.0);
#pragma unroll
    for (int l = m + 1; l < N; ++l) {
      const int base = herm_pair_base<N>(m, l);
      G[m][l] = cmake<double>(s[base], s[base + 1]);
      G[l][m] = cmake<double>(s[base], -s[base + 1]);
    }
  }
  const bool ok = gj_inverse<N>(G, nullptr);
  if (!ok && status) atomicOr(&status[b], (int)ASSX_STATUS_SINGULAR);
#pragma unroll
  for (int n = 0; n < N; ++n) {
    Cd a = cmake<double>(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < N; ++j) cfma(a, cmake<double>(s[HM + 2 * j], s[HM + 2 * j + 1]), G[j][n]);
    scale[((size_t)b * N + n) * F + f] = cmake<R>((R)a.x, (R)a.y);
  }
}

// ------------------------------------------------------------------------------------------
// host-side launch helpers
// ------------------------------------------------------------------------------------------

// number of t-splits for the wave-per-(b,f,ts) kernels: aim at >= ~8 waves per SIMD chip-wide
inline void t_split(int B, int F, int T, int* TS, int* tchunk) {
  static const int target = lab_int("ASSX_TARGET_WAVES", 8192);
  static const int forced = lab_int("ASSX_TS", 0);
  (void)B;  // the split is a function of ONE utterance's geometry: batched == per-utterance, bit for bit
  int ts = forced > 0 ? forced : (int)((target + (size_t)F - 1) / ((size_t)F));
  int max_ts = (T + 255) / 256;  // at least 4 frames per lane
  if (ts > max_ts) ts = max_ts;
  if (ts < 1) ts = 1;
  int chunk = (T + ts - 1) / ts;
  chunk = (chunk + WAVE - 1) / WAVE * WAVE;
  ts = (T + chunk - 1) / chunk;
  *TS = ts;
  *tchunk = chunk;
}

inline void f_split(int B, int F, int T, int* FS, int* fchunk) {
  static const int target = lab_int("ASSX_TARGET_WGS", 1024);
  static const int forced = lab_int("ASSX_FS", 0);
  const int TB = (T + WAVE - 1) / WAVE;
  (void)B;
  int fs = forced > 0 ? forced : (int)((target + (size_t)TB - 1) / ((size_t)TB));
  int max_fs = (F + 7) / 8;  // at least 2 bins per wave
  if (fs > max_fs) fs = max_fs;
  if (fs < 1) fs = 1;
  int chunk = (F + fs - 1) / fs;
  fs = (F + chunk - 1) / chunk;
  *FS = fs;
  *fchunk = chunk;
}

// t-ILRMA auxiliary weights (ilrma.py:946-958): Xi[n,f,t] = (nu max(R, eps) + 2 |y_n|^2) / (nu + 2), R = T V,
// y = W x with the filters BEFORE the sweep.  Materialised (B,N,F,T): the covariance kernel then runs in its
// "weights given" form.
template <typename R, int M>
__global__ void __launch_bounds__(256) tilrma_xi_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W,
                                                       const R* __restrict__ Tb, const R* __restrict__ V,
                                                       R* __restrict__ Xi, Dims d, R nu, R eps) {
  constexpr int N = M;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int f = blockIdx.y, b = blockIdx.z;
  if (t >= d.T) return;
  const size_t FT = (size_t)d.F * d.T;
  Cx<R> x[M];
#pragma unroll
  for (int m = 0; m < M; ++m) x[m] = X[((size_t)b * M + m) * FT + (size_t)f * d.T + t];
  const Cx<R>* wp = W + ((size_t)b * d.F + f) * (N * M);
  const R inv = fast_rcp(nu + (R)2);
#pragma unroll
  for (int n = 0; n < N; ++n) {
    Cx<R> y = cmake<R>(0, 0);
#pragma unroll
    for (int m = 0; m < M; ++m) cfma(y, wp[n * M + m], x[m]);
    const R* tbn = Tb + (((size_t)b * N + n) * d.F + f) * d.K;
    const R* vn = V + ((size_t)b * N + n) * d.K * d.T + t;
    R tv = 0;
    for (int k = 0; k < d.K; ++k) tv = fma(tbn[k], vn[(size_t)k * d.T], tv);
    tv = floor_eps<R>(tv, eps);
    Xi[((size_t)b * N + n) * FT + (size_t)f * d.T + t] = fma(nu, tv, (R)2 * cabs2(y)) * inv;
  }
}

// n_basis > 4 form of tilrma_xi_kernel: a thread owns one frame of WIDE_FB consecutive bins (the K activations it loads
// serve all of them, see source_variance_map_kernel)
template <typename R, int M>
__global__ void __launch_bounds__(64) tilrma_xi_wide_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W,
                                                           const R* __restrict__ Tb, const R* __restrict__ V,
                                                           R* __restrict__ Xi, Dims d, R nu, R eps) {
  constexpr int N = M;
  const int F = d.F, T = d.T, K = d.K;
  const int f0 = blockIdx.y * WIDE_FB, b = blockIdx.z;
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= T) return;
  int row[WIDE_FB];
#pragma unroll
  for (int j = 0; j < WIDE_FB; ++j) row[j] = (f0 + j < F ? f0 + j : F - 1) * K;
  R r[N][WIDE_FB];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const R* vb = V + ((size_t)b * N + n) * K * T + t;
    const R* tb = Tb + ((size_t)b * N + n) * F * K;
#pragma unroll
    for (int j = 0; j < WIDE_FB; ++j) r[n][j] = 0;
    for (int k = 0; k < K; ++k) {
      const R v = vb[(size_t)k * T];
#pragma unroll
      for (int j = 0; j < WIDE_FB; ++j) r[n][j] = fma(tb[row[j] + k], v, r[n][j]);
    }
  }
  const size_t FT = (size_t)F * T;
  const R inv = fast_rcp(nu + (R)2);
#pragma unroll
  for (int j = 0; j < WIDE_FB; ++j) {
    if (f0 + j >= F) break;
    Cx<R> w[M][M];
    load_filter<R, M>(W, (size_t)b * F + f0 + j, w);
    const Cx<R>* xb = X + (size_t)b * M * FT + (size_t)(f0 + j) * T + t;
    Cx<R> x[M], y[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = xb[m * FT];
    demix<R, M>(w, x, y);
#pragma unroll
    for (int n = 0; n < N; ++n)
      Xi[((size_t)b * N + n) * FT + (size_t)(f0 + j) * T + t] =
          fma(nu, floor_eps<R>(r[n][j], eps), (R)2 * cabs2(y[n])) * inv;
  }
}

// (f4) weights of the other callers of the covariance + IP kernels.
// GaussIDLMA.update_space_model (sss/idlma.py:182): R = dnn_output ** (2 / domain), elementwise (B,N,F,T).
template <typename R>
__global__ void __launch_bounds__(256) pow_map_kernel(const R* __restrict__ in, R* __restrict__ out, size_t n, PowSpec p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = powspec<R>(in[i], p);
}
// FastMultichannelISNMF.update_diagonalizer (bss/mnmf.py:868): R[f,t,m] = sum_n Lambda[n,f,t] g[n,f,m], laid out
// (B,M,F,T) so that channel m's variances are one weight set of the covariance kernel.  Sources are added in
// ascending order, as numpy.sum(axis=0) does.
template <typename R>
__global__ void __launch_bounds__(256) mix_variance_kernel(const R* __restrict__ Lambda, const R* __restrict__ g,
                                                          R* __restrict__ out, int M, int N, int F, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const size_t FT = (size_t)F * T;
  for (int m = 0; m < M; ++m) {
    R acc = 0;
    for (int n = 0; n < N; ++n)
      acc += Lambda[((size_t)b * N + n) * FT + (size_t)f * T + t] * g[(((size_t)b * N + n) * F + f) * M + m];
    out[((size_t)b * M + m) * FT + (size_t)f * T + t] = acc;
  }
}

struct WsLayout {  // carve-up of the caller's scratch; every region 256-byte aligned
  size_t part;     // reduction partials (largest user: activation update)
  size_t u;        // dense U (B,N,F,M,M) complex
  size_t lpart;    // double partials for losses / power
  size_t small;    // (B,N) reals etc.
  size_t map;      // n_basis > 4 only: (B,N,F,T) reals (demixed power / source variance)
  size_t nmf;      // n_basis > 4 only: scratch of the batched IS-NMF update
  size_t tmp;      // n_basis > 4 only: copies of (Tb, V) for a source-masked update
  size_t total;
};

// Flat partitions of the streaming kernels.  Deterministic by construction (MI355X geometry: 256 CUs; the
// workgroups-per-CU figures are what the kernels' VGPR budgets admit), so the summation order -- hence every
// result bit -- does not depend on a runtime occupancy query.  ASSX_ROUNDS scales the oversubscription.
constexpr int CHIP_CUS = 256;

inline long long g_target(int wgs_per_cu) {
  static const int rounds = lab_int("ASSX_ROUNDS", 1);
  const int forced = knob_int("ASSX_G", 0);  // read on every call: the tests shrink G to give small inputs long ranges
  if (forced > 0) return forced;
  return (long long)CHIP_CUS * wgs_per_cu * (rounds < 1 ? 1 : rounds);
}

inline int tblocks(int T) { return (T + WAVE - 1) / WAVE; }

// Depth of the X register ring of the n_basis <= 4 covariance / basis passes: 2.  Three slots (231 / 216 VGPRs, the basis
// pass giving up its two demixing rows in VGPRs for them) were measured in round 6 on one box, alternating libraries: one
// utterance 5601-5641 -> 5544-5574 (covariance only) / 5444-5537 it/s (both), eight utterances per launch 5515-5530 ->
// 5523-5538 / 5502-5532 utterance-it/s (profiles/r06_ring_depth_ab.txt): more bytes in flight buy nothing beyond the
// Infinity Cache either -- the passes are not waiting for memory.
// cov_stream_kernel: 1 wave per workgroup, 2 waves/SIMD; blocks of 64/LS frames
template <typename R, int M>
constexpr int cov_lane_split() { return 1; }
inline FlatPart flat_cov(int B, int F, int T, int LS) {
  const int fb = WAVE / LS, tbk = (T + fb - 1) / fb;
  return make_flat(B, (long long)F * tbk, tbk, g_target(8));
}
inline FlatPart flat_basis(int B, int F, int T) {  // basis_stream_kernel: 1 wave per workgroup, 2 waves/SIMD
  return make_flat(B, (long long)F * tblocks(T), tblocks(T), g_target(8));
}
inline FlatPart flat_act(int B, int F, int T) {    // act_stream_kernel: ACT_NH waves per workgroup, 2 waves/SIMD
  return make_flat(B, (long long)tblocks(T) * F, F, g_target(8 / ACT_NH));
}

// cov_wide_kernel: COVW_BINS waves per workgroup, one workgroup per CU; items of 64 * sb frames
inline FlatPart flat_cov_wide(int B, int F, int T, int sb = 1) {
  const int tbk = (T + WAVE * sb - 1) / (WAVE * sb);
  return make_flat(B, (long long)((F + COVW_BINS - 1) / COVW_BINS) * tbk, tbk, g_target(1));
}
// sub-blocks per item of cov_wide_kernel.  1 is what is measured fastest; ASSX_COVW_SB=2 (one barrier per 128
// frames, loads a whole item ahead) is kept for A/B runs: the kernel is instruction-issue bound, not latency bound
// (SQ counters), and the second X block in registers spills at float64 -- 124 us instead of 92.
template <typename R>
inline int cov_wide_sb(int NK) {
  static const int forced = lab_int("ASSX_COVW_SB", 0);
  if (forced == 2 && CovWideGeom<R, 2>::lds_bytes(NK) <= 144 * 1024) return 2;
  return 1;
}

inline FlatPart flat_loss(int F, int T) {  // loss_stream_kernel: the utterance index is grid.y
  return make_flat(1, (long long)F * tblocks(T), tblocks(T), g_target(8));
}

inline WsLayout ws_layout(int B, int M, int F, int T, int K, int dtype) {
  const size_t r = dtype == ASSX_F64 ? 8 : 4;
  const int Kc = K < 1 ? 1 : K;
  int TS, tchunk, FS, fchunk;
  t_split(B, F, T, &TS, &tchunk);
  f_split(B, F, T, &FS, &fchunk);
  const FlatPart fc = flat_cov(B, F, T, 1), fb = flat_basis(B, F, T), fa = flat_act(B, F, T);
  size_t p_cov = (size_t)fc.G * fc.S * (M * M * M);
  size_t p_basis = (size_t)fb.G * fb.S * (M * Kc * 2);
  size_t p_act = (size_t)fa.G * fa.S * (M * Kc * 2) * WAVE;
  size_t p_pb = (size_t)B * TS * F * (M * M + 2 * M);
  size_t p_pow = (size_t)B * M * TS * F;
  size_t p_aux = (size_t)B * FS * M * T;
  size_t pmax = p_cov;
  if (K > KU) {  // cov_wide_kernel records: [g][slot][COVW_BINS][N][M*M]
    for (int sb = 1; sb <= 2; ++sb) {  // either item size may be chosen at launch (cov_wide_sb)
      const FlatPart fw = flat_cov_wide(B, F, T, sb);
      const size_t p_wide = (size_t)fw.G * fw.S * COVW_BINS * (M * M * M);
      if (p_wide > pmax) pmax = p_wide;
    }
    const size_t p_adapt = (size_t)B * tblocks(T) * M * 2 * Kc * WAVE;  // part_adapt_act_kernel records
    if (p_adapt > pmax) pmax = p_adapt;
  }
  if (p_basis > pmax) pmax = p_basis;
  if (p_act > pmax) pmax = p_act;
  if (p_pb > pmax) pmax = p_pb;
  if (p_pow > pmax) pmax = p_pow;
  if (p_aux > pmax) pmax = p_aux;
  WsLayout L;
  L.part = 0;
  size_t off = align_up(pmax * r, 256);
  L.u = off;
  off += align_up((size_t)B * M * F * M * M * 2 * r, 256);
  L.lpart = off;
  size_t nl = (size_t)B * ((size_t)TS * F + (size_t)M * F + (size_t)M * ((T + 63) / 64) + F + 16 +
                           (size_t)flat_loss(F, T).G);
  off += align_up(nl * 8, 256);
  L.small = off;
  off += align_up((size_t)B * (M + 8) * 8, 256);
  L.map = L.nmf = L.tmp = off;
  if (K > KU) {
    off += align_up((size_t)B * M * F * T * r, 256);
    L.nmf = off;
    off += align_up(assx_nmf_workspace_bytes(B * M, F, T, K, dtype), 256);
    L.tmp = off;
    off += align_up(((size_t)B * M * F * K + (size_t)B * M * K * T) * r, 256);
  }
  L.total = off;
  return L;
}

template <typename Fn>
int dispatch_rm(assx_ctx* ctx, int dtype, int M, Fn&& fn) {
  // ASSX_DEV_ONLY_M4_F64 (build.sh with ASSX_DEV=1): instantiate the headline configuration only -- a third of the
  // compile time while a kernel is being tuned.  Never shipped: the full build is what the tests run.
  if (dtype == ASSX_F64) {
    switch (M) {
#if !defined(ASSX_DEV_ONLY_M4_F64) && !defined(ASSX_DEV_ONLY_M4_F32)
      case 2: return fn(double(), IntC<2>());
      case 3: return fn(double(), IntC<3>());
#endif
#if !defined(ASSX_DEV_ONLY_M4_F32)
      case 4: return fn(double(), IntC<4>());
#endif
    }
  } else if (dtype == ASSX_F32) {
#if !defined(ASSX_DEV_ONLY_M4_F64)
    switch (M) {
#if !defined(ASSX_DEV_ONLY_M4_F32)
      case 2: return fn(float(), IntC<2>());
      case 3: return fn(float(), IntC<3>());
#endif
      case 4: return fn(float(), IntC<4>());
    }
#endif
  } else {
    return fail(ctx, ASSX_E_ARG, "dtype must be ASSX_F32 or ASSX_F64, got %d", dtype);
  }
  return fail(ctx, ASSX_E_UNSUPPORTED,
              "this entry point supports 2 <= M <= 4 channels, got M=%d (5 <= M <= 32 runs on the wide-channel path of the "
              "model entry points; more than 32 channels are not supported)", M);
}

#define CHECK_COMMON(ctx, B, M, F, T)                                                        \
  ASSX_REQUIRE_CTX(ctx);                              \
  ASSX_REQUIRE(ctx, (B) >= 1 && (M) >= 1 && (F) >= 1 && (T) >= 1, ASSX_E_ARG,                 \
               "invalid sizes B=%d M=%d F=%d T=%d", (B), (M), (F), (T));                       \
  ASSX_REQUIRE(ctx, (long long)(M) * (F) * (T) < (1LL << 28) && (long long)(B) * (F) * (((T) + 31) / 32 + 1) < (1LL << 31), \
               ASSX_E_UNSUPPORTED,                                                                                       \
               "utterance too large: buffer offsets are 32-bit, one utterance must stay below 4 GiB in complex128 "     \
               "(M*F*T < 2^28) and B*F*T/32 below 2^31")

inline unsigned blocks_for(size_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// Launch order of a streaming pass over X (assx_stream.hpp: workgroup_range): utterance-sequential with the direction
// alternating from pass to pass when the launch holds more than one utterance; ASSX_UTT_ORDER=0 restores the legacy
// order (A/B runs).  Returns the grid size.  The order decides WHEN a range runs, never what it computes.
inline unsigned set_launch_order(FlatPart& fp, int B, int rev) {
  static const int on = lab_int("ASSX_UTT_ORDER", 1);
  if (!on || B < 2) {
    fp.Gp = 0;
    fp.rev = 0;
    return (unsigned)fp.G;
  }
  fp.Gp = (fp.Gu + N_XCD - 1) / N_XCD * N_XCD;
  fp.rev = rev ? 1 : 0;
  return (unsigned)B * (unsigned)fp.Gp;
}
inline unsigned stream_grid(assx_ctx* ctx, FlatPart& fp, int B) {
  return set_launch_order(fp, B, B >= 2 ? (int)(ctx->stream_pass++ & 1u) : 0);
}

// covariance (any weight kind): streaming partials, then dense U; returns error code
template <typename R, int M>
int run_cov_partial(assx_ctx* ctx, int wk, const void* X, const void* r, const void* Tb, const void* V, int K,
                    double domain, double eps, void* ws, int B, int F, int T, hipStream_t st, FlatPart* fp_out) {
  CovArgs<R> a;
  a.d = Dims{B, F, T, K};
  a.fp = flat_cov(B, F, T, wk == WK_NONE ? 1 : cov_lane_split<R, M>());
  a.eps = (R)eps;
  a.p2d = make_pow(2.0 / domain);
  *fp_out = a.fp;
  dim3 grid(stream_grid(ctx, a.fp, B));
  const bool d2 = a.p2d.mode == POW_ID;
#define COV_LAUNCH(WKV, K4V, D2V, LSV, DXV, DWV, MW) \
  hipLaunchKernelGGL((cov_stream_kernel<R, M, WKV, K4V, D2V, LSV, DXV, DWV, MW>), grid, dim3(64), 0, st, \
                     (const Cx<R>*)X, (const R*)r, (const R*)Tb, (const R*)V, (R*)ws, a)
  constexpr int LSX = cov_lane_split<R, M>();
  switch (wk) {
    case WK_NONE: COV_LAUNCH(WK_NONE, true, true, 1, 4, 1, 1); break;
    case WK_NT: COV_LAUNCH(WK_NT, true, true, LSX, 3, 1, 2); break;
    case WK_NFT: COV_LAUNCH(WK_NFT, true, true, LSX, 3, 1, 2); break;
    default:
      // K <= 4: activation tile through the LDS-direct ring (VTileDma), X slots refilled in place; ASSX_COV_VDMA=0
      // selects the plain vector-load form of the same kernel (kept for A/B measurements)
      if (K <= KU) {
        constexpr size_t vlds = (size_t)VDMA_SLOTS * VTileDma<R, M * KU>::TILE_BYTES;
#if ASSX_LAB
        static const int vdma = lab_int("ASSX_COV_VDMA", 1);
        if (!vdma) {
          if (d2) COV_LAUNCH(WK_TV, true, true, LSX, (sizeof(R) == 8 ? 2 : 4), 1, 2);
          else COV_LAUNCH(WK_TV, true, false, LSX, 2, 1, 1);
        } else
#endif
        if (d2)
          hipLaunchKernelGGL((cov_stream_kernel<R, M, WK_TV, true, true, 1, 2, 1, 2, true>), grid, dim3(64), vlds, st,
                             (const Cx<R>*)X, (const R*)r, (const R*)Tb, (const R*)V, (R*)ws, a);
        else
          hipLaunchKernelGGL((cov_stream_kernel<R, M, WK_TV, true, false, 1, 2, 1, 1, true>), grid, dim3(64), vlds, st,
                             (const Cx<R>*)X, (const R*)r, (const R*)Tb, (const R*)V, (R*)ws, a);
      }
      else if (d2) COV_LAUNCH(WK_TV, false, true, LSX, 4, 1, 1);
      else COV_LAUNCH(WK_TV, false, false, LSX, 2, 1, 1);
  }
#undef COV_LAUNCH
  ASSX_LAUNCH_CHECK(ctx, "cov_stream_kernel");
  return 0;
}

template <typename R, int M>
int run_cov(assx_ctx* ctx, int wk, const void* X, const void* r, const void* Tb, const void* V, int K, double domain,
            double eps, void* U, void* ws, int B, int F, int T, hipStream_t st) {
  FlatPart fp;
  int rc = run_cov_partial<R, M>(ctx, wk, X, r, Tb, V, K, domain, eps, ws, B, F, T, st, &fp);
  if (rc) return rc;
  const int N = (wk == WK_NONE) ? 1 : M;
  const size_t total = (size_t)B * N * F * M * M;
  hipLaunchKernelGGL((cov_stream_finalize_kernel<R, M>), dim3(blocks_for(total, 256)), dim3(256), 0, st, (const R*)ws,
                     (Cx<R>*)U, B, N, F, fp, (R)(1.0 / (double)T));
  ASSX_LAUNCH_CHECK(ctx, "cov_stream_finalize_kernel");
  return 0;
}

template <typename R, int M>
int run_ip(assx_ctx* ctx, const void* U, const void* part, FlatPart fp, int T, void* W, const void* C, double* pw,
           double thr, int32_t* status, int B, int F, hipStream_t st, double den_floor = 0.0, int wb = 1) {
  constexpr int GPW = WAVE / next_pow2_c(M * M);
#if ASSX_LAB
  // round 4, ASSX_IP_PAR=1 (laboratory builds only): the sources of a bin side by side in one wave (assx_group_linalg.hpp:
  // ip_par_kernel).  Parity-green on every test and SLOWER: 19.2 us against 14.9 us at config 4 (profiles/r04_ip_par.txt),
  // so the sequential sweep is the product's.  Read on every call.
  const int par = lab_int("ASSX_IP_PAR", 0);
  if (par) {
    const dim3 gp(blocks_for((size_t)B * F, GPW / M)), bp(64);
    if (part)
      hipLaunchKernelGGL((ip_par_kernel<R, M, true>), gp, bp, 0, st, (const Cx<R>*)nullptr, (const R*)part, fp,
                         1.0 / (double)T, (Cx<R>*)W, (const Cx<R>*)C, pw, thr, status, B, F, den_floor, wb);
    else
      hipLaunchKernelGGL((ip_par_kernel<R, M, false>), gp, bp, 0, st, (const Cx<R>*)U, (const R*)nullptr, fp, 1.0,
                         (Cx<R>*)W, (const Cx<R>*)C, pw, thr, status, B, F, den_floor, 1);
    ASSX_LAUNCH_CHECK(ctx, "ip_par_kernel");
    return 0;
  }
#endif
  const dim3 grid(blocks_for((size_t)B * F, GPW)), block(64);
  if (part)
    hipLaunchKernelGGL((ip_group_kernel<R, M, true>), grid, block, 0, st, (const Cx<R>*)nullptr, (const R*)part, fp,
                       1.0 / (double)T, (Cx<R>*)W, (const Cx<R>*)C, pw, thr, status, B, F, den_floor, wb);
  else
    hipLaunchKernelGGL((ip_group_kernel<R, M, false>), grid, block, 0, st, (const Cx<R>*)U, (const R*)nullptr, fp, 1.0,
                       (Cx<R>*)W, (const Cx<R>*)C, pw, thr, status, B, F, den_floor, 1);
  ASSX_LAUNCH_CHECK(ctx, "ip_group_kernel");
  return 0;
}

template <typename R, int M>
int run_iss(assx_ctx* ctx, const void* U, const void* part, FlatPart fp, int T, void* W, const void* C, double* pw,
            int B, int F, hipStream_t st) {
  constexpr int GPW = WAVE / next_pow2_c(M * M);
  const dim3 grid(blocks_for((size_t)B * F, GPW)), block(64);
  if (part)
    hipLaunchKernelGGL((iss_group_kernel<R, M, true>), grid, block, 0, st, (const Cx<R>*)nullptr, (const R*)part, fp,
                       1.0 / (double)T, (double)T, (Cx<R>*)W, (const Cx<R>*)C, pw, B, F);
  else
    hipLaunchKernelGGL((iss_group_kernel<R, M, false>), grid, block, 0, st, (const Cx<R>*)U, (const R*)nullptr, fp, 1.0,
                       (double)T, (Cx<R>*)W, (const Cx<R>*)C, pw, B, F);
  ASSX_LAUNCH_CHECK(ctx, "iss_group_kernel");
  return 0;
}

template <typename R, int M>
int run_ip2(assx_ctx* ctx, const void* U, const void* part, FlatPart fp, int T, void* W, const void* C, double* pw,
            double thr, int32_t* status, int B, int F, int pm, int pn, hipStream_t st) {
  constexpr int GPW = WAVE / next_pow2_c(M * M);
  const dim3 grid(blocks_for((size_t)B * F, GPW)), block(64);
  if (part)
    hipLaunchKernelGGL((ip2_group_kernel<R, M, true>), grid, block, 0, st, (const Cx<R>*)nullptr, (const R*)part, fp,
                       1.0 / (double)T, (Cx<R>*)W, (const Cx<R>*)C, pw, thr, status, B, F, pm, pn);
  else
    hipLaunchKernelGGL((ip2_group_kernel<R, M, false>), grid, block, 0, st, (const Cx<R>*)U, (const R*)nullptr, fp, 1.0,
                       (Cx<R>*)W, (const Cx<R>*)C, pw, thr, status, B, F, pm, pn);
  ASSX_LAUNCH_CHECK(ctx, "ip2_group_kernel");
  return 0;
}

// source-model partial sums (reduce over t): part[g][slot][n][k][num|den]
template <typename R, int MM>
int run_basis_partial(assx_ctx* ctx, const void* X, const void* W, const void* Tb, const void* V, double domain,
                      double eps, void* ws, int B, int F, int T, int K, hipStream_t st, FlatPart* fp_out,
                      double nu = -1.0 /* >= 0: t-ILRMA harmonic statistic (domain 2) */,
                      double* lpart = nullptr /* fused loss partials (see basis_loss_fusable) */, int lstride = 0) {
  NmfArgs<R> a;
  a.nu = (R)nu;
  a.d = Dims{B, F, T, K};
  a.eps = (R)eps;
  a.p1 = make_pow((domain + 2.0) / domain);
  a.fp = flat_basis(B, F, T);
  *fp_out = a.fp;
  const bool d2 = a.p1.mode == POW_SQUARE, k4 = K <= KU;
  const dim3 gb(stream_grid(ctx, a.fp, B)), bb(64);
#define BASIS_LAUNCH(K4V, D2V, DXV, DWV, MW) \
  hipLaunchKernelGGL((basis_stream_kernel<R, MM, K4V, D2V, DXV, DWV, MW>), gb, bb, 0, st, (const Cx<R>*)X, \
                     (const Cx<R>*)W, (const R*)Tb, (const R*)V, (R*)ws, a)
  constexpr size_t vlds = (size_t)VDMA_SLOTS * VTileDma<R, MM * KU>::TILE_BYTES;
#define BASIS_VD(D2V, MW, TDV) \
  hipLaunchKernelGGL((basis_stream_vd_kernel<R, MM, D2V, 2, MW, TDV>), gb, bb, vlds, st, (const Cx<R>*)X, \
                     (const Cx<R>*)W, (const R*)Tb, (const R*)V, (R*)ws, a, (double*)nullptr, 0)
  // n_basis <= 4: the activation tile rides the LDS-direct ring (basis_stream_vd_kernel).  The plain vector-load forms of
  // rounds 1-2 are laboratory builds' (ASSX_BASIS_VDMA=0); n_basis > 4 arrives here only where the matrix-core source
  // model does not apply (partitioning function with n_basis > 64)
#if ASSX_LAB
  static const int vdma = lab_int("ASSX_BASIS_VDMA", 1);
  if (k4 && !vdma) {
    if (nu >= 0.0) hipLaunchKernelGGL((basis_stream_kernel<R, MM, true, true, 3, 1, 1, true>), gb, bb, 0, st, (const Cx<R>*)X,
                                      (const Cx<R>*)W, (const R*)Tb, (const R*)V, (R*)ws, a);
    else if (d2) BASIS_LAUNCH(true, true, 3, 1, 2);
    else BASIS_LAUNCH(true, false, 2, 1, 1);
  } else
#endif
  if (nu >= 0.0) {
    if (k4) BASIS_VD(true, 2, true);
    else hipLaunchKernelGGL((basis_stream_kernel<R, MM, false, true, 3, 1, 1, true>), gb, bb, 0, st, (const Cx<R>*)X,
                            (const Cx<R>*)W, (const R*)Tb, (const R*)V, (R*)ws, a);
  } else if (k4) {
    if (d2 && lpart)
      hipLaunchKernelGGL((basis_stream_vd_kernel<R, MM, true, 2, 2, false, true>), gb, bb, vlds, st, (const Cx<R>*)X,
                         (const Cx<R>*)W, (const R*)Tb, (const R*)V, (R*)ws, a, lpart, lstride);
    else if (d2) BASIS_VD(true, 2, false);
    else BASIS_VD(false, 1, false);
  } else if (d2) BASIS_LAUNCH(false, true, 3, 1, 1);
  else BASIS_LAUNCH(false, false, 2, 1, 1);
#undef BASIS_VD
#undef BASIS_LAUNCH
  ASSX_LAUNCH_CHECK(ctx, "basis_stream_kernel");
  return 0;
}

// source-model partial sums (reduce over f): part[g][slot][n][k][num|den][64 frames]
template <typename R, int MM>
int run_act_partial(assx_ctx* ctx, const void* X, const void* W, const void* Tb, const void* V, double domain,
                    double eps, void* ws, int B, int F, int T, int K, hipStream_t st, FlatPart* fp_out,
                    double nu = -1.0) {
  NmfArgs<R> a;
  a.nu = (R)nu;
  a.d = Dims{B, F, T, K};
  a.eps = (R)eps;
  a.p1 = make_pow((domain + 2.0) / domain);
  a.fp = flat_act(B, F, T);
  *fp_out = a.fp;
  const bool d2 = a.p1.mode == POW_SQUARE, k4 = K <= KU;
  const dim3 ga(stream_grid(ctx, a.fp, B)), ba(64 * ACT_NH);
#define ACT_LAUNCH(K4V, D2V, DXV, MW) \
  hipLaunchKernelGGL((act_stream_kernel<R, MM, K4V, D2V, DXV, MW>), ga, ba, 0, st, (const Cx<R>*)X, (const Cx<R>*)W, \
                     (const R*)Tb, (const R*)V, (R*)ws, a)
#define ACT_VD(D2V, DXV, MW, TDV) \
  hipLaunchKernelGGL((act_stream_vd_kernel<R, MM, D2V, DXV, MW, TDV>), ga, ba, 0, st, (const Cx<R>*)X, (const Cx<R>*)W, \
                     (const R*)Tb, (const R*)V, (R*)ws, a)
#if ASSX_LAB
  static const int vdma = lab_int("ASSX_ACT_VDMA", 1);  // 0: the plain vector-load forms of rounds 1-2 (A/B runs)
  if (k4 && !vdma) {
    if (nu >= 0.0) hipLaunchKernelGGL((act_stream_kernel<R, MM, true, true, 3, 1, true>), ga, ba, 0, st, (const Cx<R>*)X,
                                      (const Cx<R>*)W, (const R*)Tb, (const R*)V, (R*)ws, a);
    else if (d2) ACT_LAUNCH(true, true, (sizeof(R) == 8 ? 3 : 4), 2);
    else ACT_LAUNCH(true, false, 2, 1);
  } else
#endif
  if (nu >= 0.0) {
    if (k4) ACT_VD(true, 3, 2, true);
    else hipLaunchKernelGGL((act_stream_kernel<R, MM, false, true, 4, 1, true>), ga, ba, 0, st, (const Cx<R>*)X,
                            (const Cx<R>*)W, (const R*)Tb, (const R*)V, (R*)ws, a);
  } else if (k4) {
    if (d2) ACT_VD(true, 3, 2, false);
    else ACT_VD(false, 2, 1, false);
  } else if (d2) ACT_LAUNCH(false, true, 4, 1);
  else ACT_LAUNCH(false, false, 2, 1);
#undef ACT_VD
#undef ACT_LAUNCH
  ASSX_LAUNCH_CHECK(ctx, "act_stream_kernel");
  return 0;
}

// n_basis > 4: covariance without the (N,F,T) detour where the activation tile fits LDS (cov_wide_kernel; dense U
// lands at `U_dense` and *dense is set), otherwise through the materialised source variance
// (source_variance_map_kernel + the (N,F,T)-weights form of the streaming kernel; partial records as usual)
template <typename R, int MM>
int run_cov_partial_tv(assx_ctx* ctx, const void* X, const void* Tb, const void* V, int K, double domain, double eps,
                       void* ws, void* U_dense, int B, int F, int T, int dtype, hipStream_t st, FlatPart* fp_out,
                       bool* dense, int* records_wb = nullptr /* non-null: the caller's sweep reads cov_wide_kernel's
                       records itself (ip_group_kernel, wb = COVW_BINS); the dense finalize is skipped */) {
  static const bool wide = lab_int("ASSX_WIDE_K", 1) != 0;
  static const bool fused = lab_int("ASSX_COV_WIDE", 1) != 0;
  *dense = false;
  if (records_wb) *records_wb = 1;
  if (K <= KU || !wide) return run_cov_partial<R, MM>(ctx, WK_TV, X, nullptr, Tb, V, K, domain, eps, ws, B, F, T, st, fp_out);
  const WsLayout L = ws_layout(B, MM, F, T, K, dtype);
  const PowSpec p2d = make_pow(2.0 / domain);
  // 4 < n_basis <= 16: the variance contraction on the matrix cores (assx_cov_mfma.hpp); ASSX_COV_MFMA=0 keeps round 2's
  // LDS-tile kernel for A/B runs.  Same partition, same records.
  static const bool mfma = lab_int("ASSX_COV_MFMA", 1) != 0;
  if (fused && mfma && U_dense && K <= 16 && (size_t)B * MM * K * T * sizeof(R) < 0xffffffffull) {
    const FlatPart fw = flat_cov_wide(B, F, T, 1);
    const Dims d{B, F, T, K};
    const int ks = (K + 3) / 4;
    const size_t ldsm = CovMfmaGeom<R>::lds_bytes(MM, ks);
    // timing-experiment builds (-DCOVM_TRACE=1) stamp the trips of one workgroup into the last 64 KiB of the scratch
    unsigned long long* covm_trace = COVM_TRACE ? (unsigned long long*)((char*)ws + L.total - 65536) : nullptr;
#define COVM_LAUNCH(D2V, KSV)                                                                                          \
  do {                                                                                                                 \
    if (ldsm > 64 * 1024) {                                                                                            \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cov_mfma_kernel<R, MM, D2V, KSV>),              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm);                       \
      if (e != hipSuccess) return hip_fail(ctx, e, "hipFuncSetAttribute(cov_mfma_kernel)");                            \
    }                                                                                                                  \
    hipLaunchKernelGGL((cov_mfma_kernel<R, MM, D2V, KSV>), dim3(fw.G), dim3(WAVE * COVW_BINS), ldsm, st,               \
                       (const Cx<R>*)X, (const R*)Tb, (const R*)V, (R*)ws, d, fw, (R)eps, p2d, covm_trace);            \
  } while (0)
    if (p2d.mode == POW_ID) {
      if (ks == 2) COVM_LAUNCH(true, 2);
      else if (ks == 3) COVM_LAUNCH(true, 3);
      else COVM_LAUNCH(true, 4);
    } else {
      if (ks == 2) COVM_LAUNCH(false, 2);
      else if (ks == 3) COVM_LAUNCH(false, 3);
      else COVM_LAUNCH(false, 4);
    }
#undef COVM_LAUNCH
    ASSX_LAUNCH_CHECK(ctx, "cov_mfma_kernel");
    *fp_out = fw;
    if (records_wb) {
      *records_wb = COVW_BINS;
      return 0;
    }
    hipLaunchKernelGGL((cov_wide_finalize_kernel<R, MM>), dim3(blocks_for((size_t)B * MM * F * MM * MM, 256)), dim3(256),
                       0, st, (const R*)ws, (Cx<R>*)U_dense, B, F, fw, (R)(1.0 / (double)T));
    ASSX_LAUNCH_CHECK(ctx, "cov_wide_finalize_kernel");
    *dense = true;
    return 0;
  }
  const int sb = cov_wide_sb<R>(MM * K);
  const size_t lds = sb == 2 ? CovWideGeom<R, 2>::lds_bytes(MM * K) : CovWideGeom<R, 1>::lds_bytes(MM * K);
  if (fused && U_dense && lds <= 144 * 1024 && (size_t)B * MM * K * T * sizeof(R) < 0xffffffffull) {
    const FlatPart fw = flat_cov_wide(B, F, T, sb);
    const Dims d{B, F, T, K};
#define COVW_LAUNCH(D2V, SBV)                                                                                        \
  do {                                                                                                               \
    if (lds > 64 * 1024) {                                                                                           \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cov_wide_kernel<R, MM, D2V, SBV>),            \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
      if (e != hipSuccess) return hip_fail(ctx, e, "hipFuncSetAttribute(cov_wide_kernel)");                          \
    }                                                                                                                \
    hipLaunchKernelGGL((cov_wide_kernel<R, MM, D2V, SBV>), dim3(fw.G), dim3(WAVE * COVW_BINS), lds, st,              \
                       (const Cx<R>*)X, (const R*)Tb, (const R*)V, (R*)ws, d, fw, (R)eps, p2d);                      \
  } while (0)
#if ASSX_LAB
    if (sb == 2) {  // ASSX_COVW_SB=2: one barrier per 128 frames (measured slower; A/B runs)
      if (p2d.mode == POW_ID) COVW_LAUNCH(true, 2);
      else COVW_LAUNCH(false, 2);
    } else
#endif
    if (p2d.mode == POW_ID) COVW_LAUNCH(true, 1);
    else COVW_LAUNCH(false, 1);
#undef COVW_LAUNCH
    ASSX_LAUNCH_CHECK(ctx, "cov_wide_kernel");
    if (records_wb) {
      *records_wb = COVW_BINS;
      *fp_out = fw;
      return 0;
    }
    hipLaunchKernelGGL((cov_wide_finalize_kernel<R, MM>), dim3(blocks_for((size_t)B * MM * F * MM * MM, 256)), dim3(256),
                       0, st, (const R*)ws, (Cx<R>*)U_dense, B, F, fw, (R)(1.0 / (double)T));
    ASSX_LAUNCH_CHECK(ctx, "cov_wide_finalize_kernel");
    *fp_out = fw;
    *dense = true;
    return 0;
  }
  R* rv = (R*)((char*)ws + L.map);
  const dim3 grid(blocks_for(T, 256), blocks_for(F, WIDE_FB), B * MM);
  if (p2d.mode == POW_ID)
    hipLaunchKernelGGL((source_variance_map_kernel<R, true>), grid, dim3(256), 0, st, (const R*)Tb, (const R*)V, rv, F, T,
                       K, p2d);
  else
    hipLaunchKernelGGL((source_variance_map_kernel<R, false>), grid, dim3(256), 0, st, (const R*)Tb, (const R*)V, rv, F,
                       T, K, p2d);
  ASSX_LAUNCH_CHECK(ctx, "source_variance_map_kernel");
  return run_cov_partial<R, MM>(ctx, WK_NFT, X, rv, nullptr, nullptr, 1, 2.0, eps, ws, B, F, T, st, fp_out);
}

}  // namespace

// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" {

int assx_launch_order(int B, int F, int T, int reverse, int* ranges, int capacity) {
  if (B < 1 || F < 1 || T < 1) return ASSX_E_ARG;
  FlatPart fp = flat_cov(B, F, T, 1);
  const int grid = (int)set_launch_order(fp, B, reverse);
  if (ranges) {
    if (capacity < grid) return ASSX_E_ARG;
    for (int bid = 0; bid < grid; ++bid) ranges[bid] = workgroup_range(bid, grid, fp);
  }
  return grid;
}

size_t assx_workspace_bytes(int B, int M, int F, int T, int K, int dtype) {
  if (B < 1 || M < 1 || F < 1 || T < 1) return 0;
  if (widem::handles(M)) return widem::workspace_bytes(B, M, F, T, K, dtype);
  return ws_layout(B, M, F, T, K, dtype).total;
}

int assx_demix(assx_ctx* ctx, const void* X, const void* W, const void* scale, void* Y, int B, int M, int F, int T,
               int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && Y, ASSX_E_NULL, "assx_demix: NULL array");
  if (widem::handles(M)) return widem::demix(ctx, X, W, scale, Y, B, M, F, T, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    dim3 grid(blocks_for(T, 256), F, B);
    hipLaunchKernelGGL((demix_kernel<R, MM>), grid, dim3(256), 0, st, (const Cx<R>*)X, (const Cx<R>*)W,
                       (const Cx<R>*)scale, (Cx<R>*)Y, Dims{B, F, T, 0});
    ASSX_LAUNCH_CHECK(ctx, "demix_kernel");
    return 0;
  });
}

int assx_ilrma_power_map(assx_ctx* ctx, const void* X, const void* W, void* P, int B, int M, int F, int T, int dtype,
                         void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && P, ASSX_E_NULL, "assx_ilrma_power_map: NULL array");
  hipStream_t st = (hipStream_t)stream;
  if (widem::handles(M)) return widem::power_map(ctx, X, W, P, B, M, F, T, dtype, st);
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    hipLaunchKernelGGL((demix_power_map_kernel<R, MM>), dim3(blocks_for(T, 256), F, B), dim3(256), 0, st,
                       (const Cx<R>*)X, (const Cx<R>*)W, (R*)P, Dims{B, F, T, 0});
    ASSX_LAUNCH_CHECK(ctx, "demix_power_map_kernel");
    return 0;
  });
}

int assx_cov_accumulate(assx_ctx* ctx, const void* X, const void* r, int r_kind, double eps, void* U, void* ws, int B,
                        int M, int N, int F, int T, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && U && ws, ASSX_E_NULL, "assx_cov_accumulate: NULL array");
  ASSX_REQUIRE(ctx, r_kind == ASSX_W_NONE || r_kind == ASSX_W_NT || r_kind == ASSX_W_NFT, ASSX_E_ARG,
               "assx_cov_accumulate: bad r_kind %d", r_kind);
  ASSX_REQUIRE(ctx, r_kind == ASSX_W_NONE ? (N == 1) : (N == M && r != nullptr), ASSX_E_ARG,
               "assx_cov_accumulate: N must be 1 (unweighted) or M with weights, got N=%d M=%d", N, M);
  if (widem::handles(M)) return widem::cov_accumulate(ctx, X, r, r_kind, eps, U, ws, B, M, N, F, T, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    return run_cov<R, MM>(ctx, r_kind, X, r, nullptr, nullptr, 1, 2.0, eps, U, ws, B, F, T, st);
  });
}

int assx_ip_update(assx_ctx* ctx, const void* U, void* W, double threshold, int32_t* status, int B, int M, int F,
                   int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, 1);
  ASSX_REQUIRE(ctx, U && W, ASSX_E_NULL, "assx_ip_update: NULL array");
  if (widem::handles(M)) return widem::ip_update(ctx, U, W, threshold, status, B, M, F, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    return run_ip<R, MM>(ctx, U, nullptr, FlatPart{}, 1, W, nullptr, nullptr, threshold, status, B, F, st);
  });
}

int assx_ip2_update(assx_ctx* ctx, const void* U, void* W, double threshold, int32_t* status, int pair_m, int pair_n,
                    int B, int M, int F, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, 1);
  ASSX_REQUIRE(ctx, U && W, ASSX_E_NULL, "assx_ip2_update: NULL array");
  ASSX_REQUIRE(ctx, pair_m >= 0 && pair_m < M && pair_n >= 0 && pair_n < M && pair_m != pair_n, ASSX_E_ARG,
               "bad update pair (%d, %d) for %d sources", pair_m, pair_n, M);
  if (widem::handles(M)) return widem::ip2_update(ctx, U, W, threshold, status, pair_m, pair_n, B, M, F, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    return run_ip2<R, MM>(ctx, U, nullptr, FlatPart{}, 1, W, nullptr, nullptr, threshold, status, B, F, pair_m, pair_n,
                          st);
  });
}

int assx_iss_update(assx_ctx* ctx, const void* U, void* W, int n_frames, int B, int M, int F, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, 1);
  ASSX_REQUIRE(ctx, U && W && n_frames >= 1, ASSX_E_NULL, "assx_iss_update: NULL array / bad n_frames");
  if (widem::handles(M)) return widem::iss_update(ctx, U, W, n_frames, B, M, F, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    return run_iss<R, MM>(ctx, U, nullptr, FlatPart{}, n_frames, W, nullptr, nullptr, B, F, st);
  });
}

// forward declaration (defined with the loss entry points)
static int ilrma_loss_impl(assx_ctx* ctx, const char* who, const void* X, const void* W, const void* Tb,
                           const void* V, double domain, double nu, double eps, double* loss, void* ws, int B, int M,
                           int F, int T, int K, int dtype, void* stream, void* P_out = nullptr,
                           bool* wrote_P = nullptr);

int assx_ilrma_source_update(assx_ctx* ctx, const void* X, const void* W, void* Tb, void* V, double domain, double eps,
                             unsigned source_mask, double* loss_prev, void* ws, int B, int M, int F, int T, int K,
                             int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && Tb && V && ws, ASSX_E_NULL, "assx_ilrma_source_update: NULL array");
  ASSX_REQUIRE(ctx, K >= 1, ASSX_E_ARG, "n_basis must be >= 1, got %d", K);
  ASSX_REQUIRE(ctx, domain >= 1.0 && domain <= 2.0, ASSX_E_ARG, "1 <= domain <= 2 is not satisfied (%g)", domain);
  if (widem::handles(M)) return widem::ilrma_source_update(ctx, X, W, Tb, V, domain, eps, source_mask, loss_prev, ws, B, M, F, T, K, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    const PowSpec p2 = make_pow(domain / (domain + 2.0));
    FlatPart fp;
    int rc;
    // loss of the model at entry: fused into the basis pass where that pass forms the same quantities (domain 2,
    // K <= 4, LDS-ring kernel), otherwise a pass of its own before anything is updated
    double* lpart = nullptr;
    int lstride = 0;
    static const bool wide_k = lab_int("ASSX_WIDE_K", 1) != 0;
    const bool full_mask = (source_mask & ((1u << MM) - 1u)) == ((1u << MM) - 1u);
    bool have_map = false;
    if (loss_prev && K > KU && wide_k && full_mask && domain == 2.0 && (lab_int("ASSX_FUSE_LOSS", 1) != 0)) {
      // n_basis > 4 (round 4): the loss of the model at entry rides on the X-fed basis half (assx_nmf_xfed.hpp), which
      // forms the same |w^H x|^2 and Tb V -- no pass of its own, no power map
      const int ncov = nmf_xfed_loss_partials(MM, F, T, K);
      const WsLayout L = ws_layout(B, MM, F, T, K, dtype);
      const int ls = ncov + F;  // [per-(workgroup, source) data terms | F log-det terms]
      if (ncov > 0 && (size_t)B * ls * sizeof(double) <= L.small - L.lpart) {
        double* lp = (double*)((char*)ws + L.lpart);
        hipLaunchKernelGGL((logdet_kernel<R, MM>), dim3(blocks_for((size_t)B * F, 64)), dim3(64), 0, st,
                           (const Cx<R>*)W, lp, B, F, T, ls, ncov);
        ASSX_LAUNCH_CHECK(ctx, "logdet_kernel");
        rc = nmf_update_xfed(ctx, ASSX_NMF_IS_MM, domain, 0.0, eps, X, W, Tb, V, (char*)ws + L.nmf, B, MM, F, T, K, dtype,
                             st, lp, ls);
        if (rc != ASSX_E_UNSUPPORTED) {
          if (rc) return rc;
          hipLaunchKernelGGL(ilrma_loss_finish_kernel, dim3(B), dim3(REDUCE_THREADS), 0, st, (const double*)lp, loss_prev,
                             F, ncov, ls);
          ASSX_LAUNCH_CHECK(ctx, "ilrma_loss_finish_kernel");
          return 0;
        }
      }
    }
    if (loss_prev) {
      const bool fusable = domain == 2.0 && K <= KU && (lab_int("ASSX_BASIS_VDMA", 1) != 0) && (lab_int("ASSX_FUSE_LOSS", 1) != 0);
      if (!fusable) {
        // n_basis > 4: the loss pass forms |W x|^2 anyway and leaves it behind as the map the source model needs
        void* pmap = (K > KU && wide_k) ? (void*)((char*)ws + ws_layout(B, MM, F, T, K, dtype).map) : nullptr;
        rc = ilrma_loss_impl(ctx, "assx_ilrma_source_update", X, W, Tb, V, domain, -1.0, eps, loss_prev, ws, B, MM, F, T,
                             K, dtype, stream, pmap, &have_map);
        if (rc) return rc;
      } else {
        const WsLayout L = ws_layout(B, MM, F, T, K, dtype);
        const FlatPart fb = flat_basis(B, F, T);
        const int ncov = fb.Gu;  // workgroups of one utterance (the partition is per utterance)
        lstride = ncov + F;  // [per-workgroup data terms | F log-det terms]
        ASSX_REQUIRE(ctx, (size_t)B * lstride * sizeof(double) <= L.small - L.lpart, ASSX_E_UNSUPPORTED,
                     "workspace too small for the fused loss partials");
        lpart = (double*)((char*)ws + L.lpart);
        // the F log-det terms and the final sum ride on the two finalize launches below (round 6: 9 -> 7 launches per
        // iteration with the loss recorded, the reference's default)
      }
    }
    if (K > KU && wide_k && full_mask) {
      // n_basis > 4: P = |W x|^2 once, then the batched IS-NMF MM update on the matrix cores (same update rule,
      // ilrma.py:409-430 == nmf.py:302-327 with target P)
      const WsLayout L = ws_layout(B, MM, F, T, K, dtype);
      R* pw = (R*)((char*)ws + L.map);
      if (!have_map) {
        // round 4: no map at all where the X-fed halves apply (n_basis <= 32): |w_n^H x|^2 is formed inside the two
        // matrix-core kernels from an X tile shared by the workgroup's waves (assx_nmf_xfed.hpp) -- 2 launches and
        // 2 x 268.7 MB read instead of 3 launches, 268.7 MB read + 134 MB written + 2 x 134 MB read
        rc = nmf_update_xfed(ctx, ASSX_NMF_IS_MM, domain, 0.0, eps, X, W, Tb, V, (char*)ws + L.nmf, B, MM, F, T, K, dtype, st);
        if (rc != ASSX_E_UNSUPPORTED) return rc;
        hipLaunchKernelGGL((demix_power_map_kernel<R, MM>), dim3(blocks_for(T, 256), F, B), dim3(256), 0, st,
                           (const Cx<R>*)X, (const Cx<R>*)W, pw, Dims{B, F, T, 0});
        ASSX_LAUNCH_CHECK(ctx, "demix_power_map_kernel");
      }
      NmfGroupScope grp(ctx, MM);
      return assx_nmf_update(ctx, ASSX_NMF_IS_MM, domain, eps, pw, Tb, V, (char*)ws + L.nmf, B * MM, F, T, K, dtype,
                             stream);
    }
    if (K > KU && wide_k && !full_mask && (source_mask & ((1u << MM) - 1u)) != 0u) {
      // pairwise update at n_basis > 4: every source's update is independent, so all of them are run on copies of
      // the model (matrix cores, one pass) and only the selected sources are copied back
      const WsLayout L = ws_layout(B, MM, F, T, K, dtype);
      R* pw = (R*)((char*)ws + L.map);
      const size_t nT = (size_t)B * MM * F * K, nV = (size_t)B * MM * K * T;
      R* Tt = (R*)((char*)ws + L.tmp);
      R* Vt = Tt + nT;
      if (!have_map) {
        hipLaunchKernelGGL((demix_power_map_kernel<R, MM>), dim3(blocks_for(T, 256), F, B), dim3(256), 0, st,
                           (const Cx<R>*)X, (const Cx<R>*)W, pw, Dims{B, F, T, 0});
        ASSX_LAUNCH_CHECK(ctx, "demix_power_map_kernel");
      }
      hipError_t e = hipMemcpyAsync(Tt, Tb, nT * sizeof(R), hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(Vt, V, nV * sizeof(R), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) return fail(ctx, (int)e, "hipMemcpyAsync(model copy): %s", hipGetErrorString(e));
      {
        NmfGroupScope grp(ctx, MM);
        rc = assx_nmf_update(ctx, ASSX_NMF_IS_MM, domain, eps, pw, Tt, Vt, (char*)ws + L.nmf, B * MM, F, T, K, dtype, stream);
      }
      if (rc) return rc;
      hipLaunchKernelGGL((masked_model_copy_kernel<R>), dim3(blocks_for(nT + nV, 256)), dim3(256), 0, st, (const R*)Tt,
                         (const R*)Vt, (R*)Tb, (R*)V, B, MM, (size_t)F * K, (size_t)K * T, source_mask);
      ASSX_LAUNCH_CHECK(ctx, "masked_model_copy_kernel");
      return 0;
    }
    rc = run_basis_partial<R, MM>(ctx, X, W, Tb, V, domain, eps, ws, B, F, T, K, st, &fp, -1.0, lpart, lstride);
    if (rc) return rc;
    const unsigned nb_basis = blocks_for((size_t)B * MM * F * K, 256);
    if (lpart)  // + the log-det terms of the loss (workgroups past nb_basis)
      hipLaunchKernelGGL((basis_stream_finalize_kernel<R, MM>), dim3(nb_basis + blocks_for((size_t)B * F, 256)), dim3(256), 0,
                         st, (const R*)ws, (R*)Tb, B, MM, F, K, fp, (R)eps, p2, source_mask, (int)nb_basis, (const Cx<R>*)W,
                         lpart, lstride, lstride - F, T);
    else
      hipLaunchKernelGGL((basis_stream_finalize_kernel<R>), dim3(nb_basis), dim3(256), 0, st, (const R*)ws, (R*)Tb, B, MM,
                         F, K, fp, (R)eps, p2, source_mask);
    ASSX_LAUNCH_CHECK(ctx, "basis_stream_finalize_kernel");
    FlatPart fpa;
    rc = run_act_partial<R, MM>(ctx, X, W, Tb, V, domain, eps, ws, B, F, T, K, st, &fpa);  // uses the new basis
    if (rc) return rc;
    const unsigned nb_act = (unsigned)((size_t)B * MM * K * tblocks(T));
    // + one workgroup per utterance that completes the loss (lpart: the basis pass's data terms and the log-det terms)
    hipLaunchKernelGGL((act_stream_finalize_kernel<R>), dim3(nb_act + (lpart ? (unsigned)B : 0u)), dim3(256), 0, st,
                       (const R*)ws, (R*)V, B, MM, F, K, T, fpa, (R)eps, p2, source_mask, (int)nb_act, (const double*)lpart,
                       loss_prev, lstride - F, lstride);
    ASSX_LAUNCH_CHECK(ctx, "act_stream_finalize_kernel");
    return 0;
  });
}

int assx_ilrma_expand_partitioned(assx_ctx* ctx, const void* Z, const void* Tb, const void* V, void* Teff, void* Veff,
                                  int B, int M, int F, int T, int K, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, Z && Tb && V && (Teff || Veff), ASSX_E_NULL, "assx_ilrma_expand_partitioned: NULL array");
  ASSX_REQUIRE(ctx, K >= 1, ASSX_E_ARG, "n_basis must be >= 1, got %d", K);
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)B * M * F * K + (size_t)B * M * K * T;
  if (dtype == ASSX_F64)
    hipLaunchKernelGGL((part_expand_kernel<double>), dim3(blocks_for(total, 256)), dim3(256), 0, st, (const double*)Z,
                       (const double*)Tb, (const double*)V, (double*)Teff, (double*)Veff, B, M, F, K, T);
  else if (dtype == ASSX_F32)
    hipLaunchKernelGGL((part_expand_kernel<float>), dim3(blocks_for(total, 256)), dim3(256), 0, st, (const float*)Z,
                       (const float*)Tb, (const float*)V, (float*)Teff, (float*)Veff, B, M, F, K, T);
  else
    return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
  ASSX_LAUNCH_CHECK(ctx, "part_expand_kernel");
  return 0;
}

int assx_ilrma_source_update_partitioned(assx_ctx* ctx, const void* X, const void* W, void* Z, void* Tb, void* V,
                                         void* Teff, void* Veff, double eps, void* ws, int B, int M, int F, int T,
                                         int K, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && Z && Tb && V && Teff && Veff && ws, ASSX_E_NULL,
               "assx_ilrma_source_update_partitioned: NULL array");
  ASSX_REQUIRE(ctx, K >= 1, ASSX_E_ARG, "n_basis must be >= 1, got %d", K);
  hipStream_t st = (hipStream_t)stream;
  if (widem::handles(M))
    return widem::ilrma_source_update_partitioned(ctx, X, W, Z, Tb, V, Teff, Veff, eps, ws, B, M, F, T, K, dtype, st);
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    const size_t nT = (size_t)B * MM * F * K, nV = (size_t)B * MM * K * T;
    auto expand = [&](bool with_v) -> int {
      hipLaunchKernelGGL((part_expand_kernel<R>), dim3(blocks_for(with_v ? nT + nV : nT, 256)), dim3(256), 0, st,
                         (const R*)Z, (const R*)Tb, (const R*)V, (R*)Teff, with_v ? (R*)Veff : (R*)nullptr, B, MM, F,
                         K, T);
      ASSX_LAUNCH_CHECK(ctx, "part_expand_kernel");
      return 0;
    };
    FlatPart fp;
    int rc;
    static const bool wide = lab_int("ASSX_WIDE_K", 1) != 0;
    if (K > KU && K <= 64 && wide) {
      // n_basis > 4: the three sets of per-source sums come from the NMF matrix-core kernels on P = |W x|^2 (formed
      // once: W does not move here) with the effective model, batch B*N; adapters lay them out as the records the
      // combination kernels read
      const WsLayout L = ws_layout(B, MM, F, T, K, dtype);
      R* pw = (R*)((char*)ws + L.map);
      void* nws = (char*)ws + L.nmf;
      hipLaunchKernelGGL((demix_power_map_kernel<R, MM>), dim3(blocks_for(T, 256), F, B), dim3(256), 0, st,
                         (const Cx<R>*)X, (const Cx<R>*)W, pw, Dims{B, F, T, 0});
      ASSX_LAUNCH_CHECK(ctx, "demix_power_map_kernel");
      const int TBk = tblocks(T);
      const FlatPart fpb{(long long)F * TBk, TBk, TBk, B * F, 1, F, F};      // one record per bin
      const FlatPart fpa{(long long)TBk * F, F, F, B * TBk, 1, TBk, TBk};    // one record per frame block
      auto sums = [&](int half) -> int {
        const void* np = nullptr;
        int slabs = 0;
        NmfGroupScope grp(ctx, MM);
        int r2 = nmf_half_partials(ctx, ASSX_NMF_IS_MM, 2.0, 0.0, eps, half, pw, Teff, Veff, nws, B * MM, F, T, K, dtype,
                                   st, &np, &slabs);
        if (r2) return r2;
        if (half == NMF_HALF_BASIS)
          hipLaunchKernelGGL((part_adapt_basis_kernel<R>), dim3(blocks_for((size_t)B * F * MM * 2 * K, 256)), dim3(256), 0,
                             st, (const R*)np, (R*)ws, B, MM, F, K, slabs);
        else
          hipLaunchKernelGGL((part_adapt_act_kernel<R>), dim3(blocks_for((size_t)B * TBk * MM * 2 * K * WAVE, 256)),
                             dim3(256), 0, st, (const R*)np, (R*)ws, B, MM, K, T, slabs);
        ASSX_LAUNCH_CHECK(ctx, "part_adapt_kernel");
        return 0;
      };
      if ((rc = expand(true))) return rc;
      if ((rc = sums(NMF_HALF_BASIS))) return rc;
      hipLaunchKernelGGL((part_latent_kernel<R, MM>), dim3(K, B), dim3(256), 0, st, (const R*)ws, (const R*)Tb, (R*)Z, F,
                         K, fpb, (R)eps);
      ASSX_LAUNCH_CHECK(ctx, "part_latent_kernel");
      if ((rc = expand(false))) return rc;
      if ((rc = sums(NMF_HALF_BASIS))) return rc;
      hipLaunchKernelGGL((part_basis_kernel<R>), dim3(blocks_for((size_t)B * F * K, 256)), dim3(256), 0, st,
                         (const R*)ws, (const R*)Z, (R*)Tb, B, MM, F, K, fpb, (R)eps);
      ASSX_LAUNCH_CHECK(ctx, "part_basis_kernel");
      if ((rc = expand(false))) return rc;
      if ((rc = sums(NMF_HALF_ACT))) return rc;
      hipLaunchKernelGGL((part_act_kernel<R>), dim3(blocks_for((size_t)B * K * T, 256)), dim3(256), 0, st, (const R*)ws,
                         (R*)V, B, MM, F, K, T, fpa, (R)eps);
      ASSX_LAUNCH_CHECK(ctx, "part_act_kernel");
      return expand(true);
    }
    // ---- latent variables (ilrma.py:368-387)
    if ((rc = expand(true))) return rc;
    if ((rc = run_basis_partial<R, MM>(ctx, X, W, Teff, Veff, 2.0, eps, ws, B, F, T, K, st, &fp))) return rc;
    hipLaunchKernelGGL((part_latent_kernel<R, MM>), dim3(K, B), dim3(256), 0, st, (const R*)ws, (const R*)Tb, (R*)Z, F,
                       K, fp, (R)eps);
    ASSX_LAUNCH_CHECK(ctx, "part_latent_kernel");
    // ---- bases (ilrma.py:389-397)
    if ((rc = expand(false))) return rc;
    if ((rc = run_basis_partial<R, MM>(ctx, X, W, Teff, Veff, 2.0, eps, ws, B, F, T, K, st, &fp))) return rc;
    hipLaunchKernelGGL((part_basis_kernel<R>), dim3(blocks_for((size_t)B * F * K, 256)), dim3(256), 0, st,
                       (const R*)ws, (const R*)Z, (R*)Tb, B, MM, F, K, fp, (R)eps);
    ASSX_LAUNCH_CHECK(ctx, "part_basis_kernel");
    // ---- activations (ilrma.py:399-408)
    if ((rc = expand(false))) return rc;
    if ((rc = run_act_partial<R, MM>(ctx, X, W, Teff, Veff, 2.0, eps, ws, B, F, T, K, st, &fp))) return rc;
    hipLaunchKernelGGL((part_act_kernel<R>), dim3(blocks_for((size_t)B * K * T, 256)), dim3(256), 0, st, (const R*)ws,
                       (R*)V, B, MM, F, K, T, fp, (R)eps);
    ASSX_LAUNCH_CHECK(ctx, "part_act_kernel");
    return expand(true);  // leave (Teff, Veff) consistent with the updated (Z, T, V)
  });
}

int assx_ilrma_normalize_power_bins_partitioned(assx_ctx* ctx, void* W, void* Z, void* Tb, const double* power_bins,
                                                double eps, void* ws, int B, int M, int F, int K, int dtype,
                                                void* stream) {
  CHECK_COMMON(ctx, B, M, F, 1);
  ASSX_REQUIRE(ctx, W && Z && Tb && power_bins && ws, ASSX_E_NULL,
               "assx_ilrma_normalize_power_bins_partitioned: NULL array");
  ASSX_REQUIRE(ctx, K >= 1, ASSX_E_ARG, "n_basis must be >= 1, got %d", K);
  hipStream_t st = (hipStream_t)stream;
  if (widem::handles(M))
    return widem::normalize_power_bins_partitioned(ctx, W, Z, Tb, power_bins, eps, ws, B, M, F, K, dtype, st);
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    const size_t per_b = (size_t)F * MM * MM + (size_t)F * K;
    hipLaunchKernelGGL((part_normalize_power_kernel<R, MM>), dim3(blocks_for(per_b, 256), B), dim3(256),
                       (size_t)K * sizeof(R), st, (Cx<R>*)W, (R*)ws, (const R*)Z, (R*)Tb, power_bins, F, K, (R)eps);
    ASSX_LAUNCH_CHECK(ctx, "part_normalize_power_kernel");
    hipError_t e = hipMemcpyAsync(Z, ws, (size_t)B * MM * K * sizeof(R), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail(ctx, (int)e, "hipMemcpyAsync(Z): %s", hipGetErrorString(e));
    return 0;
  });
}

int assx_ilrma_spatial_update(assx_ctx* ctx, int spatial, int pair_m, int pair_n, const void* X, void* W,
                              const void* Tb, const void* V, double domain, double eps, double threshold, void* U_out,
                              const void* C,
                              double* power_bins, int32_t* status, void* ws, int B, int M, int F, int T, int K,
                              int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && Tb && V && ws, ASSX_E_NULL, "assx_ilrma_spatial_update: NULL array");
  ASSX_REQUIRE(ctx, K >= 1, ASSX_E_ARG, "n_basis must be >= 1, got %d", K);
  ASSX_REQUIRE(ctx, domain >= 1.0 && domain <= 2.0, ASSX_E_ARG, "1 <= domain <= 2 is not satisfied (%g)", domain);
  ASSX_REQUIRE(ctx, (C == nullptr) == (power_bins == nullptr), ASSX_E_ARG,
               "assx_ilrma_spatial_update: C and power_bins must be given together");
  ASSX_REQUIRE(ctx, spatial >= ASSX_SPATIAL_IP && spatial <= ASSX_SPATIAL_IP2, ASSX_E_ARG, "bad spatial algorithm %d",
               spatial);
  ASSX_REQUIRE(ctx, spatial != ASSX_SPATIAL_IP2 || (pair_m >= 0 && pair_m < M && pair_n >= 0 && pair_n < M &&
                                                    pair_m != pair_n),
               ASSX_E_ARG, "bad update pair (%d, %d) for %d sources", pair_m, pair_n, M);
  if (widem::handles(M))
    return widem::ilrma_spatial_update(ctx, spatial, pair_m, pair_n, X, W, Tb, V, domain, eps, threshold, U_out, C, power_bins,
                                       status, ws, B, M, F, T, K, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    FlatPart fp;
    bool dense = false;
    void* Ud = U_out ? U_out : (void*)((char*)ws + ws_layout(B, MM, F, T, K, dtype).u);
    int wb = 1;  // IP without a dense-U request: the sweep reduces the covariance records itself, whichever kernel made them
    const bool direct = spatial == ASSX_SPATIAL_IP && !U_out;
    int rc = run_cov_partial_tv<R, MM>(ctx, X, Tb, V, K, domain, eps, ws, Ud, B, F, T, dtype, st, &fp, &dense,
                                       direct ? &wb : nullptr);
    if (rc) return rc;
    if (U_out && !dense) {  // dense covariance on request only; the IP sweep reduces the partial records itself
      hipLaunchKernelGGL((cov_stream_finalize_kernel<R, MM>), dim3(blocks_for((size_t)B * MM * F * MM * MM, 256)),
                         dim3(256), 0, st, (const R*)ws, (Cx<R>*)U_out, B, MM, F, fp, (R)(1.0 / (double)T));
      ASSX_LAUNCH_CHECK(ctx, "cov_stream_finalize_kernel");
    }
    const void* Uin = dense ? Ud : nullptr;
    const void* pin = dense ? nullptr : ws;
    if (spatial == ASSX_SPATIAL_ISS) return run_iss<R, MM>(ctx, Uin, pin, fp, T, W, C, power_bins, B, F, st);
    if (spatial == ASSX_SPATIAL_IP2)
      return run_ip2<R, MM>(ctx, Uin, pin, fp, T, W, C, power_bins, threshold, status, B, F, pair_m, pair_n, st);
    return run_ip<R, MM>(ctx, Uin, pin, fp, T, W, C, power_bins, threshold, status, B, F, st, 0.0, wb);
  });
}

int assx_ilrma_cov_partials(assx_ctx* ctx, const void* X, const void* Tb, const void* V, double domain, double eps,
                            void* ws, int B, int M, int F, int T, int K, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && Tb && V && ws, ASSX_E_NULL, "assx_ilrma_cov_partials: NULL array");
  ASSX_REQUIRE(ctx, K >= 1, ASSX_E_ARG, "n_basis must be >= 1, got %d", K);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    FlatPart fp;
    bool dense = false;
    return run_cov_partial_tv<R, MM>(ctx, X, Tb, V, K, domain, eps, ws, (char*)ws + ws_layout(B, MM, F, T, K, dtype).u, B, F, T,
                                     dtype, st, &fp, &dense);
  });
}

int assx_demix_power(assx_ctx* ctx, const void* X, const void* W, void* power, void* ws, int B, int M, int F, int T,
                     int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && power && ws, ASSX_E_NULL, "assx_demix_power: NULL array");
  if (widem::handles(M)) return widem::demix_power(ctx, X, W, power, ws, B, M, F, T, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    int TS, tchunk;
    t_split(B, F, T, &TS, &tchunk);
    hipLaunchKernelGGL((power_partial_kernel<R, MM>), dim3((unsigned)F * TS, B), dim3(64), 0, st, (const Cx<R>*)X,
                       (const Cx<R>*)W, (R*)ws, Dims{B, F, T, 0}, TS, tchunk);
    ASSX_LAUNCH_CHECK(ctx, "power_partial_kernel");
    hipLaunchKernelGGL((sum_reduce_kernel<R, R>), dim3(B * MM), dim3(REDUCE_THREADS), 0, st, (const R*)ws, (R*)power,
                       (size_t)TS * F, 1.0 / ((double)F * (double)T));
    ASSX_LAUNCH_CHECK(ctx, "sum_reduce_kernel");
    return 0;
  });
}

int assx_power_from_cov(assx_ctx* ctx, const void* C, const void* W, void* power, void* ws, int B, int M, int F,
                        int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, 1);
  ASSX_REQUIRE(ctx, C && W && power && ws, ASSX_E_NULL, "assx_power_from_cov: NULL array");
  if (widem::handles(M)) return widem::power_from_cov(ctx, C, W, power, ws, B, M, F, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    double* part = (double*)ws;
    hipLaunchKernelGGL((power_cov_kernel<R, MM>), dim3(blocks_for((size_t)B * F * MM, 256)), dim3(256), 0, st,
                       (const Cx<R>*)C, (const Cx<R>*)W, part, B, F);
    ASSX_LAUNCH_CHECK(ctx, "power_cov_kernel");
    // C already carries the 1/T of the mean over frames: mean_{f,t}|y|^2 = (1/F) sum_f w^H C w
    hipLaunchKernelGGL((sum_reduce_kernel<double, R>), dim3(B * MM), dim3(REDUCE_THREADS), 0, st, (const double*)part,
                       (R*)power, (size_t)F, 1.0 / (double)F);
    ASSX_LAUNCH_CHECK(ctx, "sum_reduce_kernel");
    return 0;
  });
}

int assx_ilrma_normalize_power(assx_ctx* ctx, void* W, void* Tb, const void* power, double domain, double eps, int B,
                               int M, int F, int K, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, 1);
  ASSX_REQUIRE(ctx, W && Tb && power, ASSX_E_NULL, "assx_ilrma_normalize_power: NULL array");
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)B * F * M * M + (size_t)B * M * F * K;
  const PowSpec pd = make_pow(domain);
  if (dtype == ASSX_F64)
    hipLaunchKernelGGL((normalize_power_kernel<double>), dim3(blocks_for(total, 256)), dim3(256), 0, st, (Cx<double>*)W,
                       (double*)Tb, (const double*)power, B, M, F, K, eps, pd);
  else if (dtype == ASSX_F32)
    hipLaunchKernelGGL((normalize_power_kernel<float>), dim3(blocks_for(total, 256)), dim3(256), 0, st, (Cx<float>*)W,
                       (float*)Tb, (const float*)power, B, M, F, K, (float)eps, pd);
  else
    return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
  ASSX_LAUNCH_CHECK(ctx, "normalize_power_kernel");
  return 0;
}

int assx_ilrma_normalize_power_bins(assx_ctx* ctx, void* W, void* Tb, const double* power_bins, double domain,
                                    double eps, int B, int M, int F, int K, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, 1);
  ASSX_REQUIRE(ctx, W && Tb && power_bins, ASSX_E_NULL, "assx_ilrma_normalize_power_bins: NULL array");
  ASSX_REQUIRE(ctx, M <= 32, ASSX_E_UNSUPPORTED, "more than 32 channels are not supported (M = %d)", M);
  hipStream_t st = (hipStream_t)stream;
  const size_t per_b = (size_t)F * M * M + (size_t)M * F * K;
  const PowSpec pd = make_pow(domain);
  const dim3 grid(blocks_for(per_b, 256), B);
  if (dtype == ASSX_F64)
    hipLaunchKernelGGL((normalize_power_bins_kernel<double>), grid, dim3(256), 0, st, (Cx<double>*)W, (double*)Tb,
                       power_bins, M, F, K, eps, pd);
  else if (dtype == ASSX_F32)
    hipLaunchKernelGGL((normalize_power_bins_kernel<float>), grid, dim3(256), 0, st, (Cx<float>*)W, (float*)Tb,
                       power_bins, M, F, K, (float)eps, pd);
  else
    return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
  ASSX_LAUNCH_CHECK(ctx, "normalize_power_bins_kernel");
  return 0;
}

int assx_ilrma_normalize_pb(assx_ctx* ctx, void* W, void* Tb, const void* scale, double domain, int B, int M, int F,
                            int K, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, 1);
  ASSX_REQUIRE(ctx, W && Tb && scale, ASSX_E_NULL, "assx_ilrma_normalize_pb: NULL array");
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)B * F * M * M + (size_t)B * M * F * K;
  const PowSpec pd = make_pow(domain);
  if (dtype == ASSX_F64)
    hipLaunchKernelGGL((normalize_pb_kernel<double>), dim3(blocks_for(total, 256)), dim3(256), 0, st, (Cx<double>*)W,
                       (double*)Tb, (const Cx<double>*)scale, B, M, F, K, pd);
  else if (dtype == ASSX_F32)
    hipLaunchKernelGGL((normalize_pb_kernel<float>), dim3(blocks_for(total, 256)), dim3(256), 0, st, (Cx<float>*)W,
                       (float*)Tb, (const Cx<float>*)scale, B, M, F, K, pd);
  else
    return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
  ASSX_LAUNCH_CHECK(ctx, "normalize_pb_kernel");
  return 0;
}

static int ilrma_loss_impl(assx_ctx* ctx, const char* who, const void* X, const void* W, const void* Tb,
                           const void* V, double domain, double nu, double eps, double* loss, void* ws, int B, int M,
                           int F, int T, int K, int dtype, void* stream, void* P_out, bool* wrote_P) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && Tb && V && loss && ws, ASSX_E_NULL, "%s: NULL array", who);
  ASSX_REQUIRE(ctx, K >= 1, ASSX_E_ARG, "n_basis must be >= 1, got %d", K);
  if (widem::handles(M)) {
    if (wrote_P) *wrote_P = false;
    return widem::ilrma_loss(ctx, X, W, Tb, V, domain, eps, loss, ws, B, M, F, T, K, dtype, (hipStream_t)stream, nu);
  }
  hipStream_t st = (hipStream_t)stream;
  const WsLayout L = ws_layout(B, M, F, T, K, dtype);
  double* lpart = (double*)((char*)ws + L.lpart);
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    NmfArgs<R> a;
    a.d = Dims{B, F, T, K};
    a.fp = flat_loss(F, T);
    a.eps = (R)eps;
    a.p1 = make_pow(1.0);
    a.nu = (R)nu;
    const PowSpec p2d = make_pow(2.0 / domain);
    const bool d2 = p2d.mode == POW_ID, k4 = K <= KU;
    static const bool wide = lab_int("ASSX_WIDE_K", 1) != 0;
    if (!k4 && wide) {  // n_basis > 4: bin-batched evaluation (ilrma_loss_wide_kernel)
      const dim3 gw(blocks_for(T, 64), blocks_for(F, WIDE_FB), B);
      const int nw = (int)(gw.x * gw.y), lsw = nw + F;
      if ((size_t)B * lsw * sizeof(double) <= L.small - L.lpart) {
        if (nu >= 0.0)
          hipLaunchKernelGGL((ilrma_loss_wide_kernel<R, MM, true, true>), gw, dim3(64), 0, st, (const Cx<R>*)X,
                             (const Cx<R>*)W, (const R*)Tb, (const R*)V, lpart, lsw, a.d, a.eps, p2d, (R)nu, (R*)P_out);
        else if (d2)
          hipLaunchKernelGGL((ilrma_loss_wide_kernel<R, MM, true>), gw, dim3(64), 0, st, (const Cx<R>*)X,
                             (const Cx<R>*)W, (const R*)Tb, (const R*)V, lpart, lsw, a.d, a.eps, p2d, (R)0, (R*)P_out);
        else
          hipLaunchKernelGGL((ilrma_loss_wide_kernel<R, MM, false>), gw, dim3(64), 0, st, (const Cx<R>*)X,
                             (const Cx<R>*)W, (const R*)Tb, (const R*)V, lpart, lsw, a.d, a.eps, p2d, (R)0, (R*)P_out);
        ASSX_LAUNCH_CHECK(ctx, "ilrma_loss_wide_kernel");
        if (wrote_P) *wrote_P = P_out != nullptr;
        hipLaunchKernelGGL((logdet_kernel<R, MM>), dim3(blocks_for((size_t)B * F, 64)), dim3(64), 0, st,
                           (const Cx<R>*)W, lpart, B, F, T, lsw, nw);
        ASSX_LAUNCH_CHECK(ctx, "logdet_kernel");
        hipLaunchKernelGGL((sum_reduce_kernel<double, double>), dim3(B), dim3(REDUCE_THREADS), 0, st,
                           (const double*)lpart, loss, (size_t)lsw, 1.0);
        ASSX_LAUNCH_CHECK(ctx, "sum_reduce_kernel");
        return 0;
      }
    }
    const int lstride = a.fp.G + F;  // [G data-term partials | F log-det terms] per utterance
    const dim3 grid(a.fp.G, B), blk(64);
#define LOSS_LAUNCH(K4V, D2V, DXV, DWV, MW, TDV)                                                                \
  hipLaunchKernelGGL((loss_stream_kernel<R, MM, K4V, D2V, DXV, DWV, MW, TDV>), grid, blk, 0, st, (const Cx<R>*)X, \
                     (const Cx<R>*)W, (const R*)Tb, (const R*)V, lpart, lstride, a, p2d)
    constexpr size_t vlds = (size_t)VDMA_SLOTS * VTileDma<R, MM * KU>::TILE_BYTES;
#define LOSS_VD(D2V, MW, TDV)                                                                                    \
  hipLaunchKernelGGL((loss_stream_vd_kernel<R, MM, D2V, 2, MW, TDV>), grid, blk, vlds, st, (const Cx<R>*)X,      \
                     (const Cx<R>*)W, (const R*)Tb, (const R*)V, lpart, lstride, a, p2d)
#if ASSX_LAB
    static const int vdma = lab_int("ASSX_LOSS_VDMA", 1);  // 0: the plain vector-load forms of rounds 1-2 (A/B runs)
    if (k4 && !vdma) {
      if (nu >= 0.0) LOSS_LAUNCH(true, true, 4, 1, 1, true);
      else if (d2) LOSS_LAUNCH(true, true, 4, 1, 2, false);
      else LOSS_LAUNCH(true, false, 2, 1, 1, false);
    } else
#endif
    if (k4) {
      if (nu >= 0.0) LOSS_VD(true, 2, true);
      else if (d2) LOSS_VD(true, 2, false);
      else LOSS_VD(false, 1, false);
    } else if (nu >= 0.0) LOSS_LAUNCH(false, true, 4, 1, 1, true);
    else if (d2) LOSS_LAUNCH(false, true, 4, 1, 2, false);
    else LOSS_LAUNCH(false, false, 2, 1, 1, false);
#undef LOSS_VD
#undef LOSS_LAUNCH
    ASSX_LAUNCH_CHECK(ctx, "loss_stream_kernel");
    hipLaunchKernelGGL((logdet_kernel<R, MM>), dim3(blocks_for((size_t)B * F, 64)), dim3(64), 0, st, (const Cx<R>*)W,
                       lpart, B, F, T, lstride, a.fp.G);
    ASSX_LAUNCH_CHECK(ctx, "logdet_kernel");
    hipLaunchKernelGGL((sum_reduce_kernel<double, double>), dim3(B), dim3(REDUCE_THREADS), 0, st, (const double*)lpart,
                       loss, (size_t)lstride, 1.0);
    ASSX_LAUNCH_CHECK(ctx, "sum_reduce_kernel");
    return 0;
  });
}

int assx_ilrma_loss(assx_ctx* ctx, const void* X, const void* W, const void* Tb, const void* V, double domain,
                    double eps, double* loss, void* ws, int B, int M, int F, int T, int K, int dtype, void* stream) {
  return ilrma_loss_impl(ctx, "assx_ilrma_loss", X, W, Tb, V, domain, -1.0, eps, loss, ws, B, M, F, T, K, dtype, stream);
}

// ---- t-ILRMA (ilrma.py:713-1020) -------------------------------------------------------------
int assx_tilrma_loss(assx_ctx* ctx, const void* X, const void* W, const void* Tb, const void* V, double nu, double eps,
                     double* loss, void* ws, int B, int M, int F, int T, int K, int dtype, void* stream) {
  ASSX_REQUIRE(ctx, ctx == nullptr || nu > 0.0, ASSX_E_ARG, "nu must be > 0, got %g", nu);
  return ilrma_loss_impl(ctx, "assx_tilrma_loss", X, W, Tb, V, 2.0, nu, eps, loss, ws, B, M, F, T, K, dtype, stream);
}

int assx_tilrma_source_update(assx_ctx* ctx, const void* X, const void* W, void* Tb, void* V, double nu, double eps,
                              void* ws, int B, int M, int F, int T, int K, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && Tb && V && ws, ASSX_E_NULL, "assx_tilrma_source_update: NULL array");
  ASSX_REQUIRE(ctx, K >= 1, ASSX_E_ARG, "n_basis must be >= 1, got %d", K);
  ASSX_REQUIRE(ctx, nu >= 0.0, ASSX_E_ARG, "nu must be >= 0, got %g", nu);
  hipStream_t st = (hipStream_t)stream;
  if (widem::handles(M)) return widem::tilrma_source_update(ctx, X, W, Tb, V, nu, eps, ws, B, M, F, T, K, dtype, st);
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    const PowSpec p2 = make_pow(0.5);
    FlatPart fp;
    static const bool wide = lab_int("ASSX_WIDE_K", 1) != 0;
    if (K > KU && wide) {  // n_basis > 4: P = |W x|^2 once, then the batched tNMF-type update on the matrix cores
      const WsLayout L = ws_layout(B, MM, F, T, K, dtype);
      R* pw = (R*)((char*)ws + L.map);
      int rx = nmf_update_xfed(ctx, ASSX_NMF_T_RAW, 2.0, nu, eps, X, W, Tb, V, (char*)ws + L.nmf, B, MM, F, T, K, dtype, st);
      if (rx != ASSX_E_UNSUPPORTED) return rx;
      hipLaunchKernelGGL((demix_power_map_kernel<R, MM>), dim3(blocks_for(T, 256), F, B), dim3(256), 0, st,
                         (const Cx<R>*)X, (const Cx<R>*)W, pw, Dims{B, F, T, 0});
      ASSX_LAUNCH_CHECK(ctx, "demix_power_map_kernel");
      NmfGroupScope grp(ctx, MM);
      return assx_nmf_update_ex(ctx, ASSX_NMF_T_RAW, 2.0, nu, eps, pw, Tb, V, (char*)ws + L.nmf, B * MM, F, T, K, dtype,
                                stream);
    }
    int rc = run_basis_partial<R, MM>(ctx, X, W, Tb, V, 2.0, eps, ws, B, F, T, K, st, &fp, nu);
    if (rc) return rc;
    hipLaunchKernelGGL((basis_stream_finalize_kernel<R>), dim3(blocks_for((size_t)B * MM * F * K, 256)), dim3(256), 0,
                       st, (const R*)ws, (R*)Tb, B, MM, F, K, fp, (R)eps, p2, ~0u);
    ASSX_LAUNCH_CHECK(ctx, "basis_stream_finalize_kernel");
    rc = run_act_partial<R, MM>(ctx, X, W, Tb, V, 2.0, eps, ws, B, F, T, K, st, &fp, nu);
    if (rc) return rc;
    hipLaunchKernelGGL((act_stream_finalize_kernel<R>), dim3((unsigned)((size_t)B * MM * K * tblocks(T))), dim3(256), 0, st,
                       (const R*)ws, (R*)V, B, MM, F, K, T, fp, (R)eps, p2, ~0u);
    ASSX_LAUNCH_CHECK(ctx, "act_stream_finalize_kernel");
    return 0;
  });
}

int assx_tilrma_spatial_update(assx_ctx* ctx, const void* X, void* W, const void* Tb, const void* V, double nu,
                               double eps, void* Xi, const void* C, double* power_bins, int32_t* status, void* ws,
                               int B, int M, int F, int T, int K, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && Tb && V && Xi && ws, ASSX_E_NULL, "assx_tilrma_spatial_update: NULL array");
  ASSX_REQUIRE(ctx, K >= 1, ASSX_E_ARG, "n_basis must be >= 1, got %d", K);
  ASSX_REQUIRE(ctx, nu >= 0.0, ASSX_E_ARG, "nu must be >= 0, got %g", nu);
  ASSX_REQUIRE(ctx, (C == nullptr) == (power_bins == nullptr), ASSX_E_ARG,
               "assx_tilrma_spatial_update: C and power_bins must be given together");
  hipStream_t st = (hipStream_t)stream;
  if (widem::handles(M))
    return widem::tilrma_spatial_update(ctx, X, W, Tb, V, nu, eps, Xi, C, power_bins, status, ws, B, M, F, T, K, dtype, st);
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    static const bool wide = lab_int("ASSX_WIDE_K", 1) != 0;
    if (K > KU && wide)
      hipLaunchKernelGGL((tilrma_xi_wide_kernel<R, MM>), dim3(blocks_for(T, 64), blocks_for(F, WIDE_FB), B), dim3(64), 0,
                         st, (const Cx<R>*)X, (const Cx<R>*)W, (const R*)Tb, (const R*)V, (R*)Xi, Dims{B, F, T, K},
                         (R)nu, (R)eps);
    else
      hipLaunchKernelGGL((tilrma_xi_kernel<R, MM>), dim3(blocks_for(T, 256), F, B), dim3(256), 0, st, (const Cx<R>*)X,
                         (const Cx<R>*)W, (const R*)Tb, (const R*)V, (R*)Xi, Dims{B, F, T, K}, (R)nu, (R)eps);
    ASSX_LAUNCH_CHECK(ctx, "tilrma_xi_kernel");
    FlatPart fp;
    // Xi is used as is (the reference does not floor it): eps = 0 in the covariance pass
    int rc = run_cov_partial<R, MM>(ctx, WK_NFT, X, Xi, nullptr, nullptr, 1, 2.0, 0.0, ws, B, F, T, st, &fp);
    if (rc) return rc;
    // inverse without a condition-number guard (ilrma.py:968-976); the normaliser is floored at eps
    return run_ip<R, MM>(ctx, nullptr, ws, fp, T, W, C, power_bins, INFINITY, status, B, F, st, eps);
  });
}

// ---- (f4) other callers of covariance-accumulate + IP ------------------------------------------------------------
int assx_idlma_space_update(assx_ctx* ctx, const void* X, void* W, const void* dnn_output, double domain, double eps,
                            double threshold, void* R_scratch, int32_t* status, void* ws, int B, int M, int F, int T,
                            int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && dnn_output && ws, ASSX_E_NULL, "assx_idlma_space_update: NULL array");
  ASSX_REQUIRE(ctx, domain > 0.0, ASSX_E_ARG, "domain must be > 0, got %g", domain);
  ASSX_REQUIRE(ctx, domain == 2.0 || R_scratch, ASSX_E_NULL, "assx_idlma_space_update: R_scratch is needed for domain != 2");
  hipStream_t st = (hipStream_t)stream;
  if (widem::handles(M)) {  // the same two steps on the wide-channel kernels
    const void* r = dnn_output;
    if (domain != 2.0) {
      const size_t n = (size_t)B * M * F * T;
      if (dtype == ASSX_F64)
        hipLaunchKernelGGL((pow_map_kernel<double>), dim3(blocks_for(n, 256)), dim3(256), 0, st, (const double*)dnn_output,
                           (double*)R_scratch, n, make_pow(2.0 / domain));
      else if (dtype == ASSX_F32)
        hipLaunchKernelGGL((pow_map_kernel<float>), dim3(blocks_for(n, 256)), dim3(256), 0, st, (const float*)dnn_output,
                           (float*)R_scratch, n, make_pow(2.0 / domain));
      else
        return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
      ASSX_LAUNCH_CHECK(ctx, "pow_map_kernel");
      r = R_scratch;
    }
    return widem::weighted_ip(ctx, X, r, eps, threshold, 0.0, W, status, ws, B, M, F, T, dtype, st);
  }
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    const void* r = dnn_output;
    if (domain != 2.0) {
      const size_t n = (size_t)B * MM * F * T;
      hipLaunchKernelGGL((pow_map_kernel<R>), dim3(blocks_for(n, 256)), dim3(256), 0, st, (const R*)dnn_output,
                         (R*)R_scratch, n, make_pow(2.0 / domain));
      ASSX_LAUNCH_CHECK(ctx, "pow_map_kernel");
      r = R_scratch;
    }
    FlatPart fp;
    int rc = run_cov_partial<R, MM>(ctx, WK_NFT, X, r, nullptr, nullptr, 1, 2.0, eps, ws, B, F, T, st, &fp);
    if (rc) return rc;
    return run_ip<R, MM>(ctx, nullptr, ws, fp, T, W, nullptr, nullptr, threshold, status, B, F, st);
  });
}

int assx_fastmnmf_update_diagonalizer(assx_ctx* ctx, const void* X, void* Q, const void* Lambda, const void* g,
                                      double eps, double threshold, void* R_scratch, int32_t* status, void* ws, int B,
                                      int M, int N, int F, int T, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && Q && Lambda && g && R_scratch && ws, ASSX_E_NULL, "assx_fastmnmf_update_diagonalizer: NULL array");
  ASSX_REQUIRE(ctx, N >= 1, ASSX_E_ARG, "n_sources must be >= 1, got %d", N);
  hipStream_t st = (hipStream_t)stream;
  if (widem::handles(M)) {
    if (dtype == ASSX_F64)
      hipLaunchKernelGGL((mix_variance_kernel<double>), dim3(blocks_for(T, 256), F, B), dim3(256), 0, st,
                         (const double*)Lambda, (const double*)g, (double*)R_scratch, M, N, F, T);
    else if (dtype == ASSX_F32)
      hipLaunchKernelGGL((mix_variance_kernel<float>), dim3(blocks_for(T, 256), F, B), dim3(256), 0, st,
                         (const float*)Lambda, (const float*)g, (float*)R_scratch, M, N, F, T);
    else
      return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
    ASSX_LAUNCH_CHECK(ctx, "mix_variance_kernel");
    return widem::weighted_ip(ctx, X, R_scratch, eps, threshold, eps, Q, status, ws, B, M, F, T, dtype, st);
  }
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    hipLaunchKernelGGL((mix_variance_kernel<R>), dim3(blocks_for(T, 256), F, B), dim3(256), 0, st, (const R*)Lambda,
                       (const R*)g, (R*)R_scratch, MM, N, F, T);
    ASSX_LAUNCH_CHECK(ctx, "mix_variance_kernel");
    FlatPart fp;
    int rc = run_cov_partial<R, MM>(ctx, WK_NFT, X, R_scratch, nullptr, nullptr, 1, 2.0, eps, ws, B, F, T, st, &fp);
    if (rc) return rc;
    // condition-number guard AND the eps floor on the normaliser (mnmf.py:876-884)
    return run_ip<R, MM>(ctx, nullptr, ws, fp, T, Q, nullptr, nullptr, threshold, status, B, F, st, eps);
  });
}

int assx_auxiva_weights(assx_ctx* ctx, const void* X, const void* W, int kind, double eps, void* r, double* loss,
                        void* ws, int B, int M, int F, int T, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && r && ws, ASSX_E_NULL, "assx_auxiva_weights: NULL array");
  ASSX_REQUIRE(ctx, kind == ASSX_IVA_LAPLACE || kind == ASSX_IVA_GAUSS, ASSX_E_ARG, "bad AuxIVA kind %d", kind);
  if (widem::handles(M)) return widem::auxiva_weights(ctx, X, W, kind, eps, r, loss, ws, B, M, F, T, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  const WsLayout L = ws_layout(B, M, F, T, 1, dtype);
  double* lpart = (double*)((char*)ws + L.lpart);
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    int FS, fchunk;
    f_split(B, F, T, &FS, &fchunk);
    // one loss partial per (source, frame block) -- for T a multiple of 64 exactly the blocks of the flattened (n, t)
    // index the separate finalize kernel used
    const int TBk = (int)blocks_for(T, WAVE), nblk = MM * TBk;
    const int lstride = nblk + F;
    // laboratory builds, ASSX_AUX_FOLD=1: finalize, log-det terms and loss sum inside the pass (3 launches per iteration instead of 4, 5 -> 3
    // with the loss).  OFF by default: measured SLOWER on MI355X -- config 3: 36.4 against 32.5 us per iteration, 40.6
    // against 39.1 with the loss (profiles/r04_auxiva_fold.txt) -- write-through records + ticket + read-back are three
    // memory round trips across the XCDs, more than a dependent launch (2.7 us) plus the 2 us finalize.  Read on every
    // call (tests run both forms in one process).
    int* tickets = nullptr;
#if ASSX_LAB
    const int fold = lab_int("ASSX_AUX_FOLD", 0);
    if (fold && FS > 1) {
      const int trc = ensure_tickets(ctx, (size_t)B * TBk + B, st, &tickets);
      if (trc) return trc;  // the hipError_t of the allocation, message in ctx
    }
#endif
    const bool folded = tickets != nullptr;
    hipLaunchKernelGGL((auxiva_stat_partial_kernel<R, MM>), dim3(TBk, FS, B), dim3(256), 0, st, (const Cx<R>*)X,
                       (const Cx<R>*)W, (R*)ws, Dims{B, F, T, 0}, FS, fchunk, tickets, (R*)r,
                       loss ? lpart : (double*)nullptr, folded ? loss : (double*)nullptr, kind, (R)eps, lstride);
    ASSX_LAUNCH_CHECK(ctx, "auxiva_stat_partial_kernel");
    if (!folded) {
      hipLaunchKernelGGL((auxiva_stat_finalize_kernel<R>), dim3(nblk, B), dim3(256), 0, st, (const R*)ws, (R*)r,
                         loss ? lpart : (double*)nullptr, MM, F, T, FS, kind, (R)eps, lstride, TBk);
      ASSX_LAUNCH_CHECK(ctx, "auxiva_stat_finalize_kernel");
    }
    if (loss && !folded) {
      hipLaunchKernelGGL((logdet_kernel<R, MM>), dim3(blocks_for((size_t)B * F, 64)), dim3(64), 0, st, (const Cx<R>*)W,
                         lpart, B, F, T, lstride, nblk);
      ASSX_LAUNCH_CHECK(ctx, "logdet_kernel");
      hipLaunchKernelGGL((sum_reduce_kernel<double, double>), dim3(B), dim3(REDUCE_THREADS), 0, st,
                         (const double*)lpart, loss, (size_t)lstride, 1.0);
      ASSX_LAUNCH_CHECK(ctx, "sum_reduce_kernel");
    }
    return 0;
  });
}

int assx_auxiva_spatial_update(assx_ctx* ctx, int spatial, int pair_m, int pair_n, const void* X, void* W, const void* r,
                               double eps, double threshold, void* U_out, int32_t* status, void* ws, int B, int M, int F,
                               int T, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && r && ws, ASSX_E_NULL, "assx_auxiva_spatial_update: NULL array");
  if (widem::handles(M))
    return widem::auxiva_spatial_update(ctx, spatial, pair_m, pair_n, X, W, r, eps, threshold, U_out, status, ws, B, M, F, T,
                                        dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    FlatPart fp;
    int rc = run_cov_partial<R, MM>(ctx, WK_NT, X, r, nullptr, nullptr, 1, 2.0, eps, ws, B, F, T, st, &fp);
    if (rc) return rc;
    if (U_out) {
      hipLaunchKernelGGL((cov_stream_finalize_kernel<R, MM>), dim3(blocks_for((size_t)B * MM * F * MM * MM, 256)),
                         dim3(256), 0, st, (const R*)ws, (Cx<R>*)U_out, B, MM, F, fp, (R)(1.0 / (double)T));
      ASSX_LAUNCH_CHECK(ctx, "cov_stream_finalize_kernel");
    }
    if (spatial == ASSX_SPATIAL_ISS) return run_iss<R, MM>(ctx, nullptr, ws, fp, T, W, nullptr, nullptr, B, F, st);
    if (spatial == ASSX_SPATIAL_IP2) {
      if (!(pair_m >= 0 && pair_m < MM && pair_n >= 0 && pair_n < MM && pair_m != pair_n))
        return fail(ctx, ASSX_E_ARG, "bad update pair (%d, %d) for %d sources", pair_m, pair_n, MM);
      return run_ip2<R, MM>(ctx, nullptr, ws, fp, T, W, nullptr, nullptr, threshold, status, B, F, pair_m, pair_n, st);
    }
    return run_ip<R, MM>(ctx, nullptr, ws, fp, T, W, nullptr, nullptr, threshold, status, B, F, st);
  });
}

int assx_projection_back_scale(assx_ctx* ctx, const void* X, const void* W, int ref, void* scale, int32_t* status,
                               void* ws, int B, int M, int F, int T, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, M, F, T);
  ASSX_REQUIRE(ctx, X && W && scale && ws, ASSX_E_NULL, "assx_projection_back_scale: NULL array");
  ASSX_REQUIRE(ctx, ref >= 0 && ref < M, ASSX_E_ARG, "reference_id %d out of range for %d channels", ref, M);
  if (widem::handles(M)) return widem::projection_back_scale(ctx, X, W, ref, scale, status, ws, B, M, F, T, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    int TS, tchunk;
    t_split(B, F, T, &TS, &tchunk);
    hipLaunchKernelGGL((pb_stat_partial_kernel<R, MM, true>), dim3((unsigned)F * TS, B), dim3(64), 0, st,
                       (const Cx<R>*)X, (const Cx<R>*)W, (const Cx<R>*)nullptr, ref, (R*)ws, Dims{B, F, T, 0}, TS,
                       tchunk);
    ASSX_LAUNCH_CHECK(ctx, "pb_stat_partial_kernel");
    hipLaunchKernelGGL((pb_solve_kernel<R, MM>), dim3(blocks_for((size_t)B * F, 64)), dim3(64), 0, st, (const R*)ws,
                       (Cx<R>*)scale, status, B, F, TS);
    ASSX_LAUNCH_CHECK(ctx, "pb_solve_kernel");
    return 0;
  });
}

int assx_projection_back(assx_ctx* ctx, const void* Y, const void* reference, void* scale, int32_t* status, void* ws,
                         int B, int N, int F, int T, int dtype, void* stream) {
  CHECK_COMMON(ctx, B, N, F, T);
  ASSX_REQUIRE(ctx, Y && reference && scale && ws, ASSX_E_NULL, "assx_projection_back: NULL array");
  if (widem::handles(N)) return widem::projection_back(ctx, Y, reference, scale, status, B, N, F, T, dtype, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_rm(ctx, dtype, N, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MM = decltype(mt)::value;
    int TS, tchunk;
    t_split(B, F, T, &TS, &tchunk);
    hipLaunchKernelGGL((pb_stat_partial_kernel<R, MM, false>), dim3((unsigned)F * TS, B), dim3(64), 0, st,
                       (const Cx<R>*)Y, (const Cx<R>*)nullptr, (const Cx<R>*)reference, 0, (R*)ws, Dims{B, F, T, 0}, TS,
                       tchunk);
    ASSX_LAUNCH_CHECK(ctx, "pb_stat_partial_kernel");
    hipLaunchKernelGGL((pb_solve_kernel<R, MM>), dim3(blocks_for((size_t)B * F, 64)), dim3(64), 0, st, (const R*)ws,
                       (Cx<R>*)scale, status, B, F, TS);
    ASSX_LAUNCH_CHECK(ctx, "pb_solve_kernel");
    return 0;
  });
}

}  // extern "C"

#if STREAM_TRACE
// timing-experiment builds only (tools/probes/stream_trace.py); not part of include/assx.h
extern "C" int assx_debug_stream_trace(unsigned long long* host, int clear) {
  using assx::g_stream_trace;
  constexpr size_t N = 8 * 4096;
  if (clear) {
    static unsigned long long zeros[N];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stream_trace), zeros, sizeof(zeros));
  }
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stream_trace), sizeof(unsigned long long) * N);
}
#endif
