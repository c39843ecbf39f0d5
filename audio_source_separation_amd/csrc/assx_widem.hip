// Wide-channel path (5 <= M <= 8): see assx_widem.hpp for the design.  Reference citations are next to the entry
// points in include/assx.h; the arithmetic contract (floors, exponents, Gauss-Seidel order) is the M <= 4 path's.
#include "assx_widem.hpp"
#include "assx_widem_cov.hpp"
#include "assx_widem_rt.hpp"
#include "assx_group_linalg.hpp"
#include "assx_nmf_internal.hpp"
#include "assx_partition.hpp"

namespace assx {
namespace widem {

constexpr int RB = 512;       // partial sums per reduced row (fixed: the summation order never depends on the runtime)
constexpr int RED_THREADS = 256;

inline unsigned nblk(size_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

struct Ws {
  size_t map0, map1, u, lpart, nmf, tmp, rec, total;
};

// Flat partition of src_cov_kernel (assx_widem_cov.hpp): items = (utterance, bin, 64-frame block), two workgroups'
// worth of ranges per CU (one or two are resident, depending on M and the precision: either way the ranges are equal,
// so there is no tail).  A fixed function of the geometry -- never an occupancy query -- so the summation order is too.
// ASSX_G forces the number of ranges (tests: long ranges on small inputs), read on every call.
static FlatPart flat_src_cov(int B, int F, int T) {
  const int tbk = (T + WAVE - 1) / WAVE;
  const int forced = knob_int("ASSX_G", 0);
  return make_flat(B, (long long)F * tbk, tbk, forced > 0 ? forced : 512);
}
static Ws layout(int B, int M, int F, int T, int K, int dtype) {
  const size_t r = dtype == ASSX_F64 ? 8 : 4;
  const int Kc = K < 1 ? 1 : K;
  Ws w;
  size_t off = 0;
  w.map0 = off;  // P (real) -- or, together with map1, Y (complex)
  off += align_up((size_t)B * M * F * T * r, 256);
  w.map1 = off;  // R (real)
  off += align_up((size_t)B * M * F * T * r, 256);
  w.u = off;
  off += align_up((size_t)B * M * F * M * M * 2 * r, 256);
  w.lpart = off;
  off += align_up((size_t)B * ((size_t)M * RB + (size_t)M * ((T + 255) / 256) + F + 64) * 8, 256);
  w.rec = off;  // src_cov_kernel records [g][slot][n][M*M]; ahead of the regions whose size depends on n_basis
  const FlatPart fc = flat_src_cov(B, F, T);
  // only the compile-time-M covariance kernels (M <= 8) write these records: the run-time-M path (9 <= M <= 32) never
  // does, and at M = 32 the region would be ~250 MB of scratch per utterance nobody touches
  if (M <= 8) off += align_up((size_t)fc.G * fc.S * M * M * M * r, 256);
  w.nmf = off;
  off += align_up(assx_nmf_workspace_bytes(B * M, F, T, Kc, dtype), 256);
  w.tmp = off;
  off += align_up(((size_t)B * M * F * Kc + (size_t)B * M * Kc * T) * r, 256);
  w.total = off;
  return w;
}
size_t workspace_bytes(int B, int M, int F, int T, int K, int dtype) { return layout(B, M, F, T, K, dtype).total; }

// ------------------------------------------------------------------------------------------------------------------
// y = W x per (f, t); W_f staged in LDS (M*M complex), x in registers.  Writes Y (optionally scaled) and / or |y|^2.
// ------------------------------------------------------------------------------------------------------------------
template <typename R, int M>
__global__ void __launch_bounds__(256) demix_map_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W,
                                                       const Cx<R>* __restrict__ scale, Cx<R>* __restrict__ Y,
                                                       R* __restrict__ P, int F, int T) {
  __shared__ Cx<R> w[M * M];
  const int f = blockIdx.y, b = blockIdx.z;
  if (threadIdx.x < M * M) w[threadIdx.x] = W[((size_t)b * F + f) * (M * M) + threadIdx.x];
  __syncthreads();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const size_t FT = (size_t)F * T, base = (size_t)b * M * FT + (size_t)f * T + t;
  Cx<R> x[M];
#pragma unroll
  for (int m = 0; m < M; ++m) x[m] = X[base + m * FT];
#pragma unroll
  for (int n = 0; n < M; ++n) {
    Cx<R> s = cmake<R>(0, 0);
#pragma unroll
    for (int m = 0; m < M; ++m) cfma(s, w[n * M + m], x[m]);
    if (P) P[base + n * FT] = cabs2(s);
    if (Y) {
      if (scale) s = cmul(s, scale[((size_t)b * M + n) * F + f]);
      Y[base + n * FT] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Weighted covariance of one bin per workgroup, one wave per source (a source's packed Hermitian sums fill the wave's
// registers: 36 complex at M = 8), lanes own frames; the N waves walk X_f together (HBM once, L1 for the others: with
// 4 waves taking two sources each the bin's rows were streamed from HBM twice -- 461 us at M = 8, the traffic floor).  U[b,n,f] = (1/T) sum_t x x^H / max(r_n, eps), dense output.
// ------------------------------------------------------------------------------------------------------------------
enum { RK_NONE = 0, RK_NT = 1, RK_NFT = 2 };

template <typename R, int M>
__global__ void __launch_bounds__(512) cov_bin_kernel(const Cx<R>* __restrict__ X, const R* __restrict__ r, int r_kind,
                                                     int N, R eps, Cx<R>* __restrict__ U, int F, int T, R inv_T) {
  constexpr int NH = M * (M + 1) / 2;
  const int f = blockIdx.x, b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t FT = (size_t)F * T;
  const Cx<R>* xb = X + (size_t)b * M * FT + (size_t)f * T;
  const int nwaves = blockDim.x >> 6;  // = N for the weighted forms: every source's wave walks X_f at the same time, so
                                       // the bin's rows come from HBM once and from L1 for the other N - 1 waves
  for (int n = wave; n < N; n += nwaves) {
    const R* rn = nullptr;
    if (r_kind == RK_NT) rn = r + ((size_t)b * N + n) * T;
    else if (r_kind == RK_NFT) rn = r + ((size_t)b * N + n) * FT + (size_t)f * T;
    R ar[NH], ai[NH];
#pragma unroll
    for (int q = 0; q < NH; ++q) ar[q] = ai[q] = 0;
    // one block of register prefetch: the loads of frame t + 64 travel while frame t is accumulated (without it the
    // 2 waves per SIMD exposed an L2 round trip per block: 461 us at M = 8, config-4 bins / frames)
    Cx<R> xn[M];
    R rnext = (R)1;
    if (lane < T) {
#pragma unroll
      for (int m = 0; m < M; ++m) xn[m] = xb[m * FT + lane];
      if (rn) rnext = rn[lane];
    }
    for (int t = lane; t < T; t += WAVE) {
      Cx<R> x[M];
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = xn[m];
      const R rcur = rnext;
      if (t + WAVE < T) {
#pragma unroll
        for (int m = 0; m < M; ++m) xn[m] = xb[m * FT + t + WAVE];
        if (rn) rnext = rn[t + WAVE];
      }
      const R wgt = rn ? fast_rcp(floor_eps<R>(rcur, eps)) : (R)1;
      int q = 0;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const R sx = wgt * x[i].x, sy = wgt * x[i].y;
#pragma unroll
        for (int j = i; j < M; ++j, ++q) {  // x_i conj(x_j)
          ar[q] = fma(sx, x[j].x, ar[q]);
          ar[q] = fma(sy, x[j].y, ar[q]);
          if (j != i) {
            ai[q] = fma(sy, x[j].x, ai[q]);
            ai[q] = fma(-sx, x[j].y, ai[q]);
          }
        }
      }
    }
    Cx<R>* un = U + (((size_t)b * N + n) * F + f) * (M * M);
    int q = 0;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = i; j < M; ++j, ++q) {
        const R re = wave_allreduce_sum<R>(ar[q]) * inv_T;
        const R im = (j != i) ? wave_allreduce_sum<R>(ai[q]) * inv_T : (R)0;
        if (lane == 0) {
          un[i * M + j] = cmake<R>(re, im);
          if (j != i) un[j * M + i] = cmake<R>(re, -im);
        }
      }
  }
}

// R[bn,f,t] = (sum_k Tb[bn,f,k] V[bn,k,t])^(2/domain)   (floored by the reader, ilrma.py:499-509)
template <typename R>
__global__ void __launch_bounds__(256) variance_map_kernel(const R* __restrict__ Tb, const R* __restrict__ V,
                                                          R* __restrict__ Rm, int F, int T, int K, PowSpec p2d) {
  const int f = blockIdx.y, bn = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const R* tb = Tb + ((size_t)bn * F + f) * K;
  const R* vb = V + (size_t)bn * K * T + t;
  R tv = 0;
  for (int k = 0; k < K; ++k) tv = fma(tb[k], vb[(size_t)k * T], tv);
  Rm[((size_t)bn * F + f) * T + t] = powspec<R>(tv, p2d);
}

// t-ILRMA: Xi = (nu R + 2 P) / (nu + 2) in place of R (R floored first, Xi itself is not: ilrma.py:945-966)
template <typename R>
__global__ void __launch_bounds__(256) xi_map_kernel(const R* __restrict__ P, R* __restrict__ Rm, size_t count, R nu,
                                                    R eps) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const R r = floor_eps<R>(Rm[i], eps);
  Rm[i] = fma(nu, r, (R)2 * P[i]) / (nu + (R)2);
}

// ------------------------------------------------------------------------------------------------------------------
// deterministic reductions over the maps
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double s, double* sm) {
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = RED_THREADS / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  return sm[0];
}

// out[row * ostride + blockIdx.x] = sum over this block's chunk of a[row][.]
//   TERM 1: P/R + log R of two maps (ilrma.py:648-677);  TERM 2: (1 + nu/2) log(1 + (2/nu) P/R) + log R (ilrma.py:991-1018)
template <typename R, int TERM>
__global__ void __launch_bounds__(RED_THREADS) map_sum_kernel(const R* __restrict__ a, const R* __restrict__ rm,
                                                             double* __restrict__ out, size_t elems, int ostride,
                                                             R eps, R nu) {
  __shared__ double sm[RED_THREADS];
  const size_t row = blockIdx.y;
  const size_t chunk = (elems + gridDim.x - 1) / gridDim.x;
  const size_t i0 = (size_t)blockIdx.x * chunk, i1 = i0 + chunk < elems ? i0 + chunk : elems;
  const R* pa = a + row * elems;
  const R* pr = TERM ? rm + row * elems : nullptr;
  double s = 0.0;
  for (size_t i = i0 + threadIdx.x; i < i1; i += RED_THREADS) {
    if (TERM == 1) {
      const R rr = floor_eps<R>(pr[i], eps);
      s += (double)(pa[i] / rr) + log((double)rr);
    } else if (TERM == 2) {
      const R rr = floor_eps<R>(pr[i], eps);
      s += (1.0 + 0.5 * (double)nu) * log1p((2.0 / (double)nu) * (double)(pa[i] / rr)) + log((double)rr);
    } else {
      s += (double)pa[i];
    }
  }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) out[row * (size_t)ostride + blockIdx.x] = s;
}

// out[row] = scale * sum_{i < count} in[row * stride + i], fixed order
template <typename TO>
__global__ void __launch_bounds__(RED_THREADS) row_sum_kernel(const double* __restrict__ in, TO* __restrict__ out,
                                                             int stride, int count, double scale) {
  __shared__ double sm[RED_THREADS];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += RED_THREADS) s += in[(size_t)blockIdx.x * stride + i];
  s = block_sum(s, sm);
  if (threadIdx.x == 0) out[blockIdx.x] = (TO)(s * scale);
}

template <typename R, int M>
__global__ void __launch_bounds__(64) logdet_kernel(const Cx<R>* __restrict__ W, double* __restrict__ lpart, int B, int F,
                                                   int T, int lstride, int offset) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * F) return;
  const int b = idx / F, f = idx % F;
  lpart[(size_t)b * lstride + offset + f] = neg2T_logabsdet<M, R>(W, (size_t)idx, T);
}

// AuxIVA statistic from the power map: s[bn,t] = sum_f P[bn,f,t] (ascending f); r = sqrt(s) | s / F; the loss data
// term of the block goes to lpart[b][n * TB + blockIdx.x]  (iva.py:489-491, 604-619, 722-724, 783-802)
template <typename R>
__global__ void __launch_bounds__(RED_THREADS) aux_stat_kernel(const R* __restrict__ P, R* __restrict__ r,
                                                              double* __restrict__ lpart, int kind, R eps, int N, int F,
                                                              int T, int lstride) {
  __shared__ double sm[RED_THREADS];
  const int bn = blockIdx.y, b = bn / N, n = bn - b * N;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double term = 0.0;
  if (t < T) {
    const R* p = P + (size_t)bn * F * T + t;
    R s = 0;
    for (int f = 0; f < F; ++f) s += p[(size_t)f * T];
    R rv;
    if (kind == ASSX_IVA_LAPLACE) {
      rv = sqrt(s);
      term = 2.0 * (double)rv;
    } else {
      rv = s / (R)F;
      term = (double)F * log((double)floor_eps<R>(rv, eps));
    }
    r[(size_t)bn * T + t] = rv;
  }
  if (lpart) {
    term = block_sum(term, sm);
    if (threadIdx.x == 0) lpart[(size_t)b * lstride + (size_t)n * gridDim.x + blockIdx.x] = term;
  }
}

// part[b][n][f] = Re(w_n C_f w_n^H)  (run-time M)
template <typename R>
__global__ void __launch_bounds__(256) power_cov_kernel(const Cx<R>* __restrict__ C, const Cx<R>* __restrict__ W,
                                                       double* __restrict__ part, int B, int F, int M) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * F * M) return;
  const int n = idx % M, bf = idx / M, b = bf / F, f = bf % F;
  const Cx<R>* c = C + (size_t)bf * M * M;
  const Cx<R>* w = W + (size_t)bf * M * M + n * M;
  double s = 0.0;
  for (int m = 0; m < M; ++m)
    for (int l = 0; l < M; ++l) {
      const double wr = w[m].x, wi = w[m].y, vr = w[l].x, vi = w[l].y, cr = c[m * M + l].x, ci = c[m * M + l].y;
      const double ar = wr * cr - wi * ci, ai = wr * ci + wi * cr;
      s += ar * vr + ai * vi;
    }
  part[((size_t)b * M + n) * F + f] = s;
}

template <typename R>
__global__ void __launch_bounds__(256) masked_copy_kernel(const R* __restrict__ Tsrc, const R* __restrict__ Vsrc,
                                                         R* __restrict__ Tdst, R* __restrict__ Vdst, int B, int N,
                                                         size_t FK, size_t KT, unsigned mask) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nT = (size_t)B * N * FK, nV = (size_t)B * N * KT;
  if (idx < nT) {
    if ((mask >> ((idx / FK) % N)) & 1u) Tdst[idx] = Tsrc[idx];
  } else if (idx < nT + nV) {
    const size_t j = idx - nT;
    if ((mask >> ((j / KT) % N)) & 1u) Vdst[j] = Vsrc[j];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// dispatch on (dtype, M)
// ------------------------------------------------------------------------------------------------------------------
template <typename Fn>
static int dispatch(assx_ctx* ctx, int dtype, int M, Fn&& fn) {
  if (dtype != ASSX_F64 && dtype != ASSX_F32) return fail(ctx, ASSX_E_ARG, "dtype must be ASSX_F32 or ASSX_F64, got %d", dtype);
#define WIDEM_CASE(MM)                                  \
  case MM:                                              \
    return dtype == ASSX_F64 ? fn(double(), IntC<MM>()) : fn(float(), IntC<MM>());
  switch (M) {
    WIDEM_CASE(5)
    WIDEM_CASE(6)
    WIDEM_CASE(7)
    WIDEM_CASE(8)
  }
#undef WIDEM_CASE
  // 9 <= M <= 32: IntC<0> = "M is a run-time value" (assx_widem_rt.hpp: functional, not tuned)
  if (M > MMAX && M <= RT_MMAX) return dtype == ASSX_F64 ? fn(double(), IntC<0>()) : fn(float(), IntC<0>());
  return fail(ctx, ASSX_E_UNSUPPORTED, "the wide-channel path handles 5 <= M <= %d, got M=%d", RT_MMAX, M);
}

template <typename R, int M>
static int launch_demix(assx_ctx* ctx, int Mr, const void* X, const void* W, const void* scale, void* Y, void* P, int B,
                        int F, int T, hipStream_t st) {
  if constexpr (M == 0) {  // run-time channel count (Mr > 8): x of a frame in registers, bound 12 / 16 / 24 / 32
#define DEMIX_RT(MCV)                                                                                                   \
  hipLaunchKernelGGL((demix_map_rt_kernel<R, MCV>), dim3(nblk(T, 256), F, B), dim3(256), (size_t)Mr * Mr * sizeof(Cx<R>), \
                     st, (const Cx<R>*)X, (const Cx<R>*)W, (const Cx<R>*)scale, (Cx<R>*)Y, (R*)P, F, T, Mr)
    if (Mr <= 12) DEMIX_RT(12);
    else if (Mr <= 16) DEMIX_RT(16);
    else if (Mr <= 24) DEMIX_RT(24);
    else DEMIX_RT(32);
#undef DEMIX_RT
  } else
    hipLaunchKernelGGL((demix_map_kernel<R, M>), dim3(nblk(T, 256), F, B), dim3(256), 0, st, (const Cx<R>*)X,
                       (const Cx<R>*)W, (const Cx<R>*)scale, (Cx<R>*)Y, (R*)P, F, T);
  ASSX_LAUNCH_CHECK(ctx, "widem::demix_map_kernel");
  return 0;
}

template <typename R, int M>
static int launch_cov(assx_ctx* ctx, int Mr, const void* X, const void* r, int r_kind, int N, double eps, void* U, int B,
                      int F, int T, hipStream_t st) {
  if constexpr (M == 0) {  // run-time channel count: one workgroup per bin (and 256 pairs), every source's sums per thread
    const int np = Mr * (Mr + 1) / 2;
    const int threads = np >= 256 ? 256 : (np + 63) / 64 * 64;
    // sources per workgroup: 4 up to 10 channels, 8 up to 14, 16 beyond (measured: profiles/r03_manychan_bench.txt;
    // ASSX_RT_NS overrides for A/B runs)
    static const int ns_env = lab_int("ASSX_RT_NS", 0);
    const int nsg = N <= 1 ? 1 : (ns_env > 0 ? ns_env : (Mr <= 10 ? 4 : (Mr <= 14 ? 8 : 16)));
    const int ncv = nsg <= 1 ? 1 : (nsg <= 2 ? 2 : (nsg <= 4 ? 4 : (nsg <= 8 ? 8 : 16)));
    const dim3 grid(F, ((np + threads - 1) / threads) * ((N + ncv - 1) / ncv), B);
    const size_t lds = (size_t)RT_TILE * Mr * sizeof(Cx<R>) + (size_t)RT_TILE * ncv * sizeof(R);
#define COV_RT(NCV)                                                                                                   \
  hipLaunchKernelGGL((cov_rt_kernel<R, NCV>), grid, dim3(threads), lds, st, (const Cx<R>*)X, (const R*)r, r_kind, N,   \
                     (R)eps, (Cx<R>*)U, F, T, (R)(1.0 / (double)T), Mr)
    if (ncv == 1) COV_RT(1);
    else if (ncv == 2) COV_RT(2);
    else if (ncv == 4) COV_RT(4);
    else if (ncv == 8) COV_RT(8);
    else COV_RT(16);
#undef COV_RT
  } else
    hipLaunchKernelGGL((cov_bin_kernel<R, M>), dim3(F, B), dim3(64 * (N > 1 ? N : 4)), 0, st, (const Cx<R>*)X, (const R*)r, r_kind, N,
                       (R)eps, (Cx<R>*)U, F, T, (R)(1.0 / (double)T));
  ASSX_LAUNCH_CHECK(ctx, "widem::cov_bin_kernel");
  return 0;
}

// Weighted covariance of all N = M sources, streaming (assx_widem_cov.hpp) -> dense U.  wk: WK_TV (Tb, V; n_basis K <=
// SRC_COV_KMAX and domain 2 -- callers route everything else through the variance map), WK_NT (V = r (B,N,T)),
// WK_NFT (V = r (B,N,F,T)).
template <typename R, int M, int WKV>
static int launch_src_cov_as(assx_ctx* ctx, const void* X, const void* Tb, const void* V, int K, double eps, void* rec,
                             const FlatPart& fp, int B, int F, int T, hipStream_t st) {
  const Dims d{B, F, T, K};
#if ASSX_LAB
  using GEO = SrcCovGeom<R, M, WKV>;
  static const int pairs = lab_int("ASSX_WIDEM_PAIRS", 1);  // 0: one wave per source (src_cov_kernel, round 3's form), A/B runs
  if (!pairs) {
    if (GEO::lds_bytes > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(src_cov_kernel<R, M, WKV>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEO::lds_bytes);
      if (e != hipSuccess) return hip_fail(ctx, e, "hipFuncSetAttribute(src_cov_kernel)");
    }
    hipLaunchKernelGGL((src_cov_kernel<R, M, WKV>), dim3(fp.G), dim3(WAVE * M), GEO::lds_bytes, st, (const Cx<R>*)X,
                       (const R*)Tb, (const R*)V, (R*)rec, d, fp, (R)eps);
    ASSX_LAUNCH_CHECK(ctx, "widem::src_cov_kernel");
    return 0;
  }
#endif
  {  // the Hermitian pairs split over the waves, weights exchanged through LDS
    static const int lds_pad = lab_int("ASSX_PAIR_LDS_PAD", 0);  // laboratory builds: extra dynamic LDS
    const size_t lds = pair_cov_lds_bytes<R, M, WKV>() + (size_t)lds_pad;
    if (lds > 64 * 1024) {  // > 64 KB of dynamic LDS needs the opt-in (per device: set on every launch)
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pair_cov_kernel<R, M, WKV>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return hip_fail(ctx, e, "hipFuncSetAttribute(pair_cov_kernel)");
    }
    hipLaunchKernelGGL((pair_cov_kernel<R, M, WKV>), dim3(fp.G), dim3(WAVE * M), lds, st, (const Cx<R>*)X, (const R*)Tb,
                       (const R*)V, (R*)rec, d, fp, (R)eps);
    ASSX_LAUNCH_CHECK(ctx, "widem::pair_cov_kernel");
    return 0;
  }
}
template <typename R, int M>
static int launch_src_cov(assx_ctx* ctx, const void* X, const void* Tb, const void* V, int wk, int K, double domain,
                          double eps, void* U, void* rec, int B, int F, int T, hipStream_t st) {
  if constexpr (M == 0) {  // run-time channel count: never routed here (src_cov_ok)
    return fail(ctx, ASSX_E_UNSUPPORTED, "src_cov_kernel needs a compile-time channel count");
  } else {
  const FlatPart fp = flat_src_cov(B, F, T);
  int rc;
  if (wk == WK_NT) rc = launch_src_cov_as<R, M, WK_NT>(ctx, X, nullptr, V, 1, eps, rec, fp, B, F, T, st);
  else if (wk == WK_NFT) rc = launch_src_cov_as<R, M, WK_NFT>(ctx, X, nullptr, V, 1, eps, rec, fp, B, F, T, st);
  else if (domain != 2.0 || K > SRC_COV_KMAX)
    return fail(ctx, ASSX_E_UNSUPPORTED, "src_cov_kernel rebuilds the variance for domain 2 and n_basis <= %d only", SRC_COV_KMAX);
  else rc = launch_src_cov_as<R, M, WK_TV>(ctx, X, Tb, V, K, eps, rec, fp, B, F, T, st);
  if (rc) return rc;
  hipLaunchKernelGGL((src_cov_finalize_kernel<R, M>), dim3(nblk((size_t)B * M * F * M * M, 256)), dim3(256), 0, st,
                     (const R*)rec, (Cx<R>*)U, B, F, fp, (R)(1.0 / (double)T));
  ASSX_LAUNCH_CHECK(ctx, "widem::src_cov_finalize_kernel");
  return 0;
  }
}
// the streaming kernel addresses an utterance's X and weight arrays with 32-bit byte offsets; ASSX_WIDEM_COV=0 keeps
// round 2's one-workgroup-per-bin kernel on materialised weights (A/B runs)
static bool src_cov_ok(int M, int F, int T, size_t r) {  // M = 0: run-time channel count (> 8), not served
  static const int on = lab_int("ASSX_WIDEM_COV", 1);
  return M != 0 && on != 0 && (size_t)M * F * T * 2 * r < 0xffffffffull;
}

template <typename R, int M>
static int launch_sweep(assx_ctx* ctx, int Mr, int spatial, int pm, int pn, const void* U, void* W, const void* C, double* pw,
                        double thr, int32_t* status, int B, int F, int T, hipStream_t st, double den_floor = 0.0) {
  const dim3 grid((unsigned)((size_t)B * F)), block(64);  // 64 lanes = one bin (GW = 64 for M >= 5)
  if constexpr (M == 0) {  // run-time channel count: IP and (round 6) ISS, one wave per bin, matrices in LDS
    if (spatial == ASSX_SPATIAL_ISS) {
      const size_t lds = iss_rt_lds_bytes(Mr);
      hipLaunchKernelGGL((iss_rt_kernel<R>), grid, block, lds, st, (const Cx<R>*)U, (Cx<R>*)W, (const Cx<R>*)C, pw, (double)T,
                         B, F, Mr);
      ASSX_LAUNCH_CHECK(ctx, "widem::iss_rt_kernel");
      return 0;
    }
    if (spatial == ASSX_SPATIAL_IP2) {
      const size_t lds = ip2_rt_lds_bytes(Mr);
      if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ip2_rt_kernel<R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(ctx, e, "hipFuncSetAttribute(ip2_rt_kernel)");
      }
      hipLaunchKernelGGL((ip2_rt_kernel<R>), grid, block, lds, st, (const Cx<R>*)U, (Cx<R>*)W, (const Cx<R>*)C, pw, thr, status,
                         B, F, pm, pn, Mr);
      ASSX_LAUNCH_CHECK(ctx, "widem::ip2_rt_kernel");
      return 0;
    }
    if (spatial != ASSX_SPATIAL_IP) return fail(ctx, ASSX_E_ARG, "bad spatial algorithm %d", spatial);
    const size_t lds = ip_rt_lds_bytes(Mr);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ip_rt_kernel<R>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return hip_fail(ctx, e, "hipFuncSetAttribute(ip_rt_kernel)");
    }
    hipLaunchKernelGGL((ip_rt_kernel<R>), grid, block, lds, st, (const Cx<R>*)U, (Cx<R>*)W, (const Cx<R>*)C, pw, thr, status,
                       B, F, den_floor, Mr);
    ASSX_LAUNCH_CHECK(ctx, "widem::ip_rt_kernel");
    return 0;
  } else {
  const FlatPart fp{};
  if (spatial == ASSX_SPATIAL_IP)
    hipLaunchKernelGGL((ip_group_kernel<R, M, false>), grid, block, 0, st, (const Cx<R>*)U, (const R*)nullptr, fp, 1.0,
                       (Cx<R>*)W, (const Cx<R>*)C, pw, thr, status, B, F, den_floor);
  else if (spatial == ASSX_SPATIAL_ISS)
    hipLaunchKernelGGL((iss_group_kernel<R, M, false>), grid, block, 0, st, (const Cx<R>*)U, (const R*)nullptr, fp, 1.0,
                       (double)T, (Cx<R>*)W, (const Cx<R>*)C, pw, B, F);
  else if (spatial == ASSX_SPATIAL_IP2)
    hipLaunchKernelGGL((ip2_group_kernel<R, M, false>), grid, block, 0, st, (const Cx<R>*)U, (const R*)nullptr, fp, 1.0,
                       (Cx<R>*)W, (const Cx<R>*)C, pw, thr, status, B, F, pm, pn);
  else
    return fail(ctx, ASSX_E_ARG, "bad spatial algorithm %d", spatial);
  ASSX_LAUNCH_CHECK(ctx, "widem sweep (group kernel)");
  return 0;
  }
}

template <typename R>
static int launch_variance(assx_ctx* ctx, const void* Tb, const void* V, void* Rm, double domain, int BN, int F, int T,
                           int K, hipStream_t st) {
  hipLaunchKernelGGL((variance_map_kernel<R>), dim3(nblk(T, 256), F, BN), dim3(256), 0, st, (const R*)Tb, (const R*)V,
                     (R*)Rm, F, T, K, make_pow(2.0 / domain));
  ASSX_LAUNCH_CHECK(ctx, "widem::variance_map_kernel");
  return 0;
}

// loss[b] = sum_{n,f,t} P/R + log R  -  2 T sum_f log|det W_f|, from the maps already in ws
template <typename R, int M>
static int loss_from_maps(assx_ctx* ctx, int Mr, const Ws& L, const void* W, double eps, double* loss, void* ws, int B,
                          int F, int T, hipStream_t st, double nu = -1.0) {
  double* lpart = (double*)((char*)ws + L.lpart);
  const int lstride = RB + F;
  if (nu >= 0.0)  // Student-t term (t-ILRMA)
    hipLaunchKernelGGL((map_sum_kernel<R, 2>), dim3(RB, B), dim3(RED_THREADS), 0, st, (const R*)((char*)ws + L.map0),
                       (const R*)((char*)ws + L.map1), lpart, (size_t)Mr * F * T, lstride, (R)eps, (R)nu);
  else
    hipLaunchKernelGGL((map_sum_kernel<R, 1>), dim3(RB, B), dim3(RED_THREADS), 0, st, (const R*)((char*)ws + L.map0),
                       (const R*)((char*)ws + L.map1), lpart, (size_t)Mr * F * T, lstride, (R)eps, (R)0);
  ASSX_LAUNCH_CHECK(ctx, "widem::map_sum_kernel(loss)");
  if constexpr (M == 0)  // run-time channel count: one wave per bin, LU in LDS
    hipLaunchKernelGGL((logdet_rt_kernel<R>), dim3((unsigned)((size_t)B * F)), dim3(64), (size_t)Mr * Mr * sizeof(Cd), st,
                       (const Cx<R>*)W, lpart, B, F, T, lstride, RB, Mr);
  else
    hipLaunchKernelGGL((logdet_kernel<R, M>), dim3(nblk((size_t)B * F, 64)), dim3(64), 0, st, (const Cx<R>*)W, lpart, B, F,
                       T, lstride, RB);
  ASSX_LAUNCH_CHECK(ctx, "widem::logdet_kernel");
  hipLaunchKernelGGL((row_sum_kernel<double>), dim3(B), dim3(RED_THREADS), 0, st, (const double*)lpart, loss, lstride,
                     lstride, 1.0);
  ASSX_LAUNCH_CHECK(ctx, "widem::row_sum_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// entry-point bodies
// ------------------------------------------------------------------------------------------------------------------
int demix(assx_ctx* ctx, const void* X, const void* W, const void* scale, void* Y, int B, int M, int F, int T, int dtype,
          hipStream_t st) {
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    return launch_demix<decltype(rt), decltype(mt)::value>(ctx, M, X, W, scale, Y, nullptr, B, F, T, st);
  });
}

int power_map(assx_ctx* ctx, const void* X, const void* W, void* P, int B, int M, int F, int T, int dtype,
              hipStream_t st) {
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    return launch_demix<decltype(rt), decltype(mt)::value>(ctx, M, X, W, nullptr, nullptr, P, B, F, T, st);
  });
}

int cov_accumulate(assx_ctx* ctx, const void* X, const void* r, int r_kind, double eps, void* U, void* ws, int B, int M,
                   int N, int F, int T, int dtype, hipStream_t st) {
  const Ws L = layout(B, M, F, T, 1, dtype);
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    const int rk = r_kind == ASSX_W_NONE ? RK_NONE : (r_kind == ASSX_W_NT ? RK_NT : RK_NFT);
    if (rk != RK_NONE && N == MM && ws && src_cov_ok(MT, F, T, sizeof(R)))
      return launch_src_cov<R, MT>(ctx, X, nullptr, r, rk == RK_NT ? WK_NT : WK_NFT, 1, 2.0, eps, U, (char*)ws + L.rec, B, F,
                                   T, st);
    return launch_cov<R, MT>(ctx, MM, X, r, rk, N, eps, U, B, F, T, st);
  });
}

int ip_update(assx_ctx* ctx, const void* U, void* W, double thr, int32_t* status, int B, int M, int F, int dtype,
              hipStream_t st) {
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    return launch_sweep<decltype(rt), decltype(mt)::value>(ctx, M, ASSX_SPATIAL_IP, 0, 1, U, W, nullptr, nullptr, thr, status,
                                                           B, F, 1, st);
  });
}
int iss_update(assx_ctx* ctx, const void* U, void* W, int n_frames, int B, int M, int F, int dtype, hipStream_t st) {
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    return launch_sweep<decltype(rt), decltype(mt)::value>(ctx, M, ASSX_SPATIAL_ISS, 0, 1, U, W, nullptr, nullptr, 0.0,
                                                           nullptr, B, F, n_frames, st);
  });
}
int ip2_update(assx_ctx* ctx, const void* U, void* W, double thr, int32_t* status, int pm, int pn, int B, int M, int F,
               int dtype, hipStream_t st) {
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    return launch_sweep<decltype(rt), decltype(mt)::value>(ctx, M, ASSX_SPATIAL_IP2, pm, pn, U, W, nullptr, nullptr, thr,
                                                           status, B, F, 1, st);
  });
}

int ilrma_loss(assx_ctx* ctx, const void* X, const void* W, const void* Tb, const void* V, double domain, double eps,
               double* loss, void* ws, int B, int M, int F, int T, int K, int dtype, hipStream_t st, double nu) {
  const Ws L = layout(B, M, F, T, K, dtype);
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    int rc = launch_demix<R, MT>(ctx, MM, X, W, nullptr, nullptr, (char*)ws + L.map0, B, F, T, st);
    if (rc) return rc;
    if ((rc = launch_variance<R>(ctx, Tb, V, (char*)ws + L.map1, domain, B * MM, F, T, K, st))) return rc;
    return loss_from_maps<R, MT>(ctx, MM, L, W, eps, loss, ws, B, F, T, st, nu);
  });
}

// t-ILRMA (ilrma.py:899-983) on the maps: the tNMF-type update with the raw power map as target, and IP on the
// covariance weighted by Xi = (nu R + 2 P) / (nu + 2)
int tilrma_source_update(assx_ctx* ctx, const void* X, const void* W, void* Tb, void* V, double nu, double eps, void* ws,
                         int B, int M, int F, int T, int K, int dtype, hipStream_t st) {
  const Ws L = layout(B, M, F, T, K, dtype);
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    void* P = (char*)ws + L.map0;
    int rc = launch_demix<R, MT>(ctx, MM, X, W, nullptr, nullptr, P, B, F, T, st);
    if (rc) return rc;
    NmfGroupScope grp(ctx, MM);
    return assx_nmf_update_ex(ctx, ASSX_NMF_T_RAW, 2.0, nu, eps, P, Tb, V, (char*)ws + L.nmf, B * MM, F, T, K, dtype, st);
  });
}

int tilrma_spatial_update(assx_ctx* ctx, const void* X, void* W, const void* Tb, const void* V, double nu, double eps,
                          void* Xi, const void* C, double* power_bins, int32_t* status, void* ws, int B, int M, int F,
                          int T, int K, int dtype, hipStream_t st) {
  const Ws L = layout(B, M, F, T, K, dtype);
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    void* P = (char*)ws + L.map0;
    int rc = launch_demix<R, MT>(ctx, MM, X, W, nullptr, nullptr, P, B, F, T, st);  // with the filters BEFORE the sweep
    if (rc) return rc;
    if ((rc = launch_variance<R>(ctx, Tb, V, Xi, 2.0, B * MM, F, T, K, st))) return rc;
    const size_t count = (size_t)B * MM * F * T;
    hipLaunchKernelGGL((xi_map_kernel<R>), dim3(nblk(count, 256)), dim3(256), 0, st, (const R*)P, (R*)Xi, count, (R)nu,
                       (R)eps);
    ASSX_LAUNCH_CHECK(ctx, "widem::xi_map_kernel");
    void* U = (char*)ws + L.u;
    // Xi is used as is (the reference does not floor it); no condition-number guard, normaliser floored at eps
    if (src_cov_ok(MT, F, T, sizeof(R)))
      rc = launch_src_cov<R, MT>(ctx, X, nullptr, Xi, WK_NFT, 1, 2.0, 0.0, U, (char*)ws + L.rec, B, F, T, st);
    else
      rc = launch_cov<R, MT>(ctx, MM, X, Xi, RK_NFT, MM, 0.0, U, B, F, T, st);
    if (rc) return rc;
    return launch_sweep<R, MT>(ctx, MM, ASSX_SPATIAL_IP, 0, 1, U, W, C, power_bins, INFINITY, status, B, F, T, st, eps);
  });
}

int ilrma_source_update(assx_ctx* ctx, const void* X, const void* W, void* Tb, void* V, double domain, double eps,
                        unsigned source_mask, double* loss_prev, void* ws, int B, int M, int F, int T, int K, int dtype,
                        hipStream_t st) {
  const Ws L = layout(B, M, F, T, K, dtype);
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    void* P = (char*)ws + L.map0;
    int rc = launch_demix<R, MT>(ctx, MM, X, W, nullptr, nullptr, P, B, F, T, st);  // P = |W x|^2, once
    if (rc) return rc;
    if (loss_prev) {  // the loss of the model at entry needs the same P
      if ((rc = launch_variance<R>(ctx, Tb, V, (char*)ws + L.map1, domain, B * MM, F, T, K, st))) return rc;
      if ((rc = loss_from_maps<R, MT>(ctx, MM, L, W, eps, loss_prev, ws, B, F, T, st))) return rc;
    }
    const unsigned all = MM >= 32 ? ~0u : (1u << MM) - 1u;
    if ((source_mask & all) == 0u) return 0;
    NmfGroupScope grp(ctx, MM);
    if ((source_mask & all) == all)  // ilrma.py:409-430 == nmf.py:302-327 with target P, batch B*N
      return assx_nmf_update(ctx, ASSX_NMF_IS_MM, domain, eps, P, Tb, V, (char*)ws + L.nmf, B * MM, F, T, K, dtype, st);
    // pairwise update (ilrma.py:432-481): update copies of every source, copy the selected ones back
    const size_t nT = (size_t)B * MM * F * K, nV = (size_t)B * MM * K * T;
    R* Tt = (R*)((char*)ws + L.tmp);
    R* Vt = Tt + nT;
    hipError_t e = hipMemcpyAsync(Tt, Tb, nT * sizeof(R), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(Vt, V, nV * sizeof(R), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail(ctx, (int)e, "hipMemcpyAsync(model copy): %s", hipGetErrorString(e));
    rc = assx_nmf_update(ctx, ASSX_NMF_IS_MM, domain, eps, P, Tt, Vt, (char*)ws + L.nmf, B * MM, F, T, K, dtype, st);
    if (rc) return rc;
    hipLaunchKernelGGL((masked_copy_kernel<R>), dim3(nblk(nT + nV, 256)), dim3(256), 0, st, (const R*)Tt, (const R*)Vt,
                       (R*)Tb, (R*)V, B, MM, (size_t)F * K, (size_t)K * T, source_mask);
    ASSX_LAUNCH_CHECK(ctx, "widem::masked_copy_kernel");
    return 0;
  });
}

int ilrma_spatial_update(assx_ctx* ctx, int spatial, int pm, int pn, const void* X, void* W, const void* Tb,
                         const void* V, double domain, double eps, double thr, void* U_out, const void* C,
                         double* power_bins, int32_t* status, void* ws, int B, int M, int F, int T, int K, int dtype,
                         hipStream_t st) {
  const Ws L = layout(B, M, F, T, K, dtype);
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    void* Rm = (char*)ws + L.map1;
    void* U = U_out ? U_out : (void*)((char*)ws + L.u);
    int rc;
    if (src_cov_ok(MT, F, T, sizeof(R)) && K <= SRC_COV_KMAX && domain == 2.0) {  // weights rebuilt in the kernel: no variance map
      rc = launch_src_cov<R, MT>(ctx, X, Tb, V, WK_TV, K, domain, eps, U, (char*)ws + L.rec, B, F, T, st);
    } else {
      if ((rc = launch_variance<R>(ctx, Tb, V, Rm, domain, B * MM, F, T, K, st))) return rc;
      if (src_cov_ok(MT, F, T, sizeof(R)))
        rc = launch_src_cov<R, MT>(ctx, X, nullptr, Rm, WK_NFT, 1, 2.0, eps, U, (char*)ws + L.rec, B, F, T, st);
      else
        rc = launch_cov<R, MT>(ctx, MM, X, Rm, RK_NFT, MM, eps, U, B, F, T, st);
    }
    if (rc) return rc;
    return launch_sweep<R, MT>(ctx, MM, spatial, pm, pn, U, W, C, power_bins, thr, status, B, F, T, st);
  });
}

// Partitioning function (ilrma.py:368-408) on the maps: the route the M <= 4 path takes for n_basis > 4 -- the three
// sets of per-source sums come from the NMF half kernels on P = |W x|^2 with the effective model (batch B*N), the
// adapters lay them out as one record per bin / per frame block, the combination kernels of assx_partition.hpp fold
// them.  The adapter records live in the (idle) variance-map region.
int ilrma_source_update_partitioned(assx_ctx* ctx, const void* X, const void* W, void* Z, void* Tb, void* V, void* Teff,
                                    void* Veff, double eps, void* ws, int B, int M, int F, int T, int K, int dtype,
                                    hipStream_t st) {
  const Ws L = layout(B, M, F, T, K, dtype);
  if (K > 64) return fail(ctx, ASSX_E_UNSUPPORTED, "partitioning with M = %d > 4 needs n_basis <= 64, got %d", M, K);
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    const size_t nT = (size_t)B * MM * F * K, nV = (size_t)B * MM * K * T;
    auto expand = [&](bool with_v) -> int {
      hipLaunchKernelGGL((part_expand_kernel<R>), dim3(nblk(with_v ? nT + nV : nT, 256)), dim3(256), 0, st, (const R*)Z,
                         (const R*)Tb, (const R*)V, (R*)Teff, with_v ? (R*)Veff : (R*)nullptr, B, MM, F, K, T);
      ASSX_LAUNCH_CHECK(ctx, "part_expand_kernel");
      return 0;
    };
    void* P = (char*)ws + L.map0;
    R* rec = (R*)((char*)ws + L.map1);
    void* nws = (char*)ws + L.nmf;
    int rc = launch_demix<R, MT>(ctx, MM, X, W, nullptr, nullptr, P, B, F, T, st);  // W does not move here: P once
    if (rc) return rc;
    const int TBk = (T + WAVE - 1) / WAVE;
    const FlatPart fpb{(long long)F * TBk, TBk, TBk, B * F, 1, F, F};      // one record per bin
    const FlatPart fpa{(long long)TBk * F, F, F, B * TBk, 1, TBk, TBk};    // one record per frame block
    auto sums = [&](int half) -> int {
      const void* np = nullptr;
      int slabs = 0;
      NmfGroupScope grp(ctx, MM);
      int r2 = nmf_half_partials(ctx, ASSX_NMF_IS_MM, 2.0, 0.0, eps, half, P, Teff, Veff, nws, B * MM, F, T, K, dtype, st,
                                 &np, &slabs);
      if (r2) return r2;
      if (half == NMF_HALF_BASIS)
        hipLaunchKernelGGL((part_adapt_basis_kernel<R>), dim3(nblk((size_t)B * F * MM * 2 * K, 256)), dim3(256), 0, st,
                           (const R*)np, rec, B, MM, F, K, slabs);
      else
        hipLaunchKernelGGL((part_adapt_act_kernel<R>), dim3(nblk((size_t)B * TBk * MM * 2 * K * WAVE, 256)), dim3(256), 0,
                           st, (const R*)np, rec, B, MM, K, T, slabs);
      ASSX_LAUNCH_CHECK(ctx, "part_adapt_kernel");
      return 0;
    };
    if ((rc = expand(true))) return rc;
    if ((rc = sums(NMF_HALF_BASIS))) return rc;
    if constexpr (MT != 0)
      hipLaunchKernelGGL((part_latent_kernel<R, MT>), dim3(K, B), dim3(256), 0, st, (const R*)rec, (const R*)Tb, (R*)Z, F, K,
                         fpb, (R)eps);
    else  // run-time channel count (round 6)
      hipLaunchKernelGGL((part_latent_rt_kernel<R, RT_MMAX>), dim3(K, B), dim3(256), 0, st, (const R*)rec, (const R*)Tb,
                         (R*)Z, F, K, fpb, (R)eps, MM);
    ASSX_LAUNCH_CHECK(ctx, "part_latent_kernel");
    if ((rc = expand(false))) return rc;
    if ((rc = sums(NMF_HALF_BASIS))) return rc;
    hipLaunchKernelGGL((part_basis_kernel<R>), dim3(nblk((size_t)B * F * K, 256)), dim3(256), 0, st, (const R*)rec,
                       (const R*)Z, (R*)Tb, B, MM, F, K, fpb, (R)eps);
    ASSX_LAUNCH_CHECK(ctx, "part_basis_kernel");
    if ((rc = expand(false))) return rc;
    if ((rc = sums(NMF_HALF_ACT))) return rc;
    hipLaunchKernelGGL((part_act_kernel<R>), dim3(nblk((size_t)B * K * T, 256)), dim3(256), 0, st, (const R*)rec, (R*)V, B,
                       MM, F, K, T, fpa, (R)eps);
    ASSX_LAUNCH_CHECK(ctx, "part_act_kernel");
    return expand(true);  // leave (Teff, Veff) consistent with the updated (Z, T, V)
  });
}

int normalize_power_bins_partitioned(assx_ctx* ctx, void* W, void* Z, void* Tb, const double* power_bins, double eps,
                                     void* ws, int B, int M, int F, int K, int dtype, hipStream_t st) {
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    const size_t per_b = (size_t)F * MM * MM + (size_t)F * K;
    if constexpr (MT != 0)
      hipLaunchKernelGGL((part_normalize_power_kernel<R, MT>), dim3(nblk(per_b, 256), B), dim3(256), (size_t)K * sizeof(R),
                         st, (Cx<R>*)W, (R*)ws, (const R*)Z, (R*)Tb, power_bins, F, K, (R)eps);
    else  // run-time channel count (round 6)
      hipLaunchKernelGGL((part_normalize_power_rt_kernel<R, RT_MMAX>), dim3(nblk(per_b, 256), B), dim3(256),
                         (size_t)K * sizeof(R), st, (Cx<R>*)W, (R*)ws, (const R*)Z, (R*)Tb, power_bins, F, K, (R)eps, MM);
    ASSX_LAUNCH_CHECK(ctx, "part_normalize_power_kernel");
    hipError_t e = hipMemcpyAsync(Z, ws, (size_t)B * MM * K * sizeof(R), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail(ctx, (int)e, "hipMemcpyAsync(Z): %s", hipGetErrorString(e));
    return 0;
  });
}

// covariance weighted by a given (B,N,F,T) map, then the IP sweep: the other callers of the pair (IDLMA, FastMNMF)
int weighted_ip(assx_ctx* ctx, const void* X, const void* r, double eps, double thr, double den_floor, void* W,
                int32_t* status, void* ws, int B, int M, int F, int T, int dtype, hipStream_t st) {
  const Ws L = layout(B, M, F, T, 1, dtype);
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    void* U = (char*)ws + L.u;
    int rc;
    if (src_cov_ok(MT, F, T, sizeof(R)))
      rc = launch_src_cov<R, MT>(ctx, X, nullptr, r, WK_NFT, 1, 2.0, eps, U, (char*)ws + L.rec, B, F, T, st);
    else
      rc = launch_cov<R, MT>(ctx, MM, X, r, RK_NFT, MM, eps, U, B, F, T, st);
    if (rc) return rc;
    return launch_sweep<R, MT>(ctx, MM, ASSX_SPATIAL_IP, 0, 1, U, W, nullptr, nullptr, thr, status, B, F, T, st, den_floor);
  });
}

int demix_power(assx_ctx* ctx, const void* X, const void* W, void* power, void* ws, int B, int M, int F, int T,
                int dtype, hipStream_t st) {
  const Ws L = layout(B, M, F, T, 1, dtype);
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    int rc = launch_demix<R, MT>(ctx, MM, X, W, nullptr, nullptr, (char*)ws + L.map0, B, F, T, st);
    if (rc) return rc;
    double* part = (double*)((char*)ws + L.lpart);
    hipLaunchKernelGGL((map_sum_kernel<R, 0>), dim3(RB, B * MM), dim3(RED_THREADS), 0, st,
                       (const R*)((char*)ws + L.map0), (const R*)nullptr, part, (size_t)F * T, RB, (R)0, (R)0);
    ASSX_LAUNCH_CHECK(ctx, "widem::map_sum_kernel(power)");
    hipLaunchKernelGGL((row_sum_kernel<R>), dim3(B * MM), dim3(RED_THREADS), 0, st, (const double*)part, (R*)power, RB,
                       RB, 1.0 / ((double)F * (double)T));
    ASSX_LAUNCH_CHECK(ctx, "widem::row_sum_kernel");
    return 0;
  });
}

int power_from_cov(assx_ctx* ctx, const void* C, const void* W, void* power, void* ws, int B, int M, int F, int dtype,
                   hipStream_t st) {
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    double* part = (double*)ws;
    hipLaunchKernelGGL((power_cov_kernel<R>), dim3(nblk((size_t)B * F * MM, 256)), dim3(256), 0, st, (const Cx<R>*)C,
                       (const Cx<R>*)W, part, B, F, MM);
    ASSX_LAUNCH_CHECK(ctx, "widem::power_cov_kernel");
    hipLaunchKernelGGL((row_sum_kernel<R>), dim3(B * MM), dim3(RED_THREADS), 0, st, (const double*)part, (R*)power, F, F,
                       1.0 / (double)F);
    ASSX_LAUNCH_CHECK(ctx, "widem::row_sum_kernel");
    return 0;
  });
}

int auxiva_weights(assx_ctx* ctx, const void* X, const void* W, int kind, double eps, void* r, double* loss, void* ws,
                   int B, int M, int F, int T, int dtype, hipStream_t st) {
  const Ws L = layout(B, M, F, T, 1, dtype);
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    int rc = launch_demix<R, MT>(ctx, MM, X, W, nullptr, nullptr, (char*)ws + L.map0, B, F, T, st);
    if (rc) return rc;
    const int TB = (int)nblk(T, RED_THREADS);
    const int lstride = MM * TB + F;
    double* lpart = loss ? (double*)((char*)ws + L.lpart) : nullptr;
    hipLaunchKernelGGL((aux_stat_kernel<R>), dim3(TB, B * MM), dim3(RED_THREADS), 0, st, (const R*)((char*)ws + L.map0),
                       (R*)r, lpart, kind, (R)eps, MM, F, T, lstride);
    ASSX_LAUNCH_CHECK(ctx, "widem::aux_stat_kernel");
    if (loss) {
      if constexpr (MT == 0)
        hipLaunchKernelGGL((logdet_rt_kernel<R>), dim3((unsigned)((size_t)B * F)), dim3(64), (size_t)MM * MM * sizeof(Cd), st,
                           (const Cx<R>*)W, lpart, B, F, T, lstride, MM * TB, MM);
      else
        hipLaunchKernelGGL((logdet_kernel<R, MT>), dim3(nblk((size_t)B * F, 64)), dim3(64), 0, st, (const Cx<R>*)W, lpart, B,
                           F, T, lstride, MM * TB);
      ASSX_LAUNCH_CHECK(ctx, "widem::logdet_kernel");
      hipLaunchKernelGGL((row_sum_kernel<double>), dim3(B), dim3(RED_THREADS), 0, st, (const double*)lpart, loss, lstride,
                         lstride, 1.0);
      ASSX_LAUNCH_CHECK(ctx, "widem::row_sum_kernel");
    }
    return 0;
  });
}

int auxiva_spatial_update(assx_ctx* ctx, int spatial, int pm, int pn, const void* X, void* W, const void* r, double eps,
                          double thr, void* U_out, int32_t* status, void* ws, int B, int M, int F, int T, int dtype,
                          hipStream_t st) {
  const Ws L = layout(B, M, F, T, 1, dtype);
  return dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    using R = decltype(rt);
    constexpr int MT = decltype(mt)::value;
    const int MM = MT ? MT : M;
    void* U = U_out ? U_out : (void*)((char*)ws + L.u);
    int rc;
    if (src_cov_ok(MT, F, T, sizeof(R)))
      rc = launch_src_cov<R, MT>(ctx, X, nullptr, r, WK_NT, 1, 2.0, eps, U, (char*)ws + L.rec, B, F, T, st);
    else
      rc = launch_cov<R, MT>(ctx, MM, X, r, RK_NT, MM, eps, U, B, F, T, st);
    if (rc) return rc;
    return launch_sweep<R, MT>(ctx, MM, spatial, pm, pn, U, W, nullptr, nullptr, thr, status, B, F, T, st);
  });
}

int projection_back_scale(assx_ctx* ctx, const void* X, const void* W, int ref, void* scale, int32_t* status, void* ws,
                          int B, int M, int F, int T, int dtype, hipStream_t st) {
  const Ws L = layout(B, M, F, T, 1, dtype);
  const size_t c = dtype == ASSX_F64 ? 16 : 8, plane = (size_t)F * T;
  void* Y = (char*)ws + L.map0;  // complex map: spans map0 and map1
  int rc = dispatch(ctx, dtype, M, [&](auto rt, auto mt) -> int {
    return launch_demix<decltype(rt), decltype(mt)::value>(ctx, M, X, W, nullptr, Y, nullptr, B, F, T, st);
  });
  if (rc) return rc;
  // scale[b,n,f] = (x_ref Y^H (Y Y^H)^{-1})[n]   (projection_back.py:13-21)
  return stack_gram_solve(ctx, (const char*)X + (size_t)ref * plane * c, (size_t)M * plane, 1, Y, M, scale, (size_t)M * F, 1,
                          0, (size_t)F, status, B, F, T, dtype, st);
}

int projection_back(assx_ctx* ctx, const void* Y, const void* reference, void* scale, int32_t* status, int B, int N,
                    int F, int T, int dtype, hipStream_t st) {
  return stack_gram_solve(ctx, reference, (size_t)F * T, 1, Y, N, scale, (size_t)N * F, 1, 0, (size_t)F, status, B, F, T,
                          dtype, st);
}

}  // namespace widem
}  // namespace assx

#if PAIRCOV_TRACE
// timing-experiment builds only (tools/probes/paircov_trace.py); not part of include/assx.h
extern "C" int assx_debug_paircov_trace(unsigned long long* host, int clear) {
  using assx::widem::g_paircov_trace;
  if (clear) {
    static unsigned long long zeros[1400];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_paircov_trace), zeros, sizeof(zeros));
  }
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_paircov_trace), sizeof(unsigned long long) * 1400);
}
#endif
