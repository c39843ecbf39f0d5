// GaussILRMA with the partitioning function (shared bases T (F,K), activations V (K,T), latent Z (N,K);
// ref:src/bss/ilrma.py:79-95, 368-408, 313-320, 490-495).
//
// The three multiplicative updates contract the same per-source quantities the un-partitioned updates do, once the
// model is written with the per-source *effective* basis  Teff[n,f,k] = Z[n,k] T[f,k]  and the activation replicated
// over sources:
//     Z: num[n,k] = sum_f T[f,k]  * B[n,f,k]     B = basis_stream_kernel records  (sum_t D[n,f,t] V[k,t])
//     T: num[f,k] = sum_n Z[n,k]  * B[n,f,k]
//     V: num[k,t] = sum_n           A[n,k,t]     A = act_stream_kernel records    (sum_f Teff[n,f,k] D[n,f,t])
// so the heavy passes over X are the existing streaming kernels and the kernels here only expand the model and fold
// the partial records.  domain == 2 only (the reference asserts it).
#pragma once
#include "assx_stream.hpp"

namespace assx {

template <typename R>
__global__ void __launch_bounds__(256) part_expand_kernel(const R* __restrict__ Z, const R* __restrict__ Tb,
                                                         const R* __restrict__ V, R* __restrict__ Teff,
                                                         R* __restrict__ Veff, int B, int N, int F, int K, int T) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nT = (size_t)B * N * F * K, nV = (size_t)B * N * K * T;
  if (Teff != nullptr && idx < nT) {
    const int k = idx % K;
    const int f = (idx / K) % F;
    const int n = (idx / ((size_t)K * F)) % N;
    const int b = idx / ((size_t)K * F * N);
    Teff[idx] = Z[((size_t)b * N + n) * K + k] * Tb[((size_t)b * F + f) * K + k];
  } else if (Veff != nullptr && idx >= nT && idx < nT + nV) {
    const size_t j = idx - nT;
    const size_t kt = j % ((size_t)K * T);
    const int b = j / ((size_t)K * T * N);
    Veff[j] = V[(size_t)b * K * T + kt];
  }
}

// sum of the basis-pass records covering (b, f) for source n, component k
template <typename R>
__device__ __forceinline__ void basis_records(const R* __restrict__ part, const FlatPart& fp, int N, int K, int F,
                                              int b, int f, int n, int k, R& num, R& den) {
  const long long j = (long long)b * F + f;
  int g_lo, g_hi;
  flat_cover(fp, j, g_lo, g_hi);
  num = 0;
  den = 0;
  for (int g = g_lo; g <= g_hi; ++g) {
    const int slot = flat_slot(fp, j, g);
    const R* p = part + (((size_t)g * fp.S + slot) * N + n) * (size_t)(2 * K) + k * 2;
    num += p[0];
    den += p[1];
  }
}

// Z[n,k] = sqrt(num/den), then Z /= Z.sum(axis=0)     (ilrma.py:378-387).  One workgroup per (k, utterance).
template <typename R, int N>
__global__ void __launch_bounds__(256) part_latent_kernel(const R* __restrict__ part, const R* __restrict__ Tb,
                                                         R* __restrict__ Z, int F, int K, FlatPart fp, R eps) {
  __shared__ R red[4][2 * N];
  const int k = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
  R num[N], den[N];
#pragma unroll
  for (int n = 0; n < N; ++n) num[n] = den[n] = 0;
  for (int f = threadIdx.x; f < F; f += 256) {
    const R t = Tb[((size_t)b * F + f) * K + k];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      R a, d;
      basis_records<R>(part, fp, N, K, F, b, f, n, k, a, d);
      num[n] += t * a;
      den[n] += t * d;
    }
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const R a = wave_allreduce_sum<R>(num[n]), d = wave_allreduce_sum<R>(den[n]);
    if (lane == 0) {
      red[wv][2 * n] = a;
      red[wv][2 * n + 1] = d;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    R z[N], s = 0;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const R a = (red[0][2 * n] + red[1][2 * n]) + (red[2][2 * n] + red[3][2 * n]);
      const R d = floor_eps<R>((red[0][2 * n + 1] + red[1][2 * n + 1]) + (red[2][2 * n + 1] + red[3][2 * n + 1]), eps);
      z[n] = sqrt(a / d);
      s += z[n];
    }
#pragma unroll
    for (int n = 0; n < N; ++n) Z[((size_t)b * N + n) * K + k] = z[n] / s;
  }
}

// The same with a run-time source count (round 6; more than 8 channels: csrc/assx_widem.hip): N <= NMAX sources, the loops
// unrolled to the bound and masked; same sums in the same order.
template <typename R, int NMAX>
__global__ void __launch_bounds__(256) part_latent_rt_kernel(const R* __restrict__ part, const R* __restrict__ Tb,
                                                            R* __restrict__ Z, int F, int K, FlatPart fp, R eps, int N) {
  __shared__ R red[4][2 * NMAX];
  __shared__ R zl[NMAX];
  const int k = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
  R num[NMAX], den[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) num[n] = den[n] = 0;
  for (int f = threadIdx.x; f < F; f += 256) {
    const R t = Tb[((size_t)b * F + f) * K + k];
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
      if (n < N) {
        R a, d;
        basis_records<R>(part, fp, N, K, F, b, f, n, k, a, d);
        num[n] += t * a;
        den[n] += t * d;
      }
  }
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
    if (n < N) {  // wave-uniform
      const R a = wave_allreduce_sum<R>(num[n]), d = wave_allreduce_sum<R>(den[n]);
      if (lane == 0) {
        red[wv][2 * n] = a;
        red[wv][2 * n + 1] = d;
      }
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    R s = 0;
    for (int n = 0; n < N; ++n) {
      const R a = (red[0][2 * n] + red[1][2 * n]) + (red[2][2 * n] + red[3][2 * n]);
      const R d = floor_eps<R>((red[0][2 * n + 1] + red[1][2 * n + 1]) + (red[2][2 * n + 1] + red[3][2 * n + 1]), eps);
      zl[n] = sqrt(a / d);
      s += zl[n];
    }
    for (int n = 0; n < N; ++n) Z[((size_t)b * N + n) * K + k] = zl[n] / s;
  }
}

// T[f,k] *= sqrt(num/den), contributions of all sources weighted by Z     (ilrma.py:389-397)
template <typename R>
__global__ void __launch_bounds__(256) part_basis_kernel(const R* __restrict__ part, const R* __restrict__ Z,
                                                        R* __restrict__ Tb, int B, int N, int F, int K, FlatPart fp,
                                                        R eps) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * F * K) return;
  const int k = idx % K;
  const int f = (idx / K) % F;
  const int b = idx / ((size_t)K * F);
  R num = 0, den = 0;
  for (int n = 0; n < N; ++n) {
    R a, d;
    basis_records<R>(part, fp, N, K, F, b, f, n, k, a, d);
    const R z = Z[((size_t)b * N + n) * K + k];
    num += z * a;
    den += z * d;
  }
  den = floor_eps<R>(den, eps);
  Tb[idx] = Tb[idx] * sqrt(num / den);
}

// V[k,t] *= sqrt(num/den), records of all sources summed     (ilrma.py:399-408)
template <typename R>
__global__ void __launch_bounds__(256) part_act_kernel(const R* __restrict__ part, R* __restrict__ V, int B, int N,
                                                      int F, int K, int T, FlatPart fp, R eps) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * K * T) return;
  const int t = idx % T;
  const int k = (idx / T) % K;
  const int b = idx / ((size_t)T * K);
  const int TBk = (T + WAVE - 1) / WAVE;
  const long long j = (long long)b * TBk + t / WAVE;
  const int lane = t % WAVE;
  int g_lo, g_hi;
  flat_cover(fp, j, g_lo, g_hi);
  R num = 0, den = 0;
  for (int n = 0; n < N; ++n)
    for (int g = g_lo; g <= g_hi; ++g) {
      const int slot = flat_slot(fp, j, g);
      const R* p = part + ((((size_t)g * fp.S + slot) * N + n) * (size_t)(2 * K) + k * 2) * WAVE + lane;
      num += p[0];
      den += p[WAVE];
    }
  den = floor_eps<R>(den, eps);
  V[idx] = V[idx] * sqrt(num / den);
}

// n_basis > 4: the per-source sums come from the NMF matrix-core kernels run on the demixed power with the effective
// model (batch = B*N, see nmf_half_partials); these two kernels add up the slabs and lay the result out as the
// records the kernels above read, with the trivial partition "one workgroup per bin / per frame block, one slot".
template <typename R>
__global__ void __launch_bounds__(256) part_adapt_basis_kernel(const R* __restrict__ nmf_part, R* __restrict__ out,
                                                              int B, int N, int F, int K, int slabs) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // ((b*F + f)*N + n)*2K + 2k + s
  if (idx >= (size_t)B * F * N * 2 * K) return;
  const int s = idx & 1, k = (idx >> 1) % K;
  const int n = (idx / (2 * K)) % N;
  const int f = (idx / ((size_t)2 * K * N)) % F;
  const int b = idx / ((size_t)2 * K * N * F);
  const size_t FK = (size_t)F * K, slab = (size_t)B * N * 2 * FK;
  const R* p = nmf_part + ((size_t)(b * N + n) * 2 + s) * FK + (size_t)f * K + k;
  R acc = 0;
  for (int i = 0; i < slabs; ++i) acc += p[(size_t)i * slab];
  out[idx] = acc;
}

template <typename R>
__global__ void __launch_bounds__(256) part_adapt_act_kernel(const R* __restrict__ nmf_part, R* __restrict__ out, int B,
                                                            int N, int K, int T, int slabs) {
  const int TBk = (T + WAVE - 1) / WAVE;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (((b*TBk + tb)*N + n)*2K + 2k + s)*64 + lane
  if (idx >= (size_t)B * TBk * N * 2 * K * WAVE) return;
  const int lane = idx % WAVE;
  const int s = (idx / WAVE) & 1, k = (idx / (2 * WAVE)) % K;
  const int n = (idx / ((size_t)2 * K * WAVE)) % N;
  const int tb = (idx / ((size_t)2 * K * WAVE * N)) % TBk;
  const int b = idx / ((size_t)2 * K * WAVE * N * TBk);
  const int t = tb * WAVE + lane;
  R acc = 0;
  if (t < T) {
    const size_t KT = (size_t)K * T, slab = (size_t)B * N * 2 * KT;
    const R* p = nmf_part + ((size_t)(b * N + n) * 2 + s) * KT + (size_t)k * T + t;
    for (int i = 0; i < slabs; ++i) acc += p[(size_t)i * slab];
  }
  out[idx] = acc;
}

// 'power' normalisation with a partitioning function (ilrma.py:313-320): W[n] /= a[n];  Z' = Z / a^2;
// T *= sum_n Z'[n,k];  Z = Z' / sum_n Z'.  Every workgroup derives a[] and the column sums for its utterance
// (tiny, fixed order), rescales its slice of W / T; workgroup 0 of the utterance also stores the new Z into Zout
// (scratch: other workgroups are still reading Z; the host wrapper copies Zout over Z afterwards).
template <typename R, int N>
__global__ void __launch_bounds__(256) part_normalize_power_kernel(Cx<R>* __restrict__ W, R* __restrict__ Zout,
                                                                  const R* __restrict__ Zin, R* __restrict__ Tb,
                                                                  const double* __restrict__ pbins, int F, int K,
                                                                  R eps) {
  __shared__ R anorm[N];
  extern __shared__ unsigned char smem_raw[];
  R* zsum = reinterpret_cast<R*>(smem_raw);  // [K]
  const int b = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
  for (int n = wv; n < N; n += 4) {
    const double* p = pbins + ((size_t)b * N + n) * F;
    double s = 0.0;
    for (int f = lane; f < F; f += WAVE) s += p[f];
    s = wave_allreduce_sum<double>(s);
    if (lane == 0) anorm[n] = floor_eps<R>(sqrt((R)(s / (double)F)), eps);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) {
    R s = 0;
#pragma unroll
    for (int n = 0; n < N; ++n) s += Zin[((size_t)b * N + n) * K + k] / (anorm[n] * anorm[n]);
    zsum[k] = s;
    if (blockIdx.x == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n)
        Zout[((size_t)b * N + n) * K + k] = (Zin[((size_t)b * N + n) * K + k] / (anorm[n] * anorm[n])) / s;
    }
  }
  __syncthreads();
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nW = (size_t)F * N * N, nT = (size_t)F * K;
  if (idx < nW) {
    const int n = (idx / N) % N;
    Cx<R>* w = W + (size_t)b * nW + idx;
    const R a = anorm[n];
    *w = cmake<R>(w->x / a, w->y / a);
  } else if (idx < nW + nT) {
    const size_t j = idx - nW;
    R* t = Tb + (size_t)b * nT + j;
    *t = *t * zsum[j % K];
  }
}

// run-time source count (round 6): the statements of part_normalize_power_kernel
template <typename R, int NMAX>
__global__ void __launch_bounds__(256) part_normalize_power_rt_kernel(Cx<R>* __restrict__ W, R* __restrict__ Zout,
                                                                     const R* __restrict__ Zin, R* __restrict__ Tb,
                                                                     const double* __restrict__ pbins, int F, int K,
                                                                     R eps, int N) {
  __shared__ R anorm[NMAX];
  extern __shared__ unsigned char smem_raw[];
  R* zsum = reinterpret_cast<R*>(smem_raw);  // [K]
  const int b = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
  for (int n = wv; n < N; n += 4) {
    const double* p = pbins + ((size_t)b * N + n) * F;
    double s = 0.0;
    for (int f = lane; f < F; f += WAVE) s += p[f];
    s = wave_allreduce_sum<double>(s);
    if (lane == 0) anorm[n] = floor_eps<R>(sqrt((R)(s / (double)F)), eps);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) {
    R s = 0;
    for (int n = 0; n < N; ++n) s += Zin[((size_t)b * N + n) * K + k] / (anorm[n] * anorm[n]);
    zsum[k] = s;
    if (blockIdx.x == 0)
      for (int n = 0; n < N; ++n)
        Zout[((size_t)b * N + n) * K + k] = (Zin[((size_t)b * N + n) * K + k] / (anorm[n] * anorm[n])) / s;
  }
  __syncthreads();
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nW = (size_t)F * N * N, nT = (size_t)F * K;
  if (idx < nW) {
    const int n = (idx / N) % N;
    Cx<R>* w = W + (size_t)b * nW + idx;
    const R a = anorm[n];
    *w = cmake<R>(w->x / a, w->y / a);
  } else if (idx < nW + nT) {
    const size_t j = idx - nW;
    R* t = Tb + (size_t)b * nT + j;
    *t = *t * zsum[j % K];
  }
}

}  // namespace assx
