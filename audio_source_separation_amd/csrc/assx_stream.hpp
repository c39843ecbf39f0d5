// Streaming kernels of the ILRMA / AuxIVA iteration: the three passes over X that dominate the runtime.
//
// Design (measured on MI355X):
//   * ONE WAVE PER SOURCE.  A wave keeps only its own source's accumulators (M*M = 16 reals for the
//     covariance, 2*KU = 8 for the NMF contractions) instead of all N of them, which takes the kernels
//     from ~230 VGPRs (2 waves/SIMD, latency-bound at ~2.5 TB/s) to < 100 VGPRs.  The N waves that share a
//     bin sit in ONE workgroup and walk the same frames together, so X comes from HBM once and the other
//     N-1 reads hit that CU's L1 / the XCD's L2.
//   * FLAT BALANCED PARTITION.  The (utterance, bin, 64-frame block) space is flattened and cut into G equal
//     contiguous ranges, G = a small multiple of what the chip holds concurrently, so F = 1025 bins never
//     quantise badly against 256 CUs.  A range may straddle a bin boundary: the wave then flushes its
//     accumulators (a 15-shuffle butterfly reduce-scatter) into a per-(workgroup, slot) partial record;
//     finalize kernels know which records cover a bin from the partition arithmetic alone (no atomics,
//     run-to-run bit-stable).
//   * REGISTER DOUBLE BUFFERING.  The loads of block q+1 (X row slices and the weight inputs) are issued
//     before block q is consumed.
#pragma once
#include "assx_common.hpp"

namespace assx {

enum { WK_NONE = 0, WK_NT = 1, WK_NFT = 2, WK_TV = 3 };

constexpr int KU = 4;  // k-unroll of the NMF contractions (K <= 4 is the single-chunk fast path)

struct Dims {
  int B, F, T, K;
};

struct FlatPart {
  long long NB;  // items in the flattened space
  int len;       // items per group (a group = one bin's frame blocks, or one frame block's bins)
  int L;         // items per workgroup
  int G;         // workgroups
  int S;         // partial-record slots per workgroup
};

inline FlatPart make_flat(long long NB, int len, long long G_target) {
  FlatPart p;
  p.NB = NB;
  p.len = len;
  long long G = G_target < NB ? G_target : NB;
  if (G < 1) G = 1;
  p.L = (int)((NB + G - 1) / G);
  p.G = (int)((NB + p.L - 1) / p.L);
  p.S = (p.L + len - 2) / len + 1;
  return p;
}

// 1/x: v_rcp + Newton steps instead of the 11-instruction IEEE division expansion (|rel err| < 2^-52)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  return r;
}
__device__ __forceinline__ float fast_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = fmaf(-x, r, 1.0f);
  return fmaf(r, e, r);
}

// (utterance, bin, frame-block) cursor advanced incrementally: a 64-bit division per block would cost more
// scalar instructions than the block's arithmetic.
struct Cursor {
  int b, f, tb;
};
__device__ __forceinline__ void advance(Cursor& c, int TBk, int F) {
  if (++c.tb == TBk) {
    c.tb = 0;
    if (++c.f == F) {
      c.f = 0;
      ++c.b;
    }
  }
}

constexpr int DX = 4;  // X prefetch depth in 64-frame blocks: keeps >= 4 KB of X in flight per wave (Little's law:
                       // ~50 KB per CU are needed to cover HBM latency at 6 TB/s)
constexpr int DW = 2;  // prefetch depth of the weight inputs (L2-resident)

template <int M>
__host__ __device__ constexpr int herm_pair_base(int m, int l) {  // requires m < l
  return M + 2 * (m * M - m * (m + 1) / 2 + (l - m - 1));
}

// ------------------------------------------------------------------------------------------
// (a4) weighted covariance, streaming.  part[g][slot][n][HM] packed-Hermitian un-normalised sums.
// ------------------------------------------------------------------------------------------
// Array pointers are passed as separate `const ... __restrict__` kernel parameters (not inside the struct): only
// then can the compiler prove them read-only / non-aliased and fetch wave-uniform per-bin constants with scalar
// loads into SGPRs instead of a VGPR per lane.
template <typename R>
struct CovArgs {
  Dims d;
  FlatPart fp;   // items = (b, f, tb), len = TBk
  R eps;
  PowSpec p2d;   // 2/domain
};

// D2: domain == 2 fast path (2/domain == 1, (domain+2)/domain == 2): the generic pow() expansion costs ~70 VGPRs
// even when never executed, so it is compiled only into the D2 == false instantiations.
//
// One wave per workgroup forms the M*M Hermitian products of a frame ONCE and accumulates every source's weighted
// sum (fewest instructions per frame; measured 2.7x fewer than one wave per source).
// LS: lane split.  LS == 1: a lane owns one frame and all N sources (N*M*M accumulators).  LS == 2 (M = 4, f64):
// lanes 0-31 own sources {0,1}, lanes 32-63 sources {2,3}, both halves walk the SAME 32 frames -- 32 accumulators
// and 8 weight inputs per lane instead of 64 and 16, which is what lets the TV-weighted f64 kernel keep 2 waves
// per SIMD without spilling.  The unit of the flat partition is a block of FB = 64 / LS frames.
// Latency is covered by the DXT-deep X prefetch ring (and a DWT-deep ring of the weight inputs), not by occupancy.
template <typename R, int M, int WK, bool K4, bool D2, int LS, int DXT, int DWT, int MINW = 1>
__global__ void __launch_bounds__(64, MINW)
    cov_stream_kernel(const Cx<R>* __restrict__ X, const R* __restrict__ rw /* WK_NT (B,N,T) | WK_NFT (B,N,F,T) */,
                      const R* __restrict__ Tb /* WK_TV (B,N,F,K) */, const R* __restrict__ V /* WK_TV (B,N,K,T) */,
                      R* __restrict__ part, CovArgs<R> a) {
  constexpr int N = (WK == WK_NONE) ? 1 : M;
  static_assert(N % LS == 0 && DXT % DWT == 0, "lane split / ring depths");
  constexpr int FB = WAVE / LS;   // frames per block
  constexpr int SPL = N / LS;     // sources per lane
  constexpr int HM = M * M;
  constexpr int NACC = SPL * HM;
  constexpr int NV = next_pow2_c(NACC);
  static_assert(NV <= FB, "accumulators per lane must not exceed the lanes of a group");
  constexpr int NWV = (WK == WK_TV) ? KU : 1;  // prefetched weight inputs per (source, frame)
  const int lane = threadIdx.x & (WAVE - 1);
  const int fl = lane & (FB - 1);              // frame within the block
  const int s0 = (lane / FB) * SPL;            // first source of this lane group
  const int F = a.d.F, T = a.d.T, K = a.d.K, TBk = a.fp.len;
  const size_t FT = (size_t)F * T;
  const int g = blockIdx.x;
  const long long q0 = (long long)g * a.fp.L;
  const long long q1 = (q0 + a.fp.L < a.fp.NB) ? q0 + a.fp.L : a.fp.NB;
  if (q0 >= q1) return;
  const int nblk = (int)(q1 - q0);
  const int bf_first = (int)(q0 / TBk);  // the only divisions: once per workgroup
  Cursor cc;                             // consume cursor
  cc.tb = (int)(q0 - (long long)bf_first * TBk);
  cc.b = bf_first / F;
  cc.f = bf_first - cc.b * F;
  Cursor px = cc, pw = cc;               // prefetch cursors (X ring, weight ring)
  const bool ragged = (T % FB) != 0;     // only then can a lane fall beyond the last frame

  R acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0;

  auto issue_x = [&](const Cursor& c, Vec2<R>(&x)[M]) {
    const int t = c.tb * FB + fl;
    const unsigned tc = (unsigned)(t < T ? t : T - 1);
    // row base = wave-uniform 64-bit pointer (SGPR pair), lane part = 32-bit frame offset: the loads use the
    // SGPR-base + VGPR-offset addressing form and need no per-lane 64-bit address arithmetic
    const Cx<R>* xb = X + (size_t)c.b * M * FT;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const Cx<R>* row = xb + (size_t)((unsigned)(m * F + c.f) * (unsigned)T);
      x[m] = ldv_so<R>(row, tc * (unsigned)sizeof(Cx<R>));
    }
  };
  auto issue_w = [&](const Cursor& c, R(&wv)[SPL][NWV]) {
    const int t = c.tb * FB + fl;
    const unsigned tc = (unsigned)(t < T ? t : T - 1);
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      const int n = s0 + j;
      if (WK == WK_NT) {
        const R* row = rw + ((size_t)c.b * N + n) * T;
        wv[j][0] = ld_so<R>(row, tc * (unsigned)sizeof(R));
      } else if (WK == WK_NFT) {
        const R* row = rw + (size_t)c.b * N * FT + (size_t)((unsigned)(n * F + c.f) * (unsigned)T);
        wv[j][0] = ld_so<R>(row, tc * (unsigned)sizeof(R));
      } else if (WK == WK_TV && K4) {
        const R* vb = V + (size_t)c.b * N * K * T;
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) {
          const R* row = vb + (size_t)((unsigned)(n * K + (kk < K ? kk : K - 1)) * (unsigned)T);
          wv[j][kk] = ld_so<R>(row, tc * (unsigned)sizeof(R));
        }
      }
    }
  };
  R tbr[SPL][KU];
  auto load_basis_row = [&](const Cursor& c) {
    if (WK == WK_TV && K4) {
#pragma unroll
      for (int j = 0; j < SPL; ++j) {
        const R* tbn = Tb + (((size_t)c.b * N + s0 + j) * F + c.f) * K;
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) {
          const R v = (kk < K) ? tbn[kk] : (R)0;
          tbr[j][kk] = v;
        }
      }
    }
  };

  Vec2<R> xq[DXT][M];
  R wq[DWT][SPL][NWV];
#pragma unroll
  for (int j = 0; j < DXT; ++j) {
    if (j < nblk) {
      issue_x(px, xq[j]);
      advance(px, TBk, F);
    }
  }
  if (WK != WK_NONE) {
#pragma unroll
    for (int j = 0; j < DWT; ++j) {
      if (j < nblk) {
        issue_w(pw, wq[j]);
        advance(pw, TBk, F);
      }
    }
  }
  load_basis_row(cc);

  for (int it0 = 0; it0 < nblk; it0 += DXT) {
#pragma unroll
    for (int j = 0; j < DXT; ++j) {
      const int it = it0 + j;
      if (it < nblk) {  // (a `break` here would defeat the unroll and push the ring into scratch)
        Cx<R> x[M];
        R wv[SPL][NWV];
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = tocx<R>(xq[j][m]);
#pragma unroll
        for (int q = 0; q < SPL; ++q)
#pragma unroll
          for (int i = 0; i < NWV; ++i) wv[q][i] = wq[j % DWT][q][i];
        if (it + DXT < nblk) {
          issue_x(px, xq[j]);
          advance(px, TBk, F);
        }
        if (WK != WK_NONE && it + DWT < nblk) {
          issue_w(pw, wq[j % DWT]);
          advance(pw, TBk, F);
        }
        const Cursor cur = cc;
        advance(cc, TBk, F);
        const bool more = it + 1 < nblk;

        // ---- consume block `cur`: weights first (frees the weight inputs), Hermitian products once, then one
        //      weighted accumulate per source
        const int t = cur.tb * FB + fl;
        R wgt[SPL];
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
          if (WK == WK_NONE) {
            wgt[q] = 1;
          } else {
            R r;
            if (WK == WK_TV) {
              R tv = 0;
              if (K4) {
#pragma unroll
                for (int kk = 0; kk < KU; ++kk) tv = fma(tbr[q][kk], wv[q][kk], tv);
              } else {
                const R* tbn = Tb + (((size_t)cur.b * N + s0 + q) * F + cur.f) * K;
                const R* vb = V + ((size_t)cur.b * N + s0 + q) * K * T + (t < T ? t : T - 1);
                for (int k = 0; k < K; ++k) tv = fma(tbn[k], vb[(size_t)k * T], tv);
              }
              r = D2 ? tv : powspec<R>(tv, a.p2d);  // R = (T V)^(2/domain), floored AFTER the power (ilrma.py:499-509)
            } else {
              r = wv[q][0];
            }
            wgt[q] = fast_rcp(floor_eps<R>(r, a.eps));
          }
          if (ragged && t >= T) wgt[q] = 0;
        }
        // each Hermitian product is formed once and fanned into every source's accumulator right away, so the
        // M*M products are never all live (saves ~28 VGPRs in f64 -- the margin that keeps 2 waves per SIMD)
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const R pd = cabs2(x[m]);
#pragma unroll
          for (int q = 0; q < SPL; ++q) acc[q * HM + m] = fma(wgt[q], pd, acc[q * HM + m]);
        }
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
          for (int l = m + 1; l < M; ++l) {
            const Cx<R> pr = cmulc(x[m], x[l]);
            const int hb = herm_pair_base<M>(m, l);
#pragma unroll
            for (int q = 0; q < SPL; ++q) {
              acc[q * HM + hb] = fma(wgt[q], pr.x, acc[q * HM + hb]);
              acc[q * HM + hb + 1] = fma(wgt[q], pr.y, acc[q * HM + hb + 1]);
            }
          }

        if (cc.tb == 0 || !more) {  // the bin is complete (or the range ends): flush the partial record
          const R tot = wave_reduce_scatter<R, NV, FB>(acc);
          const int i = scatter_index<NV, FB>();
          const int slot = cur.b * F + cur.f - bf_first;
          if (scatter_leader<NV, FB>() && i < NACC)
            part[(((size_t)g * a.fp.S + slot) * N + s0) * HM + i] = tot;
#pragma unroll
          for (int q = 0; q < NV; ++q) acc[q] = 0;
          if (more) load_basis_row(cc);
        }
      }
    }
  }
}

// sum the records covering each bin, scale by 1/T, expand packed Hermitian -> dense U (B,N,F,M,M)
template <typename R, int M>
__global__ void __launch_bounds__(256) cov_stream_finalize_kernel(const R* __restrict__ part, Cx<R>* __restrict__ U,
                                                                 int B, int N, int F, FlatPart fp, R inv_T) {
  constexpr int HM = M * M;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * F * HM;
  if (idx >= total) return;
  const int l = idx % M, m = (idx / M) % M;
  const int f = (idx / HM) % F;
  const int n = (idx / ((size_t)HM * F)) % N;
  const int b = idx / ((size_t)HM * F * N);
  const long long j = (long long)b * F + f;
  const int g_lo = (int)((j * fp.len) / fp.L), g_hi = (int)(((j + 1) * fp.len - 1) / fp.L);
  R re = 0, im = 0;
  for (int g = g_lo; g <= g_hi; ++g) {
    const int slot = (int)(j - ((long long)g * fp.L) / fp.len);
    const R* p = part + (((size_t)g * fp.S + slot) * N + n) * HM;
    if (m == l) {
      re += p[m];
    } else {
      const int lo = m < l ? m : l, hi = m < l ? l : m;
      const int base = herm_pair_base<M>(lo, hi);
      re += p[base];
      im += p[base + 1];
    }
  }
  if (m > l) im = -im;
  U[idx] = cmake<R>(re * inv_T, im * inv_T);
}

// ------------------------------------------------------------------------------------------
// (a2) ILRMA source model, basis half (reduce over t).  part[g][slot][n][k][{num,den}]
//      One wave per workgroup handles all N sources of its frame blocks (same structure as cov_stream_kernel).
// ------------------------------------------------------------------------------------------
template <typename R>
struct NmfArgs {
  Dims d;
  FlatPart fp;
  R eps;
  PowSpec p1;  // (domain+2)/domain
  R nu;        // t-ILRMA degree of freedom (TD instantiations only)
};

// t-ILRMA (ilrma.py:905-909): harmonic = 1 / (2/((2+nu) TV) + nu/((2+nu) P)), evaluated as (2+nu) TV P / (2 P + nu TV)
// so that P == 0 gives 0 (as 1/inf does in the reference) without an inf/NaN detour through the reciprocal.
template <typename R>
__device__ __forceinline__ R t_harmonic(R P, R tv, R nu) {
  return ((R)2 + nu) * tv * P * fast_rcp(fma(nu, tv, (R)2 * P));
}

template <typename R, int M, bool K4, bool D2, int DXT, int DWT, int MINW = 1, bool TD = false>
__global__ void __launch_bounds__(64, MINW)
    basis_stream_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W, const R* __restrict__ Tb,
                        const R* __restrict__ V, R* __restrict__ part, NmfArgs<R> a) {
  constexpr int N = M;
  constexpr int NACC = N * KU * 2;
  constexpr int NV = next_pow2_c(NACC);
  static_assert(NV <= WAVE && DXT % DWT == 0, "accumulators / ring depths");
  const int lane = threadIdx.x & (WAVE - 1);
  const int F = a.d.F, T = a.d.T, K = a.d.K, TBk = a.fp.len;
  const size_t FT = (size_t)F * T;
  const int g = blockIdx.x;
  const long long q0 = (long long)g * a.fp.L;
  const long long q1 = (q0 + a.fp.L < a.fp.NB) ? q0 + a.fp.L : a.fp.NB;
  if (q0 >= q1) return;
  const int nblk = (int)(q1 - q0);
  const int bf_first = (int)(q0 / TBk);  // the only divisions: once per workgroup
  Cursor c0;
  c0.tb = (int)(q0 - (long long)bf_first * TBk);
  c0.b = bf_first / F;
  c0.f = bf_first - c0.b * F;
  const int nchunks = K4 ? 1 : (K + KU - 1) / KU;
  const bool ragged = (T % WAVE) != 0;

  for (int c = 0; c < nchunks; ++c) {
    const int k0 = c * KU;
    Cursor cc = c0, px = c0, pw = c0;
    R acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0;

    auto issue_x = [&](const Cursor& cu, Vec2<R>(&x)[M]) {
      const int t = cu.tb * WAVE + lane;
      const unsigned tc = (unsigned)(t < T ? t : T - 1);
      const Cx<R>* xb = X + (size_t)cu.b * M * FT;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const Cx<R>* row = xb + (size_t)((unsigned)(m * F + cu.f) * (unsigned)T);  // uniform base + 32-bit lane offset
        x[m] = ldv_so<R>(row, tc * (unsigned)sizeof(Cx<R>));
      }
    };
    auto issue_v = [&](const Cursor& cu, R(&v)[N][KU]) {
      const int t = cu.tb * WAVE + lane;
      const unsigned tc = (unsigned)(t < T ? t : T - 1);
      const R* vb = V + (size_t)cu.b * N * K * T;
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) {
          const int k = k0 + kk;
          const R* row = vb + (size_t)((unsigned)(n * K + (k < K ? k : K - 1)) * (unsigned)T);
          v[n][kk] = ld_so<R>(row, tc * (unsigned)sizeof(R));
        }
    };
    Cx<R> w[N][M];
    R tbr[N][KU];
    auto load_rows = [&](const Cursor& cu) {  // wave-uniform: scalar loads
      const Cx<R>* wp = W + ((size_t)cu.b * F + cu.f) * (N * M);
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int m = 0; m < M; ++m) w[n][m] = wp[n * M + m];
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const R* tbn = Tb + (((size_t)cu.b * N + n) * F + cu.f) * K;
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) tbr[n][kk] = (k0 + kk < K) ? tbn[k0 + kk] : (R)0;
      }
    };

    Vec2<R> xq[DXT][M];
    R vq[DWT][N][KU];
#pragma unroll
    for (int j = 0; j < DXT; ++j) {
      if (j < nblk) {
        issue_x(px, xq[j]);
        advance(px, TBk, F);
      }
    }
#pragma unroll
    for (int j = 0; j < DWT; ++j) {
      if (j < nblk) {
        issue_v(pw, vq[j]);
        advance(pw, TBk, F);
      }
    }
    load_rows(cc);

    for (int it0 = 0; it0 < nblk; it0 += DXT) {
#pragma unroll
      for (int j = 0; j < DXT; ++j) {
        const int it = it0 + j;
        if (it < nblk) {
          Cx<R> x[M];
          R v[N][KU];
#pragma unroll
          for (int m = 0; m < M; ++m) x[m] = tocx<R>(xq[j][m]);
#pragma unroll
          for (int n = 0; n < N; ++n)
#pragma unroll
            for (int kk = 0; kk < KU; ++kk) v[n][kk] = vq[j % DWT][n][kk];
          if (it + DXT < nblk) {
            issue_x(px, xq[j]);
            advance(px, TBk, F);
          }
          if (it + DWT < nblk) {
            issue_v(pw, vq[j % DWT]);
            advance(pw, TBk, F);
          }
          const Cursor cur = cc;
          advance(cc, TBk, F);
          const bool more = it + 1 < nblk;

          const int t = cur.tb * WAVE + lane;
#pragma unroll
          for (int n = 0; n < N; ++n) {
            Cx<R> y = cmake<R>(0, 0);
#pragma unroll
            for (int m = 0; m < M; ++m) cfma(y, w[n][m], x[m]);
            R P = cabs2(y);
            R tv = 0;
            if (K4) {
#pragma unroll
              for (int kk = 0; kk < KU; ++kk) tv = fma(tbr[n][kk], v[n][kk], tv);
            } else {
              const R* tbn = Tb + (((size_t)cur.b * N + n) * F + cur.f) * K;
              const R* vb = V + ((size_t)cur.b * N + n) * K * T + (t < T ? t : T - 1);
              for (int k = 0; k < K; ++k) tv = fma(tbn[k], vb[(size_t)k * T], tv);
            }
            tv = floor_eps<R>(tv, a.eps);
            if (TD) P = t_harmonic<R>(P, tv, a.nu);
            R inv = fast_rcp(tv);                                 // TV_inverse
            R D = D2 ? P * inv * inv : P / powspec<R>(tv, a.p1);   // division = P / TV**((d+2)/d)
            if (ragged && t >= T) {
              inv = 0;
              D = 0;
            }
#pragma unroll
            for (int kk = 0; kk < KU; ++kk) {
              acc[(n * KU + kk) * 2 + 0] = fma(D, v[n][kk], acc[(n * KU + kk) * 2 + 0]);
              acc[(n * KU + kk) * 2 + 1] = fma(inv, v[n][kk], acc[(n * KU + kk) * 2 + 1]);
            }
          }

          if (cc.tb == 0 || !more) {
            const R tot = wave_reduce_scatter<R, NV>(acc);
            const int i = scatter_index<NV>();
            const int slot = cur.b * F + cur.f - bf_first;
            const int n = i / (2 * KU), k = k0 + (i >> 1) % KU;
            if (scatter_leader<NV>() && i < NACC && k < K)
              part[(((size_t)g * a.fp.S + slot) * N + n) * (size_t)(2 * K) + k * 2 + (i & 1)] = tot;
#pragma unroll
            for (int q = 0; q < NV; ++q) acc[q] = 0;
            if (more) load_rows(cc);
          }
        }
      }
    }
  }
}

// T *= (num / max(den, eps)) ** (d/(d+2))      (ilrma.py:417-419)
template <typename R>
__global__ void __launch_bounds__(256) basis_stream_finalize_kernel(const R* __restrict__ part, R* __restrict__ Tb,
                                                                   int B, int N, int F, int K, FlatPart fp, R eps,
                                                                   PowSpec p2, unsigned src_mask) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * F * K;
  if (idx >= total) return;
  const int k = idx % K;
  const int f = (idx / K) % F;
  const int n = (idx / ((size_t)K * F)) % N;
  if (!((src_mask >> n) & 1u)) return;  // pairwise source-model update: only the selected sources move
  const int b = idx / ((size_t)K * F * N);
  const long long j = (long long)b * F + f;
  const int g_lo = (int)((j * fp.len) / fp.L), g_hi = (int)(((j + 1) * fp.len - 1) / fp.L);
  R num = 0, den = 0;
  for (int g = g_lo; g <= g_hi; ++g) {
    const int slot = (int)(j - ((long long)g * fp.L) / fp.len);
    const R* p = part + (((size_t)g * fp.S + slot) * N + n) * (size_t)(2 * K) + k * 2;
    num += p[0];
    den += p[1];
  }
  den = floor_eps<R>(den, eps);
  Tb[idx] = Tb[idx] * powspec<R>(num / den, p2);
}

// ------------------------------------------------------------------------------------------
// (a7) ILRMA negative log-likelihood, data term: sum_{n,t} P/R + log R per workgroup range (ilrma.py:672-675).
//      Same streaming structure as the basis kernel, one float64 accumulator per lane.  The flat partition is
//      per utterance (blockIdx.y) so a partial never mixes utterances: lpart[b][g].
// ------------------------------------------------------------------------------------------
template <typename R, int M, bool K4, bool D2, int DXT, int DWT, int MINW = 1, bool TD = false>
__global__ void __launch_bounds__(64, MINW)
    loss_stream_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W, const R* __restrict__ Tb,
                       const R* __restrict__ V, double* __restrict__ lpart, int lstride, NmfArgs<R> a, PowSpec p2d) {
  constexpr int N = M;
  static_assert(DXT % DWT == 0, "ring depths");
  const int lane = threadIdx.x & (WAVE - 1);
  const int F = a.d.F, T = a.d.T, K = a.d.K, TBk = a.fp.len;
  const size_t FT = (size_t)F * T;
  const int g = blockIdx.x, b = blockIdx.y;
  const long long q0 = (long long)g * a.fp.L;
  const long long q1 = (q0 + a.fp.L < a.fp.NB) ? q0 + a.fp.L : a.fp.NB;
  double acc = 0.0;
  // sum log R is carried as log(prod R): running mantissa product + integer exponent (frexp once per block), ONE
  // log per lane at the end instead of N per frame.
  double lm = 1.0;
  int le = 0;
  // t-ILRMA (ilrma.py:1015-1016): sum (1 + nu/2) log(1 + (2/nu) P/R) carried the same way
  double tm = 1.0;
  int te = 0;
  if (q0 < q1) {
    const int nblk = (int)(q1 - q0);
    Cursor cc;
    cc.f = (int)(q0 / TBk);
    cc.tb = (int)(q0 - (long long)cc.f * TBk);
    cc.b = b;
    Cursor px = cc, pw = cc;
    const Cx<R>* xb = X + (size_t)b * M * FT;
    const R* vb = V + (size_t)b * N * K * T;
    auto issue_x = [&](const Cursor& cu, Vec2<R>(&x)[M]) {
      const int t = cu.tb * WAVE + lane;
      const unsigned tc = (unsigned)(t < T ? t : T - 1);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const Cx<R>* row = xb + (size_t)((unsigned)(m * F + cu.f) * (unsigned)T);
        x[m] = ldv_so<R>(row, tc * (unsigned)sizeof(Cx<R>));
      }
    };
    auto issue_v = [&](const Cursor& cu, R(&v)[N][KU]) {
      const int t = cu.tb * WAVE + lane;
      const unsigned tc = (unsigned)(t < T ? t : T - 1);
      if (K4) {
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
          for (int kk = 0; kk < KU; ++kk) {
            const R* row = vb + (size_t)((unsigned)(n * K + (kk < K ? kk : K - 1)) * (unsigned)T);
            v[n][kk] = ld_so<R>(row, tc * (unsigned)sizeof(R));
          }
      }
    };
    Cx<R> w[N][M];
    R tbr[N][KU];
    auto load_rows = [&](const Cursor& cu) {  // wave-uniform: scalar loads, once per bin
      const Cx<R>* wp = W + ((size_t)b * F + cu.f) * (N * M);
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int m = 0; m < M; ++m) w[n][m] = wp[n * M + m];
      if (K4) {
#pragma unroll
        for (int n = 0; n < N; ++n) {
          const R* tbn = Tb + (((size_t)b * N + n) * F + cu.f) * K;
#pragma unroll
          for (int kk = 0; kk < KU; ++kk) tbr[n][kk] = (kk < K) ? tbn[kk] : (R)0;
        }
      }
    };
    Vec2<R> xq[DXT][M];
    R vq[DWT][N][KU];
#pragma unroll
    for (int j = 0; j < DXT; ++j) {
      if (j < nblk) {
        issue_x(px, xq[j]);
        advance(px, TBk, F);
      }
    }
#pragma unroll
    for (int j = 0; j < DWT; ++j) {
      if (j < nblk) {
        issue_v(pw, vq[j]);
        advance(pw, TBk, F);
      }
    }
    load_rows(cc);
    for (int it0 = 0; it0 < nblk; it0 += DXT) {
#pragma unroll
      for (int j = 0; j < DXT; ++j) {
        const int it = it0 + j;
        if (it < nblk) {
          Cx<R> x[M];
          R v[N][KU];
#pragma unroll
          for (int m = 0; m < M; ++m) x[m] = tocx<R>(xq[j][m]);
#pragma unroll
          for (int n = 0; n < N; ++n)
#pragma unroll
            for (int kk = 0; kk < KU; ++kk) v[n][kk] = vq[j % DWT][n][kk];
          if (it + DXT < nblk) {
            issue_x(px, xq[j]);
            advance(px, TBk, F);
          }
          if (it + DWT < nblk) {
            issue_v(pw, vq[j % DWT]);
            advance(pw, TBk, F);
          }
          const Cursor cur = cc;
          advance(cc, TBk, F);
          const int t = cur.tb * WAVE + lane;
          double term = 0.0, rprod = 1.0, tprod = 1.0;
#pragma unroll
          for (int n = 0; n < N; ++n) {
            Cx<R> y = cmake<R>(0, 0);
#pragma unroll
            for (int m = 0; m < M; ++m) cfma(y, w[n][m], x[m]);
            R tv = 0;
            if (K4) {
#pragma unroll
              for (int kk = 0; kk < KU; ++kk) tv = fma(tbr[n][kk], v[n][kk], tv);
            } else {
              const unsigned tc = (unsigned)(t < T ? t : T - 1);
              const R* tbn = Tb + (((size_t)b * N + n) * F + cur.f) * K;
              for (int k = 0; k < K; ++k) tv = fma(tbn[k], vb[(unsigned)(n * K + k) * (unsigned)T + tc], tv);
            }
            const R r = floor_eps<R>(D2 ? tv : powspec<R>(tv, p2d), a.eps);
            if (TD) tprod *= fma((double)((R)2 * fast_rcp(a.nu)), (double)(cabs2(y) * fast_rcp(r)), 1.0);
            else term += (double)(cabs2(y) * fast_rcp(r));
            rprod *= (double)r;
          }
          if (t < T) {
            acc += term;
            int e;
            lm = frexp(lm * rprod, &e);
            le += e;
            if (TD) {
              tm = frexp(tm * tprod, &e);
              te += e;
            }
          }
          if (cc.tb == 0 && it + 1 < nblk) load_rows(cc);
        }
      }
    }
  }
  acc += (double)le * 0.6931471805599453 + log(lm);
  if (TD) acc += (1.0 + 0.5 * (double)a.nu) * ((double)te * 0.6931471805599453 + log(tm));
  acc = wave_allreduce_sum<double>(acc);
  if (lane == 0) lpart[(size_t)b * lstride + g] = acc;
}

// ------------------------------------------------------------------------------------------
// (a2) activation half (reduce over f).  Lanes own 64 frames; a workgroup = ACT_NH waves walking interleaved
//      bins of the same frame block, each wave handling all N sources; the streams are combined through LDS.
//      items = (b, tb, f), len = F.   part[g][slot][n][k][{num,den}][64]
// ------------------------------------------------------------------------------------------
constexpr int ACT_NH = 2;

template <typename R, int M, bool K4, bool D2, int DXT, int MINW = 1, bool TD = false>
__global__ void __launch_bounds__(64 * ACT_NH, MINW)
    act_stream_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W, const R* __restrict__ Tb,
                      const R* __restrict__ V, R* __restrict__ part, NmfArgs<R> a) {
  constexpr int N = M;
  constexpr int NACC = N * KU * 2;
  __shared__ R lds[(ACT_NH - 1) * NACC * WAVE];
  const int lane = threadIdx.x & (WAVE - 1);
  const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int F = a.d.F, T = a.d.T, K = a.d.K;
  const int TBk = (T + WAVE - 1) / WAVE;
  const size_t FT = (size_t)F * T;
  const int g = blockIdx.x;
  const long long q0 = (long long)g * a.fp.L;
  const long long q1 = (q0 + a.fp.L < a.fp.NB) ? q0 + a.fp.L : a.fp.NB;
  if (q0 >= q1) return;
  const long long bt_first = q0 / F;
  const long long bt_last = (q1 - 1) / F;
  const int nchunks = K4 ? 1 : (K + KU - 1) / KU;

  for (int c = 0; c < nchunks; ++c) {
    const int k0 = c * KU;
    for (long long bt = bt_first; bt <= bt_last; ++bt) {  // segments of the range, one frame block each
      const int b = (int)(bt / TBk), tb = (int)(bt - (long long)b * TBk);
      const int fa = (int)((q0 > bt * F ? q0 : bt * F) - bt * F);
      const int fb = (int)((q1 < (bt + 1) * F ? q1 : (bt + 1) * F) - bt * F);
      const int t = tb * WAVE + lane;
      const bool valid = t < T;
      const unsigned tc = (unsigned)(valid ? t : T - 1);
      const Cx<R>* xb = X + (size_t)b * M * FT;
      const R* vb = V + (size_t)b * N * K * T;
      R v[N][KU];
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) {
          const int k = k0 + kk;
          v[n][kk] = vb[(unsigned)(n * K + (k < K ? k : K - 1)) * (unsigned)T + tc];
        }
      R acc[NACC];
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = 0;

      // this wave's bins: fa + h, fa + h + ACT_NH, ...
      const int nmine = (fb - fa - h + ACT_NH - 1) / ACT_NH;
      auto issue_x = [&](int f, Vec2<R>(&x)[M]) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const Cx<R>* row = xb + (size_t)((unsigned)(m * F + f) * (unsigned)T);  // uniform base + 32-bit lane offset
          x[m] = ldv_so<R>(row, tc * (unsigned)sizeof(Cx<R>));
        }
      };
      Vec2<R> xq[DXT][M];
#pragma unroll
      for (int j = 0; j < DXT; ++j)
        if (j < nmine) issue_x(fa + h + j * ACT_NH, xq[j]);

      for (int it0 = 0; it0 < nmine; it0 += DXT) {
#pragma unroll
        for (int j = 0; j < DXT; ++j) {
          const int it = it0 + j;
          if (it < nmine) {
            const int f = fa + h + it * ACT_NH;
            Cx<R> x[M];
#pragma unroll
            for (int m = 0; m < M; ++m) x[m] = tocx<R>(xq[j][m]);
            if (it + DXT < nmine) issue_x(f + DXT * ACT_NH, xq[j]);
            const Cx<R>* wp = W + ((size_t)b * F + f) * (N * M);  // wave-uniform: scalar loads
#pragma unroll
            for (int n = 0; n < N; ++n) {
              Cx<R> y = cmake<R>(0, 0);
#pragma unroll
              for (int m = 0; m < M; ++m) cfma(y, wp[n * M + m], x[m]);
              R P = cabs2(y);
              const R* tbn = Tb + (((size_t)b * N + n) * F + f) * K;
              R tk[KU];
              R tv = 0;
              if (K4) {
#pragma unroll
                for (int kk = 0; kk < KU; ++kk) {
                  tk[kk] = (kk < K) ? tbn[kk] : (R)0;
                  tv = fma(tk[kk], v[n][kk], tv);
                }
              } else {
                const R* vn = vb + (size_t)n * K * T + tc;
                for (int k = 0; k < K; ++k) tv = fma(tbn[k], vn[(size_t)k * T], tv);
#pragma unroll
                for (int kk = 0; kk < KU; ++kk) tk[kk] = (k0 + kk < K) ? tbn[k0 + kk] : (R)0;
              }
              tv = floor_eps<R>(tv, a.eps);
              if (TD) P = t_harmonic<R>(P, tv, a.nu);
              const R inv = fast_rcp(tv);
              const R D = D2 ? P * inv * inv : P / powspec<R>(tv, a.p1);
#pragma unroll
              for (int kk = 0; kk < KU; ++kk) {
                acc[(n * KU + kk) * 2 + 0] = fma(tk[kk], D, acc[(n * KU + kk) * 2 + 0]);
                acc[(n * KU + kk) * 2 + 1] = fma(tk[kk], inv, acc[(n * KU + kk) * 2 + 1]);
              }
            }
          }
        }
      }
      // combine the ACT_NH bin streams through LDS, then one coalesced partial record per (n, k, num|den)
      if (h > 0) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) lds[((h - 1) * NACC + i) * WAVE + lane] = acc[i];
      }
      __syncthreads();
      if (h == 0) {
        const int slot = (int)(bt - bt_first);
        R* out = part + ((size_t)g * a.fp.S + slot) * (size_t)(N * 2 * K) * WAVE + lane;
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          R tot = acc[i];
#pragma unroll
          for (int hh = 1; hh < ACT_NH; ++hh) tot += lds[((hh - 1) * NACC + i) * WAVE + lane];
          const int n = i / (2 * KU), k = k0 + (i >> 1) % KU;
          if (k < K) out[(size_t)(n * 2 * K + k * 2 + (i & 1)) * WAVE] = tot;
        }
      }
      __syncthreads();
    }
  }
}

// V *= (num / max(den, eps)) ** (d/(d+2))      (ilrma.py:426-428)
template <typename R>
__global__ void __launch_bounds__(256) act_stream_finalize_kernel(const R* __restrict__ part, R* __restrict__ V, int B,
                                                                 int N, int F, int K, int T, FlatPart fp, R eps,
                                                                 PowSpec p2, unsigned src_mask) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * K * T;
  if (idx >= total) return;
  const int t = idx % T;
  const int k = (idx / T) % K;
  const int n = (idx / ((size_t)T * K)) % N;
  if (!((src_mask >> n) & 1u)) return;
  const int b = idx / ((size_t)T * K * N);
  const int TBk = (T + WAVE - 1) / WAVE;
  const long long j = (long long)b * TBk + t / WAVE;
  const int lane = t % WAVE;
  const int g_lo = (int)((j * fp.len) / fp.L), g_hi = (int)(((j + 1) * fp.len - 1) / fp.L);
  R num = 0, den = 0;
  for (int g = g_lo; g <= g_hi; ++g) {
    const int slot = (int)(j - ((long long)g * fp.L) / fp.len);
    const R* p = part + ((((size_t)g * fp.S + slot) * N + n) * (size_t)(2 * K) + k * 2) * WAVE + lane;
    num += p[0];
    den += p[WAVE];
  }
  den = floor_eps<R>(den, eps);
  V[idx] = V[idx] * powspec<R>(num / den, p2);
}

}  // namespace assx
