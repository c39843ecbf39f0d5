// NMF half-updates on the matrix cores (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32), n_basis <= 64.
//
// One half-update is three chained products per 16x16 sub-tile of the spectrogram, all register to register:
//   (1) TV   = Tb V                       MFMA, K = n_basis          (or its transpose, see below)
//   (2) A    = X * g(TV),  Bm = h(TV)     elementwise, in the accumulator layout
//   (3) num += A . V^T,   den += Bm . V^T  (basis)      |  num += Tb^T . A, den += Tb^T . Bm  (activation)
// The trick that avoids any LDS transpose: MFMA's accumulator register r of lane l holds row crow(r,l) and column
// l&15, and the NEXT product reduces over exactly the rows (basis: frames, via the transposed product TV^T;
// activation: bins) -- a reduction may take its terms in any order, so accumulator register r is fed straight back
// as the k-slice r of the A (basis) / B (activation) operand, with the other operand loaded in the matching order.
// Operand layout (probed on gfx950, tools/probes/mfma_f64_probe.hip): A lane l -> A[l&15][l>>4], B lane l ->
// B[l>>4][l&15]; C/D f64: row (l>>4)+4r, f32: row 4(l>>4)+r; column l&15.
// All global reads are 32- or 128-byte row segments whose cache lines are fully consumed by the same wave.
#pragma once
#include "assx_common.hpp"

#ifndef NMF_SKIP
#define NMF_SKIP 0  // timing experiments only (tools/probes/nmf_parts.sh; results are wrong by construction): 1 product (1),
                    // 2 element terms, 4 product (3), 8 X loads, 16 operand-tile loads and staging
#endif

namespace assx {

typedef double v4d_t __attribute__((ext_vector_type(4)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

template <typename R>
struct Mfma16;
template <>
struct Mfma16<double> {
  using acc_t = v4d_t;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int r, int lane) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma16<float> {
  using acc_t = v4f_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int r, int lane) { return 4 * (lane >> 4) + r; }
};

struct TermSpec {
  int kind;
  PowSpec pa, pb;  // exponents applied to TV for the numerator / denominator weights
  double p0;       // the kind's own parameter (tNMF: nu)
};

template <typename R>
__device__ __forceinline__ R pow0(R x, PowSpec p) {  // x**0 == 1 exactly as numpy (domain=2 EUC/KL numerators)
  return (p.mode == POW_GENERIC && p.e == 0.0) ? (R)1 : powspec<R>(x, p);
}

// numerator / denominator weights of one element (only TV is floored here: nmf.py:312-316)
// D2K: domain == 2 specialisation for the kind given (ASSX_NMF_EUC / _KL / _IS_MM (= both IS rules); -1 = generic, exponents at run time).
// With domain 2 every exponent is 0, 1 or 2; keeping pow() out of the kernel takes it from 220 to ~130 VGPRs,
// i.e. from 2 to 3-4 waves per SIMD to overlap one wave's elementwise / LDS phase with another's MFMA chain.
template <typename R, int D2K = -1>
__device__ __forceinline__ void nmf_terms(const TermSpec& s, R x, R tv, R eps, R& a, R& bm) {
  if (D2K < 0 && s.kind >= ASSX_NMF_T) {  // tNMF / CauchyNMF (domain 2), floors exactly where the reference has them
    if (s.kind == ASSX_NMF_CAUCHY_MM_FAST) {  // nmf.py:577-588: T V is NOT floored
      const R c = fma(tv, tv, (R)2 * x);
      a = x / floor_eps<R>(c * tv, eps);
      bm = tv / floor_eps<R>(c, eps);
      return;
    }
    if (s.kind == ASSX_NMF_CAUCHY_ME) {  // nmf.py:548-556: T V is NOT floored; "num" carries B, "den" carries A
      a = (R)1 / tv;
      bm = (R)0.75 * (tv / floor_eps<R>(fma(tv, tv, x), eps));
      return;
    }
    tv = floor_eps<R>(tv, eps);
    if (s.kind == ASSX_NMF_T || s.kind == ASSX_NMF_T_RAW) {  // nmf.py:410-417; T_RAW: ilrma.py:903-906 (P as is)
      const R nu = (R)s.p0, z = (s.kind == ASSX_NMF_T && x < eps) ? eps : x;
      const R harmonic = (R)1 / ((R)2 / (((R)2 + nu) * tv) + nu / (((R)2 + nu) * z));
      a = harmonic / (tv * tv);
      bm = (R)1 / tv;
    } else {  // CAUCHY_NAIVE / CAUCHY_MM, nmf.py:478-486
      a = (R)1 / tv;
      bm = (R)3 * (tv / floor_eps<R>(fma(tv, tv, (R)2 * x), eps));
    }
    return;
  }
  tv = floor_eps<R>(tv, eps);
  if (D2K == ASSX_NMF_EUC) {  // X * TV^0 ; TV^1
    a = x;
    bm = tv;
    return;
  }
  if (D2K == ASSX_NMF_KL) {  // X / TV ; TV^0
    a = x * fast_rcp(tv);
    bm = (R)1;
    return;
  }
  if (D2K == ASSX_NMF_IS_MM) {  // X / TV^2 ; 1 / TV   (v_rcp + Newton: 5 instructions instead of the ~25 of an IEEE divide)
    bm = fast_rcp(tv);
    a = x * bm * bm;
    return;
  }
  if (s.kind == ASSX_NMF_EUC) {  // X * TV^((2-d)/d) ; TV^((4-d)/d)
    a = x * pow0<R>(tv, s.pa);
    bm = pow0<R>(tv, s.pb);
  } else if (s.kind == ASSX_NMF_KL) {  // X / TV ; TV^((2-d)/d)
    a = x / tv;
    bm = pow0<R>(tv, s.pb);
  } else {  // IS: X / TV^((d+2)/d) ; 1 / TV
    bm = (R)1 / tv;
    a = (s.pa.mode == POW_SQUARE) ? x * bm * bm : x / pow0<R>(tv, s.pa);
  }
}

// ---------------------------------------------------------------------------------------------------------
// basis half: num|den (F,K) = [A|Bm] (F,T) . V^T, reduced over this workgroup's frame range.
//   grid (ceil(F/64), TS, B), 4 waves, wave w owns bins f0 = (4*blockIdx.x + w)*16 .. +15.
//   part[ts][b*2 + s][f*K + k]
// NS sub-tiles (of 16 frames) may advance TOGETHER through the three products (sub-tile s accumulates into its own
// num/den set, the sets are added in ascending s at the end).  Measured in round 3 and NOT used (NS = 1 everywhere): the
// parts of a trip add up -- tools/probes/nmf_parts.sh: compiling out any one part saves only its own share,
// profiles/r03_nmf_parts.txt -- and a dependent v_mfma_f64_16x16x4 issues ~184 cycles after its predecessor while the pipe
// takes one every 64-80, so twin chains looked like the cure; but NS = 2 costs 288 registers at n_basis 32 (176 at
// n_basis <= 16): one wave per SIMD, one workgroup per CU, i.e. a second round of workgroups -- config 2 took 123 us
// instead of 66, the n_basis 10 ILRMA source update 232 us instead of 200.
// ---------------------------------------------------------------------------------------------------------
template <typename R, int KT, int D2K = -1, int NS = 1>
__global__ void __launch_bounds__(256)
    nmf_basis_mfma_kernel(const R* __restrict__ X, const R* __restrict__ Tb, const R* __restrict__ V,
                          R* __restrict__ part, int B, int F, int T, int K, int tchunk, R eps, TermSpec s) {
  using MM = Mfma16<R>;
  using acc_t = typename MM::acc_t;
  constexpr int KS = KT * 4;      // k-slices of 4 in product (1)
  constexpr int KP = KT * 16;     // n_basis padded to the tile
  constexpr int LD = 17;          // padded row of the staged tile (bank-conflict-free column reads)
  constexpr int NLD = KP * 16 / 64;  // staged elements per lane and sub-tile
  // The V tile (KP x 16 frames) is read in two layouts -- as A operand of (1) and as B operand of (3).  Each wave
  // stages it once through its private LDS slice with coalesced 128-byte row reads instead of issuing 16 narrow
  // global loads per sub-tile (the CU's single L1 pipe made the first version load-issue bound).
  __shared__ R vt[4][NS][KP][LD];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int b = blockIdx.z, ts = blockIdx.y;
  const int f0 = (blockIdx.x * 4 + wv) * 16;
  if (f0 >= F) return;  // no workgroup barriers in this kernel (LDS slices are wave-private)
  const bool fvalid = f0 + li < F;
  const int f = fvalid ? f0 + li : F - 1;
  const R* xrow = X + ((size_t)b * F + f) * T;
  const R* vb = V + (size_t)b * K * T;

  R tb[KS];  // B operand of product (1): Tb^T[k = 4j + lk][f]
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    const int k = 4 * j + lk;
    tb[j] = (k < K) ? Tb[((size_t)b * F + f) * K + k] : (R)0;
  }
  acc_t num[NS][KT], den[NS][KT];
#pragma unroll
  for (int q = 0; q < NS; ++q)
#pragma unroll
    for (int c = 0; c < KT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        num[q][c][r] = 0;
        den[q][c][r] = 0;
      }

  const int ta = ts * tchunk;
  const int te = min(T, ta + tchunk);
  // staged element e = i*64 + lane  ->  row k = e / 16, frame t0 + e % 16 (16 lanes cover one 128-byte row segment)
  R stage[NS][NLD];
  R xq[NS][4];  // X of the next sub-tiles, in the accumulator layout (HBM latency hidden behind the current ones)
  auto fetch = [&](int t0) {
#pragma unroll
    for (int q = 0; q < NS; ++q) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int e = i * 64 + lane;
        const int k = e >> 4, tt = min(t0 + 16 * q + (e & 15), T - 1);
        stage[q][i] = (NMF_SKIP & 16) ? (R)0.5 : ((k < K) ? vb[(size_t)k * T + tt] : (R)0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        xq[q][r] = (NMF_SKIP & 8) ? (R)1 : xrow[min(t0 + 16 * q + MM::crow(r, lane), T - 1)];
    }
  };
  fetch(ta);
  for (int t0 = ta; t0 < te; t0 += 16 * NS) {
    R xc[NS][4];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) xc[q][r] = xq[q][r];
    if (!(NMF_SKIP & 16)) {
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          const int e = i * 64 + lane;
          vt[wv][q][e >> 4][e & 15] = stage[q][i];
        }
    }
    if (t0 + 16 * NS < te) fetch(t0 + 16 * NS);  // the next tiles travel while these are consumed
    __builtin_amdgcn_wave_barrier();
    // (1) TV^T sub-tiles: rows = frames t0 + 16 q + crow, columns = bins f0 + li; the NS chains alternate
    acc_t tv[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[q][r] = 0;
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int q = 0; q < NS; ++q)
        if (4 * j < K && !(NMF_SKIP & 1))
          tv[q] = MM::mma(vt[wv][q][4 * j + lk][li], tb[j], tv[q]);  // A operand: V^T[t = li][k]; all-padding k-slices skipped
    if (NMF_SKIP & 1)
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) tv[q][r] = xc[q][r] + (R)1;
    // (2) elementwise in the accumulator layout
    R a[NS][4], bm[NS][4];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = t0 + 16 * q + MM::crow(r, lane);
        if (NMF_SKIP & 2) {
          a[q][r] = xc[q][r];
          bm[q][r] = tv[q][r];
        } else
          nmf_terms<R, D2K>(s, xc[q][r], tv[q][r], eps, a[q][r], bm[q][r]);
        if (!(fvalid && t < te)) {
          a[q][r] = 0;
          bm[q][r] = 0;
        }
      }
    // (3) num[f, kb] += sum_t a[f,t] V[kb,t]: accumulator register r is k-slice r of the A operand
#pragma unroll
    for (int c = 0; c < KT; ++c) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          if (NMF_SKIP & 4) {
            num[q][c][r] += a[q][r];
            den[q][c][r] += bm[q][r];
            continue;
          }
          const R vv = vt[wv][q][16 * c + li][MM::crow(r, lane)];  // B operand: V^T[t-pos of slice r][kb]; rows >= K are 0
          num[q][c] = MM::mma(a[q][r], vv, num[q][c]);
          den[q][c] = MM::mma(bm[q][r], vv, den[q][c]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // D[row = f0 + crow][col = kb]; the NS sets in ascending order
  const size_t FK = (size_t)F * K;
  R* pn = part + ((size_t)ts * B * 2 + (size_t)b * 2) * FK;
#pragma unroll
  for (int c = 0; c < KT; ++c) {
    const int kb = 16 * c + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int fo = f0 + MM::crow(r, lane);
      R n = num[0][c][r], dd = den[0][c][r];
#pragma unroll
      for (int q = 1; q < NS; ++q) {
        n += num[q][c][r];
        dd += den[q][c][r];
      }
      if (fo < F && kb < K) {
        pn[(size_t)fo * K + kb] = n;
        pn[FK + (size_t)fo * K + kb] = dd;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// activation half: num|den (K,T) = Tb^T (K,F) . [A|Bm] (F,T), reduced over this workgroup's bin range.
//   grid (ceil(T/16), FS, B); the 4 waves stride over the bin range in sub-tiles of 16 and are combined through LDS.
//   part[fs][b*2 + s][k*T + t]
// NS sub-tiles of 16 bins (64 bins apart: a wave's consecutive sub-tiles) advance together, as in the basis half.
// ---------------------------------------------------------------------------------------------------------
template <typename R, int KT, int D2K = -1, int NS = 1>
__global__ void __launch_bounds__(256)
    nmf_act_mfma_kernel(const R* __restrict__ X, const R* __restrict__ Tb, const R* __restrict__ V, R* __restrict__ part,
                        int B, int F, int T, int K, int fchunk, R eps, TermSpec s) {
  using MM = Mfma16<R>;
  using acc_t = typename MM::acc_t;
  constexpr int KS = KT * 4;
  constexpr int KP = KT * 16;
  constexpr int LD = KP + 4;          // padded row of the staged 16 x KP basis tile
  constexpr int NLD = KP * 16 / 64;   // staged elements per lane and sub-tile
  __shared__ R tt_[4][NS][16][LD];    // wave-private basis tiles (read as A operand of (1) and of (3))
  __shared__ R red[3][KT * 2 * 4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int b = blockIdx.z, fs = blockIdx.y;
  const int t0 = blockIdx.x * 16;
  const bool tvalid = t0 + li < T;
  const int t = tvalid ? t0 + li : T - 1;
  const R* tbb = Tb + (size_t)b * F * K;
  const R* xb = X + (size_t)b * F * T;

  R vbr[KS];  // B operand of product (1): V[k = 4j + lk][t]
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    const int k = 4 * j + lk;
    vbr[j] = (k < K) ? V[((size_t)b * K + k) * T + t] : (R)0;
  }
  acc_t num[NS][KT], den[NS][KT];
#pragma unroll
  for (int q = 0; q < NS; ++q)
#pragma unroll
    for (int c = 0; c < KT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        num[q][c][r] = 0;
        den[q][c][r] = 0;
      }

  const int fa = fs * fchunk;
  const int fe = min(F, fa + fchunk);
  // staged element e = i*64 + lane -> bin f0 + e / KP, basis e % KP: consecutive lanes read consecutive k of a row
  R stage[NS][NLD];
  R xq[NS][4];  // X of the next sub-tiles, in the accumulator layout
  auto fetch = [&](int f0) {
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const int fq = f0 + 64 * q;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int e = i * 64 + lane;
        const int fr = fq + e / KP, k = e % KP;
        stage[q][i] = (NMF_SKIP & 16) ? (R)0.5 : ((k < K && fr < fe) ? tbb[(size_t)fr * K + k] : (R)0);  // bins beyond the range contribute 0
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        xq[q][r] = (NMF_SKIP & 8) ? (R)1 : xb[(size_t)min(fq + MM::crow(r, lane), F - 1) * T + t];
    }
  };
  int f0 = fa + 16 * wv;
  if (f0 < fe) fetch(f0);
  for (; f0 < fe; f0 += 64 * NS) {
    R xc[NS][4];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) xc[q][r] = xq[q][r];
    if (!(NMF_SKIP & 16)) {
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          const int e = i * 64 + lane;
          tt_[wv][q][e / KP][e % KP] = stage[q][i];
        }
    }
    if (f0 + 64 * NS < fe) fetch(f0 + 64 * NS);
    __builtin_amdgcn_wave_barrier();
    // (1) TV sub-tiles: rows = bins f0 + 64 q + crow, columns = frames t0 + li; the NS chains alternate
    acc_t tv[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[q][r] = 0;
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int q = 0; q < NS; ++q)
        if (4 * j < K && !(NMF_SKIP & 1))
          tv[q] = MM::mma(tt_[wv][q][li][4 * j + lk], vbr[j], tv[q]);  // A operand: Tb[f = li][k]; padding slices skipped
    if (NMF_SKIP & 1)
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) tv[q][r] = xc[q][r] + (R)1;
    // (2)
    R a[NS][4], bm[NS][4];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int fr = f0 + 64 * q + MM::crow(r, lane);
        if (NMF_SKIP & 2) {
          a[q][r] = xc[q][r];
          bm[q][r] = tv[q][r];
        } else
          nmf_terms<R, D2K>(s, xc[q][r], tv[q][r], eps, a[q][r], bm[q][r]);
        if (!(tvalid && fr < fe)) {
          a[q][r] = 0;
          bm[q][r] = 0;
        }
      }
    // (3) num[kb, t] += sum_f Tb[f,kb] a[f,t]: accumulator register r is k-slice r of the B operand
#pragma unroll
    for (int c = 0; c < KT; ++c) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          if (NMF_SKIP & 4) {
            num[q][c][r] += a[q][r];
            den[q][c][r] += bm[q][r];
            continue;
          }
          const R tv3 = tt_[wv][q][MM::crow(r, lane)][16 * c + li];  // A operand: Tb^T[kb][f-pos of slice r]
          num[q][c] = MM::mma(tv3, a[q][r], num[q][c]);
          den[q][c] = MM::mma(tv3, bm[q][r], den[q][c]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the NS sets in ascending order, then the 4 waves
#pragma unroll
  for (int c = 0; c < KT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 1; q < NS; ++q) {
        num[0][c][r] += num[q][c][r];
        den[0][c][r] += den[q][c][r];
      }
  if (wv > 0) {
#pragma unroll
    for (int c = 0; c < KT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[wv - 1][(c * 2 + 0) * 4 + r][lane] = num[0][c][r];
        red[wv - 1][(c * 2 + 1) * 4 + r][lane] = den[0][c][r];
      }
  }
  __syncthreads();
  if (wv == 0 && tvalid) {
    const size_t KTt = (size_t)K * T;
    R* pn = part + ((size_t)fs * B * 2 + (size_t)b * 2) * KTt;
#pragma unroll
    for (int c = 0; c < KT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        R n = num[0][c][r], d = den[0][c][r];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          n += red[w][(c * 2 + 0) * 4 + r][lane];
          d += red[w][(c * 2 + 1) * 4 + r][lane];
        }
        const int kb = 16 * c + MM::crow(r, lane);  // D[row = kb][col = t]
        if (kb < K) {
          pn[(size_t)kb * T + t] = n;
          pn[KTt + (size_t)kb * T + t] = d;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// one element of criterion(in = (Tb V)^(2/domain), x)  (nmf.py:170-174, 229-233, 288-292; divergence.py:21-45;
// t_divergence nmf.py:369-373, cauchy_divergence nmf.py:435-443)
__device__ __forceinline__ double nmf_criterion(int kind, double in, double x, double eps, double p0) {
  if (kind == ASSX_NMF_EUC) return (x - in) * (x - in);
  const double _in = in + eps, _tg = x + eps;  // divergence.py:26-27, 39-40
  const double ratio = _tg / _in;
  if (kind == ASSX_NMF_KL) return _tg * log(ratio) + _in - _tg;
  if (kind == ASSX_NMF_T) return log(_in) + 0.5 * (2.0 + p0) * log(1.0 + (2.0 / p0) * ratio);
  if (kind >= ASSX_NMF_CAUCHY_NAIVE) return log(ratio) + 1.5 * log((2.0 * _tg * _tg + _in * _in) / (3.0 * _tg * _tg));
  return ratio - log(ratio) - 1.0;
}

// loss: sum_{f,t} criterion((Tb V)^(2/domain), X)
//   Same tiling as the activation kernel (TV sub-tiles by MFMA, elementwise in the accumulator layout);
//   one float64 partial per workgroup: lpart[b][blockIdx.y * gridDim.x + blockIdx.x]
// ---------------------------------------------------------------------------------------------------------
template <typename R, int KT>
__global__ void __launch_bounds__(256)
    nmf_loss_mfma_kernel(const R* __restrict__ X, const R* __restrict__ Tb, const R* __restrict__ V,
                         double* __restrict__ lpart, int F, int T, int K, int fchunk, int kind, double eps,
                         PowSpec p2d, double p0) {
  using MM = Mfma16<R>;
  using acc_t = typename MM::acc_t;
  constexpr int KS = KT * 4;
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int b = blockIdx.z, fs = blockIdx.y;
  const int t0 = blockIdx.x * 16;
  const bool tvalid = t0 + li < T;
  const int t = tvalid ? t0 + li : T - 1;
  const R* tbb = Tb + (size_t)b * F * K;
  const R* xb = X + (size_t)b * F * T;
  R vbr[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    const int k = 4 * j + lk;
    vbr[j] = (k < K) ? V[((size_t)b * K + k) * T + t] : (R)0;
  }
  const int fa = fs * fchunk;
  const int fe = min(F, fa + fchunk);
  double acc = 0.0;
  for (int f0 = fa + 16 * wv; f0 < fe; f0 += 64) {
    R xc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xc[r] = xb[(size_t)min(f0 + MM::crow(r, lane), F - 1) * T + t];
    acc_t tv;
#pragma unroll
    for (int r = 0; r < 4; ++r) tv[r] = 0;
    const int fA = min(f0 + li, F - 1);
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      if (4 * j >= K) break;  // all-padding k-slice
      const int k = 4 * j + lk;
      const R ta = (k < K) ? tbb[(size_t)fA * K + k] : (R)0;
      tv = MM::mma(ta, vbr[j], tv);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int fr = f0 + MM::crow(r, lane);
      if (tvalid && fr < fe) {
        const double in = (double)powspec<R>(tv[r], p2d);  // (T V) ** (2 / domain), not floored
        acc += nmf_criterion(kind, in, (double)xc[r], eps, p0);
      }
    }
  }
  acc = wave_allreduce_sum<double>(acc);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    lpart[(size_t)b * gridDim.y * gridDim.x + (size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace assx
