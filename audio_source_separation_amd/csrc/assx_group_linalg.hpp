// Per-bin small linear algebra spread over a GROUP of lanes (one lane per matrix element): the IP, ISS and IP2
// sweeps of the spatial update.  Shared by the streaming-kernel path (assx_bss.hip, M <= 4, covariance arriving as
// partial records or dense) and the wide-channel path (assx_widem.hip, 5 <= M <= 8, dense covariance).
#pragma once
#include "assx_common.hpp"
#include "assx_small_linalg.hpp"
#include "assx_stream.hpp"

namespace assx {

// ------------------------------------------------------------------------------------------
// (a5) IP sweep.  A group of GW = next_pow2(M*M) lanes owns one bin (b, f): lane (i, j) holds element (i, j) of
//      W and of the working matrix, rows/columns travel by in-group shuffles.  Gauss-Seidel over the sources
//      (sequential), Gauss-Jordan with partial pivoting inside, all in float64.  The 1025 bins of one utterance
//      become 257 waves spread over the chip instead of 17 latency-bound single-lane waves (30 us -> a few us).
//      FROM_PART: the covariance arrives as the streaming kernel's partial records (cov_stream_kernel) and is
//      reduced here (saves the separate finalize launch); otherwise as dense U (B,N,F,M,M).
//      Optionally emits pw[b][n][f] = w_n^H C_f w_n (the per-bin share of the power normalisation statistic).
// ------------------------------------------------------------------------------------------
// ---- data movement inside a lane group without the LDS pipe (round 4).  __shfl is ds_bpermute_b32: every use is a round
//      trip through the LDS unit behind an s_waitcnt, and the per-bin sweeps are single dependent chains on an otherwise
//      idle SIMD -- their time is the SUM of such latencies.  The patterns that are fixed at compile time and stay inside a
//      DPP row (16 lanes) move with DPP instead: row_newbcast (one lane of every row to the whole row), quad_perm (a lane
//      of every quad to the whole quad, the xor-1 / xor-2 exchanges) and row_ror.  Pure data movement: no result changes.
//      ASSX_GROUP_DPP=0 (A/B builds, tools/probes/ip_dpp_probe.hip) keeps the ds_bpermute forms.
#ifndef ASSX_GROUP_DPP
#define ASSX_GROUP_DPP 1
#endif
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_ROR = 0x120, DPP_ROW_NEWBCAST = 0x150;
constexpr int dpp_quad_bcast(int k) { return k * 0x55; }
template <int CTRL, int BANKS = 0xF>
__device__ __forceinline__ double dpp_mov(double v, double old = 0.0) {  // lanes of the banks (quads) not in BANKS keep `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xF, BANKS, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xF, BANKS, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {  // l wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// butterfly sum over the group, every lane ends with the total.  The pairing is that of v += shfl_xor(v, off), off = GW/2
// .. 1, in both forms: after the steps with off >= 16 the values repeat with period 16, so lane e ^ 8 holds what the lane
// 8 further round its row holds, and after off = 8 (period 8) lane e ^ 4 what the lane 4 further round holds.
template <int GW>
__device__ __forceinline__ double group_sum(double v) {
  static_assert(GW == 1 || GW == 4 || GW >= 16, "a group is a quad or whole DPP rows");
#if ASSX_GROUP_DPP
#pragma unroll
  for (int off = GW / 2; off >= 16; off >>= 1) v += __shfl_xor(v, off, GW);
  if (GW >= 16) v += dpp_mov<DPP_ROW_ROR + 8>(v);
  if (GW >= 8) v += dpp_mov<DPP_ROW_ROR + 4>(v);
  if (GW >= 4) v += dpp_mov<DPP_QUAD_XOR2>(v);
  if (GW >= 2) v += dpp_mov<DPP_QUAD_XOR1>(v);
#else
#pragma unroll
  for (int off = GW / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, GW);
#endif
  return v;
}
template <int GW>
__device__ __forceinline__ Cd group_shfl(Cd v, int src) {
  return cmake<double>(__shfl(v.x, src, GW), __shfl(v.y, src, GW));
}
// lane L (compile time) of every group, to the whole group
template <int GW, int L>
__device__ __forceinline__ double group_bcast(double v) {
#if ASSX_GROUP_DPP
  if constexpr (GW == 64) return readlane_f64(v, L);
  else if constexpr (GW == 16) return dpp_mov<DPP_ROW_NEWBCAST + L>(v);
  else if constexpr (GW == 4) return dpp_mov<dpp_quad_bcast(L)>(v);
  else
#endif
    return __shfl(v, L, GW);
}
template <int GW, int L>
__device__ __forceinline__ Cd group_bcast(Cd v) {
  return cmake<double>(group_bcast<GW, L>(v.x), group_bcast<GW, L>(v.y));
}
// element (i, C) of an M x M matrix spread one element per lane (lane i * M + j of the group), to lane (i, j): column C
// (compile time) along the rows
template <int M, int GW, int C>
__device__ __forceinline__ double group_row_bcast(double v, int i) {
#if ASSX_GROUP_DPP
  if constexpr (M == 4) return dpp_mov<dpp_quad_bcast(C)>(v);  // a matrix row is a quad
  else if constexpr (M == 2) return dpp_mov<(C) | (C << 2) | ((2 + C) << 4) | ((2 + C) << 6)>(v);  // half a quad
  else if constexpr (M == 8) {  // a matrix row is half a DPP row: banks 0-1 and 2-3
    const double lo = dpp_mov<DPP_ROW_NEWBCAST + C, 0x3>(v, v);
    return dpp_mov<DPP_ROW_NEWBCAST + 8 + C, 0xC>(v, lo);
  } else
#endif
    return __shfl(v, i * M + C, GW);
}
template <int M, int GW, int C>
__device__ __forceinline__ Cd group_row_bcast(Cd v, int i) {
  return cmake<double>(group_row_bcast<M, GW, C>(v.x, i), group_row_bcast<M, GW, C>(v.y, i));
}

// the same with a wave-uniform column known at run time only (the source index of a sweep): M uniform branches
template <int M, int GW>
__device__ __forceinline__ Cd group_row_bcast_rt(Cd v, int i, int col) {
#if ASSX_GROUP_DPP
  if constexpr (M == 2 || M == 4 || M == 8) {
    Cd r = v;
    static_for<M>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if (col == k) r = group_row_bcast<M, GW, k>(v, i);
    });
    return r;
  } else
#endif
    return group_shfl<GW>(v, i * M + col);
}

// In-group Gauss-Jordan inverse with partial pivoting (LAPACK's pivot rule: first max of |re|+|im|) of the M x M
// matrix whose element (i, j) lives in lane (i, j) of a GW-lane group.  Returns this lane's element of the inverse.
template <int M, int GW>
__device__ __forceinline__ Cd group_gj_inverse(Cd a, int i, int j, bool& singular) {
  int piv[M];
  static_for<M>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    int p = c;
    double best = -1.0;
    const double m1own = cabs1(a);  // the metric travels (one real), not the element
    static_for<M - c>([&](auto rc) {
      constexpr int r = c + decltype(rc)::value;
      const double m1 = group_bcast<GW, r * M + c>(m1own);
      if (m1 > best) {
        best = m1;
        p = r;
      }
    });
    if (!(best > 0.0)) singular = true;
    piv[c] = p;
    const int src_i = (i == c) ? p : ((i == p) ? c : i);
    a = group_shfl<GW>(a, src_i * M + j);  // row interchange c <-> p
    const Cd pv = group_bcast<GW, c * M + c>(a);
    const Cd ipv = crcp_fast(pv);
    // (fetching the pivot row in the round trip of the interchange -- row p before it -- was measured: no faster, and the
    // moved loads made the compiler contract the products below differently: last-bit changes in float64.  Not kept.)
    const Cd acj = group_shfl<GW>(a, c * M + j);
    const Cd rcj = cmul((j == c) ? cmake<double>(1.0, 0.0) : acj, ipv);
    const Cd fic = group_row_bcast<M, GW, c>(a, i);
    if (i == c) {
      a = rcj;
    } else {
      const Cd base = (j == c) ? cmake<double>(0.0, 0.0) : a;
      a = cmake<double>(base.x - (fic.x * rcj.x - fic.y * rcj.y), base.y - (fic.x * rcj.y + fic.y * rcj.x));
    }
  });
  // undo the row interchanges as column interchanges (c = M-1 .. 0): composed on the column index first (integer
  // selects), then ONE shuffle instead of M dependent ones
  int jj = j;
#pragma unroll
  for (int c = 0; c < M; ++c) {
    const int p = piv[c];
    jj = (jj == c) ? p : ((jj == p) ? c : jj);
  }
  return group_shfl<GW>(a, i * M + jj);
}

// In-group ADJUGATE (round 6), 2 <= M <= 4: adj[i][j] = cofactor(j, i) -- every lane its own cofactor from the elements it
// fetches in ONE round of in-group shuffles (all independent) -- and det = sum_i a[j][i] adj[i][j] by two rotations of the
// group.  A^-1 = adj / det.  The Gauss-Jordan form above is M pivot steps of ~28 DEPENDENT float64 operations and two shuffle
// round trips each, a complex reciprocal in every one of them; a per-bin sweep is a single dependent chain on an otherwise
// idle SIMD (~32 cycles per dependent instruction), so its time IS that depth.  The IP step needs only the DIRECTION of
// column n of the inverse (the new row is normalised by sqrt(w^H U w)): column n of the adjugate, ~8 operations and one round
// trip deep, no reciprocal at all.  No pivoting, hence no a-priori error bound: ip_group_kernel checks the backward error
// of that column a posteriori, A adj[:, n] = det e_n, and runs the pivoted elimination for the bins that fail
// (group_adj_column_ok).
// MEASURED in round 6 and NOT the default (profiles/r06_ip_adjugate_ab.txt): the sweep kernel is not only a dependent chain,
// it is also ~1700 instructions issued by ONE wave per SIMD at ~8 cycles each -- the column check, the guard's inverse
// adj / det and the phase of det that the normalised row needs (conj(x) / sqrt(x^H U x) keeps the phase of a complex scale)
// add back most of what the elimination's four reciprocal chains cost.  Headline, alternating libraries on one box:
// 5576-5653 it/s with the elimination, 5613-5649 with this form (float64); float32 models 8819-9020 -> 8412-8438: their
// W U_n fail the column check in many bins, and a wave with one such bin runs BOTH forms -- the kernel ends with its
// slowest wave.  (On the headline input kappa_F(W U_n) grows to 1e8 ... 1e11 as the model converges -- one small singular
// value, the case the adjugate handles well -- tools/probes/ip_cond_hist.py.)  Kept for A/B builds: -DASSX_IP_ADJ=1.
#ifndef ASSX_IP_ADJ
#define ASSX_IP_ADJ 0
#endif
#if ASSX_IP_ADJ
template <int M, int GW>
__device__ __forceinline__ Cd group_adjugate(Cd a, int i, int j, bool active, Cd& det) {
  static_assert(M >= 2 && M <= 4, "closed-form adjugate: 2 <= M <= 4");
  Cd adj;
  if constexpr (M == 2) {
    // adj[i][j] = (-1)^(i+j) a[1-j][1-i]: lane (i, j) = 2 i + j reads lane 2 (1-j) + (1-i): quad_perm [3, 1, 2, 0]
    const Cd p = cmake<double>(dpp_mov<3 | (1 << 2) | (2 << 4) | (0 << 6)>(a.x), dpp_mov<3 | (1 << 2) | (2 << 4) | (0 << 6)>(a.y));
    adj = (i != j) ? cmake<double>(-p.x, -p.y) : p;
  } else if constexpr (M == 3) {
    // cyclic indices carry the sign: adj[i][j] = a[j+1][i+1] a[j+2][i+2] - a[j+1][i+2] a[j+2][i+1]  (indices mod 3)
    const int r1 = (j + 1) % 3, r2 = (j + 2) % 3, c1 = (i + 1) % 3, c2 = (i + 2) % 3;
    const Cd e11 = group_shfl<GW>(a, r1 * 3 + c1), e22 = group_shfl<GW>(a, r2 * 3 + c2);
    const Cd e12 = group_shfl<GW>(a, r1 * 3 + c2), e21 = group_shfl<GW>(a, r2 * 3 + c1);
    const Cd p = cmul(e11, e22), q = cmul(e12, e21);
    adj = cmake<double>(p.x - q.x, p.y - q.y);
  } else {
    // the 3 x 3 minor without row j and column i, rows / columns in ascending order; sign (-1)^(i+j)
    int r[3], c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      r[k] = k + (k >= j ? 1 : 0);
      c[k] = k + (k >= i ? 1 : 0);
    }
    Cd e[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = 0; l < 3; ++l) e[k][l] = group_shfl<GW>(a, r[k] * 4 + c[l]);
    auto m2 = [&](int l0, int l1) {  // rows 1, 2; columns l0, l1
      const Cd p = cmul(e[1][l0], e[2][l1]), q = cmul(e[1][l1], e[2][l0]);
      return cmake<double>(p.x - q.x, p.y - q.y);
    };
    const Cd t0 = cmul(e[0][0], m2(1, 2)), t1 = cmul(e[0][1], m2(0, 2)), t2 = cmul(e[0][2], m2(0, 1));
    const Cd m = cmake<double>(t0.x - t1.x + t2.x, t0.y - t1.y + t2.y);
    adj = ((i + j) & 1) ? cmake<double>(-m.x, -m.y) : m;
  }
  // det = sum_i a[j][i] adj[i][j] (expansion along row j), over the lanes (i, j) of this lane's column j
  const Cd at = group_shfl<GW>(a, j * M + i);
  Cd t = cmul(at, adj);
  if (!active) t = cmake<double>(0.0, 0.0);
  if constexpr (M == 4) {  // lanes i * 4 + j, i = 0..3: the lanes 4, 8, 12 further round the 16-lane row
    t = cmake<double>(t.x + dpp_mov<DPP_ROW_ROR + 8>(t.x), t.y + dpp_mov<DPP_ROW_ROR + 8>(t.y));
    det = cmake<double>(t.x + dpp_mov<DPP_ROW_ROR + 4>(t.x), t.y + dpp_mov<DPP_ROW_ROR + 4>(t.y));
  } else if constexpr (M == 2) {  // lanes j and 2 + j of the quad
    det = cmake<double>(t.x + dpp_mov<DPP_QUAD_XOR2>(t.x), t.y + dpp_mov<DPP_QUAD_XOR2>(t.y));
  } else {
    const Cd s0 = group_shfl<GW>(t, j), s1 = group_shfl<GW>(t, 3 + j), s2 = group_shfl<GW>(t, 6 + j);
    det = cmake<double>(s0.x + s1.x + s2.x, s0.y + s1.y + s2.y);
  }
  det = group_bcast<GW, 0>(det);  // one value for the whole matrix (the M column sums differ in their last bits)
  return adj;
}
// sum over the lanes (i, 0..M-1) of this lane's matrix row
template <int M, int GW>
__device__ __forceinline__ double group_row_sum(double v, int i) {
  if constexpr (M == 4) {  // a row is a quad
    v += dpp_mov<DPP_QUAD_XOR1>(v);
    return v + dpp_mov<DPP_QUAD_XOR2>(v);
  } else if constexpr (M == 2) {  // a row is half a quad
    return v + dpp_mov<DPP_QUAD_XOR1>(v);
  } else {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) s += __shfl(v, i * M + k, GW);
    return s;
  }
}
// Is column n of the adjugate an acceptable solution of  A x = det e_n ?  wj = adj[j][n] at lane (i, j).  Normwise backward
// error  ||A x - det e_n||_2 <= 1e-14 ||A||_F ||x||_2  (the pivoted elimination reaches ~1e-16; a cofactor column that lost
// digits to cancellation -- several small singular values -- fails by orders of magnitude), every quantity a finite normal
// number.  NaN fails.
template <int M, int GW>
__device__ __forceinline__ bool group_adj_column_ok(Cd a, Cd wj, Cd det, int i, int n, bool active) {
  const Cd t = active ? cmul(a, wj) : cmake<double>(0.0, 0.0);
  Cd r = cmake<double>(group_row_sum<M, GW>(t.x, i), group_row_sum<M, GW>(t.y, i));  // (A x)[i], in every lane of row i
  if (i == n) r = cmake<double>(r.x - det.x, r.y - det.y);
  const double rr = group_sum<GW>(active ? cabs2(r) : 0.0);    // M ||r||^2
  const double xx = group_sum<GW>(active ? cabs2(wj) : 0.0);   // M ||x||^2
  const double aa = group_sum<GW>(active ? cabs2(a) : 0.0);    // ||A||_F^2
  const double bound = 1e-28 * aa * xx;
  return rr <= bound && bound > 1e-280 && bound < 1e280 && xx < 1e280 && aa < 1e280;
}
#endif  // ASSX_IP_ADJ

// Largest singular value of the M x M matrix whose element (i, j) lives in lane (i, j) of the group: lambda_max of
// the Gram matrix by repeated squaring with trace normalisation, every product made of in-group shuffles.  Same
// arithmetic as spectral_norm_slow (assx_small_linalg.hpp) but with NO per-lane arrays: the single-thread form keeps
// three M x M complex matrices in scratch, and although this is a rare path its 3.6 KB per lane are reserved for
// every wave of the kernel -- at 8 utterances per launch the scratch ceiling throttled the waves in flight and the IP
// sweep took 405 us instead of 67.
template <int M, int GW>
__device__ __forceinline__ double group_spectral_norm(Cd a, int i, int j, bool active) {
  Cd g = cmake<double>(0.0, 0.0);
#pragma unroll
  for (int k = 0; k < M; ++k) cfma(g, cconj(group_shfl<GW>(a, k * M + i)), group_shfl<GW>(a, k * M + j));  // (A^H A)[i][j]
  const double tr = group_sum<GW>((active && i == j) ? g.x : 0.0);
  if (!(tr > 0.0) || !isfinite(tr)) return tr > 0.0 ? tr : 0.0;
  g = cscale(g, 1.0 / tr);
  const Cd g0 = g;
  // G <- G^2 / tr(G^2): tr(G) = 1 throughout, and tr(G^2) -> 1 as G approaches the projector on the dominant eigenvector
  // (1 - tr(G^2) falls quadratically: (l2/l1)^(2^k)).  Every group of the wave has converged once none of them is more than
  // 1e-14 away from 1 (the Rayleigh quotient below is then within ~5e-15 relative of lambda_max: for the mixed state G_k its
  // error is linear in 1 - tr(G^2), ample for a guard that compares against a threshold) -- typically after 5-8 squarings;
  // the bound of 24 is what round 2 always ran (float32 models put many bins into the ambiguous band of the condition guard,
  // and 2 x 24 squarings per such wave doubled the sweep).  A group whose trace came out 0 (a wholly inactive group of a
  // tail wave) counts as converged and is not divided by: before, such a wave ran all 24 squarings on NaNs.
  for (int it = 0; it < 24; ++it) {
    Cd h = cmake<double>(0.0, 0.0);
    static_for<M>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      cfma(h, group_row_bcast<M, GW, k>(g, i), group_shfl<GW>(g, k * M + j));
    });
    const double t2 = group_sum<GW>((active && i == j) ? h.x : 0.0);
    const bool dead = t2 == 0.0;
    g = cscale(h, dead ? 0.0 : 1.0 / t2);
    if (!__any(!(dead || 1.0 - t2 < 1e-14))) break;  // wave-uniform (NaN keeps iterating to the bound, as before)
  }
  // Rayleigh quotient tr(G0 Gk) / tr(Gk), tr(Gk) = 1
  const Cd gt = group_shfl<GW>(g, j * M + i);
  const double lam = group_sum<GW>(active ? (g0.x * gt.x - g0.y * gt.y) : 0.0);
  return sqrt(lam * tr);
}

// cond_2(A) < thr decided for the groups inside the ambiguous band, from A (a0) and its inverse (ainv), element (i, j)
// per lane (round 4; replaces  ||A||_2 ||A^-1||_2 < thr  from two calls of group_spectral_norm, which is kept for
// degenerate traces and as the arithmetic the many-channel path restates).  Both Gram matrices are squared in ONE loop
// (two independent chains for a lone wave to interleave: the squarings are dependent-latency bound), and the loop ends
// when the answer is known instead of when the norms have converged:
//   G_0 = A^H A / tr,  G_{k+1} = G_k^2 s_k  (s_k ~ 1 / t_k, t_k = tr G_k^2; any s_k would do, see below),
//   mu_k = lambda_max(G_k):  mu_{k+1} = mu_k^2 s_k  and, since tr G_k = 1 and G_k >= 0,   t_k <= mu_k <= sqrt(t_k).
// With P_k = mu_k(A) mu_k(A^-1) and tau_k = t_k(A) t_k(A^-1):  cond_2^2 = tr tr' P_0 < thr^2  <=>  P_0 < theta_0 = thr^2 /
// (tr tr'), and P_{k+1} < theta_{k+1} = theta_k^2 s_k s'_k  <=>  P_k < theta_k  (the SAME factors scale both sides: the
// accuracy of the reciprocals does not enter).  theta_k / P_k is squared by every step, so it leaves the interval
// [tau_k, sqrt(tau_k)] / P_k, which shrinks to a point, after 2-4 steps unless cond_2 sits within rounding of thr:
//   tau_k < theta_k^2  =>  P_k <= sqrt(tau_k) < theta_k : below;      tau_k >= theta_k  =>  P_k >= theta_k : not below.
// theta_k stays in [1 / M^2, 1] until then (it starts there: tr tr' is the squared Frobenius product, and the band is
// thr^2 <= tr tr' < M^2 thr^2).  Undecided after 24 steps, or a trace that is not a positive normal number: the old form.
template <int M, int GW>
__device__ __forceinline__ bool group_cond_band(Cd a0, Cd ainv, int i, int j, bool active, bool amb, double thr) {
  Cd g1 = cmake<double>(0.0, 0.0), g2 = g1;
#pragma unroll
  for (int k = 0; k < M; ++k) {
    cfma(g1, cconj(group_shfl<GW>(a0, k * M + i)), group_shfl<GW>(a0, k * M + j));
    cfma(g2, cconj(group_shfl<GW>(ainv, k * M + i)), group_shfl<GW>(ainv, k * M + j));
  }
  const bool diag = active && i == j;
  const double tr1 = group_sum<GW>(diag ? g1.x : 0.0), tr2 = group_sum<GW>(diag ? g2.x : 0.0);
  const double thr2 = thr * thr, den = tr1 * tr2;
  const bool plain = tr1 > 1e-290 && tr1 < 1e290 && tr2 > 1e-290 && tr2 < 1e290 && den > 1e-290 && den < 1e290 && thr2 < 1e290;
  bool undecided = amb && plain, odd = amb && !plain, ok = false;
  double theta = thr2 * fast_rcp(plain ? den : 1.0);
  g1 = cscale(g1, fast_rcp(plain ? tr1 : 1.0));
  g2 = cscale(g2, fast_rcp(plain ? tr2 : 1.0));
  for (int it = 0; it < 24 && __any(undecided); ++it) {
    Cd h1 = cmake<double>(0.0, 0.0), h2 = h1;
    static_for<M>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      cfma(h1, group_row_bcast<M, GW, k>(g1, i), group_shfl<GW>(g1, k * M + j));
      cfma(h2, group_row_bcast<M, GW, k>(g2, i), group_shfl<GW>(g2, k * M + j));
    });
    const double t1 = group_sum<GW>(diag ? h1.x : 0.0), t2 = group_sum<GW>(diag ? h2.x : 0.0);
    const double tau = t1 * t2, th2 = theta * theta;
    if (undecided && tau < th2) {
      ok = true;
      undecided = false;
    } else if (undecided && tau >= theta) {
      undecided = false;
    }
    const bool live = t1 > 0.0 && t2 > 0.0;  // not a wholly inactive group of a tail wave (nor a NaN)
    const double s1 = live ? fast_rcp(t1) : 0.0, s2 = live ? fast_rcp(t2) : 0.0;
    g1 = cscale(h1, s1);
    g2 = cscale(h2, s2);
    theta = th2 * s1 * s2;
  }
  if (__any(undecided || odd)) {  // rounding-close to the threshold, or extreme magnitudes: the converged norms
    const double s = group_spectral_norm<M, GW>(a0, i, j, active) * group_spectral_norm<M, GW>(ainv, i, j, active);
    if (undecided || odd) ok = s < thr;
  }
  return ok;
}

// cond_2(A) < thr from A (a0) and its inverse (ainv), element (i, j) per lane: Frobenius bounds, exact spectral
// norms only inside the factor-M band (rare; whole groups take the slow path together).
template <int M, int GW>
__device__ __forceinline__ bool group_cond_below(Cd a0, Cd ainv, bool active, bool singular, double thr,
                                                 double* frob2 = nullptr /* out: ||A||_F^2 ||A^-1||_F^2 */) {
  const double nA2 = group_sum<GW>(active ? cabs2(a0) : 0.0);
  const double nI2 = group_sum<GW>(active ? cabs2(ainv) : 0.0);
  if (frob2) *frob2 = nA2 * nI2;
  // compared as squares: the two square roots would sit in the middle of the sweep's dependent chain
  // (outside the range where the product of the squared norms is a normal number: the square-root form)
  double c2 = nA2 * nI2, thr2 = thr * thr, m2 = (double)(M * M);
  if (!(c2 > 1e-290 && c2 < 1e290 && thr2 < 1e290)) {
    c2 = sqrt(nA2) * sqrt(nI2);
    thr2 = thr;
    m2 = (double)M;
  }
  const bool amb = !singular && (c2 == c2) && c2 >= thr2 && c2 < thr2 * m2;
  bool ok = !singular && (c2 == c2) && c2 < thr2;
  if (__any(amb)) {
    const int lane = threadIdx.x & (GW - 1);
    const int i = active ? lane / M : 0, j = active ? lane % M : 0;
    const bool below = group_cond_band<M, GW>(a0, ainv, i, j, active, amb, thr);
    if (amb) ok = below;
  }
  return ok;
}

template <typename R, int M, bool FROM_PART>
__global__ void __launch_bounds__(64)
    ip_group_kernel(const Cx<R>* __restrict__ U, const R* __restrict__ part, FlatPart fp, double inv_T,
                    Cx<R>* __restrict__ W, const Cx<R>* __restrict__ C, double* __restrict__ pw, double thr,
                    int32_t* __restrict__ status, int B, int F, double den_floor, int wb = 1) {
  // wb (FROM_PART): bins per record group -- 1 for cov_stream_kernel's records [g][slot][n][HM], COVW_BINS for
  // cov_wide_kernel's [g][slot][bin in group][n][HM] (same packed-Hermitian layout inside)
  constexpr int N = M;
  constexpr int MM = M * M;
  constexpr int GW = next_pow2_c(MM);
  constexpr int GPW = WAVE / GW;  // groups per wave
  const int lane = threadIdx.x & (WAVE - 1);
  const int e = lane & (GW - 1);
  const long long grp = (long long)blockIdx.x * GPW + lane / GW;
  const bool in_range = grp < (long long)B * F;
  const long long bf = in_range ? grp : (long long)B * F - 1;  // out-of-range groups shadow the last bin, store nothing
  const bool active = e < MM;
  const int i = active ? e / M : 0, j = active ? e % M : 0;
  const int b = (int)(bf / F), f = (int)(bf - (long long)b * F);

  Cd w;
  {
    const Cx<R> v = W[(size_t)bf * MM + i * M + j];
    w = cmake<double>((double)v.x, (double)v.y);
  }
  // the mixture covariance of the power statistic is only needed after the sweep: requested here, its round trip
  // hides behind the sweep instead of following it
  Cx<R> cv = cmake<R>((R)0, (R)0);
  if (pw) cv = C[(size_t)bf * MM + i * M + j];
  int flags = 0;
  // partial records covering this bin (FROM_PART): computed once, 32-bit arithmetic (NB < 2^31 is checked on the host)
  int g_lo = 0, g_hi = -1, base = 0;
  long long jrec = bf;
  int sub = 0;
  if (FROM_PART) {
    if (wb > 1) {
      jrec = (long long)b * ((F + wb - 1) / wb) + f / wb;
      sub = f % wb;
    }
    flat_cover(fp, jrec, g_lo, g_hi);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    base = (i == j) ? i : M + 2 * (lo * M - lo * (lo + 1) / 2 + (hi - lo - 1));
  }

  // ---- this lane's element of every U_n, fetched up front: the loads of the N sources are independent, the sweep
  //      below is a serial chain -- inside the loop each source would pay its own round trip to L2 / HBM (with 8
  //      utterances per launch the records no longer fit the L2: 51 us per utterance instead of 21)
  Cd uall[N];
  if (FROM_PART) {
#pragma unroll
    for (int n = 0; n < N; ++n) uall[n] = cmake<double>(0.0, 0.0);
    // records in chunks of RC with every load of a chunk issued before the first add (a bin is covered by 2-3
    // workgroups at benchmark size: one round trip instead of one per record); same summation order
    constexpr int RC = 4;
    for (int g0 = g_lo; g0 <= g_hi; g0 += RC) {
      R vx[RC][N], vy[RC][N];
#pragma unroll
      for (int c = 0; c < RC; ++c) {
        const int g = min(g0 + c, g_hi);
        const int slot = flat_slot(fp, jrec, g);
        const R* p = part + (((size_t)g * fp.S + slot) * wb + sub) * N * MM;
#pragma unroll
        for (int n = 0; n < N; ++n) {
          vx[c][n] = p[n * MM + base];
          vy[c][n] = (i != j) ? p[n * MM + base + 1] : (R)0;
        }
      }
#pragma unroll
      for (int c = 0; c < RC; ++c)
        if (g0 + c <= g_hi) {
#pragma unroll
          for (int n = 0; n < N; ++n) {
            uall[n].x += (double)vx[c][n];
            if (i != j) uall[n].y += (double)vy[c][n];
          }
        }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      if (i > j) uall[n].y = -uall[n].y;
      uall[n] = cmake<double>(uall[n].x * inv_T, uall[n].y * inv_T);
    }
  } else {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const Cx<R> v = U[(((size_t)b * N + n) * F + f) * MM + i * M + j];
      uall[n] = cmake<double>((double)v.x, (double)v.y);
    }
  }

#pragma unroll 1
  for (int n = 0; n < N; ++n) {
    Cd u = uall[0];
#pragma unroll
    for (int q = 1; q < N; ++q)
      if (q == n) u = uall[q];
    // ---- A = W @ U_n
    Cd a = cmake<double>(0.0, 0.0);
    static_for<M>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      cfma(a, group_row_bcast<M, GW, k>(w, i), group_shfl<GW>(u, k * M + j));
    });
    const Cd a0 = a;
    bool singular = false, ok = false;
    Cd wnew = cmake<double>(0.0, 0.0);
    // everything behind the inverse: w = (WU)^{-1} e_n ; den = sqrt(w^H U_n w) ; W[n,:] = conj(w) / den ; the guard
    auto finish = [&](Cd inv, bool sing, Cd& wnew_o, bool& ok_o, double* frob2) {
      const Cd wi = group_row_bcast_rt<M, GW>(inv, i, n);
      const Cd wj = group_shfl<GW>(inv, j * M + n);
      Cd term = cmul(cmul(cconj(wi), u), wj);
      if (!active) term = cmake<double>(0.0, 0.0);
      const Cd q = cmake<double>(group_sum<GW>(term.x), group_sum<GW>(term.y));
      Cd den = csqrt_fast(q);
      if (den.x < den_floor) den = cmake<double>(den_floor, 0.0);  // t-ILRMA only (ilrma.py:974-975); Gauss: floor 0
      wnew_o = cdiv_fast(cconj(wj), den);
      // ---- cond_2(WU) < threshold ?  LAST in program order (round 4): its two norm sums depend only on a0 and the inverse,
      // and an in-order wave overlaps them with the chain above only if they sit in the same basic block -- i.e. ahead of
      // the wave vote of the guard's slow path; evaluated first, they stood in front of the row's whole chain.
      ok_o = group_cond_below<M, GW>(a0, inv, active, sing, thr, frob2);
    };
#if ASSX_IP_ADJ
    if constexpr (M <= 4) {
      // Column n of the ADJUGATE gives the direction of w = (WU)^{-1} e_n at a third of the elimination's dependent depth,
      // and the new row conj(w) / sqrt(w^H U w) does not depend on its scale: no reciprocal of the determinant on the
      // chain (a floor on the normaliser -- t-ILRMA, FastMNMF -- does depend on it: those callers pay the reciprocal).
      // Off the chain, in the same basic block: the backward error of the column (group_adj_column_ok) and the guard,
      // from adj / det.  A wave with a bin that fails the check (or whose determinant is 0 / not finite: a singular W U_n)
      // runs the pivoted elimination as well and those groups take ITS result, flag and decision -- the reference's zgesv
      // is pivoted too, and its LinAlgError comes from there.
      Cd det;
      const Cd adj = group_adjugate<M, GW>(a0, i, j, active, det);
      Cd wi = group_row_bcast_rt<M, GW>(adj, i, n);
      Cd wj = group_shfl<GW>(adj, j * M + n);
      const Cd idet = crcp_fast(det);
      // x = det w: conj(x) / sqrt(x^H U x) = (conj(det) / |det|) conj(w) / sqrt(w^H U w) -- the modulus of det drops out,
      // its PHASE does not: the row is turned back by det / |det| at the end (one product; the factor is formed off the chain)
      const double dn2 = cabs2(det);
      const double rdn = rsqrt(dn2);
      const Cd phase = cmake<double>(det.x * rdn, det.y * rdn);
      if (den_floor > 0.0) {  // wave-uniform: the floor needs w itself
        wi = cmul(wi, idet);
        wj = cmul(wj, idet);
      }
      Cd term = cmul(cmul(cconj(wi), u), wj);
      if (!active) term = cmake<double>(0.0, 0.0);
      const Cd q = cmake<double>(group_sum<GW>(term.x), group_sum<GW>(term.y));
      Cd den = csqrt_fast(q);
      if (den.x < den_floor) den = cmake<double>(den_floor, 0.0);
      wnew = cdiv_fast(cconj(wj), den);
      if (!(den_floor > 0.0)) wnew = cmul(wnew, phase);
      double c2 = 0.0;
      ok = group_cond_below<M, GW>(a0, cmul(adj, idet), active, false, thr, &c2);
      const bool good = group_adj_column_ok<M, GW>(a0, group_shfl<GW>(adj, j * M + n), det, i, n, active) && c2 < 1e300 &&
                        q.x > 1e-280 && q.x < 1e280 && dn2 > 1e-280 && dn2 < 1e280;  // false for NaN
      if (__any(!good)) {
        bool sing_g = false, ok_g = false;
        Cd wnew_g;
        finish(group_gj_inverse<M, GW>(a0, i, j, sing_g), sing_g, wnew_g, ok_g, nullptr);
        if (!good) {
          wnew = wnew_g;
          ok = ok_g;
          singular = sing_g;
        }
      }
    } else
#endif
    {
      a = group_gj_inverse<M, GW>(a, i, j, singular);
      finish(a, singular, wnew, ok, nullptr);
    }
    if (singular) flags |= ASSX_STATUS_SINGULAR;       // numpy.linalg.solve raises here
    else if (!ok) flags |= ASSX_STATUS_COND_REJECT;    // keep the old row (np.where(condition, ..., w_n_Hermite))
    if (ok && !singular && i == n) w = wnew;
  }

  if (in_range && active) W[(size_t)bf * MM + i * M + j] = cmake<R>((R)w.x, (R)w.y);
  if (pw) {  // per-bin share of mean|y_n|^2 = mean_f w_n^H C_f w_n
    const Cd c = cmake<double>((double)cv.x, (double)cv.y);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const Cd wni = group_shfl<GW>(w, n * M + i);
      const Cd wnj = group_shfl<GW>(w, n * M + j);
      const Cd t1 = cmul(wni, c);
      double term = t1.x * wnj.x + t1.y * wnj.y;  // Re(W[n,i] C[i,j] conj(W[n,j]))
      if (!active) term = 0.0;
      const double sum = group_sum<GW>(term);
      if (in_range && e == 0) pw[((size_t)b * N + n) * F + f] = sum;
    }
  }
  if (flags && status && in_range && e == 0) atomicOr(&status[b], flags);
}

// ------------------------------------------------------------------------------------------
// (a5) IP sweep, sources side by side (round 4; M <= 4).  ip_group_kernel walks the sources one after the other and
//      every step is a full Gauss-Jordan inversion of W U_n by in-group shuffles: 4 x ~6500 cycles of dependent
//      latency on a chip that is otherwise idle (16.7 us for the 1025 bins of config 4).  But
//          (W U_n)^{-1} e_n = U_n^{-1} (W^{-1} e_n),
//      and U_n^{-1} does not depend on the sweep: here the M sources of a bin are M lane groups of ONE wave; group n
//      inverts U_n, every group inverts W (the two eliminations interleave), and the Gauss-Seidel part is left with,
//      per source:  P = U_n^{-1} A  (A = W^{-1} as updated so far: P IS (W U_n)^{-1}, its column n the solution, its
//      norm the condition guard's), den^2 = a_n^H P e_n (= w^H U_n w: U_n is Hermitian), the new row conj(w) / den, and
//      a rank-one update of A (Sherman-Morrison; the new pivot r^T a_n equals den up to rounding) -- products and
//      gathers by shuffles, two rounds deep, instead of an elimination.  The step of source n is computed by every
//      group (SIMD) and taken from group n, which broadcasts the new W and A.  Same guard (Frobenius bounds, exact
//      spectral norms in the ambiguous band, group n only), same flags, same floor on the normaliser (t-ILRMA).
//      Rounding differs from the elimination form by O(cond eps), like any two solvers.
//      MEASURED (profiles/r04_ip_par.txt): 19.2 us against 14.9 us for ip_group_kernel at config 4 -- the estimate
//      above counted shuffle rounds; what a lone wave on a SIMD pays is ~32 cycles per DEPENDENT f64 instruction, and
//      the step still strings ~100 of them together (two 4-deep complex product chains, a complex square root, three
//      complex reciprocals) behind two eliminations that the compiler does not interleave.  Off by default
//      (ASSX_IP_PAR=1 selects it; both forms are tested).
// ------------------------------------------------------------------------------------------
template <int M, int GW>
__device__ __forceinline__ bool group_cond_below_if(bool want, Cd a0, Cd ainv, bool active, bool singular, double thr) {
  const double nA2 = group_sum<GW>(active ? cabs2(a0) : 0.0);
  const double nI2 = group_sum<GW>(active ? cabs2(ainv) : 0.0);
  double c2 = nA2 * nI2, thr2 = thr * thr, m2 = (double)(M * M);
  if (!(c2 > 1e-290 && c2 < 1e290 && thr2 < 1e290)) {
    c2 = sqrt(nA2) * sqrt(nI2);
    thr2 = thr;
    m2 = (double)M;
  }
  const bool amb = want && !singular && (c2 == c2) && c2 >= thr2 && c2 < thr2 * m2;
  bool ok = !singular && (c2 == c2) && c2 < thr2;
  if (__any(amb)) {
    const int lane = threadIdx.x & (GW - 1);
    const int i = active ? lane / M : 0, j = active ? lane % M : 0;
    const bool below = group_cond_band<M, GW>(a0, ainv, i, j, active, amb, thr);
    if (amb) ok = below;
  }
  return ok;
}

template <typename R, int M, bool FROM_PART>
__global__ void __launch_bounds__(64)
    ip_par_kernel(const Cx<R>* __restrict__ U, const R* __restrict__ part, FlatPart fp, double inv_T,
                  Cx<R>* __restrict__ W, const Cx<R>* __restrict__ C, double* __restrict__ pw, double thr,
                  int32_t* __restrict__ status, int B, int F, double den_floor, int wb = 1) {
  constexpr int N = M;
  constexpr int MM = M * M;
  constexpr int GW = next_pow2_c(MM);
  constexpr int GPW = WAVE / GW;   // lane groups per wave
  constexpr int BPW = GPW / N;     // bins per wave (M = 4, 3: 1; M = 2: 8)
  static_assert(BPW >= 1, "a wave holds the sources of at least one bin");
  const int lane = threadIdx.x & (WAVE - 1);
  const int e = lane & (GW - 1);
  const int grp = lane / GW;
  const bool spare = grp >= BPW * N;                    // M = 3: the fourth group shadows source 2, stores nothing
  const int slot = spare ? BPW - 1 : grp / N;           // bin of the wave this group works on
  const int src = spare ? N - 1 : grp - slot * N;       // ... and its source
  const int lane0 = slot * N * GW;                      // first lane of the bin's groups
  const long long bin = (long long)blockIdx.x * BPW + slot;
  const bool in_range = bin < (long long)B * F;
  const long long bf = in_range ? bin : (long long)B * F - 1;
  const bool active = e < MM;
  const int i = active ? e / M : 0, j = active ? e % M : 0;
  const int b = (int)(bf / F), f = (int)(bf - (long long)b * F);

  Cd w;
  {
    const Cx<R> v = W[(size_t)bf * MM + i * M + j];
    w = cmake<double>((double)v.x, (double)v.y);
  }
  Cx<R> cv = cmake<R>((R)0, (R)0);
  if (pw) cv = C[(size_t)bf * MM + i * M + j];
  int flags = 0;
  // ---- this lane's element of U_src
  Cd u = cmake<double>(0.0, 0.0);
  if (FROM_PART) {
    long long jrec = bf;
    int sub = 0;
    if (wb > 1) {
      jrec = (long long)b * ((F + wb - 1) / wb) + f / wb;
      sub = f % wb;
    }
    int g_lo, g_hi;
    flat_cover(fp, jrec, g_lo, g_hi);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int base = (i == j) ? i : M + 2 * (lo * M - lo * (lo + 1) / 2 + (hi - lo - 1));
    constexpr int RC = 4;  // records in chunks whose loads are all in flight together; ascending order
    for (int g0 = g_lo; g0 <= g_hi; g0 += RC) {
      R vx[RC], vy[RC];
#pragma unroll
      for (int c = 0; c < RC; ++c) {
        const int g = min(g0 + c, g_hi);
        const R* p = part + (((size_t)g * fp.S + flat_slot(fp, jrec, g)) * wb + sub) * N * MM + src * MM + base;
        vx[c] = p[0];
        vy[c] = (i != j) ? p[1] : (R)0;
      }
#pragma unroll
      for (int c = 0; c < RC; ++c)
        if (g0 + c <= g_hi) {
          u.x += (double)vx[c];
          if (i != j) u.y += (double)vy[c];
        }
    }
    if (i > j) u.y = -u.y;
    u = cmake<double>(u.x * inv_T, u.y * inv_T);
  } else {
    const Cx<R> v = U[(((size_t)b * N + src) * F + f) * MM + i * M + j];
    u = cmake<double>((double)v.x, (double)v.y);
  }

  // ---- the two inversions that do not wait for the sweep
  bool sU = false, sW = false;
  const Cd Bm = group_gj_inverse<M, GW>(u, i, j, sU);  // U_src^{-1}
  Cd A = group_gj_inverse<M, GW>(w, i, j, sW);         // W^{-1}: identical in every group of the bin

#pragma unroll 1
  for (int n = 0; n < N; ++n) {
    const bool mine = src == n;
    // P = Bm A (= (W U_src)^{-1}); a0 = W U_src for the guard
    Cd P = cmake<double>(0.0, 0.0), a0 = cmake<double>(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < M; ++k) {
      cfma(P, group_shfl<GW>(Bm, i * M + k), group_shfl<GW>(A, k * M + j));
      cfma(a0, group_shfl<GW>(w, i * M + k), group_shfl<GW>(u, k * M + j));
    }
    const bool singular = sU || sW;
    // gathers along column n: s = a_n^H P e_n, zt_j = sum_k conj(P_kn) A_kj
    Cd sq = cmake<double>(0.0, 0.0), zt = cmake<double>(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const Cd pk = group_shfl<GW>(P, k * M + n), akn = group_shfl<GW>(A, k * M + n), akj = group_shfl<GW>(A, k * M + j);
      cfma(sq, cconj(akn), pk);
      cfma(zt, cconj(pk), akj);
    }
    const Cd wj = group_shfl<GW>(P, j * M + n);   // solution component j
    const Cd ain = group_shfl<GW>(A, i * M + n);  // column n of A, row i
    Cd den = csqrt_fast(sq);
    if (den.x < den_floor) den = cmake<double>(den_floor, 0.0);  // t-ILRMA only (ilrma.py:974-975); Gauss: floor 0
    const Cd rj = cdiv_fast(cconj(wj), den);                      // new row n, element j
    // A' = A - a_n (r^T A - e_n^T) / (r^T a_n),  r^T A = zt / den,  r^T a_n = conj(s) / den
    const Cd zj = cdiv_fast(zt, den), piv = cdiv_fast(cconj(sq), den);
    const Cd zd = cmake<double>(zj.x - (j == n ? 1.0 : 0.0), zj.y);
    const Cd corr = cmul(ain, cdiv_fast(zd, piv));
    // the guard LAST in program order: its norms (two group sums) are independent of everything above and an in-order wave
    // only overlaps what the compiler can interleave inside one basic block -- ahead of the wave vote of its slow path
    const bool ok = group_cond_below_if<M, GW>(mine, a0, P, active, singular, thr);
    const bool upd = ok && !singular;
    const Cd An = upd ? cmake<double>(A.x - corr.x, A.y - corr.y) : A;
    const Cd Wn = (upd && i == n) ? rj : w;
    if (mine && !spare) {
      if (singular) flags |= ASSX_STATUS_SINGULAR;
      else if (!ok) flags |= ASSX_STATUS_COND_REJECT;
    }
    // everybody takes the step from group n of its bin
    const int from = lane0 + n * GW + e;
    A = cmake<double>(__shfl(An.x, from, WAVE), __shfl(An.y, from, WAVE));
    w = cmake<double>(__shfl(Wn.x, from, WAVE), __shfl(Wn.y, from, WAVE));
  }

  if (in_range && active && src == 0 && !spare) W[(size_t)bf * MM + i * M + j] = cmake<R>((R)w.x, (R)w.y);
  if (pw) {  // per-bin share of mean|y_n|^2 = mean_f w_n^H C_f w_n: group src does source src
    const Cd c = cmake<double>((double)cv.x, (double)cv.y);
    const Cd wni = group_shfl<GW>(w, src * M + i);
    const Cd wnj = group_shfl<GW>(w, src * M + j);
    const Cd t1 = cmul(wni, c);
    double term = t1.x * wnj.x + t1.y * wnj.y;  // Re(W[n,i] C[i,j] conj(W[n,j]))
    if (!active) term = 0.0;
    const double sum = group_sum<GW>(term);
    if (in_range && e == 0 && !spare) pw[((size_t)b * N + src) * F + f] = sum;
  }
  if (flags && status && in_range && e == 0) atomicOr(&status[b], flags);
}

// ------------------------------------------------------------------------------------------
// (f1) ISS sweep (ilrma.py:537-564, iva.py:525-542, 758-775).  The reference applies the rank-1 updates to
//      Y (N passes that read and rewrite Y).  Y = W X stays linear in W, so the statistics are quadratic forms of
//      the SAME weighted covariances the IP path uses:
//          sum_t y_s conj(y_n) / r_s = T w_s U_s w_n^H ,   sum_t |y_n|^2 / r_s = T w_n U_s w_n^H
//      (sums, not means: the reference omits the 1/T of the paper) and the update is W[s,:] -= v_s W[n,:].
//      One pass over X (cov_stream_kernel) + this per-bin kernel, same lane-group layout as ip_group_kernel.
// ------------------------------------------------------------------------------------------
template <typename R, int M, bool FROM_PART>
__global__ void __launch_bounds__(64)
    iss_group_kernel(const Cx<R>* __restrict__ U, const R* __restrict__ part, FlatPart fp, double inv_T,
                     double n_frames, Cx<R>* __restrict__ W, const Cx<R>* __restrict__ C, double* __restrict__ pw,
                     int B, int F) {
  constexpr int N = M;
  constexpr int MM = M * M;
  constexpr int GW = next_pow2_c(MM);
  constexpr int GPW = WAVE / GW;
  const int lane = threadIdx.x & (WAVE - 1);
  const int e = lane & (GW - 1);
  const long long grp = (long long)blockIdx.x * GPW + lane / GW;
  const bool in_range = grp < (long long)B * F;
  const long long bf = in_range ? grp : (long long)B * F - 1;
  const bool active = e < MM;
  const int i = active ? e / M : 0, j = active ? e % M : 0;
  const int b = (int)(bf / F), f = (int)(bf - (long long)b * F);

  Cd w;
  {
    const Cx<R> v = W[(size_t)bf * MM + i * M + j];
    w = cmake<double>((double)v.x, (double)v.y);
  }
  // this lane's element (i, j) of every source's covariance (the weights are fixed during the sweep)
  Cd u[N];
  if (FROM_PART) {
    int g_lo, g_hi;
    flat_cover(fp, bf, g_lo, g_hi);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int base = (i == j) ? i : M + 2 * (lo * M - lo * (lo + 1) / 2 + (hi - lo - 1));
#pragma unroll
    for (int s = 0; s < N; ++s) {
      double re = 0.0, im = 0.0;
      for (int g = g_lo; g <= g_hi; ++g) {
        const int slot = flat_slot(fp, bf, g);
        const R* p = part + (((size_t)g * fp.S + slot) * N + s) * MM;
        re += (double)p[base];
        if (i != j) im += (double)p[base + 1];
      }
      if (i > j) im = -im;
      u[s] = cmake<double>(re * inv_T, im * inv_T);
    }
  } else {
#pragma unroll
    for (int s = 0; s < N; ++s) {
      const Cx<R> v = U[(((size_t)b * N + s) * F + f) * MM + i * M + j];
      u[s] = cmake<double>((double)v.x, (double)v.y);
    }
  }

#pragma unroll 1
  for (int n = 0; n < N; ++n) {
    const Cd wn_i = group_shfl<GW>(w, n * M + i);
    const Cd wn_jc = cconj(group_shfl<GW>(w, n * M + j));
    Cd vmine = cmake<double>(0.0, 0.0);  // v_s for this lane's row s = i
#pragma unroll
    for (int s = 0; s < N; ++s) {
      const Cd ws_i = group_shfl<GW>(w, s * M + i);
      const Cd uw = cmul(u[s], wn_jc);          // U_s[i][j] conj(W[n][j])
      Cd tq = cmul(ws_i, uw);                   // W[s][i] U_s[i][j] conj(W[n][j])
      Cd td = cmul(wn_i, uw);                   // W[n][i] U_s[i][j] conj(W[n][j])
      if (!active) {
        tq = cmake<double>(0.0, 0.0);
        td = tq;
      }
      const Cd q = cmake<double>(group_sum<GW>(tq.x), group_sum<GW>(tq.y));
      const double d = group_sum<GW>(td.x);     // Hermitian form: real
      Cd v;
      if (s == n) v = cmake<double>(1.0 - 1.0 / sqrt(n_frames * d), 0.0);
      else v = cmake<double>(q.x / d, q.y / d);
      if (i == s) vmine = v;
    }
    const Cd wn_j = cconj(wn_jc);
    const Cd dlt = cmul(vmine, wn_j);           // all rows use the OLD row n (Y - V_n Y[n] is evaluated at once)
    w = cmake<double>(w.x - dlt.x, w.y - dlt.y);
  }

  if (in_range && active) W[(size_t)bf * MM + i * M + j] = cmake<R>((R)w.x, (R)w.y);
  if (pw) {
    const Cx<R> cv = C[(size_t)bf * MM + i * M + j];
    const Cd c = cmake<double>((double)cv.x, (double)cv.y);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const Cd wni = group_shfl<GW>(w, n * M + i);
      const Cd wnj = group_shfl<GW>(w, n * M + j);
      const Cd t1 = cmul(wni, c);
      double term = t1.x * wnj.x + t1.y * wnj.y;
      if (!active) term = 0.0;
      const double sum = group_sum<GW>(term);
      if (in_range && e == 0) pw[((size_t)b * N + n) * F + f] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------
// (f1) IP2 / pairwise update of rows (pm, pn) (ilrma.py:566-633, iva.py:544-599).  Same lane-group layout.
//      P_x = (W U_x)^{-1} [e_pm e_pn];  V_x = P_x^H U_x P_x (2x2);  eig(V_pn^{-1} V_pm), eigenvectors sorted by
//      descending eigenvalue with LAPACK zgeev's convention (unit 2-norm, largest component real) so that W, not
//      only |W|, matches the reference;  w_x = conj(P_x v_x / sqrt(v_x^H V_x v_x)).  Both rows use the OLD W.
// ------------------------------------------------------------------------------------------
template <typename R, int M, bool FROM_PART>
__global__ void __launch_bounds__(64)
    ip2_group_kernel(const Cx<R>* __restrict__ U, const R* __restrict__ part, FlatPart fp, double inv_T,
                     Cx<R>* __restrict__ W, const Cx<R>* __restrict__ C, double* __restrict__ pw, double thr,
                     int32_t* __restrict__ status, int B, int F, int pm, int pn) {
  constexpr int N = M;
  constexpr int MM = M * M;
  constexpr int GW = next_pow2_c(MM);
  constexpr int GPW = WAVE / GW;
  const int lane = threadIdx.x & (WAVE - 1);
  const int e = lane & (GW - 1);
  const long long grp = (long long)blockIdx.x * GPW + lane / GW;
  const bool in_range = grp < (long long)B * F;
  const long long bf = in_range ? grp : (long long)B * F - 1;
  const bool active = e < MM;
  const int i = active ? e / M : 0, j = active ? e % M : 0;
  const int b = (int)(bf / F), f = (int)(bf - (long long)b * F);

  Cd w;
  {
    const Cx<R> v = W[(size_t)bf * MM + i * M + j];
    w = cmake<double>((double)v.x, (double)v.y);
  }
  auto load_u = [&](int src) -> Cd {
    if (FROM_PART) {
      int g_lo, g_hi;
      flat_cover(fp, bf, g_lo, g_hi);
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      const int base = (i == j) ? i : M + 2 * (lo * M - lo * (lo + 1) / 2 + (hi - lo - 1));
      double re = 0.0, im = 0.0;
      for (int g = g_lo; g <= g_hi; ++g) {
        const int slot = flat_slot(fp, bf, g);
        const R* p = part + (((size_t)g * fp.S + slot) * N + src) * MM;
        re += (double)p[base];
        if (i != j) im += (double)p[base + 1];
      }
      if (i > j) im = -im;
      return cmake<double>(re * inv_T, im * inv_T);
    }
    const Cx<R> v = U[(((size_t)b * N + src) * F + f) * MM + i * M + j];
    return cmake<double>((double)v.x, (double)v.y);
  };
  int flags = 0;
  const int col[2] = {pm, pn};
  Cd ux[2], inv[2];
  bool okx[2];
  Cd V[2][2][2];  // V[x][a][b], x = 0 -> source pm, 1 -> source pn
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    ux[x] = load_u(col[x]);
    Cd a = cmake<double>(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < M; ++k) cfma(a, group_shfl<GW>(w, i * M + k), group_shfl<GW>(ux[x], k * M + j));
    const Cd a0 = a;
    bool singular = false;
    a = group_gj_inverse<M, GW>(a, i, j, singular);
    okx[x] = group_cond_below<M, GW>(a0, a, active, singular, thr);
    if (singular) flags |= ASSX_STATUS_SINGULAR;  // numpy.linalg.inv raises
    else if (!okx[x]) flags |= ASSX_STATUS_COND_REJECT;
    inv[x] = a;
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const Cd pia = group_shfl<GW>(a, i * M + col[aa]);
        const Cd pjb = group_shfl<GW>(a, j * M + col[bb]);
        Cd term = cmul(cmul(cconj(pia), ux[x]), pjb);
        if (!active) term = cmake<double>(0.0, 0.0);
        V[x][aa][bb] = cmake<double>(group_sum<GW>(term.x), group_sum<GW>(term.y));
      }
  }
  // VV = V_pn^{-1} V_pm  (2x2)
  const Cd detn = csub(cmul(V[1][0][0], V[1][1][1]), cmul(V[1][0][1], V[1][1][0]));
  if (detn.x == 0.0 && detn.y == 0.0) flags |= ASSX_STATUS_SINGULAR;
  const Cd idet = cdiv(cmake<double>(1.0, 0.0), detn);
  const Cd ni[2][2] = {{cmul(V[1][1][1], idet), cmul(cmake<double>(-V[1][0][1].x, -V[1][0][1].y), idet)},
                       {cmul(cmake<double>(-V[1][1][0].x, -V[1][1][0].y), idet), cmul(V[1][0][0], idet)}};
  Cd VV[2][2];
#pragma unroll
  for (int aa = 0; aa < 2; ++aa)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) VV[aa][bb] = cadd(cmul(ni[aa][0], V[0][0][bb]), cmul(ni[aa][1], V[0][1][bb]));
  // eigenvalues of the 2x2
  const Cd htr = cscale(cadd(VV[0][0], VV[1][1]), 0.5);
  const Cd det = csub(cmul(VV[0][0], VV[1][1]), cmul(VV[0][1], VV[1][0]));
  const Cd disc = csqrt_principal(csub(cmul(htr, htr), det));
  Cd lam[2] = {cadd(htr, disc), csub(htr, disc)};
  // numpy argsort of complex = lexicographic (real, imag); order[::-1] -> largest first
  const bool first_big = (lam[0].x > lam[1].x) || (lam[0].x == lam[1].x && lam[0].y >= lam[1].y);
  if (!first_big) {
    const Cd t = lam[0];
    lam[0] = lam[1];
    lam[1] = t;
  }
  Cd wrow[2];  // this lane's new W[pm][j] / W[pn][j]
#pragma unroll
  for (int x = 0; x < 2; ++x) {  // x = 0: eigenvector of the larger eigenvalue -> row pm; x = 1 -> row pn
    // eigenvector of VV for lam[x]: the better conditioned of [b, lam - a] and [lam - d, c]
    const Cd c1[2] = {VV[0][1], csub(lam[x], VV[0][0])};
    const Cd c2[2] = {csub(lam[x], VV[1][1]), VV[1][0]};
    const double n1 = cabs2(c1[0]) + cabs2(c1[1]), n2 = cabs2(c2[0]) + cabs2(c2[1]);
    Cd v[2] = {n1 >= n2 ? c1[0] : c2[0], n1 >= n2 ? c1[1] : c2[1]};
    const double nrm = sqrt(n1 >= n2 ? n1 : n2);
    v[0] = cscale(v[0], 1.0 / nrm);
    v[1] = cscale(v[1], 1.0 / nrm);
    // zgeev: rotate so that the component of largest modulus is real (first one on ties)
    const int kbig = (cabs2(v[1]) > cabs2(v[0])) ? 1 : 0;
    const double mag = sqrt(cabs2(v[kbig]));
    const Cd rot = cscale(cconj(v[kbig]), 1.0 / mag);
    v[0] = cmul(v[0], rot);
    v[1] = cmul(v[1], rot);
    v[kbig].y = 0.0;
    // normalise by sqrt(v^H V_x v)
    Cd q = cmake<double>(0.0, 0.0);
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) cfma(q, cmul(cconj(v[aa]), V[x][aa][bb]), v[bb]);
    const Cd den = csqrt_principal(q);
    v[0] = cdiv(v[0], den);
    v[1] = cdiv(v[1], den);
    // w_x[c] = conj(P_x[c][0] v0 + P_x[c][1] v1), c = channel = this lane's column j
    const Cd p0 = group_shfl<GW>(inv[x], j * M + pm);
    const Cd p1 = group_shfl<GW>(inv[x], j * M + pn);
    wrow[x] = cconj(cadd(cmul(p0, v[0]), cmul(p1, v[1])));
  }
  if (i == pm && okx[0] && !(flags & ASSX_STATUS_SINGULAR)) w = wrow[0];
  if (i == pn && okx[1] && !(flags & ASSX_STATUS_SINGULAR)) w = wrow[1];

  if (in_range && active) W[(size_t)bf * MM + i * M + j] = cmake<R>((R)w.x, (R)w.y);
  if (pw) {
    const Cx<R> cv = C[(size_t)bf * MM + i * M + j];
    const Cd c = cmake<double>((double)cv.x, (double)cv.y);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const Cd wni = group_shfl<GW>(w, n * M + i);
      const Cd wnj = group_shfl<GW>(w, n * M + j);
      const Cd t1 = cmul(wni, c);
      double term = t1.x * wnj.x + t1.y * wnj.y;
      if (!active) term = 0.0;
      const double sum = group_sum<GW>(term);
      if (in_range && e == 0) pw[((size_t)b * N + n) * F + f] = sum;
    }
  }
  if (flags && status && in_range && e == 0) atomicOr(&status[b], flags);
}


// -2 T log|det W_f| (ilrma.py:675): LU with partial pivoting in registers, one thread per bin
template <int M, typename R>
__device__ __forceinline__ double neg2T_logabsdet(const Cx<R>* __restrict__ W, size_t bf, int T) {
  Cd A[M][M];
  const Cx<R>* p = W + bf * (M * M);
#pragma unroll
  for (int n = 0; n < M; ++n)
#pragma unroll
    for (int m = 0; m < M; ++m) A[n][m] = cmake<double>((double)p[n * M + m].x, (double)p[n * M + m].y);
  Cd det = lu_det<M>(A);
  return -2.0 * (double)T * log(hypot(det.x, det.y));
}

}  // namespace assx
