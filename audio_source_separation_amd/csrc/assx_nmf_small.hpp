// IS-NMF MM half-updates for n_basis <= 4 on the vector ALU (domain 2; nmf.py:302-327, ilrma.py:409-430 on P = |W x|^2).
//
// The matrix-core kernels (assx_nmf_mfma.hpp) pad the rank to 16: at n_basis <= 4 three quarters of every product are
// padding, and a half-update over the wide-channel path's power map (8 x 1025 x 4096 at M = 8) took 119 + 80 us where
// the map is 268 MB -- 50 us at the speed of its bytes.  With four basis vectors everything fits one lane: a lane owns a
// frame, the element terms (nmf_terms<IS_MM>: r = 1 / max(tv, eps) by rcp + Newton, x r r, r) cost ~20 instructions, and
// the two contractions are 2 n_basis multiply-adds into per-lane sums -- the kernels stream the map once, coalesced.
//   basis half:      a wave owns SMALL_BPW consecutive bins (their basis rows in SGPRs) and walks a frame range; the
//                    activation rows of a 64-frame block are loaded ONCE for the wave's bins; the 2 n_basis sums per bin
//                    are reduced over the lanes at the end (butterfly reduce-scatter) -> part[ts][b*2 + s][f*K + k]
//   activation half: a wave owns one 64-frame block (its activation values in registers) and walks a bin range, four bins
//                    in flight; sums per lane = per frame                  -> part[fs][b*2 + s][k*T + t]
// Same record layout and the same nmf_finalize_kernel as the matrix-core path; slab counts depend on ONE problem's geometry
// and assx_ctx::nmf_group only (batch-invariant).  Summation order differs from the matrix-core kernels (rounding only).
#pragma once
#include "assx_common.hpp"

namespace assx {

constexpr int SMALL_K = 4;    // largest n_basis served.  The kernels are written for KC = 4, 8, 12 or 16 rows per lane and
                              // were measured at all of them (profiles/r03_nmf_small_rank.txt): beyond 4 the matrix cores win
                              // (n_basis 6: 192-194 us per ILRMA source update against 187; n_basis 10: 229 against 196;
                              // n_basis 16: 271 against 208), so only KC = 4 is instantiated
constexpr int small_bpw(int KC) { return KC <= 4 ? 4 : 2; }  // bins per wave in the basis half (their rows live in SGPRs)
constexpr int small_uf(int KC) { return KC <= 4 ? 4 : 2; }   // bins in flight in the activation half
inline int small_kc(int K) { return 4; }

template <typename R, int KC>
__global__ void __launch_bounds__(256)
    nmf_basis_small_kernel(const R* __restrict__ X, const R* __restrict__ Tb, const R* __restrict__ V, R* __restrict__ part,
                           int B, int F, int T, int K, int tchunk, R eps) {
  constexpr int NB = small_bpw(KC), NACC = 2 * NB * KC, NV = next_pow2_c(NACC);  // 32 sums at KC = 4, 64 at KC = 16
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int b = blockIdx.z, ts = blockIdx.y;
  const int f0 = (blockIdx.x * 4 + wv) * NB;
  if (f0 >= F) return;  // no workgroup barriers in this kernel
  const int ta = ts * tchunk, te = min(T, ta + tchunk);
  R tb[NB][KC];  // wave-uniform basis rows (0 beyond n_basis / beyond the last bin)
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int k = 0; k < KC; ++k)
      tb[i][k] = (f0 + i < F && k < K) ? Tb[((size_t)b * F + f0 + i) * K + k] : (R)0;
  const R* xrow[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) xrow[i] = X + ((size_t)b * F + min(f0 + i, F - 1)) * T;
  const R* vb = V + (size_t)b * K * T;
  R acc[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) acc[q] = 0;
  // two blocks in flight (register sets A / B, the loop unrolled by two so that each copy names its set): with one block
  // ahead a wave had 4 KB in flight and the kernel sat at 3.6 TB/s of its map
  R xa[NB], va[KC], xb[NB], vbb[KC];
  auto fetch = [&](int t0, R (&xd)[NB], R (&vd)[KC]) {
    const int t = min(t0 + lane, T - 1);
#pragma unroll
    for (int i = 0; i < NB; ++i) xd[i] = xrow[i][t];
#pragma unroll
    for (int k = 0; k < KC; ++k) vd[k] = vb[(size_t)min(k, K - 1) * T + t];  // rows past n_basis: their basis is 0
  };
  auto consume = [&](int t0, const R (&x)[NB], const R (&v)[KC]) {
    const bool live = t0 + lane < te;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      R tv = 0;
#pragma unroll
      for (int k = 0; k < KC; ++k) tv = fma(tb[i][k], v[k], tv);
      const R bm = live ? fast_rcp(floor_eps<R>(tv, eps)) : (R)0;
      const R a = x[i] * bm * bm;
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        acc[(i * 2 + 0) * KC + k] = fma(a, v[k], acc[(i * 2 + 0) * KC + k]);
        acc[(i * 2 + 1) * KC + k] = fma(bm, v[k], acc[(i * 2 + 1) * KC + k]);
      }
    }
  };
  if (ta < te) fetch(ta, xa, va);
  if (ta + WAVE < te) fetch(ta + WAVE, xb, vbb);
  for (int t0 = ta; t0 < te; t0 += 2 * WAVE) {
    {
      R x[NB], v[KC];
#pragma unroll
      for (int i = 0; i < NB; ++i) x[i] = xa[i];
#pragma unroll
      for (int k = 0; k < KC; ++k) v[k] = va[k];
      if (t0 + 2 * WAVE < te) fetch(t0 + 2 * WAVE, xa, va);  // two blocks ahead
      consume(t0, x, v);
    }
    if (t0 + WAVE < te) {
      R x[NB], v[KC];
#pragma unroll
      for (int i = 0; i < NB; ++i) x[i] = xb[i];
#pragma unroll
      for (int k = 0; k < KC; ++k) v[k] = vbb[k];
      if (t0 + 3 * WAVE < te) fetch(t0 + 3 * WAVE, xb, vbb);
      consume(t0 + WAVE, x, v);
    }
  }
  const R tot = wave_reduce_scatter<R, NV>(acc);
  const int q = scatter_index<NV>();
  if (scatter_leader<NV>() && q < NACC) {
    const int i = q / (2 * KC), s = (q / KC) & 1, k = q % KC;
    const size_t FK = (size_t)F * K;
    if (f0 + i < F && k < K) part[((size_t)ts * B * 2 + (size_t)b * 2 + s) * FK + (size_t)(f0 + i) * K + k] = tot;
  }
}

template <typename R, int KC>
__global__ void __launch_bounds__(64)
    nmf_act_small_kernel(const R* __restrict__ X, const R* __restrict__ Tb, const R* __restrict__ V, R* __restrict__ part,
                         int B, int F, int T, int K, int fchunk, R eps) {
  constexpr int UF = small_uf(KC);  // bins in flight
  const int lane = threadIdx.x;
  const int b = blockIdx.z, fs = blockIdx.y;
  const int t = blockIdx.x * WAVE + lane;
  const bool live = t < T;
  const int tc = live ? t : T - 1;
  R v[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) v[k] = V[((size_t)b * K + min(k, K - 1)) * T + tc];
  R num[KC], den[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) num[k] = den[k] = 0;
  const int fa = fs * fchunk, fe = min(F, fa + fchunk);
  const R* xb = X + (size_t)b * F * T + tc;
  const R* tbb = Tb + (size_t)b * F * K;
  for (int f0 = fa; f0 < fe; f0 += UF) {
    R x[UF], tk[UF][KC];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int f = min(f0 + u, F - 1);
      x[u] = xb[(size_t)f * T];
#pragma unroll
      for (int k = 0; k < KC; ++k)  // wave-uniform addresses: scalar loads
        tk[u][k] = (f0 + u < fe && k < K) ? tbb[(size_t)f * K + k] : (R)0;  // bins past the range contribute 0
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      R tv = 0;
#pragma unroll
      for (int k = 0; k < KC; ++k) tv = fma(tk[u][k], v[k], tv);
      const R bm = fast_rcp(floor_eps<R>(tv, eps));
      const R a = x[u] * bm * bm;
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        num[k] = fma(tk[u][k], a, num[k]);
        den[k] = fma(tk[u][k], bm, den[k]);
      }
    }
  }
  if (live) {
    const size_t KTt = (size_t)K * T;
    R* pn = part + ((size_t)fs * B * 2 + (size_t)b * 2) * KTt;
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (k < K) {
        pn[(size_t)k * T + t] = num[k];
        pn[KTt + (size_t)k * T + t] = den[k];
      }
  }
}

}  // namespace assx
