// Covariance accumulate for n_basis > 4 without materialising the source variance.
//
// With K basis vectors the variance of one (source, bin, frame) costs K activation reads; one bin per wave (the
// K <= 4 streaming kernel's structure) makes that K * N L2 reads per frame and bin -- L2-bound (333 us at K = 10).
// Here a workgroup of COVW_BINS waves owns COVW_BINS consecutive bins of the SAME frame block: the activation tile
// (N*K rows x 64 frames) is fetched once per block by LDS-direct loads (no registers, double-buffered, requested one
// block ahead) and serves all the bins; wave w forms r_n = sum_k T[n, f0+w, k] V[n, k, t] from LDS (the basis rows
// sit in LDS too, read as broadcasts), then the M*M Hermitian products of its bin and the N weighted accumulates.
// Work is handed out as a flat partition of (utterance, bin group, frame block) items like the streaming kernels
// (exact balance: F = 1025 gives 129 bin groups, any fixed tiling of which leaves half a round of the chip idle);
// a record is flushed when the range leaves a bin group:  part[g][slot][w][n][M*M].
#pragma once
#include "assx_stream.hpp"

namespace assx {

constexpr int COVW_BINS = 8;

// SB = 64-frame sub-blocks per work item.  With SB = 2 a workgroup passes ONE barrier per 128 frames and the loads
// of the next item are issued a whole item's arithmetic (~1.7 us) ahead -- about the HBM latency -- where SB = 1
// waited ~2 us per 64-frame block for loads issued 0.8 us earlier (97 us per pass at config-4 size, K = 10).
template <typename R, int SB = 1>
struct CovWideGeom {
  static constexpr int FBW = WAVE * SB;                    // frames per item
  static constexpr int ROW_BYTES = FBW * (int)sizeof(R);
  static constexpr int LPR = ROW_BYTES / 16;               // lanes per row (16 bytes per lane)
  static constexpr int RPI = WAVE / LPR;                   // rows per LDS-direct instruction
  static_assert(LPR >= 1 && LPR <= WAVE && WAVE % LPR == 0, "row geometry");
  static __host__ __device__ int rows_padded(int NK) { return (NK + RPI - 1) / RPI * RPI; }
  static __host__ __device__ size_t tile_bytes(int NK) { return (size_t)rows_padded(NK) * ROW_BYTES; }
  static __host__ __device__ size_t lds_bytes(int NK) { return 2 * tile_bytes(NK) + (size_t)COVW_BINS * NK * sizeof(R); }
};

template <typename R, int M, bool D2, int SB = 1>
__global__ void __launch_bounds__(WAVE * COVW_BINS)
    cov_wide_kernel(const Cx<R>* __restrict__ X, const R* __restrict__ Tb, const R* __restrict__ V, R* __restrict__ part,
                    Dims d, FlatPart fp, R eps, PowSpec p2d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int N = M, HM = M * M, WB = COVW_BINS, NACC = N * HM, NV = next_pow2_c(NACC);
  using GEO = CovWideGeom<R, SB>;
  constexpr int FBW = GEO::FBW;
  const int F = d.F, T = d.T, K = d.K, NK = N * K, TBk = fp.len, FG = (F + WB - 1) / WB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned tile_bytes = (unsigned)GEO::tile_bytes(NK);
  R* Tl = reinterpret_cast<R*>(smem + 2 * tile_bytes);
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = (int)blockIdx.x;
  long long q0, q1;
  flat_range(fp, g, q0, q1);
  if (q0 >= q1) return;
  const int nblk = (int)(q1 - q0);
  const int jg_first = (int)(q0 / TBk);
  Cursor cc;  // .f counts bin groups here, .tb items of FBW frames
  cc.tb = (int)(q0 - (long long)jg_first * TBk);
  cc.b = jg_first / FG;
  cc.f = jg_first - cc.b * FG;
  const size_t FT = (size_t)F * T;
  const BufRsrc rv = make_rsrc_sized(V, (size_t)d.B * NK * T * sizeof(R));
  const int NI = GEO::rows_padded(NK) / GEO::RPI;

  auto issue_tile = [&](const Cursor& c, int buf) {
    for (int ri = w; ri < NI; ri += WB) {
      int row = ri * GEO::RPI + lane / GEO::LPR;
      row = row < NK ? row : NK - 1;
      const unsigned voff = (unsigned)(((size_t)(c.b * NK + row) * T + (size_t)c.tb * FBW) * sizeof(R)) +
                            (unsigned)(lane % GEO::LPR) * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rv, (__attribute__((address_space(3))) void*)(smem + (unsigned)buf * tile_bytes + (unsigned)ri * GEO::RPI * GEO::ROW_BYTES),
          16, (int)voff, 0, 0, 0);
    }
  };
  auto load_rows = [&](const Cursor& c) {  // basis rows of the group's bins: Tl[ww][n*K + k]
    for (int i = tid; i < WB * NK; i += WAVE * WB) {
      const int ww = i / NK, nk = i - ww * NK, n = nk / K, k = nk - n * K;
      const int ff = min(c.f * WB + ww, F - 1);
      Tl[i] = Tb[(((size_t)c.b * N + n) * F + ff) * K + k];
    }
  };
  auto load_x = [&](const Cursor& c, Cx<R> (&x)[SB][M]) {
    const int ff = min(c.f * WB + w, F - 1);
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
      const int t = min(c.tb * FBW + sb * WAVE + lane, T - 1);
      const Cx<R>* xb = X + (size_t)c.b * M * FT + (size_t)ff * T + t;
#pragma unroll
      for (int m = 0; m < M; ++m) x[sb][m] = xb[m * FT];
    }
  };

  R acc[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) acc[q] = 0;
  Cx<R> xn[SB][M];
  load_rows(cc);
  issue_tile(cc, 0);
  load_x(cc, xn);
  for (int it = 0; it < nblk; ++it) {
    const Cursor cur = cc;
    advance(cc, TBk, FG);
    const bool more = it + 1 < nblk;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of tile `it` (and its X item) has landed
    __syncthreads();                                  // ... everyone's has; tile it-1 and the old rows are no longer read
    Cx<R> x[SB][M];
#pragma unroll
    for (int sb = 0; sb < SB; ++sb)
#pragma unroll
      for (int m = 0; m < M; ++m) x[sb][m] = xn[sb][m];
    if (more) {
      issue_tile(cc, (it + 1) & 1);
      load_x(cc, xn);
    }
    const R* tl = Tl + w * NK;
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
      const R* vl = reinterpret_cast<const R*>(smem + (unsigned)(it & 1) * tile_bytes) + sb * WAVE + lane;
      const bool live = (cur.f * WB + w < F) && (cur.tb * FBW + sb * WAVE + lane < T);
      // r_n = sum_k T[n,f,k] V[n,k,t], k ascending per source.  The N sources advance together and k in pairs, so
      // 4N LDS reads are in flight per wait instead of 2.
      R tvv[N];
#pragma unroll
      for (int n = 0; n < N; ++n) tvv[n] = 0;
      int k = 0;
      for (; k + 2 <= K; k += 2) {
        R t0[N], t1[N], v0[N], v1[N];
#pragma unroll
        for (int n = 0; n < N; ++n) {
          t0[n] = tl[n * K + k];
          t1[n] = tl[n * K + k + 1];
          v0[n] = vl[(n * K + k) * FBW];
          v1[n] = vl[(n * K + k + 1) * FBW];
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
          tvv[n] = fma(t0[n], v0[n], tvv[n]);
          tvv[n] = fma(t1[n], v1[n], tvv[n]);
        }
      }
      if (k < K) {
#pragma unroll
        for (int n = 0; n < N; ++n) tvv[n] = fma(tl[n * K + k], vl[(n * K + k) * FBW], tvv[n]);
      }
      R wgt[N];
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const R r = floor_eps<R>(D2 ? tvv[n] : powspec<R>(tvv[n], p2d), eps);  // floored AFTER the power (ilrma.py:499-509)
        wgt[n] = live ? fast_rcp(r) : (R)0;
      }
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const R pd = cabs2(x[sb][m]);
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n * HM + m] = fma(wgt[n], pd, acc[n * HM + m]);
      }
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int l = m + 1; l < M; ++l) {
          const Cx<R> pr = cmulc(x[sb][m], x[sb][l]);
          const int hb = herm_pair_base<M>(m, l);
#pragma unroll
          for (int n = 0; n < N; ++n) {
            acc[n * HM + hb] = fma(wgt[n], pr.x, acc[n * HM + hb]);
            acc[n * HM + hb + 1] = fma(wgt[n], pr.y, acc[n * HM + hb + 1]);
          }
        }
    }
    if (cc.tb == 0 || !more) {  // the bin group is complete (or the range ends): flush, take the next group's rows
      const R tot = wave_reduce_scatter<R, NV>(acc);
      const int i = scatter_index<NV>();
      const int slot = cur.b * FG + cur.f - jg_first;
      if (scatter_leader<NV>() && i < NACC) part[(((size_t)g * fp.S + slot) * WB + w) * NACC + i] = tot;
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = 0;
      if (more) {
        __syncthreads();  // every wave has read the old rows
        load_rows(cc);
      }
    }
  }
#endif
}

// sum the records covering each bin group, scale by 1/T, expand packed Hermitian -> dense U (B,N,F,M,M)
template <typename R, int M>
__global__ void __launch_bounds__(256) cov_wide_finalize_kernel(const R* __restrict__ part, Cx<R>* __restrict__ U, int B,
                                                               int F, FlatPart fp, R inv_T) {
  constexpr int N = M, HM = M * M, WB = COVW_BINS;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * F * HM;
  if (idx >= total) return;
  const int l = idx % M, m = (idx / M) % M;
  const int f = (idx / HM) % F;
  const int n = (idx / ((size_t)HM * F)) % N;
  const int b = idx / ((size_t)HM * F * N);
  const int FG = (F + WB - 1) / WB;
  const long long j = (long long)b * FG + f / WB;
  int g_lo, g_hi;
  flat_cover(fp, j, g_lo, g_hi);
  R re = 0, im = 0;
  for (int g = g_lo; g <= g_hi; ++g) {
    const int slot = flat_slot(fp, j, g);
    const R* p = part + ((((size_t)g * fp.S + slot) * WB + f % WB) * N + n) * HM;
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

}  // namespace assx
