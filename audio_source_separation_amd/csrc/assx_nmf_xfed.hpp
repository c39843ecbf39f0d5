// The ILRMA source model for n_basis > 4 (ilrma.py:356-366, 409-430) straight from the mixture: the two matrix-core
// NMF halves of assx_nmf_mfma.hpp with their target P_n = |w_n^H x|^2 formed ON THE FLY.
//
// Until round 3 the source model went  X --demix_power_map_kernel--> P (B,N,F,T)  --two NMF halves--> Tb, V :  268.7 MB
// read + 134.3 MB written by the map (62.6 us at config 4), then 134.3 MB read by each half.  An X-fed half was priced in
// round 3 and dropped ("needs all four sources in one wave to read X once -- ~300 VGPRs -- or re-reads X once per
// source").  There is a third way: the sources of an utterance become the WAVES of a workgroup.  Wave n owns source n;
// all waves walk the same 16 x 16 sub-tiles of the spectrogram, the sub-tile's samples of the M channels are staged
// ONCE per workgroup in LDS (each wave fetches one channel: 16 rows x 256 B, coalesced), every wave forms its own
// |w_n^H x|^2 from that tile in the accumulator layout and feeds it to the very chain of products of the map-fed
// kernels.  X is read once per half (268.7 MB), the map and its launch are gone, and no cross-wave combine is left:
// a wave's accumulators are already one source's sums.
//
// Work partition, records, tickets and the multiplicative step are those of assx_nmf_mfma.hpp (NmfPart over one
// utterance's (block, step) space; a step is done by all M waves at once; record matrices are indexed b * N + n like the
// batch of the map-fed kernels, so the slab layout -- and nmf_ws -- are shared).  Double-buffered tile + one workgroup
// barrier per step.
#pragma once
#include "assx_nmf_mfma.hpp"

namespace assx {

// ---------------------------------------------------------------------------------------------------------
// basis half.  grid (G, 1, B), M waves; block = 16 bins, step = 16 frames.  part[slab][(b*M + n)*2 + s][f*K + k]
// ---------------------------------------------------------------------------------------------------------
// LOSS (domain 2, Gaussian model: D2K = IS_MM): the pass also accumulates the data term of the negative log-likelihood
// of the model it reads -- sum_{n,f,t} P / R + log R with the very P = |w_n^H x|^2 and R = max(Tb V, eps) it forms anyway
// (ilrma.py:672-675), every (n, f, t) exactly once -- so recording the loss of iteration i costs a dozen instructions per
// element of iteration i+1's basis half instead of a pass over X of its own (the n_basis <= 4 kernels do the same,
// assx_stream.hpp).  sum log R travels as a mantissa product + exponent.  One partial per (workgroup, source):
// lpart[b * lstride + g * M + n]; the caller adds the log-det terms and sums (ilrma_loss_finish_kernel).
template <typename R, int M, int KT, int D2K, bool LOSS = false>
__global__ void __launch_bounds__(64 * M)
    nmf_basis_xfed_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W, R* Tb, const R* __restrict__ V,
                          R* part, int* tickets, NmfPart pt, int B, int F, int T, int K, R eps, TermSpec s, PowSpec pe,
                          double* __restrict__ lpart, int lstride) {
  // (Round 4 also tried leaving the demixed power behind for a map-fed activation half: the 134 MB of 32-byte stores took
  // this half from 67 to 106 us, the map-fed activation half gave back 33 -- profiles/r04_xfed_act_route.txt -- and the
  // per-element store branches cut the block the compiler schedules.  Removed.)
  static_assert(!LOSS || D2K == ASSX_NMF_IS_MM, "the fused loss is the domain-2 Gaussian one");
  using MM = Mfma16<R>;
  using acc_t = typename MM::acc_t;
  constexpr int N = M;
  constexpr int KS = KT * 4, KP = KT * 16, LD = 17, NLD = KP * 16 / 64;
  constexpr int XLD = 17;  // padded row (16-byte units) of the staged X tile: conflict-free b128 reads down a column
  __shared__ R vts[M * KP * LD];
  __shared__ __attribute__((aligned(16))) Vec2<R> xt[2][M][16][XLD];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, n = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave = source = channel fetched
  const int li = lane & 15, lk = lane >> 4;
  const int b = blockIdx.z, g = blockIdx.x;
  const bool last_slice = K > 4 * (KS - 1);  // the last k-slice of product (1) holds a basis vector (wave-uniform)
  const size_t bn = (size_t)b * N + n, BN = (size_t)B * N;
  R(*vt)[LD] = reinterpret_cast<R(*)[LD]>(vts + n * KP * LD);
  const R* vb = V + bn * K * T;
  const size_t FT = (size_t)F * T, FK = (size_t)F * K;
  const Cx<R>* xm = X + ((size_t)b * M + n) * FT;  // the channel this wave stages
  unsigned voff[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) voff[i] = (unsigned)((min(4 * i + lk, K - 1) * (size_t)T + li) * sizeof(R));
  // X tile piece i of this lane: row lk + 4 i of the block's 16 bins, frames li of the step's 16 (one 16 / 8-byte sample)
  const unsigned xvoff = (unsigned)(((size_t)lk * T + li) * sizeof(Cx<R>));
  const unsigned xrow4 = 4u * (unsigned)T * (unsigned)sizeof(Cx<R>);
  const BufRsrc vrs = make_rsrc(vb), xrs = make_rsrc(xm);

  double ltot = 0.0;  // LOSS: this wave's share, summed over the blocks of its range

  unsigned lo, hi;
  nmf_part_range(pt, g, lo, hi);
  for (int blk = (int)(lo / (unsigned)pt.nstep); (unsigned)blk * (unsigned)pt.nstep < hi; ++blk) {
    const unsigned base = (unsigned)blk * (unsigned)pt.nstep;
    const int s0 = lo > base ? (int)(lo - base) : 0;
    const int s1 = hi - base < (unsigned)pt.nstep ? (int)(hi - base) : pt.nstep;
    const int gf = pt.w > 0 ? blk * pt.w : nmf_part_owner(pt, base);  // aligned: the block's w workgroups
    const int members = pt.w > 0 ? pt.w : nmf_part_owner(pt, base + pt.nstep - 1) - gf + 1;
    const int slot = g - gf;
    const int f0 = blk * 16;
    const int f = min(f0 + li, F - 1);  // rows past F feed only output rows that are never written
    const bool rows_in = f0 + 16 <= F;
    double lacc = 0.0, lm = 1.0;  // LOSS: sum P/R of this lane's bin; sum log R as mantissa product + exponent
    int le = 0;

    R tb[KS];  // B operand of product (1): Tb^T[k = 4j + lk][f]
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int k = 4 * j + lk;
      tb[j] = (k < K) ? Tb[(bn * F + f) * K + k] : (R)0;
    }
    Vec2<R> w[M];  // this source's demixing row of the lane's bin
#pragma unroll
    for (int m = 0; m < M; ++m) w[m] = ldv<R>(W + (((size_t)b * F + f) * N + n) * M + m);
    acc_t num[KT], den[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        num[c][r] = 0;
        den[c][r] = 0;
      }
    const int te = min(T, s1 * 16);
    R stage[NLD];
    Vec2<R> xs[4];  // this wave's channel of the NEXT step's tile
    auto fetch = [&](int t0) {
      if (t0 + 16 <= T && rows_in) {  // whole tile inside the matrix (workgroup-uniform)
        const unsigned soff = (unsigned)t0 * (unsigned)sizeof(R);
#pragma unroll
        for (int i = 0; i < NLD; ++i) stage[i] = buf_ld<R>(vrs, voff[i], soff);
        const unsigned xso = ((unsigned)f0 * (unsigned)T + (unsigned)t0) * (unsigned)sizeof(Cx<R>);
#pragma unroll
        for (int i = 0; i < 4; ++i) xs[i] = buf_ldv<R>(xrs, xvoff, xso + i * xrow4);
      } else {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          const int e = i * 64 + lane;
          const int k = min(e >> 4, K - 1), tt = min(t0 + (e & 15), T - 1);
          stage[i] = vb[(size_t)k * T + tt];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          xs[i] = ldv<R>(xm + (size_t)min(f0 + lk + 4 * i, F - 1) * T + min(t0 + li, T - 1));
      }
    };
    auto compute = [&](auto masked, int t0, int buf) {
      acc_t tv;
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[r] = 0;
#pragma unroll
      for (int j = 0; j < KS; ++j)
        if (j < KS - 1 || last_slice) tv = MM::mma(vt[4 * j + lk][li], tb[j], tv);  // n_basis 9..12 (the reference's 10): 3 of 4 slices
      R a[4], bm[4];
      double lprod = 1.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Vec2<R> x[M];
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = xt[buf][m][li][MM::crow(r, lane)];
        Cx<R> y = cmake<R>(0, 0);
#pragma unroll
        for (int m = 0; m < M; ++m) cfma(y, tocx<R>(w[m]), tocx<R>(x[m]));
        const R P = cabs2(y);
        nmf_terms<R, D2K>(s, P, tv[r], eps, a[r], bm[r]);
        const bool dead = decltype(masked)::value && t0 + MM::crow(r, lane) >= T;
        if (LOSS && !dead) {  // D2K = IS_MM: bm = 1 / max(tv, eps)
          lacc += (double)(P * bm[r]);
          lprod *= (double)floor_eps<R>(tv[r], eps);
        }
        if (dead) {
          a[r] = 0;
          bm[r] = 0;
        }
      }
      if (LOSS) {
        int e;
        lm = frexp(lm * lprod, &e);
        le += e;
      }
#pragma unroll
      for (int c = 0; c < KT; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const R vv = vt[16 * c + li][MM::crow(r, lane)];
          num[c] = MM::mma(a[r], vv, num[c]);
          den[c] = MM::mma(bm[r], vv, den[c]);
        }
      }
    };
    auto step = [&](auto masked, int t0, int buf) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) vt[(i * 64 + lane) >> 4][lane & 15] = stage[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) xt[buf][n][lk + 4 * i][li] = xs[i];
      if (t0 + 16 < te) fetch(t0 + 16);  // the next tile travels while this one is consumed
      __syncthreads();  // the M channels of the tile are in place; buffer `buf ^ 1` is free again once everybody is here
      compute(masked, t0, buf);
    };
    int t0 = s0 * 16, it = 0;
    if (t0 < te) fetch(t0);
    for (; t0 + 16 <= te; t0 += 16, ++it) step(IntC<0>(), t0, it & 1);
    if (t0 < te) step(IntC<1>(), t0, it & 1);

    if (LOSS && f0 + li < F)  // a lane's elements all belong to one bin: rows past F (copies of row F-1) drop out here
      ltot += lacc + (double)le * 0.6931471805599453 + log(lm);
    // ---- end of this workgroup's share of the block: every wave holds its own source's sums
    R* pn = part + ((size_t)slot * BN * 2 + bn * 2) * FK;
    const bool direct = members == 1;
    // two straight-line forms of the output: in one loop body the `direct` path's read of Tb made the compiler wait for
    // ALL memory operations between the record stores -- they went out one round trip after the other (assx_nmf_mfma.hpp)
    if (direct) {
#pragma unroll
      for (int c = 0; c < KT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kb = 16 * c + li, fo = f0 + MM::crow(r, lane);
          if (fo < F && kb < K) {
            R* tp = Tb + bn * FK + (size_t)fo * K + kb;
            *tp = nmf_apply<R, D2K>(*tp, num[c][r], den[c][r], eps, pe);
          }
        }
    } else {
#pragma unroll
      for (int c = 0; c < KT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kb = 16 * c + li, fo = f0 + MM::crow(r, lane);
          if (fo < F && kb < K) {
            const size_t o = (size_t)fo * K + kb;
            st_agent(pn + o, num[c][r]);
            st_agent(pn + FK + o, den[c][r]);
          }
        }
    }
    if (!direct) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's records are out ...
      __syncthreads();                                   // ... and so are the other sources'
      if (n == 0) {
        const bool last = take_ticket(tickets + (size_t)b * pt.nblk + blk, members);
        if (lane == 0) s_last = last;
      }
      __syncthreads();
      if (s_last) {
        // holder of the last ticket: wave n sums the slabs of source n's 16 x K block and updates Tb in place (every
        // member read its rows of Tb at the top of the block, before it took its ticket)
        // numerator and denominator sums of an output in flight together (slab_sum4_n: same additions, same bits; 16 loads
        // per trip as before -- this kernel runs three workgroups per CU and has no registers to spare for more)
        const R* p0 = part + bn * 2 * FK;
        R* tbo = Tb + bn * FK;
        const size_t slab = BN * 2 * FK;
        for (int o = lane; o < 16 * K; o += 64) {
          const int fo = f0 + o / K;
          unsigned idx[1];
          bool on[1];
          on[0] = fo < F;
          idx[0] = on[0] ? (unsigned)fo * (unsigned)K + (unsigned)(o % K) : 0u;
          const R old = on[0] ? tbo[idx[0]] : (R)0;
          R sum[2];
          slab_sum4_n<R, 1, 8>(p0, p0 + FK, idx, on, slab, members, sum);
          if (on[0]) tbo[idx[0]] = nmf_apply<R, D2K>(old, sum[0], sum[1], eps, pe);
        }
      }
    }
    __syncthreads();  // the tile buffers and s_last are reused by the next block of this range
  }
  if (LOSS) {
    ltot = wave_allreduce_sum<double>(ltot);
    if (lane == 0) lpart[(size_t)b * lstride + (size_t)g * M + n] = ltot;
  }
}

// ---------------------------------------------------------------------------------------------------------
// activation half.  grid (G, 1, B), M waves; block = 16 frames, step = 16 bins.  part[slab][(b*M + n)*2 + s][k*T + t]
// The demixing rows of the step's 16 bins (all sources: 16 x M x M samples, contiguous in W) are staged with the tile.
// ---------------------------------------------------------------------------------------------------------
template <typename R, int M, int KT, int D2K>
__global__ void __launch_bounds__(64 * M)
    nmf_act_xfed_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W, const R* __restrict__ Tb, R* V,
                        R* part, int* tickets, NmfPart pt, int B, int F, int T, int K, R eps, TermSpec s, PowSpec pe) {
  using MM = Mfma16<R>;
  using acc_t = typename MM::acc_t;
  constexpr int N = M;
  constexpr int KS = KT * 4, KP = KT * 16, LD = KP + 4, NLD = KP * 16 / 64;
  constexpr int XLD = 17;
  constexpr int WPB = M * M;                    // samples of W per bin
  constexpr int WPT = (16 * WPB + 64 * M - 1) / (64 * M);  // pieces of the W tile per thread (1 for M = 2..4)
  __shared__ R tts[M * 16 * LD];
  __shared__ __attribute__((aligned(16))) Vec2<R> xt[2][M][16][XLD];
  __shared__ __attribute__((aligned(16))) Vec2<R> wt[2][16 * WPB];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, n = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int b = blockIdx.z, g = blockIdx.x;
  const bool last_slice = K > 4 * (KS - 1);  // the last k-slice of product (1) holds a basis vector (wave-uniform)
  const size_t bn = (size_t)b * N + n, BN = (size_t)B * N;
  R(*tt_)[LD] = reinterpret_cast<R(*)[LD]>(tts + n * 16 * LD);
  const R* tbb = Tb + bn * F * K;
  const size_t FT = (size_t)F * T, KTt = (size_t)K * T;
  const Cx<R>* xm = X + ((size_t)b * M + n) * FT;
  const Cx<R>* wb = W + (size_t)b * F * WPB;
  constexpr bool TUNI = (64 % KP) == 0;
  unsigned toff[TUNI ? 1 : NLD];
#pragma unroll
  for (int i = 0; i < (TUNI ? 1 : NLD); ++i) {
    const int e = i * 64 + lane;
    toff[i] = (unsigned)(((size_t)(e / KP) * K + min(e % KP, K - 1)) * sizeof(R));
  }
  const unsigned tstep = (unsigned)(64 / KP) * (unsigned)K * (unsigned)sizeof(R);
  const unsigned xvoff = (unsigned)(((size_t)lk * T + li) * sizeof(Cx<R>));
  const unsigned xrow4 = 4u * (unsigned)T * (unsigned)sizeof(Cx<R>);
  const BufRsrc trs = make_rsrc(tbb), xrs = make_rsrc(xm);

  unsigned lo, hi;
  nmf_part_range(pt, g, lo, hi);
  for (int blk = (int)(lo / (unsigned)pt.nstep); (unsigned)blk * (unsigned)pt.nstep < hi; ++blk) {
    const unsigned base = (unsigned)blk * (unsigned)pt.nstep;
    const int s0 = lo > base ? (int)(lo - base) : 0;
    const int s1 = hi - base < (unsigned)pt.nstep ? (int)(hi - base) : pt.nstep;
    const int gf = pt.w > 0 ? blk * pt.w : nmf_part_owner(pt, base);  // aligned: the block's w workgroups
    const int members = pt.w > 0 ? pt.w : nmf_part_owner(pt, base + pt.nstep - 1) - gf + 1;
    const int slot = g - gf;
    const int t0 = blk * 16;
    const int t = min(t0 + li, T - 1);
    const bool tvalid = t0 + li < T;
    const bool cols_in = t0 + 16 <= T;

    R vbr[KS];  // B operand of product (1): V[k = 4j + lk][t]
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int k = 4 * j + lk;
      vbr[j] = (k < K) ? V[(bn * K + k) * T + t] : (R)0;
    }
    acc_t num[KT], den[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        num[c][r] = 0;
        den[c][r] = 0;
      }
    const int fe = min(F, s1 * 16);
    R stage[NLD];
    Vec2<R> xs[4];
    Vec2<R> ws_[WPT];
    auto fetch = [&](int f0) {
      if (f0 + 16 <= fe && cols_in) {
        const unsigned tso = (unsigned)f0 * (unsigned)K * (unsigned)sizeof(R);
#pragma unroll
        for (int i = 0; i < NLD; ++i) stage[i] = TUNI ? buf_ld<R>(trs, toff[0], tso + i * tstep) : buf_ld<R>(trs, toff[TUNI ? 0 : i], tso);
        const unsigned xso = ((unsigned)f0 * (unsigned)T + (unsigned)t0) * (unsigned)sizeof(Cx<R>);
#pragma unroll
        for (int i = 0; i < 4; ++i) xs[i] = buf_ldv<R>(xrs, xvoff, xso + i * xrow4);
      } else {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          const int e = i * 64 + lane;
          const int fr = f0 + e / KP, k = min(e % KP, K - 1);
          stage[i] = (fr < fe) ? tbb[(size_t)fr * K + k] : (R)0;  // bins beyond the range contribute 0
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          xs[i] = ldv<R>(xm + (size_t)min(f0 + lk + 4 * i, F - 1) * T + min(t0 + li, T - 1));
      }
#pragma unroll
      for (int q = 0; q < WPT; ++q) {  // the 16 bins' demixing rows: piece (thread) of a contiguous run of W
        const int e = q * 64 * M + (int)threadIdx.x;
        const int fr = min(f0 + e / WPB, F - 1);
        ws_[q] = (e < 16 * WPB) ? ldv<R>(wb + (size_t)fr * WPB + e % WPB) : Vec2<R>{0, 0};
      }
    };
    auto compute = [&](auto masked, int f0, int buf) {
      acc_t tv;
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[r] = 0;
#pragma unroll
      for (int j = 0; j < KS; ++j)
        if (j < KS - 1 || last_slice) tv = MM::mma(tt_[li][4 * j + lk], vbr[j], tv);
      R a[4], bm[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = MM::crow(r, lane);  // bin of the sub-tile this accumulator register belongs to
        Cx<R> y = cmake<R>(0, 0);
#pragma unroll
        for (int m = 0; m < M; ++m) cfma(y, tocx<R>(wt[buf][row * WPB + n * M + m]), tocx<R>(xt[buf][m][row][li]));
        nmf_terms<R, D2K>(s, cabs2(y), tv[r], eps, a[r], bm[r]);
        if (decltype(masked)::value && f0 + row >= fe) {
          a[r] = 0;
          bm[r] = 0;
        }
      }
#pragma unroll
      for (int c = 0; c < KT; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const R tv3 = tt_[MM::crow(r, lane)][16 * c + li];
          num[c] = MM::mma(tv3, a[r], num[c]);
          den[c] = MM::mma(tv3, bm[r], den[c]);
        }
      }
    };
    auto step = [&](auto masked, int f0, int buf) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int e = i * 64 + lane;
        tt_[e / KP][e % KP] = stage[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) xt[buf][n][lk + 4 * i][li] = xs[i];
#pragma unroll
      for (int q = 0; q < WPT; ++q) {
        const int e = q * 64 * M + (int)threadIdx.x;
        if (e < 16 * WPB) wt[buf][e] = ws_[q];
      }
      if (f0 + 16 < fe) fetch(f0 + 16);
      __syncthreads();
      compute(masked, f0, buf);
    };
    int f0 = s0 * 16, it = 0;
    if (f0 < fe) fetch(f0);
    for (; f0 + 16 <= fe; f0 += 16, ++it) step(IntC<0>(), f0, it & 1);
    if (f0 < fe) step(IntC<1>(), f0, it & 1);

    R* pn = part + ((size_t)slot * BN * 2 + bn * 2) * KTt;
    const bool direct = members == 1;
    if (direct) {  // two straight-line forms, as in the basis half
#pragma unroll
      for (int c = 0; c < KT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kb = 16 * c + MM::crow(r, lane);  // D[row = kb][col = t]
          if (kb < K && tvalid) {
            R* vp = V + bn * KTt + (size_t)kb * T + t;
            *vp = nmf_apply<R, D2K>(*vp, num[c][r], den[c][r], eps, pe);
          }
        }
    } else {
#pragma unroll
      for (int c = 0; c < KT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kb = 16 * c + MM::crow(r, lane);
          if (kb < K && tvalid) {
            const size_t o = (size_t)kb * T + t;
            st_agent(pn + o, num[c][r]);
            st_agent(pn + KTt + o, den[c][r]);
          }
        }
    }
    if (!direct) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (n == 0) {
        const bool last = take_ticket(tickets + (size_t)b * pt.nblk + blk, members);
        if (lane == 0) s_last = last;
      }
      __syncthreads();
      if (s_last) {
        const R* p0 = part + bn * 2 * KTt;
        R* vo = V + bn * KTt;
        const size_t slab = BN * 2 * KTt;
        for (int o = lane; o < 16 * K; o += 128) {
          unsigned idx[2];
          bool on[2];
          R old[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int ou = o + 64 * u, tc = t0 + (ou & 15);
            on[u] = ou < 16 * K && tc < T;
            idx[u] = on[u] ? (unsigned)(ou >> 4) * (unsigned)T + (unsigned)tc : 0u;
            old[u] = on[u] ? vo[idx[u]] : (R)0;
          }
          R sum[4];
          slab_sum4_n<R, 2, 4>(p0, p0 + KTt, idx, on, slab, members, sum);  // 16 loads per trip, as before
#pragma unroll
          for (int u = 0; u < 2; ++u)
            if (on[u]) vo[idx[u]] = nmf_apply<R, D2K>(old[u], sum[2 * u], sum[2 * u + 1], eps, pe);
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace assx
