// ILRMA source model (IS-NMF MM update of basis, then activation, on P = |W x|^2; ref: src/bss/ilrma.py:356-366,
// 409-430) as two STREAMING passes over X, one wave per source -- for the shapes whose source model went through a
// materialised power map: n_basis 5..16 at M <= 4 (the reference's default is 10) and every n_basis <= 16 at 5 <= M <= 8.
//
// The map route costs a pass that reads X and writes P (N F T reals: 63 us at config 4, 141 us at M = 8) before two
// matrix-core NMF halves read P back (63 + 43 us at n_basis 10 -- on a rank padded to 16 --, 119 + 80 us at M = 8, n_basis
// 4).  Here each half is ONE pass over X: a workgroup of N waves walks (bin, 64-frame block) items, wave n forms
// y_n = sum_m W[f,n,m] x_m of its own source from the item's rows of X, P = |y_n|^2, the variance
// tv = sum_k Tb[n,f,k] V[n,k,t] (k ascending, fused multiply-adds, floored), and accumulates the two sums of the update,
//     basis half:       num[f,k] += P / tv^2 * V[k,t]      den[f,k] += 1 / tv * V[k,t]      (over t)
//     activation half:  num[k,t] += Tb[f,k] * P / tv^2     den[k,t] += Tb[f,k] / tv         (over f)
// -- the element terms are those of the matrix-core kernels (nmf_terms<IS_MM>: r = 1 / max(tv, eps) by rcp + Newton,
// P * r * r, r) and the sums are applied by the same assx_nmf_apply_sums; only the summation order differs.
// Memory pipeline: that of src_cov_kernel (assx_widem_cov.hpp).  Everything an item needs rides ONE LDS ring DXS items
// deep filled by LDS-direct loads: the M rows of X (wave n carries row n) and, wave-private, the source's demixing row
// (M complex) and basis row (n_basis reals; lanes past n_basis fetch beyond the array and land as zeros) as
// dword-per-lane loads, and -- basis half only -- the n_basis activation rows of the item's frame block.  One counted
// wait per trip, vmcnt((DXS - 2) * C).  The two halves walk the items in different orders so that each sum stays in
// registers: the basis half frame block fastest (a bin's 2 n_basis sums per lane, butterfly reduce-scatter when the bin
// ends), the activation half bin fastest (a frame block's sums per lane = per frame, no cross-lane step; the activation
// values stay in registers for the whole frame block).  Work is the flat balanced partition of the streaming kernels
// (FlatPart), records are reduced in workgroup order by two small kernels: deterministic and batch-invariant.
#pragma once
#include "assx_stream.hpp"
#include "assx_widem_cov.hpp"  // the inline-asm LDS read helpers

namespace assx {

enum { SRC_NMF_BASIS = 0, SRC_NMF_ACT = 1 };
constexpr int SRC_NMF_KMAX = 16;

template <typename R, int M, int KR, int PASS>
struct SrcNmfGeom {
  static constexpr int RB = WAVE * 2 * (int)sizeof(R);   // bytes of one row of X: 64 complex frames
  static constexpr int LPR = RB / 16;                    // lanes that carry an X row
  static constexpr int RBV = WAVE * (int)sizeof(R);      // bytes of one activation row: 64 reals
  static constexpr int NVI = PASS == SRC_NMF_BASIS ? (KR * RBV + 1023) / 1024 : 0;  // activation-row instructions
  static constexpr int VBYTES = NVI * 1024;
  static constexpr int ROWB = 4 * WAVE;                  // landing area of a dword-per-lane row
  static constexpr int C = 1 + NVI + 2;                  // VMEM instructions of one item's request, per wave
  static constexpr int WSLOT = VBYTES + 2 * ROWB;        // wave-private bytes per slot: activation rows, W row, Tb row
  static constexpr int SLOT = M * RB + M * WSLOT;
  static constexpr int TWO = (78 * 1024) / SLOT;         // slots when two workgroups share a CU's 160 KB
  static constexpr int ONE = (156 * 1024) / SLOT;
  static constexpr int DXS = TWO >= 3 ? (TWO > 6 ? 6 : TWO) : (ONE > 6 ? 6 : ONE);
  static constexpr bool FITS = DXS >= 3;                 // the ring needs three slots (n_basis 16 at M >= 6 in float64 does not)
  static constexpr size_t lds_bytes = (size_t)DXS * SLOT;
  static constexpr int NA = 2 * KR;                      // sums per (source, group): num[0..KR), den[0..KR)
};

// KR: activation / basis rows kept (n_basis <= KR <= SRC_NMF_KMAX).  PASS: SRC_NMF_BASIS -- fp over (bin, frame block)
// items, groups = bins, part[g][slot][n][NA]; SRC_NMF_ACT -- fp over (frame block, bin) items, groups = frame blocks,
// part[g][slot][n][NA][64].
template <typename R, int M, int KR, int PASS>
__global__ void __launch_bounds__(WAVE * M)
    src_nmf_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ Wf, const R* __restrict__ Tb,
                   const R* __restrict__ V, R* __restrict__ part, Dims d, FlatPart fp, R eps) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int N = M;
  using GEO = SrcNmfGeom<R, M, KR, PASS>;
  constexpr int RB = GEO::RB, DXS = GEO::DXS, NA = GEO::NA, NV = next_pow2_c(NA);
  constexpr unsigned SLOT = (unsigned)GEO::SLOT;
  constexpr bool BASIS = PASS == SRC_NMF_BASIS;
  const int F = d.F, T = d.T, K = d.K;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int n = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's source
  const int g = (int)blockIdx.x;
  int b0, grp0, it0, nblk;
  if (!flat_start(fp, g, b0, grp0, it0, nblk)) return;  // whole workgroup: the range is wave-uniform
  // item cursor: (group, item in group) = (bin, frame block) in the basis half, (frame block, bin) in the activation half
  struct Item {
    int grp, it;
  };
  const int glen = fp.len;
  auto next = [&](Item& c) {
    if (++c.it == glen) {
      c.it = 0;
      ++c.grp;
    }
  };
  auto bin_of = [&](const Item& c) { return BASIS ? c.grp : c.it; };
  auto blk_of = [&](const Item& c) { return BASIS ? c.it : c.grp; };
  const size_t FT = (size_t)F * T;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned wpriv = (unsigned)M * RB + (unsigned)n * GEO::WSLOT;  // this wave's private part of a slot

  // ---- requests (the range never leaves utterance b0; offsets inside an utterance's arrays are 32-bit: host check)
  const BufRsrc rx = make_rsrc_sized(X + (size_t)b0 * M * FT, (size_t)M * FT * sizeof(Cx<R>));
  const unsigned xlane = (unsigned)((size_t)n * FT * sizeof(Cx<R>)) + (unsigned)lane * 16u;  // wave n carries row n
  const BufRsrc rvb = make_rsrc_sized(V + (size_t)b0 * N * K * T, (size_t)N * K * T * sizeof(R));
  constexpr int LPV = GEO::RBV / 16;  // lanes per activation row of an LDS-direct instruction
  unsigned vlane[GEO::NVI > 0 ? GEO::NVI : 1];
#pragma unroll
  for (int j = 0; j < GEO::NVI; ++j) {  // rows past n_basis repeat the last one (their basis entry is 0)
    const int row = min(j * (WAVE / LPV) + lane / LPV, K - 1);
    vlane[j] = (unsigned)((size_t)row * T * sizeof(R)) + (unsigned)(lane % LPV) * 16u;
  }
  auto words = [&](const void* p, size_t bytes) {
    buf_u4 r = make_rsrc_words(p, bytes);
    r.x = __builtin_amdgcn_readfirstlane(r.x);
    r.y = __builtin_amdgcn_readfirstlane(r.y);
    r.z = __builtin_amdgcn_readfirstlane(r.z);
    r.w = __builtin_amdgcn_readfirstlane(r.w);
    return r;
  };
  const buf_u4 rw = words(Wf + (size_t)b0 * F * N * M, (size_t)F * N * M * sizeof(Cx<R>));
  const buf_u4 rt = words(Tb + (size_t)b0 * N * F * K, (size_t)N * F * K * sizeof(R));
  constexpr unsigned WPR = (unsigned)sizeof(R) / 4u;  // dwords per real
  const unsigned wvoff = (unsigned)lane * 4u;         // the demixing row: 2 M reals, whatever follows is never read
  const unsigned tvoff = (unsigned)lane < (unsigned)K * WPR ? (unsigned)lane * 4u : 0x7ffffff0u;  // past n_basis: zeros
  auto request = [&](const Item& c, int sl) {
    const unsigned sbase = (unsigned)sl * SLOT;
    const int f = bin_of(c), tb = blk_of(c);
    const unsigned priv = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + sbase + wpriv + GEO::VBYTES));
    const unsigned wso = (unsigned)__builtin_amdgcn_readfirstlane((int)((((size_t)f * N + n) * M) * sizeof(Cx<R>)));
    const unsigned tso = (unsigned)__builtin_amdgcn_readfirstlane((int)((((size_t)n * F + f) * K) * sizeof(R)));
    buf_dword_to_lds(priv, rw, wvoff, wso);
    buf_dword_to_lds(priv + GEO::ROWB, rt, tvoff, tso);
    if constexpr (BASIS) {
      const unsigned vsoff = (unsigned)(((size_t)n * K * T + (size_t)tb * WAVE) * sizeof(R));
#pragma unroll
      for (int j = 0; j < GEO::NVI; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rvb, (__attribute__((address_space(3))) void*)(smem + sbase + wpriv + (unsigned)j * 1024u), 16, (int)vlane[j],
            (int)vsoff, 0, 0);
    }
    const unsigned xoff = xlane + (unsigned)(((size_t)f * T + (size_t)tb * WAVE) * sizeof(Cx<R>));
    if (GEO::LPR == WAVE || lane < GEO::LPR)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rx, (__attribute__((address_space(3))) void*)(smem + sbase + (unsigned)n * RB), 16, (int)xoff, 0, 0, 0);
  };

  // ---- activation values of this lane's frame: from the ring per item (basis half) or held for a whole frame block
  R vv[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) vv[k] = 0;
  const buf_u4 rvw = words(V + (size_t)b0 * N * K * T, (size_t)N * K * T * sizeof(R));
  auto load_v_block = [&](int tb) {  // activation half: the block's n_basis rows, once per frame block (rare: drains the ring)
    const unsigned row0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(((size_t)n * K * T + (size_t)tb * WAVE) * sizeof(R)));
    const unsigned tstep = (unsigned)__builtin_amdgcn_readfirstlane((int)((size_t)T * sizeof(R)));
    unsigned so = row0;
    static_for<KR>([&](auto kc) {
      buf_ld_tied(vv[decltype(kc)::value], rvw, (unsigned)lane * (unsigned)sizeof(R), so);
      so = sgpr_opaque(so + (decltype(kc)::value + 1 < K ? tstep : 0u));
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < KR; ++k) asm volatile("" : "+v"(vv[k]));
  };

  R acc[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) acc[q] = 0;
  Item c0;
  c0.grp = grp0;
  c0.it = it0;
  if (!BASIS) load_v_block(blk_of(c0));  // before the ring starts: nothing to drain yet
  // prologue: items 0 .. DXS-2 requested into slots 0 .. DXS-2 (past the end of the range: item 0 again -- the queue keeps
  // its shape); slot DXS-1 is the free one the first trip fills
  Item cr = c0;  // the next item to request
  {
    const Item first = c0;
#pragma unroll
    for (int i = 0; i < DXS - 1; ++i) {
      request(i < nblk ? cr : first, i);
      next(cr);
    }
  }
  Item c1 = c0;
  next(c1);
  wait_vmcnt<(DXS - 2) * GEO::C>();  // item 0 has landed (this wave's share: the barrier publishes the rest)
  int sl = 0;                        // ring slot of item `it`
  for (int it = 0; it < nblk; ++it) {
    const bool more = it + 1 < nblk;
    // every wave's row of item `it` has landed (each waited for its own share at the end of the previous trip) and nobody
    // reads the slot of item it-1 any more.  A bare barrier: only LDS traffic crosses waves.
    asm volatile("s_barrier" ::: "memory");
    // ---- LDS reads of the item (inline asm, see assx_widem_cov.hpp): demixing row, basis row, activation, X
    const unsigned base = lds0 + (unsigned)sl * SLOT;
    const unsigned pa = base + wpriv + GEO::VBYTES;
    Vec2<R> wrow[M];  // broadcast reads: every lane holds the row
    static_for<M>([&](auto mc) { widem::lds_read_cx<decltype(mc)::value * 2 * (int)sizeof(R)>(pa, wrow[decltype(mc)::value]); });
    Vec2<R> trow[KR / 2];
    static_for<KR / 2>([&](auto kc) { widem::lds_read_cx<GEO::ROWB + decltype(kc)::value * 2 * (int)sizeof(R)>(pa, trow[decltype(kc)::value]); });
    Vec2<R> vrow[BASIS ? KR / 2 : 1];
    if constexpr (BASIS) {
      const unsigned va = base + wpriv + (unsigned)lane * (unsigned)sizeof(R);
      static_for<KR / 2>([&](auto kc) { widem::lds_read_rows2<decltype(kc)::value>(va, vrow[decltype(kc)::value]); });
    }
    Vec2<R> x[M];
    const unsigned xa = base + (unsigned)lane * (unsigned)sizeof(Cx<R>);
    static_for<M>([&](auto mc) { widem::lds_read_cx<decltype(mc)::value * RB>(xa, x[decltype(mc)::value]); });
    // the request of item it+DXS-1 reuses the slot of item it-1
    request(it + DXS - 1 < nblk ? cr : c0, sl == 0 ? DXS - 1 : sl - 1);  // past the end: the current item again
    widem::lds_wait<0>();
#pragma unroll
    for (int m = 0; m < M; ++m) asm volatile("" : "+v"(wrow[m]), "+v"(x[m]));
#pragma unroll
    for (int k = 0; k < KR / 2; ++k) asm volatile("" : "+v"(trow[k]));
    if constexpr (BASIS) {
#pragma unroll
      for (int k = 0; k < KR / 2; ++k) {
        asm volatile("" : "+v"(vrow[k]));
        vv[2 * k] = vrow[k].x;
        vv[2 * k + 1] = vrow[k].y;
      }
    }
    // ---- y = W x of this source, P = |y|^2, the element terms of the IS update (nmf_terms<IS_MM>, domain 2)
    R yr = 0, yi = 0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      yr = fma(wrow[m].x, x[m].x, yr);
      yr = fma(-wrow[m].y, x[m].y, yr);
      yi = fma(wrow[m].x, x[m].y, yi);
      yi = fma(wrow[m].y, x[m].x, yi);
    }
    const R P = fma(yr, yr, yi * yi);
    R tv = 0;
#pragma unroll
    for (int k = 0; k < KR; ++k) tv = fma(k & 1 ? trow[k / 2].y : trow[k / 2].x, vv[k], tv);  // entries past n_basis are 0 * finite
    const bool live = blk_of(c0) * WAVE + lane < T;
    const R bm = live ? fast_rcp(floor_eps<R>(tv, eps)) : (R)0;
    const R a = P * bm * bm;
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const R tk = k & 1 ? trow[k / 2].y : trow[k / 2].x;
      acc[k] = BASIS ? fma(a, vv[k], acc[k]) : fma(tk, a, acc[k]);
      acc[KR + k] = BASIS ? fma(bm, vv[k], acc[KR + k]) : fma(tk, bm, acc[KR + k]);
    }
#pragma unroll
    for (int q = 0; q < NA; ++q) asm volatile("" : "+v"(acc[q]));  // the arithmetic stays above the wait
    // this wave's share of item it+1 has landed once only the requests of items it+2 .. it+DXS-1 are in flight
    wait_vmcnt<(DXS - 2) * GEO::C>();
    const bool group_ends = c1.it == 0 || !more;
    if (group_ends) {
      const int slot = c0.grp - grp0;
      if (BASIS) {  // a bin's sums: reduce over the 64 frames of the lanes
        const R tot = wave_reduce_scatter<R, NV>(acc);
        const int i = scatter_index<NV>();
        if (scatter_leader<NV>() && i < NA) part[(((size_t)g * fp.S + slot) * N + n) * NA + i] = tot;
      } else {  // a frame block's sums: one per lane = per frame
#pragma unroll
        for (int q = 0; q < NA; ++q) part[((((size_t)g * fp.S + slot) * N + n) * NA + q) * WAVE + lane] = acc[q];
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = 0;
      if (!BASIS && more) load_v_block(blk_of(c1));  // the next item belongs to the next frame block
    }
    sl = sl + 1 == DXS ? 0 : sl + 1;
    c0 = c1;
    next(c1);
    next(cr);
  }
  wait_vmcnt<0>();  // nothing may land in LDS after the workgroup has gone
#endif
}

// records -> sums (2, B N, F K) of the basis half: sums[s][b N + n][f K + k], records added in workgroup order
template <typename R, int KR>
__global__ void __launch_bounds__(256) src_nmf_basis_sums_kernel(const R* __restrict__ part, R* __restrict__ sums, int B,
                                                                int N, int F, int K, FlatPart fp) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * F * K;
  if (idx >= total) return;
  constexpr int NA = 2 * KR;
  const int k = idx % K, f = (idx / K) % F;
  const int n = (idx / ((size_t)K * F)) % N;
  const int b = idx / ((size_t)K * F * N);
  const long long j = (long long)b * F + f;
  int g_lo, g_hi;
  flat_cover(fp, j, g_lo, g_hi);
  R num = 0, den = 0;
  for (int g = g_lo; g <= g_hi; ++g) {
    const R* p = part + (((size_t)g * fp.S + flat_slot(fp, j, g)) * N + n) * NA;
    num += p[k];
    den += p[KR + k];
  }
  sums[idx] = num;
  sums[total + idx] = den;
}

// records -> sums (2, B N, K T) of the activation half: sums[s][b N + n][k T + t]
template <typename R, int KR>
__global__ void __launch_bounds__(256) src_nmf_act_sums_kernel(const R* __restrict__ part, R* __restrict__ sums, int B,
                                                              int N, int K, int T, FlatPart fp) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * K * T;
  if (idx >= total) return;
  constexpr int NA = 2 * KR;
  const int t = idx % T, k = (idx / T) % K;
  const int n = (idx / ((size_t)T * K)) % N;
  const int b = idx / ((size_t)T * K * N);
  const int TBk = (T + WAVE - 1) / WAVE;
  const long long j = (long long)b * TBk + t / WAVE;
  int g_lo, g_hi;
  flat_cover(fp, j, g_lo, g_hi);
  R num = 0, den = 0;
  for (int g = g_lo; g <= g_hi; ++g) {
    const R* p = part + (((size_t)g * fp.S + flat_slot(fp, j, g)) * N + n) * NA * WAVE + (t % WAVE);
    num += p[(size_t)k * WAVE];
    den += p[(size_t)(KR + k) * WAVE];
  }
  sums[idx] = num;
  sums[total + idx] = den;
}

// ---- host side -------------------------------------------------------------------------------------------------
struct SrcNmfPlan {
  FlatPart fb, fa;          // basis-half / activation-half partitions
  size_t rec_bytes, sums_bytes;
};
inline int src_nmf_kr(int K) { return K <= 4 ? 4 : (K <= 8 ? 8 : (K <= 10 ? 10 : (K <= 12 ? 12 : 16))); }
inline SrcNmfPlan src_nmf_plan(int B, int N, int F, int T, int K, size_t r, long long g_target) {
  SrcNmfPlan p;
  const int tbk = (T + WAVE - 1) / WAVE, kr = src_nmf_kr(K);
  p.fb = make_flat(B, (long long)F * tbk, tbk, g_target);
  p.fa = make_flat(B, (long long)tbk * F, F, g_target);
  const size_t rb = (size_t)p.fb.G * p.fb.S * N * 2 * kr, ra = (size_t)p.fa.G * p.fa.S * N * 2 * kr * WAVE;
  p.rec_bytes = align_up((rb > ra ? rb : ra) * r, 256);
  const size_t sb = (size_t)2 * B * N * F * K, sa = (size_t)2 * B * N * K * T;
  p.sums_bytes = align_up((sb > sa ? sb : sa) * r, 256);
  return p;
}
inline bool src_nmf_ok(int M, int F, int T, int K, double domain, size_t r) {
  return K <= SRC_NMF_KMAX && domain == 2.0 && (size_t)M * F * T * 2 * r < 0xffffffffull && (size_t)M * K * T * r < 0xffffffffull;
}

constexpr int SRC_NMF_NO_FIT = -1000;  // internal: this (M, n_basis, precision) does not fit the LDS ring -- take the map route

template <typename R, int M, int KR, int PASS>
int src_nmf_launch_pass(assx_ctx* ctx, const void* X, const void* W, const void* Tb, const void* V, void* rec,
                        const FlatPart& fp, int B, int F, int T, int K, double eps, hipStream_t st) {
  using GEO = SrcNmfGeom<R, M, KR, PASS>;
  if (GEO::lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(src_nmf_kernel<R, M, KR, PASS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEO::lds_bytes);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipFuncSetAttribute(src_nmf_kernel)");
  }
  const Dims d{B, F, T, K};
  hipLaunchKernelGGL((src_nmf_kernel<R, M, KR, PASS>), dim3(fp.G), dim3(WAVE * M), GEO::lds_bytes, st, (const Cx<R>*)X,
                     (const Cx<R>*)W, (const R*)Tb, (const R*)V, (R*)rec, d, fp, (R)eps);
  ASSX_LAUNCH_CHECK(ctx, "src_nmf_kernel");
  return 0;
}

// One source-model update (basis, then activation) of all N = M sources.  rec / sums: scratch of plan.rec_bytes /
// plan.sums_bytes.  Requires src_nmf_ok().
template <typename R, int M, int KR>
int src_nmf_update_kr(assx_ctx* ctx, const void* X, const void* W, void* Tb, void* V, double eps, void* rec, void* sums,
                      const SrcNmfPlan& p, int B, int F, int T, int K, int dtype, hipStream_t st) {
  if constexpr (!SrcNmfGeom<R, M, KR, SRC_NMF_BASIS>::FITS || !SrcNmfGeom<R, M, KR, SRC_NMF_ACT>::FITS) {
    return SRC_NMF_NO_FIT;
  } else {
  int rc = src_nmf_launch_pass<R, M, KR, SRC_NMF_BASIS>(ctx, X, W, Tb, V, rec, p.fb, B, F, T, K, eps, st);
  if (rc) return rc;
  hipLaunchKernelGGL((src_nmf_basis_sums_kernel<R, KR>), dim3((unsigned)(((size_t)B * M * F * K + 255) / 256)), dim3(256), 0,
                     st, (const R*)rec, (R*)sums, B, M, F, K, p.fb);
  ASSX_LAUNCH_CHECK(ctx, "src_nmf_basis_sums_kernel");
  if ((rc = assx_nmf_apply_sums(ctx, ASSX_NMF_IS_MM, 2.0, eps, Tb, sums, B * M, F * K, dtype, st))) return rc;
  if ((rc = src_nmf_launch_pass<R, M, KR, SRC_NMF_ACT>(ctx, X, W, Tb, V, rec, p.fa, B, F, T, K, eps, st))) return rc;  // the new basis
  hipLaunchKernelGGL((src_nmf_act_sums_kernel<R, KR>), dim3((unsigned)(((size_t)B * M * K * T + 255) / 256)), dim3(256), 0, st,
                     (const R*)rec, (R*)sums, B, M, K, T, p.fa);
  ASSX_LAUNCH_CHECK(ctx, "src_nmf_act_sums_kernel");
  return assx_nmf_apply_sums(ctx, ASSX_NMF_IS_MM, 2.0, eps, V, sums, B * M, K * T, dtype, st);
  }
}
// dispatch on the row count; SRC_NMF_NO_FIT: nothing was launched
template <typename R, int M>
int src_nmf_update(assx_ctx* ctx, const void* X, const void* W, void* Tb, void* V, double eps, void* rec, void* sums,
                   const SrcNmfPlan& p, int B, int F, int T, int K, int dtype, hipStream_t st) {
  switch (src_nmf_kr(K)) {
    case 4: return src_nmf_update_kr<R, M, 4>(ctx, X, W, Tb, V, eps, rec, sums, p, B, F, T, K, dtype, st);
    case 8: return src_nmf_update_kr<R, M, 8>(ctx, X, W, Tb, V, eps, rec, sums, p, B, F, T, K, dtype, st);
    case 10: return src_nmf_update_kr<R, M, 10>(ctx, X, W, Tb, V, eps, rec, sums, p, B, F, T, K, dtype, st);
    case 12: return src_nmf_update_kr<R, M, 12>(ctx, X, W, Tb, V, eps, rec, sums, p, B, F, T, K, dtype, st);
    default: return src_nmf_update_kr<R, M, 16>(ctx, X, W, Tb, V, eps, rec, sums, p, B, F, T, K, dtype, st);
  }
}

}  // namespace assx
