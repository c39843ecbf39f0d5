// Wide-channel (5 <= M <= 8) weighted covariance, streaming:  U[b,n,f] = (1/T) sum_t x x^H / max(r_n(f,t), eps)
// (ref: src/bss/ilrma.py:497-511, src/bss/iva.py:489-499, 722-732).
//
// Round 2's cov_bin_kernel gave every bin to one workgroup (1025 bins on 256 CUs = 4.004 rounds, i.e. five), read the
// source variance from a materialised (N,F,T) map -- a 268 MB write + read per iteration at M = 8 -- and had every wave
// fetch the bin's rows from L1/L2 itself behind one block of register prefetch: 395 us + 100 us for the map, 0.23 of
// the HBM peak for the iteration.  Here:
//   * one WAVE PER SOURCE, M waves per workgroup: a source's packed Hermitian sums (M*M reals per lane) are all a wave's
//     registers hold in float64, and every wave needs the same rows of X;
//   * work is the flat balanced partition of the streaming kernels over (utterance, bin, 64-frame block) items
//     (assx_stream.hpp: FlatPart): no quantisation against the CU count, a record per (workgroup, bin crossed, source),
//     reduced to dense U by a small finalize in fixed order (deterministic, batch-invariant);
//   * EVERYTHING an item needs rides one LDS ring, DXS items deep, filled by LDS-direct loads (no registers): the M
//     rows of X (wave n carries row n: HBM is read once per workgroup, the waves read their frames back with M ds_read),
//     and -- wave-private -- the weight inputs of the wave's source: for the ILRMA form with n_basis <= 4 the four
//     activation rows and the bin's basis row (the weight is rebuilt in the kernel: r = sum_k Tb[n,f,k] V[n,k,t], k
//     ascending, fused multiply-adds, floor, 1/x exactly as the map + reader pair did; no variance map), otherwise one row
//     of the caller's weights ((N,T) for AuxIVA; (N,F,T) for t-ILRMA's Xi, IDLMA, FastMNMF, ILRMA with n_basis > 4 or
//     domain != 2, whose variance map costs half of X's bytes -- re-reading n_basis activation rows per 64 frames of ONE
//     bin would cost more from L2 than that from HBM).
// Why the weight inputs go through the ring as well: VMEM returns in order.  A weight input requested a trip ahead is
// the YOUNGEST entry of the queue when it is needed, so waiting for it drains every X request behind which it queued:
// the first version of this kernel (three X slots, weight inputs as register loads one item ahead) never had more than
// one item of X per workgroup in flight and ran at 2.3-2.6 TB/s.  Requested together, DXS - 1 items ahead, all of an
// item's inputs have the same age, and one counted wait per trip -- vmcnt((DXS - 2) * C), C = the instructions of one
// item's request, the same for every wave -- leaves DXS - 2 items in flight behind the one the next trip reads.
// Trip `it` runs the arithmetic of item `it` from REGISTERS (x, wgt: read / formed a trip earlier) and threads through it
// the LDS reads of item it+1 and its weight chain (a dozen dependent instructions), then requests item it+DXS into the
// slot item `it` has left.  All loads in the loop are LDS-direct or inline asm, the LDS reads are inline asm (the compiler
// would otherwise drain vmcnt before any LDS read that might alias an LDS-direct load), every wait is explicit.
#pragma once
#include "assx_stream.hpp"

namespace assx {
namespace widem {

constexpr int SRC_COV_KMAX = 4;  // largest n_basis whose variance is rebuilt in the kernel

template <typename R, int M, int WK, int DX = 0>
struct SrcCovGeom {
  static constexpr int RB = WAVE * 2 * (int)sizeof(R);     // bytes of one row of X: 64 complex frames
  static constexpr int LPR = RB / 16;                      // lanes that carry an X row (f64: 64, f32: the lower 32)
  static constexpr int RBV = WAVE * (int)sizeof(R);        // bytes of one weight-input row: 64 reals
  static constexpr int VR = WK == WK_TV ? SRC_COV_KMAX : 1;  // weight-input rows per wave and item
  static constexpr int VBYTES = VR * RBV;
  static constexpr int NVI = (VBYTES + 1023) / 1024;       // instructions that carry them (f64 ILRMA form: 2, else 1)
  static constexpr int VLANES = VBYTES >= 1024 ? WAVE : VBYTES / 16;  // active lanes of such an instruction
#ifndef ASSX_COV_TLANES
#define ASSX_COV_TLANES 8
#endif
  static constexpr int TLANES = ASSX_COV_TLANES;           // lanes that fetch the basis row: 32 bytes = n_basis <= 4 reals
  static constexpr int TBYTES = WK == WK_TV ? 4 * TLANES : 0;  // its landing area (one dword per lane)
  static constexpr int C = 1 + NVI + (WK == WK_TV ? 1 : 0);  // VMEM instructions of one item's request, per wave
  static constexpr int WSLOT = VBYTES + TBYTES;            // wave-private bytes per slot
  static constexpr int SLOT = M * RB + M * WSLOT;          // bytes per ring slot
#ifndef ASSX_COV_DXS
#define ASSX_COV_DXS 0  // A/B builds: ring depth for M >= 7
#endif
  static constexpr int XBYTES = 2 * M * WAVE * (int)sizeof(R);  // pair_cov_kernel's weight exchange
  static constexpr int FIT = (160 * 1024 - XBYTES) / SLOT;
  static constexpr int DXS = DX ? DX : (M <= 6 ? 4 : (ASSX_COV_DXS ? ASSX_COV_DXS : (FIT < 6 ? FIT : 6)));  // slots (M = 8, f64, ILRMA form: 6 x 24.3 KB of the CU's 160 KB)
  static_assert(DXS >= 3 && DXS <= FIT, "ring does not fit the CU's LDS");
  static constexpr size_t lds_bytes = (size_t)DXS * SLOT;
};

// ---- LDS reads (inline asm: see the header).  Values are readable after the matching lds_wait<N>().
template <int OFF>
__device__ __forceinline__ void lds_read_cx(unsigned addr, Vec2<double>& x) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_cx(unsigned addr, Vec2<float>& x) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(OFF) : "memory");
}
// rows 2p and 2p+1 of the wave's weight-input tile (rows are 64 reals apart)
template <int P>
__device__ __forceinline__ void lds_read_rows2(unsigned addr, Vec2<double>& v) {
  asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(2 * P), "n"(2 * P + 1) : "memory");
}
template <int P>
__device__ __forceinline__ void lds_read_rows2(unsigned addr, Vec2<float>& v) {
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(2 * P), "n"(2 * P + 1) : "memory");
}
__device__ __forceinline__ void lds_read_real(unsigned addr, double& v) {
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
}
__device__ __forceinline__ void lds_read_real(unsigned addr, float& v) {
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
}
// the first four reals of the basis row (every lane reads the same address: a broadcast), as the RAW registers the reads
// fill: nothing may touch them before the matching lds_wait -- not even a copy.  (Until the end of round 3 this helper
// unpacked into a scalar array right after the asm; where the register allocator could not coalesce that, the unpacking
// became v_mov_b64 instructions executed BEFORE the wait -- copies of registers the LDS had not written yet.  They
// sat ~100 cycles after the reads, so the data was usually there: every test passed until two workgroups shared a CU at
// M = 5 in float64 and the LDS got slower, see launch_src_cov_as in csrc/assx_widem.hip.)
template <typename R>
struct Row4Raw;
template <>
struct Row4Raw<double> {
  Vec2<double> a, b;
};
template <>
struct Row4Raw<float> {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 q;
};
__device__ __forceinline__ void lds_read_row4(unsigned addr, Row4Raw<double>& t) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(t.a), "=&v"(t.b) : "v"(addr) : "memory");
}
__device__ __forceinline__ void lds_read_row4(unsigned addr, Row4Raw<float>& t) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(t.q) : "v"(addr) : "memory");
}
__device__ __forceinline__ void row4_fence(Row4Raw<double>& t) { asm volatile("" : "+v"(t.a), "+v"(t.b)); }
__device__ __forceinline__ void row4_fence(Row4Raw<float>& t) { asm volatile("" : "+v"(t.q)); }
template <int I>
__device__ __forceinline__ double row4_get(const Row4Raw<double>& t) {
  return I == 0 ? t.a.x : (I == 1 ? t.a.y : (I == 2 ? t.b.x : t.b.y));
}
template <int I>
__device__ __forceinline__ float row4_get(const Row4Raw<float>& t) {
  return I == 0 ? t.q.x : (I == 1 ? t.q.y : (I == 2 ? t.q.z : t.q.w));
}
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// WK: WK_TV (Tb (B,N,F,K), V (B,N,K,T); n_basis <= SRC_COV_KMAX, domain 2), WK_NT (V = r (B,N,T)), WK_NFT (V = r (B,N,F,T))
template <typename R, int M, int WK>
__global__ void __launch_bounds__(WAVE * M)
    src_cov_kernel(const Cx<R>* __restrict__ X, const R* __restrict__ Tb, const R* __restrict__ V, R* __restrict__ part,
                   Dims d, FlatPart fp, R eps) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int N = M, HM = M * M, NV = next_pow2_c(HM);
  using GEO = SrcCovGeom<R, M, WK>;
  constexpr int RB = GEO::RB, DXS = GEO::DXS;
  constexpr unsigned SLOT = (unsigned)GEO::SLOT;
  const int F = d.F, T = d.T, K = WK == WK_TV ? d.K : 1, TBk = fp.len;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int n = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's source
  const int g = (int)blockIdx.x;
  int b0, f0, tb0, nblk;
  if (!flat_start(fp, g, b0, f0, tb0, nblk)) return;  // whole workgroup: the range is wave-uniform
  const size_t FT = (size_t)F * T;
  // slot layout: [M rows of X][wave 0: weight-input rows, basis row][wave 1: ...]
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned wpriv = (unsigned)M * RB + (unsigned)n * GEO::WSLOT;  // this wave's private part of a slot

  // ---- requests.  The range never leaves utterance b0; offsets inside an utterance's arrays are 32-bit (host check).
  const BufRsrc rx = make_rsrc_sized(X + (size_t)b0 * M * FT, (size_t)M * FT * sizeof(Cx<R>));
  const unsigned xlane = (unsigned)((size_t)n * FT * sizeof(Cx<R>)) + (unsigned)lane * 16u;  // wave n carries row n
  const size_t vrows = WK == WK_TV ? (size_t)N * K : (WK == WK_NT ? (size_t)N : (size_t)N * F);
  const BufRsrc rvb = make_rsrc_sized(V + (size_t)b0 * vrows * T, vrows * T * sizeof(R));
  // lane -> (row, 16-byte piece) of a weight-input instruction; rows past n_basis repeat the last one (their basis is 0)
  constexpr int LPV = GEO::RBV / 16;  // lanes per weight-input row
  unsigned vlane[GEO::NVI];
#pragma unroll
  for (int j = 0; j < GEO::NVI; ++j) {
    const int row = min(j * (WAVE / LPV) + lane / LPV, K - 1);
    vlane[j] = (unsigned)((size_t)row * T * sizeof(R)) + (unsigned)(lane % LPV) * 16u;
  }
  buf_u4 rt = make_rsrc_words(Tb + (size_t)b0 * N * F * K, (size_t)N * F * K * sizeof(R));
  rt.x = __builtin_amdgcn_readfirstlane(rt.x);
  rt.y = __builtin_amdgcn_readfirstlane(rt.y);
  rt.z = __builtin_amdgcn_readfirstlane(rt.z);
  rt.w = __builtin_amdgcn_readfirstlane(rt.w);
  // one item's inputs into ring slot sl: C instructions, the same for every wave.  Frames past T read into the next row
  // (zeros past the end of the array): their weight is 0.
  auto request = [&](const Cursor& c, int sl) {
    const unsigned sbase = (unsigned)sl * SLOT;
    if (WK == WK_TV) {  // basis row: lane L's dword lands at +4L (only the first n_basis reals are used)
      const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((((size_t)n * F + c.f) * K) * sizeof(R)));
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + sbase + wpriv + GEO::VBYTES));
      if (lane < GEO::TLANES) buf_dword_to_lds(dst, rt, (unsigned)lane * 4u, soff);
    }
    const size_t vrow = WK == WK_TV ? (size_t)n * K : (WK == WK_NT ? (size_t)n : (size_t)n * F + c.f);
    const unsigned vsoff = (unsigned)((vrow * T + (size_t)c.tb * WAVE) * sizeof(R));
#pragma unroll
    for (int j = 0; j < GEO::NVI; ++j)
      if (GEO::VLANES == WAVE || lane < GEO::VLANES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rvb, (__attribute__((address_space(3))) void*)(smem + sbase + wpriv + (unsigned)j * 1024u), 16, (int)vlane[j],
            (int)vsoff, 0, 0);
    const unsigned xoff = xlane + (unsigned)(((size_t)c.f * T + (size_t)c.tb * WAVE) * sizeof(Cx<R>));
    if (GEO::LPR == WAVE || lane < GEO::LPR)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rx, (__attribute__((address_space(3))) void*)(smem + sbase + (unsigned)n * RB), 16, (int)xoff, 0, 0, 0);
  };
  // ---- LDS reads of a landed item: weight inputs first (their wait leaves the M X reads in flight), then X
  struct WIn {
    Vec2<R> v01, v23;  // ILRMA form: activation rows 0..3 at this lane's frame ...
    Row4Raw<R> t;      // ... and the basis row (broadcast), raw
    R r;               // weights given: this lane's value
  };
  auto read_w = [&](int sl, WIn& w) {
    const unsigned base = lds0 + (unsigned)sl * SLOT + wpriv;
    if (WK == WK_TV) {
      const unsigned va = base + (unsigned)lane * (unsigned)sizeof(R);
      lds_read_rows2<0>(va, w.v01);
      lds_read_rows2<1>(va, w.v23);
      lds_read_row4(base + GEO::VBYTES, w.t);
    } else {
      lds_read_real(base + (unsigned)lane * (unsigned)sizeof(R), w.r);
    }
  };
  auto read_x = [&](int sl, Vec2<R> (&xr)[M]) {
    const unsigned xa = lds0 + (unsigned)sl * SLOT + (unsigned)lane * (unsigned)sizeof(Cx<R>);
    static_for<M>([&](auto mc) { lds_read_cx<decltype(mc)::value * RB>(xa, xr[decltype(mc)::value]); });
  };
  auto fence_w = [&](WIn& w) {
    if (WK == WK_TV) {
      asm volatile("" : "+v"(w.v01), "+v"(w.v23));
      row4_fence(w.t);
    }
    else asm volatile("" : "+v"(w.r));
  };
  auto weight = [&](const Cursor& c, const WIn& w) -> R {  // 1 / max(r, eps) of this lane's frame, 0 past T
    R tv;
    if (WK == WK_TV) {  // k ascending from 0 (variance_map_kernel's order); entries past n_basis contribute 0 * finite
      tv = fma(row4_get<0>(w.t), w.v01.x, (R)0);
      tv = fma(K > 1 ? row4_get<1>(w.t) : (R)0, w.v01.y, tv);
      tv = fma(K > 2 ? row4_get<2>(w.t) : (R)0, w.v23.x, tv);
      tv = fma(K > 3 ? row4_get<3>(w.t) : (R)0, w.v23.y, tv);
    } else {
      tv = w.r;
    }
    const R rr = floor_eps<R>(tv, eps);
    return c.tb * WAVE + lane < T ? fast_rcp(rr) : (R)0;
  };

  R acc[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) acc[q] = 0;
  Cursor c0;  // item `it`
  c0.b = b0;
  c0.f = f0;
  c0.tb = tb0;
  // prologue: items 0 .. DXS-1 requested (past the end of the range: item 0 again -- the queue keeps its shape)
  Cursor cr = c0;  // item it + DXS, the next one to request
  {
    const Cursor first = c0;
#pragma unroll
    for (int i = 0; i < DXS; ++i) {
      request(i < nblk ? cr : first, i);
      advance(cr, TBk, F);
    }
  }
  Cursor c1 = c0;
  advance(c1, TBk, F);
  wait_vmcnt<(DXS - 2) * GEO::C>();  // items 0 and 1 have landed (this wave's share: the barriers publish the rest)
  asm volatile("s_barrier" ::: "memory");
  Vec2<R> x[M];
  R wgt;
  {
    WIn w;
    read_w(0, w);
    read_x(0, x);
    lds_wait<0>();
    fence_w(w);
#pragma unroll
    for (int m = 0; m < M; ++m) asm volatile("" : "+v"(x[m]));
    wgt = weight(c0, w);
  }
  int sl = 0;  // ring slot of item `it`
  for (int it = 0; it < nblk; ++it) {
    const bool more = it + 1 < nblk;
    // every wave's row of item it+1 has landed (each waited for its own at the end of the previous trip) and every wave
    // has item `it` in registers.  A bare barrier: only LDS traffic crosses waves, and __syncthreads() would drain the
    // requests that have to stay in flight.
    asm volatile("s_barrier" ::: "memory");
    const int sl1 = sl + 1 == DXS ? 0 : sl + 1;
    WIn wn;
    Vec2<R> xn[M];
    read_w(sl1, wn);
    read_x(sl1, xn);
    request(it + DXS < nblk ? cr : c0, sl);  // past the end: the current item again (an L2 hit nobody reads)
    lds_wait<M>();  // LDS returns in order: the weight inputs are here, the X reads may still travel
    fence_w(wn);
    R wgt_n = weight(more ? c1 : c0, wn);
    // x_i conj(x_j) for j >= i, weighted: diagonal terms in acc[i], pairs (re, im) at herm_pair_base
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const R sx = wgt * x[i].x, sy = wgt * x[i].y;
      acc[i] = fma(sx, x[i].x, acc[i]);
      acc[i] = fma(sy, x[i].y, acc[i]);
#pragma unroll
      for (int j = i + 1; j < M; ++j) {
        const int hb = herm_pair_base<M>(i, j);  // compile-time after unrolling
        acc[hb] = fma(sx, x[j].x, acc[hb]);
        acc[hb] = fma(sy, x[j].y, acc[hb]);
        acc[hb + 1] = fma(sy, x[j].x, acc[hb + 1]);
        acc[hb + 1] = fma(-sx, x[j].y, acc[hb + 1]);
      }
    }
#pragma unroll
    for (int q = 0; q < HM; ++q) asm volatile("" : "+v"(acc[q]));  // the arithmetic stays HERE, above the waits (left alone
                                                                   // it is sunk below them; the padding of acc stays foldable) ...
    asm volatile("" : "+v"(wgt_n));  // ... and so does the weight chain of the next item
    lds_wait<0>();
#pragma unroll
    for (int m = 0; m < M; ++m) asm volatile("" : "+v"(xn[m]));
    // this wave's share of item it+2 has landed once only the requests of items it+3 .. it+DXS are in flight
    wait_vmcnt<(DXS - 2) * GEO::C>();
    if (c1.tb == 0 || !more) {  // the bin is complete (or the range ends): flush
      const R tot = wave_reduce_scatter<R, NV>(acc);
      const int i = scatter_index<NV>();
      const int slot = c0.f - f0;
      if (scatter_leader<NV>() && i < HM) part[(((size_t)g * fp.S + slot) * N + n) * HM + i] = tot;
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = 0;
    }
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = xn[m];
    wgt = wgt_n;
    sl = sl1;
    c0 = c1;
    advance(c1, TBk, F);
    advance(cr, TBk, F);
  }
  wait_vmcnt<0>();  // nothing may land in LDS after the workgroup has gone
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// The same pass with the PAIRS of the Hermitian matrix split over the waves instead of the sources (round 3, second
// form).  src_cov_kernel's wave forms all M (M + 1) / 2 products of its source itself: 16 + 16 + 112 f64 instructions per
// item at M = 8, i.e. every wave repeats x_i conj(x_j) and only the weight differs -- the pass is bound by the f64 vector
// rate (profiles/r03_sq_src_cov_m8.txt: 198 VALU per wave and item, 0.64 of them issuing).  Here wave w owns M reals of
// the packed matrix (a few pairs (i, j), i <= j, chosen so that it needs few rows of X: pair_list) for ALL sources:
//   p = x_i conj(x_j)   once per item and pair       (2 instructions on the diagonal, 4 off it)
//   acc[n][p] += weight_n p                          (1 / 2 fused multiply-adds per source)
// = 16 + 64 instructions at M = 8 instead of 144.  The weights travel through LDS: wave n still carries row n of X and the
// weight inputs of source n on the ring, forms weight_n of item it+1 during trip `it` (as before) and publishes its 64
// values in wbuf[(it+1) & 1][n]; the barrier that opens trip it+1 makes all N of them readable by every wave.  Records,
// partition and finalize kernel are src_cov_kernel's (acc index n*M + r -> packed index by pair_real_index at the flush).
// ---------------------------------------------------------------------------------------------------------------
struct PairList {
  int n;         // pairs of this wave
  int i[8], j[8];  // i <= j; the wave's reals follow in this order, 1 per diagonal entry, (re, im) per pair off it
};
template <int M>
__host__ __device__ constexpr PairList pair_list(int w) {
  PairList L{};
  auto put = [&](int a, int b) {
    L.i[L.n] = a < b ? a : b;
    L.j[L.n] = a < b ? b : a;
    ++L.n;
  };
  if (M % 2 == 1) {  // cyclic: (w, w), (w, w+1), ..., (w, w + (M-1)/2): every pair once, (M + 1) / 2 consecutive rows
    for (int dlt = 0; dlt <= (M - 1) / 2; ++dlt) put(w, (w + dlt) % M);
  } else if (M == 6) {  // five waves with three pairs off the diagonal, one with the diagonal
    if (w == 0) put(0, 1), put(0, 2), put(1, 2);
    else if (w == 1) put(3, 4), put(3, 5), put(4, 5);
    else if (w < 5) put(w - 2, 3), put(w - 2, 4), put(w - 2, 5);
    else for (int m = 0; m < 6; ++m) put(m, m);
  } else {  // M == 8: 2 x 2 blocks of rows {0,1} {2,3} {4,5} {6,7}; the diagonal blocks two by two
    constexpr int ba[6] = {0, 0, 0, 1, 1, 2}, bb[6] = {1, 2, 3, 2, 3, 3};
    if (w < 6) {
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) put(2 * ba[w] + a, 2 * bb[w] + b);
    } else {
      for (int h = 0; h < 2; ++h) {
        const int r0 = (w - 6) * 4 + 2 * h;
        put(r0, r0), put(r0 + 1, r0 + 1), put(r0, r0 + 1);
      }
    }
  }
  return L;
}
template <int M>
constexpr bool pair_lists_cover() {  // every (i, j), i <= j, owned by exactly one wave; M reals per wave
  int seen[8][8] = {};
  for (int w = 0; w < M; ++w) {
    const PairList L = pair_list<M>(w);
    int reals = 0;
    for (int q = 0; q < L.n; ++q) {
      if (L.i[q] > L.j[q] || L.j[q] >= M) return false;
      ++seen[L.i[q]][L.j[q]];
      reals += L.i[q] == L.j[q] ? 1 : 2;
    }
    if (reals != M) return false;
  }
  for (int i = 0; i < M; ++i)
    for (int j = i; j < M; ++j)
      if (seen[i][j] != 1) return false;
  return true;
}
static_assert(pair_lists_cover<5>() && pair_lists_cover<6>() && pair_lists_cover<7>() && pair_lists_cover<8>(),
              "pair_list must partition the upper triangle");

template <int M, int W>
struct WavePairs {
  static constexpr PairList L = pair_list<M>(W);
  static constexpr int real_offset(int s) {  // first real of pair s
    int o = 0;
    for (int q = 0; q < s; ++q) o += L.i[q] == L.j[q] ? 1 : 2;
    return o;
  }
  static constexpr bool needs_row(int m) {
    for (int q = 0; q < L.n; ++q)
      if (L.i[q] == m || L.j[q] == m) return true;
    return false;
  }
  static constexpr int rows() {
    int c = 0;
    for (int m = 0; m < M; ++m) c += needs_row(m) ? 1 : 0;
    return c;
  }
  static constexpr int nth_row(int k) {  // the k-th row (ascending) this wave reads
    for (int m = 0; m < M; ++m)
      if (needs_row(m) && k-- == 0) return m;
    return 0;
  }
  static constexpr int packed_index(int r) {  // real r of this wave -> index in the packed Hermitian record
    for (int q = 0; q < L.n; ++q) {
      const int o = real_offset(q);
      if (L.i[q] == L.j[q]) {
        if (r == o) return L.i[q];
      } else if (r == o || r == o + 1) {
        return herm_pair_base<M>(L.i[q], L.j[q]) + (r - o);
      }
    }
    return 0;
  }
};

__device__ __forceinline__ void lds_write_real(unsigned addr, double v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_real(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <typename WP, int M, typename V, int I = 0>
__device__ __forceinline__ void fence_rows(V (&xr)[M]) {  // value fence on the rows of X a wave reads
  if constexpr (I < M) {
    if constexpr (WP::needs_row(I)) asm volatile("" : "+v"(xr[I]));
    fence_rows<WP, M, V, I + 1>(xr);
  }
}

#if defined(PAIRCOV_TRACE) && PAIRCOV_TRACE && !defined(ASSX_PROBE_BUILD)
#error "PAIRCOV_TRACE adds a debug entry point and stamps: build it with -DASSX_PROBE_BUILD into a probe library, never into libassx.so"
#endif
#ifndef PAIRCOV_TRACE
#define PAIRCOV_TRACE 0
#endif
// PAIRCOV_SKIP (probe builds only; results are WRONG, the time is what is read -- tools/probes/paircov_knockout.sh,
// profiles/r04_paircov_knockout.txt): 1 no fan-out, 2 no requests inside the loop, 4 no LDS reads of X rows / published
// weights, 8 no barrier, 16 no weight chain at all (its LDS reads, its arithmetic, the publication); parts of it alone: 32 the
// arithmetic, 64 the LDS reads of the weight inputs, 128 the publication, 256 the wait in front of the chain
#if defined(PAIRCOV_SKIP) && PAIRCOV_SKIP && !defined(ASSX_PROBE_BUILD)
#error "PAIRCOV_SKIP removes parts of pair_cov_kernel: build it with -DASSX_PROBE_BUILD into a probe library, never into libassx.so"
#endif
#ifndef PAIRCOV_SKIP
#define PAIRCOV_SKIP 0
#endif
#if PAIRCOV_TRACE
__device__ unsigned long long g_paircov_trace[1400];  // timing-experiment builds only
#endif

#ifndef ASSX_PAIR_DXS_LE6
#define ASSX_PAIR_DXS_LE6 4  // ring depth of pair_cov_kernel for M <= 6 (two workgroups per CU at M = 5; 5 and 6 measured: slower at M = 5, the same at M = 6)
#endif
// Ring geometry of pair_cov_kernel.  A slot is M blocks of WBLK bytes, one per wave: [row of X][weight-input rows][basis row]
// -- everything wave n requests for an item is CONTIGUOUS, so that all LDS-direct loads of a request use ONE value of M0
// and select their landing place with the instruction's immediate offset (which is added to the LDS address AND to the
// buffer offset: each load goes through a descriptor whose base is moved back by its immediate).  This layout was built
// while hunting a failure of <double, 5, WK_TV> with two workgroups on a CU, whose cause turned out to be elsewhere
// (lds_read_row4 above: registers copied before their wait); it is kept: one M0 write per request instead of four,
// nothing else changed in time or results.
template <typename R, int M, int WK>
struct PairCovGeom {
  using G0 = SrcCovGeom<R, M, WK>;
  static constexpr int RB = G0::RB, LPR = G0::LPR, RBV = G0::RBV, VBYTES = G0::VBYTES, NVI = G0::NVI, VLANES = G0::VLANES;
  static constexpr int TBYTES = WK == WK_TV ? 32 : 0;  // n_basis <= 4 reals: two lanes of a dwordx4 load
  static constexpr int C = G0::C;
  static constexpr int OFF_V = RB, OFF_T = RB + VBYTES;  // immediates of the weight-input and basis-row loads
  static constexpr int WBLK = RB + VBYTES + TBYTES;
  static_assert(OFF_T + 32 <= 4096, "immediate offsets are 12 bits");
  static constexpr int SLOT = M * WBLK;
  static constexpr int XBYTES = 2 * M * WAVE * (int)sizeof(R);  // the weight exchange
  static constexpr int FIT = (160 * 1024 - XBYTES) / SLOT;
  static constexpr int DXS = M <= 6 ? ASSX_PAIR_DXS_LE6 : (ASSX_COV_DXS ? ASSX_COV_DXS : (FIT < 6 ? FIT : 6));
  static_assert(DXS >= 4 && DXS <= FIT, "ring does not fit the CU's LDS");
  static constexpr size_t lds_bytes = (size_t)DXS * SLOT;
};
template <typename R, int M, int WK>
constexpr size_t pair_cov_lds_bytes() {
  return PairCovGeom<R, M, WK>::lds_bytes + (size_t)PairCovGeom<R, M, WK>::XBYTES;
}

template <typename R, int M, int WK>
__global__ void __launch_bounds__(WAVE * M)
    pair_cov_kernel(const Cx<R>* __restrict__ X, const R* __restrict__ Tb, const R* __restrict__ V, R* __restrict__ part,
                    Dims d, FlatPart fp, R eps) {
#if defined(__HIP_DEVICE_COMPILE__)
  [[maybe_unused]] int trace_n = 0;
  auto stamp = [&]() {  // -DPAIRCOV_TRACE=1: shader-clock stamps of workgroup 100, waves 0 and 5 (tools/probes/paircov_trace.py)
#if PAIRCOV_TRACE
    if (blockIdx.x == 100 && (threadIdx.x == 0 || threadIdx.x == 320) && trace_n < 700)
      g_paircov_trace[(threadIdx.x ? 700 : 0) + trace_n] = __builtin_amdgcn_s_memtime();
#endif
    ++trace_n;
  };
  constexpr int N = M, HM = M * M, NV = next_pow2_c(HM);
  using GEO = PairCovGeom<R, M, WK>;
  constexpr int DXS = GEO::DXS;
  constexpr unsigned SLOT = (unsigned)GEO::SLOT;
  constexpr unsigned WROW = WAVE * (unsigned)sizeof(R);                  // one source's published weights
  const int F = d.F, T = d.T, K = WK == WK_TV ? d.K : 1, TBk = fp.len;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int n = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave: row n of X, weight of source n, pairs of pair_list(n)
  const int g = (int)blockIdx.x;
  int b0, f0, tb0, nblk;
  if (!flat_start(fp, g, b0, f0, tb0, nblk)) return;
  const size_t FT = (size_t)F * T;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  constexpr unsigned WBLK = (unsigned)GEO::WBLK;
  const unsigned wblk = (unsigned)n * WBLK;           // this wave's block of a slot
  const unsigned wbuf = lds0 + (unsigned)DXS * SLOT;  // [2][N][WAVE] reals

  const BufRsrc rx = make_rsrc_sized(X + (size_t)b0 * M * FT, (size_t)M * FT * sizeof(Cx<R>));
  const unsigned xlane = (unsigned)((size_t)n * FT * sizeof(Cx<R>)) + (unsigned)lane * 16u;
  const size_t vrows = WK == WK_TV ? (size_t)N * K : (WK == WK_NT ? (size_t)N : (size_t)N * F);
  // descriptors moved back by the immediate offset their load carries (the bytes in front of the arrays are never
  // addressed: every offset is >= the immediate)
  const char* vbase = reinterpret_cast<const char*>(V + (size_t)b0 * vrows * T);
  BufRsrc rvb[GEO::NVI];
#pragma unroll
  for (int j = 0; j < GEO::NVI; ++j)
    rvb[j] = make_rsrc_sized(vbase - (GEO::OFF_V + j * 1024), vrows * T * sizeof(R) + (size_t)(GEO::OFF_V + j * 1024));
  constexpr int LPV = GEO::RBV / 16;
  unsigned vlane[GEO::NVI];
#pragma unroll
  for (int j = 0; j < GEO::NVI; ++j) {
    const int row = min(j * (WAVE / LPV) + lane / LPV, K - 1);
    vlane[j] = (unsigned)((size_t)row * T * sizeof(R)) + (unsigned)(lane % LPV) * 16u;
  }
  const BufRsrc rtb = make_rsrc_sized(reinterpret_cast<const char*>(Tb + (size_t)b0 * N * F * K) - GEO::OFF_T,
                                      (size_t)N * F * K * sizeof(R) + (size_t)GEO::OFF_T);
  // one item's inputs into ring slot sl: C instructions, the same for every wave, ONE LDS base (M0) for all of them.
  // Frames past T read into the next row (zeros past the end of the array): their weight is 0.
  auto request = [&](const Cursor& c, int sl) {
    __attribute__((address_space(3))) void* const blk =
        (__attribute__((address_space(3))) void*)(smem + (unsigned)sl * SLOT + wblk);
    if (WK == WK_TV) {  // basis row: 32 bytes = the first n_basis <= 4 reals (and what follows them)
      const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((((size_t)n * F + c.f) * K) * sizeof(R)));
      if (lane < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rtb, blk, 16, (int)((unsigned)lane * 16u), (int)soff, GEO::OFF_T, 0);
    }
    const size_t vrow = WK == WK_TV ? (size_t)n * K : (WK == WK_NT ? (size_t)n : (size_t)n * F + c.f);
    const unsigned vsoff = (unsigned)((vrow * T + (size_t)c.tb * WAVE) * sizeof(R));
    static_for<GEO::NVI>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (GEO::VLANES == WAVE || lane < GEO::VLANES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rvb[j], blk, 16, (int)vlane[j], (int)vsoff, GEO::OFF_V + j * 1024, 0);
    });
    const unsigned xoff = xlane + (unsigned)(((size_t)c.f * T + (size_t)c.tb * WAVE) * sizeof(Cx<R>));
    if (GEO::LPR == WAVE || lane < GEO::LPR) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, blk, 16, (int)xoff, 0, 0, 0);
  };
  struct WIn {
    Vec2<R> v01, v23;
    Row4Raw<R> t;
    R r;
  };
  auto read_w = [&](int sl, WIn& w) {  // WIN_READS instructions
    const unsigned base = lds0 + (unsigned)sl * SLOT + wblk + (unsigned)GEO::OFF_V;
    if (WK == WK_TV) {
      const unsigned va = base + (unsigned)lane * (unsigned)sizeof(R);
      lds_read_rows2<0>(va, w.v01);
      lds_read_rows2<1>(va, w.v23);
      lds_read_row4(base + GEO::VBYTES, w.t);
    } else {
      lds_read_real(base + (unsigned)lane * (unsigned)sizeof(R), w.r);
    }
  };
  auto fence_w = [&](WIn& w) {
    if (WK == WK_TV) {
      asm volatile("" : "+v"(w.v01), "+v"(w.v23));
      row4_fence(w.t);
    }
    else asm volatile("" : "+v"(w.r));
  };
  auto weight = [&](int tb, const WIn& w) -> R {  // src_cov_kernel's chain, bit for bit
    R tv;
    if (WK == WK_TV) {
      tv = fma(row4_get<0>(w.t), w.v01.x, (R)0);
      tv = fma(K > 1 ? row4_get<1>(w.t) : (R)0, w.v01.y, tv);
      tv = fma(K > 2 ? row4_get<2>(w.t) : (R)0, w.v23.x, tv);
      tv = fma(K > 3 ? row4_get<3>(w.t) : (R)0, w.v23.y, tv);
    } else {
      tv = w.r;
    }
    const R rr = floor_eps<R>(tv, eps);
    return tb * WAVE + lane < T ? fast_rcp(rr) : (R)0;
  };
  // the N published weights of one item: sources 2q, 2q+1 per instruction (rows are WAVE reals apart)
  constexpr int WALL_READS = (N + 1) / 2;
  auto read_wall = [&](int par, Vec2<R> (&wl)[WALL_READS]) {
    const unsigned a = wbuf + (unsigned)par * (unsigned)N * WROW + (unsigned)lane * (unsigned)sizeof(R);
    static_for<N / 2>([&](auto qc) { lds_read_rows2<decltype(qc)::value>(a, wl[decltype(qc)::value]); });
  };

  // Pipeline: trip `it` runs the arithmetic of item `it` from registers (x: read during trip it-1; wl: the N weights,
  // read during trip it-1 as well), reads the rows of X and the published weights of item it+1, forms this wave's weight of
  // item it+2 and publishes it in wbuf[it & 1] (whose previous content, the weights of item `it`, every wave holds in
  // registers since the barrier that opened this trip), and requests item it+DXS into the slot item `it` has left.  Nothing
  // a trip reads from LDS is needed before its end: no wait stands between the barrier and the arithmetic.
  Cursor first;
  first.b = b0;
  first.f = f0;
  first.tb = tb0;
  Cursor cr = first;  // item it + DXS, the next one to request
#pragma unroll
  for (int i = 0; i < DXS; ++i) {
    request(i < nblk ? cr : first, i);
    advance(cr, TBk, F);
  }
  int tb1 = tb0 + 1 == TBk ? 0 : tb0 + 1;  // frame block of item it+1 ...
  int tb2 = tb1 + 1 == TBk ? 0 : tb1 + 1;  // ... and of item it+2
  wait_vmcnt<(DXS - 3) * GEO::C>();  // items 0, 1 and 2 have landed (this wave's share: the barriers publish the rest)
#ifdef ASSX_PAIR_WAIT0
  asm volatile("s_sleep 64" ::: "memory");  // experiment: time between the counted wait and the first reads
#endif
  asm volatile("s_barrier" ::: "memory");
  {  // weights of items 0 and 1 -> wbuf[0][n], wbuf[1][n]
    WIn w0, w1;
    read_w(0, w0);
    read_w(1, w1);
    lds_wait<0>();
    fence_w(w0);
    fence_w(w1);
    const unsigned wa = wbuf + (unsigned)n * WROW + (unsigned)lane * (unsigned)sizeof(R);
    lds_write_real(wa, weight(tb0, w0));
    lds_write_real(wa + (unsigned)N * WROW, weight(tb1, w1));
    lds_wait<0>();
  }
  asm volatile("s_barrier" ::: "memory");

  auto run = [&](auto wc) {
    constexpr int W = decltype(wc)::value;
    using WP = WavePairs<M, W>;
    static_assert(WP::real_offset(WP::L.n) == M, "every wave owns M reals");
    constexpr int NXR = WP::rows();
    // what a trip issues besides its arithmetic, placed BETWEEN the slices of the fan-out (one slice = one source): issued
    // together after the barrier, the 8 waves' LDS reads (12 x 1 KB each) filled the LDS queue and every wave stood at its
    // reads until they were accepted (profiles/r03_paircov_trace.txt)
    constexpr int E_WALL = 0, E_WIN = 1, E_REQ = 2, E_X = 3, NE = 3 + NXR;
    auto event_slice = [](int e) constexpr { return e * N / NE; };  // the slice after which event e is issued
    constexpr int CHAIN_AT = N / 2 + 1;  // the slice in front of which the weight chain (item it+2) starts
    constexpr int WRITE_AT = N - 1;      // ... and in front of which its result is published
    auto x_reads_before = [=](int sl_) constexpr {
      int c = 0;
      for (int e = E_X; e < NE; ++e) c += event_slice(e) < sl_ ? 1 : 0;
      return c;
    };
    static_assert(event_slice(E_WIN) < CHAIN_AT && CHAIN_AT < WRITE_AT, "the weight inputs are read before the chain starts");
    auto fence_x = [&](Vec2<R> (&xr)[M]) { fence_rows<WP, M>(xr); };
    // two register sets (rows of X, weights), alternating from trip to trip: the loop is unrolled by two so that no trip
    // ends with copies
    Vec2<R> xs[2][M];
    Vec2<R> wls[2][WALL_READS];
    R wlasts[2] = {0, 0};  // N odd: the last source's weights on their own
    {
      const unsigned xa = lds0 + (unsigned)lane * (unsigned)sizeof(Cx<R>);
      static_for<M>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (WP::needs_row(m)) lds_read_cx<m * (int)WBLK>(xa, xs[0][m]);
      });
      read_wall(0, wls[0]);
      if (N % 2) lds_read_real(wbuf + (unsigned)(N - 1) * WROW + (unsigned)lane * (unsigned)sizeof(R), wlasts[0]);
      lds_wait<0>();
      fence_x(xs[0]);
    }
    R acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = 0;
    int sl = 0, slot = 0, it = 0;
    auto trip = [&](auto hc) {
      constexpr int H = decltype(hc)::value;
      Vec2<R>(&x)[M] = xs[H];
      Vec2<R>(&xn)[M] = xs[1 - H];
      Vec2<R>(&wl)[WALL_READS] = wls[H];
      Vec2<R>(&wln)[WALL_READS] = wls[1 - H];
      R& wlast = wlasts[H];
      R& wlastn = wlasts[1 - H];
      const bool more = it + 1 < nblk;
      stamp();
      if (!(PAIRCOV_SKIP & 8))
      asm volatile("s_barrier" ::: "memory");  // items it+1 (rows of X) and it+2 (weight inputs) have landed for every wave;
                                                // the weights of item it+1 are published
      stamp();
      const int sl1 = sl + 1 == DXS ? 0 : sl + 1;
      const int sl2 = sl1 + 1 == DXS ? 0 : sl1 + 1;
      WIn wn;
      // x_i conj(x_j) of the wave's pairs
      R p[M];
      static_for<WP::L.n>([&](auto sc) {
        constexpr int s = decltype(sc)::value, i = WP::L.i[s], j = WP::L.j[s], o = WP::real_offset(s);
        if constexpr (i == j) {
          p[o] = fma(x[i].y, x[i].y, x[i].x * x[i].x);
        } else {
          p[o] = fma(x[i].y, x[j].y, x[i].x * x[j].x);
          p[o + 1] = fma(-x[i].x, x[j].y, x[i].y * x[j].x);
        }
      });
#pragma unroll
      for (int r = 0; r < M; ++r) asm volatile("" : "+v"(p[r]));
      stamp();
      R wgt_n = 0;
      int tb1n = tb1, tb2n = tb2;
      Cursor crn = cr;
      const unsigned xa1 = lds0 + (unsigned)sl1 * SLOT + (unsigned)lane * (unsigned)sizeof(Cx<R>);
      static_for<N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s == CHAIN_AT && !(PAIRCOV_SKIP & 16)) {
          if (!(PAIRCOV_SKIP & 256)) lds_wait<x_reads_before(CHAIN_AT)>();  // LDS returns in order: the weight inputs (issued before any row of X) are here
          fence_w(wn);
          if (!(PAIRCOV_SKIP & 32)) wgt_n = weight(tb2, wn);
        }
        if constexpr (s == WRITE_AT && !(PAIRCOV_SKIP & (16 | 128))) {
          asm volatile("" : "+v"(wgt_n));
          lds_write_real(wbuf + (unsigned)(it & 1) * (unsigned)N * WROW + (unsigned)n * WROW +
                             (unsigned)lane * (unsigned)sizeof(R),
                         wgt_n);
        }
        const R ws = (N % 2 && s == N - 1) ? wlast : ((s & 1) ? wl[s >> 1].y : wl[s >> 1].x);
        if (!(PAIRCOV_SKIP & 1))
#pragma unroll
        for (int r = 0; r < M; ++r) acc[s * M + r] = fma(ws, p[r], acc[s * M + r]);
#pragma unroll
        for (int r = 0; r < M; ++r) asm volatile("" : "+v"(acc[s * M + r]));
        static_for<NE>([&](auto ec) {
          constexpr int e = decltype(ec)::value;
          if constexpr (event_slice(e) == s) {
            if constexpr (e == E_WALL) {
              if (!(PAIRCOV_SKIP & 4)) {
                read_wall((it + 1) & 1, wln);
                if (N % 2)
                  lds_read_real(wbuf + (unsigned)(((it + 1) & 1) * N + N - 1) * WROW + (unsigned)lane * (unsigned)sizeof(R), wlastn);
              }
            } else if constexpr (e == E_WIN) {
              if (!(PAIRCOV_SKIP & (16 | 64))) read_w(sl2, wn);
            } else if constexpr (e == E_REQ) {
              if (!(PAIRCOV_SKIP & 2)) request(it + DXS < nblk ? cr : first, sl);
              // the cursors of the next trip, here where the scalar unit is idle
              advance(crn, TBk, F);
              tb1n = tb1 + 1 == TBk ? 0 : tb1 + 1;
              tb2n = tb2 + 1 == TBk ? 0 : tb2 + 1;
            } else if (!(PAIRCOV_SKIP & 4)) {
              lds_read_cx<WP::nth_row(e - E_X) * (int)WBLK>(xa1, xn[WP::nth_row(e - E_X)]);
            }
          }
        });
#pragma unroll
        for (int r = 0; r < M; ++r) asm volatile("" : "+v"(p[r]));
      });
      stamp();
      lds_wait<0>();
      fence_x(xn);
#pragma unroll
      for (int q = 0; q < N / 2; ++q) asm volatile("" : "+v"(wln[q]));
      asm volatile("" : "+v"(wlastn));
      stamp();
#ifdef ASSX_PAIR_WAIT0
      wait_vmcnt<0>();
#else
      wait_vmcnt<(DXS - 3) * GEO::C>();  // this wave's share of item it+3 has landed
#endif
      stamp();
      if (tb1 == 0 || !more) {  // the bin is complete (or the range ends): flush
        const R tot = wave_reduce_scatter<R, NV>(acc);
        const int q = scatter_index<NV>();
        if (scatter_leader<NV>() && q < HM) {
          const int s = q / M, r = q - s * M;
          int hm = 0;
          static_for<M>([&](auto rc) {
            if (r == decltype(rc)::value) hm = WP::packed_index(decltype(rc)::value);
          });
          part[(((size_t)g * fp.S + slot) * N + s) * HM + hm] = tot;
        }
        ++slot;
#pragma unroll
        for (int q2 = 0; q2 < NV; ++q2) acc[q2] = 0;
      }
      sl = sl1;
      tb1 = tb1n;
      tb2 = tb2n;
      cr = crn;
      ++it;
    };
    while (it < nblk) {
      trip(IntC<0>());
      if (it >= nblk) break;
      trip(IntC<1>());
    }
  };
  static_for<M>([&](auto wc) {
    if (n == decltype(wc)::value) run(wc);
  });
  wait_vmcnt<0>();  // nothing may land in LDS after the workgroup has gone
#endif
}

// sum the records covering each bin in workgroup order, scale by 1/T, expand packed Hermitian -> dense U (B,N,F,M,M)
template <typename R, int M>
__global__ void __launch_bounds__(256) src_cov_finalize_kernel(const R* __restrict__ part, Cx<R>* __restrict__ U, int B,
                                                              int F, FlatPart fp, R inv_T) {
  constexpr int N = M, HM = M * M;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * F * HM;
  if (idx >= total) return;
  const int l = idx % M, m = (idx / M) % M;
  const int f = (idx / HM) % F;
  const int n = (idx / ((size_t)HM * F)) % N;
  const int b = idx / ((size_t)HM * F * N);
  const long long j = (long long)b * F + f;
  int g_lo, g_hi;
  flat_cover(fp, j, g_lo, g_hi);
  R re = 0, im = 0;
  for (int g = g_lo; g <= g_hi; ++g) {
    const int slot = flat_slot(fp, j, g);
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

}  // namespace widem
}  // namespace assx
