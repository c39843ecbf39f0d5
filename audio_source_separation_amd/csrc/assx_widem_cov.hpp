// Wide-channel (5 <= M <= 8) weighted covariance, streaming:  U[b,n,f] = (1/T) sum_t x x^H / max(r_n(f,t), eps)
// (ref: src/bss/ilrma.py:497-511, src/bss/iva.py:489-499, 722-732).
//
// Round 2's cov_bin_kernel gave every bin to one workgroup (1025 bins on 256 CUs = 4.004 rounds, i.e. five), read the
// source variance from a materialised (N,F,T) map -- a 268 MB write + read per iteration at M = 8 -- and had every wave
// fetch the bin's rows from L1/L2 itself behind one block of register prefetch: 395 us + 100 us for the map, 0.23 of
// the HBM peak for the iteration.  Here:
//   * one WAVE PER SOURCE, M waves per workgroup: a source's packed Hermitian sums (M*M reals per lane) are all a wave's
//     registers hold in float64, and every wave needs the same rows of X;
//   * X rides a three-slot LDS ring filled by LDS-direct loads (one instruction of one wave per row of 64 frames, no
//     registers): the rows of item i+2 are requested while item i is consumed; HBM is read once, the waves read their
//     frames back with M ds_read per item;
//   * the weights are rebuilt in the kernel from (Tb, V) -- r = sum_k Tb[n,f,k] V[n,k,t], k ascending, fused
//     multiply-adds, then ^(2/domain), floor, 1/x exactly as the map + reader pair did -- for n_basis <= 16 (the basis
//     row sits one value per lane and is broadcast with v_readlane; the activation values of item i+1 are requested
//     before the arithmetic of item i and turned into the weight after it).  Larger ranks and the callers that own a
//     weight map (t-ILRMA's Xi, IDLMA, FastMNMF) use the same kernel in its "map given" form, AuxIVA in its (N,T) form;
//   * work is the flat balanced partition of the streaming kernels over (utterance, bin, 64-frame block) items
//     (assx_stream.hpp: FlatPart): no quantisation against the CU count, a record per (workgroup, bin crossed, source),
//     reduced to dense U by a small finalize in fixed order (deterministic, batch-invariant).
// Memory pipeline.  Per trip a wave issues [activation / weight loads of item i+1][its X row of item i+2] and, after the
// arithmetic, waits until at most that one X request is in flight: the weight inputs have landed -- and so has the row
// of item i+1 it requested a trip earlier (VMEM returns in order), which is what the barrier at the top of the next trip
// publishes to the other waves.  The same barrier frees the slot item i-1 was read from for the request of item i+2.
// All loads inside the loop are inline asm or LDS-direct (invisible to the compiler's wait model), the LDS reads are
// inline asm as well (the compiler would otherwise drain vmcnt before any LDS read that might alias an LDS-direct load).
#pragma once
#include "assx_stream.hpp"

namespace assx {
namespace widem {

template <typename R>
struct SrcCovGeom {
  static constexpr int DXS = 3;                         // X ring slots (items)
  static constexpr int RB = WAVE * 2 * (int)sizeof(R);  // bytes of one row of X: 64 complex frames
  static constexpr int LPR = RB / 16;                   // lanes per row of a 16-byte-per-lane LDS-direct load
  static constexpr int RPI = WAVE / LPR;                // rows per instruction (f64: 1, f32: 2)
  static __host__ __device__ constexpr int nxi(int M) { return (M + RPI - 1) / RPI; }  // instructions per item
  static __host__ __device__ constexpr int mp(int M) { return nxi(M) * RPI; }          // rows per slot, padded
  static __host__ __device__ constexpr size_t lds_bytes(int M) { return (size_t)DXS * mp(M) * RB; }
};

constexpr int SRC_COV_KMAX = 16;  // largest n_basis whose variance is rebuilt in the kernel

// this lane's frame of the row at byte offset OFF of a landed item; the value is readable after xrow_wait()
template <int OFF>
__device__ __forceinline__ void xrow_read_at(unsigned addr, Vec2<double>& x) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void xrow_read_at(unsigned addr, Vec2<float>& x) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(OFF) : "memory");
}
template <typename R, int M>
__device__ __forceinline__ void xrow_wait(Vec2<R> (&x)[M]) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int m = 0; m < M; ++m) asm volatile("" : "+v"(x[m]));
}
__device__ __forceinline__ double lane_value(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_value(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// WK: WK_TV (Tb (B,N,F,K), V (B,N,K,T)), WK_NT (V = r (B,N,T)), WK_NFT (V = r (B,N,F,T)).  KC: activation values kept
// per lane (4 or SRC_COV_KMAX; n_basis <= KC).  D2: no power (domain 2, or a weight that is used as given).
template <typename R, int M, int WK, bool D2, int KC>
__global__ void __launch_bounds__(WAVE * M)
    src_cov_kernel(const Cx<R>* __restrict__ X, const R* __restrict__ Tb, const R* __restrict__ V, R* __restrict__ part,
                   Dims d, FlatPart fp, R eps, PowSpec p2d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int N = M, HM = M * M, NV = next_pow2_c(HM);
  using GEO = SrcCovGeom<R>;
  constexpr int NXI = GEO::nxi(M), MP = GEO::mp(M), RB = GEO::RB, DXS = GEO::DXS;
  constexpr unsigned SLOT_BYTES = (unsigned)MP * RB;
  const int F = d.F, T = d.T, K = WK == WK_TV ? d.K : 1, TBk = fp.len;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int n = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's source
  const int g = (int)blockIdx.x;
  int b0, f0, tb0, nblk;
  if (!flat_start(fp, g, b0, f0, tb0, nblk)) return;  // whole workgroup: the range is wave-uniform
  Cursor c0;
  c0.b = b0;
  c0.f = f0;
  c0.tb = tb0;
  const size_t FT = (size_t)F * T;
  const bool mine = n < NXI;  // this wave carries one X instruction per item

  // ---- X requests (the range never leaves utterance b0)
  const BufRsrc rx = make_rsrc_sized(X + (size_t)b0 * M * FT, (size_t)M * FT * sizeof(Cx<R>));
  const int xrow = min(n * GEO::RPI + lane / GEO::LPR, M - 1);  // the padding row of an odd M repeats row M-1
  const unsigned xlane = (unsigned)((size_t)xrow * FT * sizeof(Cx<R>)) + (unsigned)(lane % GEO::LPR) * 16u;
  auto request_x = [&](const Cursor& c, int sl) {
    if (mine) {
      // frames past T read into the next row (zeros past the end of the utterance): their weight is 0
      const unsigned voff = xlane + (unsigned)(((size_t)c.f * T + (size_t)c.tb * WAVE) * sizeof(Cx<R>));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rx, (__attribute__((address_space(3))) void*)(smem + (unsigned)sl * SLOT_BYTES + (unsigned)n * GEO::RPI * RB), 16,
          (int)voff, 0, 0, 0);
    }
  };
  const unsigned xread0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem + (unsigned)lane * (unsigned)sizeof(Cx<R>);

  // ---- weight inputs of one item, in place (tied loads): vv[j] = V[b, n, j, t] (TV) or r[b, n, (f,) t] in vv[0]
  const size_t vrows = WK == WK_TV ? (size_t)N * K : (WK == WK_NT ? (size_t)N : (size_t)N * F);
  buf_u4 rv = make_rsrc_words(V + (size_t)b0 * vrows * T, vrows * T * sizeof(R));
  rv.x = __builtin_amdgcn_readfirstlane(rv.x);
  rv.y = __builtin_amdgcn_readfirstlane(rv.y);
  rv.z = __builtin_amdgcn_readfirstlane(rv.z);
  rv.w = __builtin_amdgcn_readfirstlane(rv.w);
  const unsigned vlane = (unsigned)lane * (unsigned)sizeof(R);
  R vv[KC];
#pragma unroll
  for (int j = 0; j < KC; ++j) vv[j] = 0;
  auto request_w = [&](const Cursor& c) {
    if (WK == WK_TV) {
      const unsigned row0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(((size_t)n * K * T + (size_t)c.tb * WAVE) * sizeof(R)));
      const unsigned tstep = (unsigned)__builtin_amdgcn_readfirstlane((int)((size_t)T * sizeof(R)));
      static_for<KC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (j < K) buf_ld_tied(vv[j], rv, vlane, row0 + (unsigned)j * tstep);
      });
    } else {
      const size_t row = WK == WK_NT ? (size_t)n : (size_t)n * F + c.f;
      const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((row * T + (size_t)c.tb * WAVE) * sizeof(R)));
      buf_ld_tied(vv[0], rv, vlane, soff);
    }
  };
  // basis row of (source n, bin f): lane k holds Tb[b, n, f, k] (0 beyond n_basis).  A tied load like the others (a
  // compiler-managed one would put its own vmcnt(0) -- draining the X request -- at the merge point of every trip); the
  // mask is applied by row_value() after the trip's wait.
  buf_u4 rt = make_rsrc_words(Tb + (size_t)b0 * N * F * K, (size_t)N * F * K * sizeof(R));
  rt.x = __builtin_amdgcn_readfirstlane(rt.x);
  rt.y = __builtin_amdgcn_readfirstlane(rt.y);
  rt.z = __builtin_amdgcn_readfirstlane(rt.z);
  rt.w = __builtin_amdgcn_readfirstlane(rt.w);
  auto request_row = [&](const Cursor& c, R& dst) {
    if (WK == WK_TV) {
      const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((((size_t)n * F + c.f) * K) * sizeof(R)));
      buf_ld_tied(dst, rt, vlane, soff);
    }
  };
  auto row_value = [&](R raw) -> R { return WK != WK_TV ? (R)1 : (lane < K ? raw : (R)0); };
  auto weight = [&](const Cursor& c, R tbl) -> R {  // 1 / max(r^(2/domain), eps) of this lane's frame, 0 past T
    R tv;
    if (WK == WK_TV) {
      tv = 0;
#pragma unroll
      for (int j = 0; j < KC; ++j) tv = fma(lane_value(tbl, j), vv[j], tv);  // k ascending; rows past n_basis add 0 * 0
    } else {
      tv = vv[0];
    }
    const R rr = floor_eps<R>(D2 ? tv : powspec<R>(tv, p2d), eps);  // floored AFTER the power (ilrma.py:499-509)
    return c.tb * WAVE + lane < T ? fast_rcp(rr) : (R)0;
  };

  R acc[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) acc[q] = 0;
  Cursor c1 = c0;
  advance(c1, TBk, F);
  Cursor c2 = c1;
  advance(c2, TBk, F);
  // prologue: weight of item 0; X of items 0 and 1 on their way
  R traw = 0;
  request_row(c0, traw);
  request_w(c0);
  request_x(c0, 0);
  request_x(nblk > 1 ? c1 : c0, 1);
  if (mine) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < KC; ++j) asm volatile("" : "+v"(vv[j]));
  asm volatile("" : "+v"(traw));
  R tbl = row_value(traw);
  R wgt = weight(c0, tbl);
  int sl = 0;
  for (int it = 0; it < nblk; ++it) {
    const bool more = it + 1 < nblk, more2 = it + 2 < nblk;
    // every wave's row of item `it` has landed (each waited for its own at the end of the previous trip) and nobody reads
    // the slot of item it-1 any more.  A bare barrier: only LDS traffic crosses waves, and __syncthreads() would drain
    // the X request that has to stay in flight.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const Cursor c1v = more ? c1 : c0, c2v = more2 ? c2 : c0;
    const bool new_bin = more && c1.tb == 0;  // the next item starts a new bin: its basis row
    if (new_bin) request_row(c1, traw);
    request_w(c1v);
    request_x(c2v, sl == 0 ? DXS - 1 : sl - 1);  // the slot item it-1 has left
    Vec2<R> xv[M];
    const unsigned xaddr = xread0 + (unsigned)sl * SLOT_BYTES;
    static_for<M>([&](auto mc) { xrow_read_at<decltype(mc)::value * RB>(xaddr, xv[decltype(mc)::value]); });
    xrow_wait(xv);
    // x_i conj(x_j) for j >= i, weighted: diagonal terms in acc[i], pairs (re, im) at herm_pair_base
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const R sx = wgt * xv[i].x, sy = wgt * xv[i].y;
      acc[i] = fma(sx, xv[i].x, acc[i]);
      acc[i] = fma(sy, xv[i].y, acc[i]);
#pragma unroll
      for (int j = i + 1; j < M; ++j) {
        const int hb = herm_pair_base<M>(i, j);  // compile-time after unrolling
        acc[hb] = fma(sx, xv[j].x, acc[hb]);
        acc[hb] = fma(sy, xv[j].y, acc[hb]);
        acc[hb + 1] = fma(sy, xv[j].x, acc[hb + 1]);
        acc[hb + 1] = fma(-sx, xv[j].y, acc[hb + 1]);
      }
    }
    // the weight inputs of item it+1 -- and this wave's X row of item it+1, requested a trip ago -- have landed once at
    // most this trip's X request is in flight
    if (mine) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < KC; ++j) asm volatile("" : "+v"(vv[j]));
    asm volatile("" : "+v"(traw));
    if (c1.tb == 0 || !more) {  // the bin is complete (or the range ends): flush
      const R tot = wave_reduce_scatter<R, NV>(acc);
      const int i = scatter_index<NV>();
      const int slot = c0.f - f0;
      if (scatter_leader<NV>() && i < HM) part[(((size_t)g * fp.S + slot) * N + n) * HM + i] = tot;
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = 0;
    }
    if (new_bin) tbl = row_value(traw);
    wgt = weight(c1v, tbl);
    sl = sl + 1 == DXS ? 0 : sl + 1;
    c0 = c1;
    c1 = c2;
    advance(c2, TBk, F);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may land in LDS after the workgroup has gone
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
