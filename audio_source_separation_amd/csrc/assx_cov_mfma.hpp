// Covariance accumulate for 4 < n_basis <= 16 (the reference's default is 10, ilrma.py:183) with the variance
// contraction on the matrix cores and X on a three-item LDS ring.
//
// Round 2's cov_wide_kernel (assx_cov_wide.hpp) takes 2.2-2.7 us per item (8 bins x 64 frames) whatever is trimmed from
// it.  Two things hold it there.  (1) LDS: every one of a workgroup's 8 waves reads the whole activation tile and its
// basis row out of LDS for its own bin -- 80 ds_read per wave and item, 8 x 80 x 4 cycles of the CU's single LDS pipe
// (1.07 us), in a phase that never overlaps the VALU phase of the Hermitian fan-out because the per-item barrier
// lock-steps the waves.  (2) Latency: the loads of item i+1 are issued after the barrier of item i and waited for at
// the barrier of item i+1, all 256 workgroups doing so in bursts: an item can never take less than one loaded memory
// round trip (measured: with (1) removed and nothing else changed the item still takes 2.6 us).
//
// Here (1) the variance  r[n, f, t] = sum_k Tb[n,f,k] V[n,k,t]  of the workgroup's 8 bins x 64 frames x N sources is ONE
// set of small matrix products per item, v_mfma_f64_16x16x4 (rows = bins, columns = frames, K = n_basis in slices of 4),
// shared by the 8 waves: wave w computes the tiles (source w/4 [+2], frames 16 (w%4) ...) from registers -- the
// activation operand arrives from L2 by vector loads two items ahead, the basis operand comes zero-padded out of an LDS
// image of the bin group's rows -- turns them into the reciprocal weights 1 / max(r^(2/domain), eps) and stores those
// to LDS; after the item's barrier every wave reads the N weights of its own bin and frame (N ds_read instead of 80)
// and runs the Hermitian fan-out of the K <= 4 kernel.  The products of item i+1 are threaded through the fan-out of
// item i (the matrix pipe works beside the vector pipe), the rcp/Newton sequence is evaluated once per (source, bin,
// frame) by exactly one lane, and no activation tile lives in LDS.  The bits are those of cov_wide_kernel: the matrix
// core accumulates k in ascending order with fused multiply-adds, as the vector form did (tools/covw_ab.py digests).
// (2) X rides a wave-private three-slot LDS ring filled by LDS-direct loads (no registers): the rows of item i+2 are
// requested while item i is consumed, so two items' worth of X per CU (64 KB) is in flight and the per-item barrier no
// longer gates the memory pipeline.  Those loads -- and the tied register loads of the activation operand -- are
// invisible to the compiler's wait-count model: the one VMEM wait per item is explicit, vmcnt(instructions of one X
// request) -- issue order per trip is [activation operand of item i+2][X of item i+2], so that leaves exactly the
// youngest X request in flight.  Records, partition and finalize are those of cov_wide_kernel
// (part[g][slot][w][n][M*M]).
#pragma once
#include "assx_cov_wide.hpp"
#include "assx_nmf_mfma.hpp"

#ifndef COVM_TRACE
#define COVM_TRACE 0  // 1: workgroup 100 writes shader-clock stamps of every trip to the tail of the caller's scratch (tools/probes/covm_trace.py)
#endif
#ifndef COVM_SKIP
#define COVM_SKIP 0  // timing experiments only (tools/probes/covm_parts.sh): 1 products, 2 fan-out, 4 X requests, 8 publish, 16 barrier, 32 operand requests
#endif

#if (COVM_SKIP || COVM_TRACE) && !defined(ASSX_PROBE_BUILD)
#error "COVM_SKIP / COVM_TRACE are timing experiments (wrong results by construction): build them with -DASSX_PROBE_BUILD into a probe library, never into libassx.so"
#endif

namespace assx {

// which bin of the group a matrix-core row stands for (rows the accumulator layout cannot hand to a lane in its first
// two registers carry no bin: -1), and the bin behind accumulator register r (0 or 1) of a lane
template <typename R>
struct CovMfmaRows;
template <>
struct CovMfmaRows<double> {  // accumulator row of (lane, r) = (lane >> 4) + 4 r
  static __device__ __forceinline__ int bin_of_row(int i) { return i < COVW_BINS ? i : -1; }
  static __device__ __forceinline__ int bin_of_reg(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct CovMfmaRows<float> {  // accumulator row of (lane, r) = 4 (lane >> 4) + r
  static __device__ __forceinline__ int bin_of_row(int i) { return (i & 3) < 2 ? 2 * (i >> 2) + (i & 3) : -1; }
  static __device__ __forceinline__ int bin_of_reg(int lane, int r) { return 2 * (lane >> 4) + r; }
};

// the p-th pair (m, l > m) in the order m ascending, l ascending
template <int M>
__host__ __device__ constexpr int herm_pair_m(int p) {
  int m = 0;
  while (p >= M - 1 - m) {
    p -= M - 1 - m;
    ++m;
  }
  return m;
}
template <int M>
__host__ __device__ constexpr int herm_pair_l(int p) {
  int m = 0;
  while (p >= M - 1 - m) {
    p -= M - 1 - m;
    ++m;
  }
  return m + 1 + p;
}

template <typename R>
struct CovMfmaGeom {
  static constexpr int DXS = 3;                               // X ring slots (items)
  static constexpr int RB = WAVE * 2 * (int)sizeof(R);        // bytes of one row of X: 64 complex frames
  static constexpr int LPR = RB / 16;                         // lanes per row of a 16-byte-per-lane LDS-direct load
  static constexpr int RPI = WAVE / LPR;                      // rows per instruction (f64: 1, f32: 2)
  static __host__ __device__ constexpr int nxi(int M) { return (M + RPI - 1) / RPI; }  // instructions per X request
  static __host__ __device__ constexpr int mp(int M) { return nxi(M) * RPI; }          // rows per wave, padded
  static __host__ __device__ constexpr size_t x_bytes(int M) { return (size_t)DXS * COVW_BINS * mp(M) * RB; }
  static __host__ __device__ constexpr size_t lds_bytes(int M, int KS) {
    return x_bytes(M) + ((size_t)2 * M * COVW_BINS * WAVE + (size_t)2 * M * 16 * 4 * KS) * sizeof(R);
  }
};

// this lane's frame of the M rows of a landed X item (rows RB bytes apart): the reads are only ISSUED here, so that the
// weight and basis-row reads that follow share one LDS round trip with them; xrows_wait() makes the values readable
template <int RB>
__device__ __forceinline__ void xrows_read(unsigned addr, Vec2<double> (&x)[4]) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%5\n\tds_read_b128 %2, %4 offset:%6\n\t"
               "ds_read_b128 %3, %4 offset:%7"
               : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
               : "v"(addr), "n"(RB), "n"(2 * RB), "n"(3 * RB)
               : "memory");
}
template <int RB>
__device__ __forceinline__ void xrows_read(unsigned addr, Vec2<double> (&x)[3]) {
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:%4\n\tds_read_b128 %2, %3 offset:%5"
               : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2])
               : "v"(addr), "n"(RB), "n"(2 * RB)
               : "memory");
}
template <int RB>
__device__ __forceinline__ void xrows_read(unsigned addr, Vec2<double> (&x)[2]) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3"
               : "=&v"(x[0]), "=&v"(x[1])
               : "v"(addr), "n"(RB)
               : "memory");
}
template <int RB>
__device__ __forceinline__ void xrows_read(unsigned addr, Vec2<float> (&x)[4]) {
  asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:%5\n\tds_read_b64 %2, %4 offset:%6\n\t"
               "ds_read_b64 %3, %4 offset:%7"
               : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
               : "v"(addr), "n"(RB), "n"(2 * RB), "n"(3 * RB)
               : "memory");
}
template <int RB>
__device__ __forceinline__ void xrows_read(unsigned addr, Vec2<float> (&x)[3]) {
  asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:%4\n\tds_read_b64 %2, %3 offset:%5"
               : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2])
               : "v"(addr), "n"(RB), "n"(2 * RB)
               : "memory");
}
template <int RB>
__device__ __forceinline__ void xrows_read(unsigned addr, Vec2<float> (&x)[2]) {
  asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:%3"
               : "=&v"(x[0]), "=&v"(x[1])
               : "v"(addr), "n"(RB)
               : "memory");
}

template <typename R, int M>
__device__ __forceinline__ void xrows_wait(Vec2<R> (&x)[M]) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int m = 0; m < M; ++m) asm volatile("" : "+v"(x[m]));
}

// wait until at most N VMEM operations are in flight; the NV values become readable here and not before
template <int N, typename R, int NV>
__device__ __forceinline__ void wait_values(R (&v)[NV]) {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#pragma unroll
  for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i]));
}

// KS = k-slices of 4 (n_basis <= 4 KS)
template <typename R, int M, bool D2, int KS>
__global__ void __launch_bounds__(WAVE * COVW_BINS)
    cov_mfma_kernel(const Cx<R>* __restrict__ X, const R* __restrict__ Tb, const R* __restrict__ V, R* __restrict__ part,
                    Dims d, FlatPart fp, R eps, PowSpec p2d, unsigned long long* trace) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int N = M, HM = M * M, WB = COVW_BINS, NACC = N * HM, NV = next_pow2_c(NACC);
  constexpr int KP = 4 * KS;                   // padded rank
  constexpr int NT = (4 * N + WB - 1) / WB;    // variance tiles per wave and item (4 N tiles of 16 frames, 8 waves)
  constexpr int NB = NT * KS;                  // variance products = activation-operand loads per wave and item
  using MM = Mfma16<R>;
  using acc_t = typename MM::acc_t;
  using ROWS = CovMfmaRows<R>;
  using GEO = CovMfmaGeom<R>;
  constexpr int NXI = GEO::nxi(M), MP = GEO::mp(M), RB = GEO::RB, DXS = GEO::DXS;
  int trace_n = 0;
  auto stamp = [&]() {
    if (COVM_TRACE && trace && blockIdx.x == 100 && (threadIdx.x == 0 || threadIdx.x == 320) && trace_n < 400)
      trace[(threadIdx.x ? 400 : 0) + trace_n] = __builtin_amdgcn_s_memtime();
    ++trace_n;
  };
  const int F = d.F, T = d.T, K = d.K, TBk = fp.len, FG = (F + WB - 1) / WB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  R* const Wt = reinterpret_cast<R*>(smem + GEO::x_bytes(M));  // [2][N][WB][WAVE] reciprocal weights, consumed / produced
  R* const Tl = Wt + 2 * N * WB * WAVE;                        // [2][N][16][KP] basis rows of the current / next bin group
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int g = (int)blockIdx.x;
  long long q0, q1;
  flat_range(fp, g, q0, q1);
  if (q0 >= q1) return;
  const int nblk = (int)(q1 - q0);
  const int jg_first = (int)(q0 / TBk);
  Cursor c0;  // .f counts bin groups, .tb 64-frame items
  c0.tb = (int)(q0 - (long long)jg_first * TBk);
  c0.b = jg_first / FG;
  c0.f = jg_first - c0.b * FG;
  const size_t FT = (size_t)F * T;
  // wave-private X ring: slot s holds this wave's MP rows of 64 frames
  unsigned char* const xring = smem + (size_t)w * MP * RB;
  const unsigned xread0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)xring + (unsigned)lane * (unsigned)sizeof(Cx<R>);
  constexpr unsigned SLOT_BYTES = (unsigned)WB * MP * RB;

  // this wave's variance tiles: tile tau = w + 8 i  ->  source tau / 4, frames 16 (tau % 4) .. + 15
  const int jt = w & 3;
  auto tile_src = [&](int i) { return (w >> 2) + 2 * i; };
  auto tile_on = [&](int i) { return (4 * N) % WB == 0 || w + WB * i < 4 * N; };  // compile-time true for N = 2, 4

  auto load_rows = [&](const Cursor& c, int par) {  // Tl[par][n][row][k] = Tb[b, n, f0 + bin(row), k] or 0
    for (int i = tid; i < N * 16 * KP; i += WAVE * WB) {
      const int n = i / (16 * KP), rem = i - n * 16 * KP, row = rem / KP, k = rem - row * KP;
      const int bin = ROWS::bin_of_row(row), ff = c.f * WB + bin;
      R v = 0;
      if (bin >= 0 && ff < F && k < K) v = Tb[(((size_t)c.b * N + n) * F + ff) * K + k];
      Tl[par * (N * 16 * KP) + i] = v;
    }
  };
  // ---- X requests: instruction j of an item's request carries rows j RPI .. of this wave's bin into the ring slot.
  // The per-lane part of the offset is loop invariant (xlane; xlast for the last instruction, whose padding row repeats
  // row M-1), the rest is wave-uniform.  The WHOLE offset rides in the vector register, so the descriptor's range check
  // covers it: frames past T read into the next row, zeros past the end of the utterance; their weights are 0.
  const unsigned xlane = (unsigned)((size_t)(lane / GEO::LPR) * FT * sizeof(Cx<R>)) + (unsigned)(lane % GEO::LPR) * 16u;
  const unsigned xlast = (unsigned)((size_t)(min((NXI - 1) * GEO::RPI + lane / GEO::LPR, M - 1) - (NXI - 1) * GEO::RPI) * FT * sizeof(Cx<R>)) +
                         (unsigned)(lane % GEO::LPR) * 16u;
  struct XReq {
    BufRsrc rx;
    unsigned row;  // wave-uniform byte offset of (bin, first frame) inside the utterance
    unsigned char* slot;
  };
  auto x_item = [&](const Cursor& c, int sl) {
    XReq q;
    q.rx = make_rsrc_sized(X + (size_t)c.b * M * FT, (size_t)M * FT * sizeof(Cx<R>));
    q.row = (unsigned)(((size_t)min(c.f * WB + w, F - 1) * T + (size_t)c.tb * WAVE) * sizeof(Cx<R>));
    q.slot = xring + (unsigned)sl * SLOT_BYTES;
    return q;
  };
  auto request_x1 = [&](const XReq& q, auto jc) {
    constexpr int j = decltype(jc)::value;
    const unsigned voff = (j == NXI - 1 ? xlast : xlane) + (q.row + (unsigned)((size_t)j * GEO::RPI * FT * sizeof(Cx<R>)));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(q.rx, (__attribute__((address_space(3))) void*)(q.slot + j * GEO::RPI * RB), 16,
                                             (int)voff, 0, 0, 0);
  };
  // ---- activation-operand requests (B operand of the variance product), in place: element (i, s) is
  // V[b, n_i, 4 s + lk, t0 + 16 jt + li].  Rows past n_basis repeat the last row (their basis entries are zero, the value
  // only has to be finite): only the last slice can reach them, so vlane serves every slice but the last.  Frames past
  // T read into the next row (zeros past the end of the utterance's V): masked in publish.
  unsigned vlane[NT], vlastl[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int n = min(tile_src(i), N - 1);
    vlane[i] = (unsigned)(((size_t)(n * K + lk) * T + 16 * jt + li) * sizeof(R));
    vlastl[i] = (unsigned)(((size_t)(n * K + min(4 * (KS - 1) + lk, K - 1)) * T + 16 * jt + li) * sizeof(R));
  }
  struct VReq {
    buf_u4 rv;
    unsigned col;  // wave-uniform byte offset of the item's first frame inside a row
  };
  auto v_item = [&](const Cursor& c) {
    VReq q;
    q.rv = make_rsrc_words(V + (size_t)c.b * N * K * T, (size_t)N * K * T * sizeof(R));
    q.rv.x = __builtin_amdgcn_readfirstlane(q.rv.x);  // the cursor is wave-uniform but comes out of 64-bit vector-ALU
    q.rv.y = __builtin_amdgcn_readfirstlane(q.rv.y);  // divisions: an "s" operand of inline asm is not legalised for us
    q.rv.z = __builtin_amdgcn_readfirstlane(q.rv.z);
    q.rv.w = __builtin_amdgcn_readfirstlane(q.rv.w);
    q.col = (unsigned)__builtin_amdgcn_readfirstlane((int)((size_t)c.tb * WAVE * sizeof(R)));
    return q;
  };
  auto request_v1 = [&](const VReq& q, R (&bv)[NB], auto ic, auto sc) {
    constexpr int i = decltype(ic)::value, s2 = decltype(sc)::value;
    const unsigned voff = (s2 == KS - 1 ? vlastl[i] : vlane[i] + (unsigned)((size_t)4 * s2 * T * sizeof(R))) + q.col;
    buf_ld_tied(bv[i * KS + s2], q.rv, voff, 0u);
  };
  auto request_v = [&](const Cursor& c, R (&bv)[NB]) {
    const VReq q = v_item(c);
    static_for<NT>([&](auto ic) { static_for<KS>([&](auto sc) { request_v1(q, bv, ic, sc); }); });
  };
  auto request_x = [&](const Cursor& c, int sl) {
    const XReq q = x_item(c, sl);
    static_for<NXI>([&](auto jc) { request_x1(q, jc); });
  };
  auto read_rows = [&](int par, R (&av)[NT][KS]) {  // A operand: the bin group's basis rows, [row = li][k = 4 s + lk]
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) av[i][s2] = Tl[((par * N + min(tile_src(i), N - 1)) * 16 + li) * KP + 4 * s2 + lk];
  };
  auto variance = [&](int par, const R (&bv)[NB], acc_t (&tv)[NT]) {  // prologue form: all products of an item at once
    R av[NT][KS];
    read_rows(par, av);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[i][r] = 0;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
      for (int i = 0; i < NT; ++i)
        if (tile_on(i)) tv[i] = MM::mma(av[i][s2], bv[i * KS + s2], tv[i]);
  };
  // weights of item c from its variance tiles -> Wt[buf].  The 2 NT reciprocal chains advance one Newton step at a time,
  // side by side: a dependent f64 instruction issues ~30 cycles after its predecessor, four chains one after the other
  // took 700-900 cycles.  `between(step)` lets the caller thread other instructions through.
  auto publish = [&](const Cursor& c, int buf, const acc_t (&tv)[NT], auto&& between) {
    constexpr int NC = 2 * NT;
    const bool tlive = c.tb * WAVE + 16 * jt + li < T;
    R rr[NC], rc[NC], er[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q)  // floored AFTER the power (ilrma.py:499-509)
      rr[q] = floor_eps<R>(D2 ? tv[q / 2][q % 2] : powspec<R>(tv[q / 2][q % 2], p2d), eps);
    between(IntC<0>());
#pragma unroll
    for (int q = 0; q < NC; ++q) rc[q] = sizeof(R) == 8 ? (R)__builtin_amdgcn_rcp((double)rr[q]) : (R)__builtin_amdgcn_rcpf((float)rr[q]);
    between(IntC<1>());
#pragma unroll
    for (int q = 0; q < NC; ++q) er[q] = fma(-rr[q], rc[q], (R)1);
#pragma unroll
    for (int q = 0; q < NC; ++q) rc[q] = fma(rc[q], er[q], rc[q]);
    between(IntC<2>());
    if (sizeof(R) == 8) {  // second Newton step (float64 only: fast_rcp)
#pragma unroll
      for (int q = 0; q < NC; ++q) er[q] = fma(-rr[q], rc[q], (R)1);
#pragma unroll
      for (int q = 0; q < NC; ++q) rc[q] = fma(rc[q], er[q], rc[q]);
    }
    between(IntC<3>());
#pragma unroll
    for (int q = 0; q < NC; ++q) asm volatile("" : "+v"(rc[q]));  // evaluated for every lane: no branch around a chain
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const int i = q / 2, bin = ROWS::bin_of_reg(lane, q % 2);
      if (tile_on(i)) Wt[((buf * N + tile_src(i)) * WB + bin) * WAVE + 16 * jt + li] = (tlive && c.f * WB + bin < F) ? rc[q] : (R)0;
    }
  };
  auto nothing = [](auto) {};

  R acc[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) acc[q] = 0;
  R bn[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) bn[i] = 0;
  acc_t tv[NT], tvp[NT];
  Cursor c1 = c0;
  advance(c1, TBk, FG);
  Cursor c2 = c1;
  advance(c2, TBk, FG);
  Cursor c3 = c2;
  advance(c3, TBk, FG);
  // Software pipeline (trip `it` consumes item `it`):  X of item i is requested in trip i-2, its activation operand in
  // trip i-3, its variance products are issued in trip i-2 and turned into weights in trip i-1 -- every consumer's input
  // was produced at least one trip earlier, so no trip waits for a matrix-core result or a memory round trip of its own.
  int par = 0;  // Tl slot of the bin group the NEXT variance product belongs to
  // prologue: weights of item 0; products of item 1 (-> tvp); operand of item 2; X of items 0 and 1
  load_rows(c0, 0);
  request_v(c0, bn);
  request_x(c0, 0);
  wait_values<0>(bn);
  __syncthreads();
  variance(0, bn, tv);
  publish(c0, 0, tv, nothing);
  {
    const Cursor c1v = nblk > 1 ? c1 : c0, c2v = nblk > 2 ? c2 : c0;
    if (nblk > 1 && c1.tb == 0) {  // item 1 already belongs to the next bin group
      par = 1;
      load_rows(c1, 1);
      __syncthreads();
    }
    request_v(c1v, bn);
    wait_values<0>(bn);
    variance(par, bn, tvp);
    if (nblk > 2 && c2.tb == 0) {  // so does item 2 (after a barrier in the case above: slot 0 is no longer read)
      par ^= 1;
      load_rows(c2, par);
    }
    request_v(c2v, bn);  // loop order: [activation operand][X]
    request_x(c1v, 1);
  }
  int sl = 0;  // ring slot of item `it`
  for (int it = 0; it < nblk; ++it) {
    const bool more = it + 1 < nblk, more2 = it + 2 < nblk, more3 = it + 3 < nblk;
    stamp();
    // X of item `it` (requested two trips ago) and the activation operand of item it+2 (first requests of the previous
    // trip) have landed once at most the previous trip's X request -- the NXI youngest operations -- is in flight
    wait_values<(COVM_SKIP & 4) ? 0 : NXI>(bn);
    stamp();
    Vec2<R> xv[M];
    xrows_read<RB>(xread0 + (unsigned)sl * SLOT_BYTES, xv);  // wave-private: needs no barrier, overlaps the wait for one
    // weights of item `it` (and rows stored during the previous trip) are visible; Wt[(it+1)&1] is free.  A BARE barrier:
    // __syncthreads() is a fence + barrier and the fence drains vmcnt to 0, i.e. the X request that must stay in flight.
    // Only LDS traffic crosses waves here, so this wave's LDS writes (lgkmcnt) are all the barrier has to wait for.
    if (!(COVM_SKIP & 16)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamp();
    // The trip body is ONE basic block (no branch on `more`): past the end of the range it re-requests item `it` and
    // produces weights nobody reads.  All eight waves run it in lock step, so whatever unit a phase leans on -- LDS,
    // the load path, the matrix pipe, a chain of dependent f64 instructions -- is hit by all of them at once while the
    // others idle (in-kernel stamps: the trip was the SUM of its phases).  The order below therefore threads the phases
    // through each other, pinned with scheduling barriers because the scheduler's own choice groups like with like.
    Cursor c1v = c1, c2v = c2, c3v = c3;
    if (!more) c1v = c0;
    if (!more2) c2v = c0;
    if (!more3) c3v = c0;
    const int sl2 = sl == 0 ? DXS - 1 : sl - 1;  // slot of item it+2 = the one item it-1 has left
    R wgt[N];
#pragma unroll
    for (int n = 0; n < N; ++n) wgt[n] = Wt[(((it & 1) * N + n) * WB + w) * WAVE + lane];
    R av[NT][KS];
    read_rows(par, av);
    const VReq vq = v_item(c3v);
    const XReq xq = x_item(c2v, sl2);
    __builtin_amdgcn_sched_barrier(0);
    // (1) weights of item it+1 from the previous trip's products (long complete) while the LDS reads above travel
    if (!(COVM_SKIP & 8)) publish(c1v, (it + 1) & 1, tvp, [&](auto) { __builtin_amdgcn_sched_barrier(0); });
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    xrows_wait(xv);  // LDS returns in order: the weights and rows requested after X have arrived as well
    Cx<R> x[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = cmake<R>(xv[m].x, xv[m].y);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[i][r] = 0;
    // (2) the NB variance products of item it+2, one at a time; behind each the request that refills the operand it has
    // just read (item it+3) and a share of the vector fan-out of item `it` (this wave's bin, one frame per lane); the X
    // requests of item it+2 ride behind the last shares.  VMEM order per trip: [operand][X], as the wait above assumes.
    constexpr int NU = M + M * (M - 1) / 2;  // fan-out units: M diagonal terms, then the pairs (m, l > m)
    auto fan_unit = [&](auto uc) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u < M) {
        const R pd = cabs2(x[u]);
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n * HM + u] = fma(wgt[n], pd, acc[n * HM + u]);
      } else {
        constexpr int m = herm_pair_m<M>(u - M), l = herm_pair_l<M>(u - M);
        const Cx<R> pr = cmulc(x[m], x[l]);
        constexpr int hb = herm_pair_base<M>(m, l);
#pragma unroll
        for (int n = 0; n < N; ++n) {
          acc[n * HM + hb] = fma(wgt[n], pr.x, acc[n * HM + hb]);
          acc[n * HM + hb + 1] = fma(wgt[n], pr.y, acc[n * HM + hb + 1]);
        }
      }
    };
    constexpr int NSTEP = NB + NXI;  // shares of the fan-out: one per product, one per X request
    static_for<NSTEP>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if constexpr (q < NB) {
        constexpr int sq = q / NT, iq = q % NT;  // slice-major: a tile's chain is NT products apart
        if (!(COVM_SKIP & 1))
          if (tile_on(iq)) tv[iq] = MM::mma(av[iq][sq], bn[iq * KS + sq], tv[iq]);
        if (!(COVM_SKIP & 32)) request_v1(vq, bn, IntC<iq>(), IntC<sq>());
      } else {
        if (!(COVM_SKIP & 4)) request_x1(xq, IntC<q - NB>());
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(COVM_SKIP & 2)) static_for<(q + 1) * NU / NSTEP - q * NU / NSTEP>([&](auto jc) { fan_unit(IntC<q * NU / NSTEP + decltype(jc)::value>()); });
      __builtin_amdgcn_sched_barrier(0);
    });
    value_fence(acc);  // the fan-out stays HERE (left alone it is sunk below load_rows / the flush)
    stamp();
#pragma unroll
    for (int i = 0; i < NT; ++i) tvp[i] = tv[i];
    if (more3 && c3.tb == 0) load_rows(c3, par ^ 1);  // rows of the group after next trip's: read from the trip after next on
    if (c1.tb == 0 || !more) {  // the bin group is complete (or the range ends): flush
      const R tot = wave_reduce_scatter<R, NV>(acc);
      const int i = scatter_index<NV>();
      const int slot = c0.b * FG + c0.f - jg_first;
      if (scatter_leader<NV>() && i < NACC) part[(((size_t)g * fp.S + slot) * WB + w) * NACC + i] = tot;
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = 0;
    }
    if (more3 && c3.tb == 0) par ^= 1;  // next trip's product (item it+3) starts a new bin group
    sl = sl + 1 == DXS ? 0 : sl + 1;
    c0 = c1;
    c1 = c2;
    c2 = c3;
    advance(c3, TBk, FG);
    stamp();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may land in LDS after the workgroup has gone
#endif
}

}  // namespace assx
