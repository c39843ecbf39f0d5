// Streaming kernels of the ILRMA / AuxIVA iteration: the three passes over X that dominate the runtime.
//
// Design (measured on MI355X; DESIGN.md sections 4.1, 4.1.1, 4.2):
//   * ALL SOURCES IN ONE WAVE, one wave per workgroup.  A lane owns one frame of a 64-frame block; the wave forms the
//     M*M Hermitian products of its frames ONCE and fans each product into the N sources' accumulators (N*M*M reals
//     per lane at M = 4; measured 2.7x fewer instructions than one wave per source, which repeats the products and
//     the address arithmetic per source).  Per-bin rows (demixing filter, basis) are wave-uniform: SGPRs / broadcast
//     loads.
//   * FLAT BALANCED PARTITION (FlatPart below).  An utterance's (bin, 64-frame block) items are cut into equal
//     contiguous ranges, a small multiple of what the chip holds concurrently, so F = 1025 bins never quantise badly
//     against 256 CUs.  A range may straddle a bin boundary: the wave then flushes its accumulators (butterfly
//     reduce-scatter) into a per-(workgroup, slot) partial record; the consumers know which records cover a bin from
//     the partition arithmetic alone (no atomics, run-to-run bit-stable).  When the workgroup budget is a whole
//     number of parts per group the partition is aligned to the groups instead (one segment, one record).
//   * EXPLICIT MEMORY PIPELINE.  X blocks ride a register ring that is refilled IN PLACE by tied inline-asm buffer
//     loads (depth = the kernel's D template parameter, 2-3 blocks); the activation tile of block i+2 lands in a
//     wave-private LDS ring through LDS-direct loads (VTileDma, VDMA_SLOTS deep) and costs no registers.  Neither is
//     visible to the compiler's wait-count model, so every `s_waitcnt vmcnt(N)` is written out and the distances are
//     part of the kernels' contract with themselves (tests shrink the partition with ASSX_G to walk every trip kind).
#pragma once
#include <cstdlib>
#include "assx_common.hpp"

namespace assx {

enum { WK_NONE = 0, WK_NT = 1, WK_NFT = 2, WK_TV = 3 };

constexpr int KU = 4;  // k-unroll of the NMF contractions (K <= 4 is the single-chunk fast path)

struct Dims {
  int B, F, T, K;
};

// Flat partition of a (utterance, group, item-in-group) space among workgroups.  The partition is made PER UTTERANCE
// and replicated: an utterance's NB items are cut into Gu contiguous ranges of L items (the last one shorter), and
// workgroup g works on range g % Gu of utterance g / Gu.  A range therefore never crosses an utterance, and -- the
// point -- which items are summed together, and in which order, depends only on the utterance's own geometry, not
// on how many utterances share the launch: a batched call is bit-identical to per-utterance calls, and a batch
// sharded over ranks is bit-identical to the same batch on one rank.
struct FlatPart {
  long long NB;  // items of ONE utterance
  int len;       // items per group (a group = one bin's frame blocks, or one frame block's bins)
  int L;         // items per workgroup
  int G;         // workgroups in the launch (= B * Gu)
  int S;         // partial-record slots per workgroup
  int Gu;        // workgroups per utterance
  int Ju;        // groups per utterance (NB / len)
  int P;         // 0: flat ranges of L items (may cross groups).  > 0: ALIGNED -- every group is cut into P parts of
                 // L items, workgroup gl = group * P + part; a range never crosses a group, so a workgroup has ONE
                 // segment / flush and one record slot
  // ---- launch order (which workgroup of the grid takes which range; never changes what a range computes)
  int Gp;        // 0: legacy order (grid = G workgroups, each XCD takes a contiguous eighth of ALL ranges).  > 0:
                 // utterance-sequential order: grid = B * Gp workgroups, Gp = Gu rounded up to a multiple of the XCD
                 // count; the utterances are walked one after the other and inside an utterance each XCD takes a
                 // contiguous eighth of its ranges (launch slots beyond Gu exit at once)
  int rev;       // utterance-sequential order only: walk the utterances last to first
};

// A workgroup whose range crosses a group boundary pays a second prologue / flush (3-5 us measured, on the critical
// path: the kernel ends with its slowest workgroup).  When the workgroup budget is (almost) a whole number of parts
// per group the partition is therefore aligned to the groups -- the activation pass at config 4: 64 frame blocks x 8
// parts = the 512 resident workgroups exactly.  The covariance / basis passes (1025 bins, 2048 workgroups) cannot be.
inline FlatPart make_flat(int B, long long NB, int len, long long G_target) {
  FlatPart p;
  p.Gp = 0;
  p.rev = 0;
  p.NB = NB;
  p.len = len;
  p.Ju = (int)(NB / len);
  p.P = 0;
  long long G = G_target < NB ? G_target : NB;
  if (G < 1) G = 1;
  const long long parts = G / p.Ju;
  static const bool align = lab_int("ASSX_ALIGN", 1) != 0;  // laboratory builds: A/B switch
  if (align && parts >= 1 && parts <= len && p.Ju * parts * 100 >= G * 97) {
    // L first, then only as many parts as are not empty (len = 31 into 31 parts of L = 2 would leave 15 parts
    // without an item: idle workgroups, and record / loss-partial slots nobody writes)
    p.L = (int)((len + parts - 1) / parts);
    p.P = (len + p.L - 1) / p.L;
    p.Gu = p.Ju * p.P;
    p.G = B * p.Gu;
    p.S = 1;
    return p;
  }
  p.L = (int)((NB + G - 1) / G);
  p.Gu = (int)((NB + p.L - 1) / p.L);
  p.G = B * p.Gu;
  p.S = (p.L + len - 2) / len + 1;
  return p;
}

// first item (within the utterance) and item count of local workgroup gl; false: nothing to do
__host__ __device__ __forceinline__ bool flat_local(const FlatPart& fp, unsigned gl, unsigned& lo, unsigned& n) {
  const unsigned L = (unsigned)fp.L;
  if (fp.P > 0) {
    const unsigned grp = gl / (unsigned)fp.P, part = gl - grp * (unsigned)fp.P, len = (unsigned)fp.len;
    const unsigned off = part * L;
    if (off >= len) return false;
    lo = grp * len + off;
    n = off + L < len ? L : len - off;
    return true;
  }
  const unsigned nb = (unsigned)fp.NB;
  lo = gl * L;
  if (lo >= nb) return false;
  n = lo + L < nb ? L : nb - lo;
  return true;
}

// producer side: the GLOBAL item range [q0, q1) of workgroup g (global item = utterance * NB + item in utterance)
__host__ __device__ __forceinline__ void flat_range(const FlatPart& fp, int g, long long& q0, long long& q1) {
  const int b = g / fp.Gu;
  unsigned lo = 0, n = 0;
  if (!flat_local(fp, (unsigned)(g - b * fp.Gu), lo, n)) n = 0;
  q0 = (long long)b * fp.NB + lo;
  q1 = q0 + n;
}
// the same start as (utterance, group in the utterance, item in the group, item count), in 32-bit arithmetic: an
// utterance's NB items fit an int (checked on the host), so the 64-bit division a global item index needs -- ~150
// instructions at the head of every streaming kernel, before its first load is issued -- is not needed
__device__ __forceinline__ bool flat_start(const FlatPart& fp, int g, int& b, int& grp, int& item, int& n) {
  b = g / fp.Gu;
  unsigned lo, cnt;
  if (!flat_local(fp, (unsigned)(g - b * fp.Gu), lo, cnt)) return false;
  n = (int)cnt;
  grp = (int)(lo / (unsigned)fp.len);
  item = (int)(lo - (unsigned)grp * (unsigned)fp.len);
  return true;
}
// consumer side: the workgroups g_lo..g_hi whose records cover GLOBAL group j (= utterance * Ju + group), and the
// slot of that group inside workgroup g's block of records
__host__ __device__ __forceinline__ void flat_cover(const FlatPart& fp, long long j, int& g_lo, int& g_hi) {
  const int b = (int)(j / fp.Ju);
  const unsigned jl = (unsigned)(j - (long long)b * fp.Ju);
  if (fp.P > 0) {
    g_lo = b * fp.Gu + (int)(jl * (unsigned)fp.P);
    g_hi = g_lo + (fp.len + fp.L - 1) / fp.L - 1;
    return;
  }
  const unsigned lo = jl * (unsigned)fp.len;
  g_lo = b * fp.Gu + (int)(lo / (unsigned)fp.L);
  g_hi = b * fp.Gu + (int)((lo + (unsigned)fp.len - 1u) / (unsigned)fp.L);
}
__host__ __device__ __forceinline__ int flat_slot(const FlatPart& fp, long long j, int g) {
  if (fp.P > 0) return 0;
  const int b = g / fp.Gu, gl = g - b * fp.Gu;
  return (int)(j - (long long)b * fp.Ju) - (int)(((unsigned)gl * (unsigned)fp.L) / (unsigned)fp.len);
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

// Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each XCD with its own 4 MiB L2.  The flat partitions
// hand out contiguous ranges of (utterance, bin, block) items, so giving XCD x the x-th contiguous eighth of the
// ranges keeps what is re-read from L2 -- the activation tiles, the per-bin rows -- local: with 8 utterances in
// flight every XCD then sees one utterance's V (0.5 MB) instead of all eight (4.2 MB, more than its L2).
constexpr int N_XCD = 8;
__host__ __device__ __forceinline__ int xcd_local_range(int bid, int G) {
  const int x = bid % N_XCD, j = bid / N_XCD;
  const int q = G / N_XCD, r = G % N_XCD;
  return x * q + (x < r ? x : r) + j;
}

// Range of workgroup `bid` of the launch; < 0: a padding slot of the utterance-sequential order.
// Utterance-sequential order (fp.Gp > 0): with more than one utterance per launch the workgroups run in rounds, and
// X (B x 268.7 MB at config 4) streams through the 256 MiB Infinity Cache once per pass.  In the legacy order every
// XCD works on a different utterance at the same time, so nothing of X survives from one pass to the next.  Here all
// XCDs work on the same utterance (each on a contiguous eighth of its ranges, which keeps the per-bin rows and the
// utterance's activation local to an XCD's L2 as before), utterance after utterance, and consecutive passes alternate
// the direction: a pass starts with the utterance the previous pass ended with -- the one the cache still holds.
__host__ __device__ __forceinline__ int workgroup_range(int bid, int grid, const FlatPart& fp) {
  if (fp.Gp == 0) return xcd_local_range(bid, grid);
  const int Gp = fp.Gp, B = grid / Gp;
  int b = bid / Gp;
  const int r = bid - b * Gp;
  const int x = r % N_XCD, j = r / N_XCD;  // Gp is a multiple of N_XCD: x is the XCD this workgroup is dealt to
  const int q = fp.Gu / N_XCD, rem = fp.Gu % N_XCD;
  const int cnt = q + (x < rem ? 1 : 0);  // ranges of this XCD's eighth
  if (j >= cnt) return -1;
  const int first = x * q + (x < rem ? x : rem);
  if (fp.rev) {  // everything backwards: the most recently streamed lines are read first (a forward re-read of a set
    b = B - 1 - b;  // slightly larger than the cache evicts every line just before it is needed)
    return b * fp.Gu + first + (cnt - 1 - j);
  }
  return b * fp.Gu + first + j;
}


// ------------------------------------------------------------------------------------------
// Activation tile through LDS-direct loads (gfx950 `buffer_load_dwordx4 ... lds`).
//
// The streaming kernels need, per 64-frame block, NROW = N*KU rows of V (one value per lane and row).  As ordinary
// vector loads these are the youngest entries of the in-order vmcnt queue exactly when they are needed, so waiting
// for them also drains every older X load: the X ring never has more than one block in flight however deep it is
// declared, and a deeper V ring costs 32 VGPRs per stage that the f64 kernels do not have.  Landing the tile in a
// wave-private LDS ring instead costs no registers: the tile of block i+2 is requested while block i is consumed,
// `s_waitcnt vmcnt(NI + 2M)` then leaves the X loads of the next two blocks and the next tile in flight.
// 16 bytes per lane: one instruction covers RPI rows of 64 frames; lane L of instruction j lands at
// tile + j*RPI*ROW_BYTES + L*16, i.e. the LDS image is simply [row][frame].  Frames past the end of a row read into
// the next row (or zeros past the end of the buffer: the descriptor carries the exact size) and belong to lanes the
// kernels mask anyway.
// ------------------------------------------------------------------------------------------
template <typename R, int NROW>
struct VTileDma {
  static constexpr int LPR = WAVE * (int)sizeof(R) / 16;  // lanes per row
  static constexpr int RPI = WAVE / LPR;                   // rows per instruction
  static constexpr int NI = NROW / RPI;                    // instructions per tile
  static constexpr int ROW_BYTES = WAVE * (int)sizeof(R);
  static constexpr int TILE_BYTES = NROW * ROW_BYTES;
  static constexpr int NP = NROW / 2;                      // ds_read2st64 pairs per lane
  static_assert(NROW % RPI == 0 && NROW % 2 == 0 && (NP == 4 || NP == 6 || NP == 8), "tile geometry");
  unsigned vro[NI];  // this lane's byte offset (row start + piece) inside the utterance's V, per instruction

  // LDS row rho = n*KU + kk holds V row n*K + min(kk, K-1) (rows beyond K duplicate the last one; their basis is 0)
  __device__ __forceinline__ void init(int lane, int K, unsigned t_row) {
    const int grp = lane / LPR, piece = lane % LPR;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int rho = j * RPI + grp;
      const int n = rho / KU, kk = rho % KU;
      vro[j] = (unsigned)(n * K + (kk < K ? kk : K - 1)) * t_row + (unsigned)piece * 16u;
    }
  }
  // request the tile of frame block tb (soff = tb * 64 * sizeof(R)) into the LDS slot at `slot` (wave-uniform)
  __device__ __forceinline__ void issue(BufRsrc rv, unsigned char* slot, unsigned soff) const {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < NI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(slot + j * RPI * ROW_BYTES),
                                               16, (int)vro[j], (int)soff, 0, 0);
#endif
  }
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// this lane's value of every row of a landed tile: o[p] = (row 2p, row 2p+1).  One asm block: the reads and the
// lgkmcnt wait must not be separated from each other by the scheduler.
__device__ __forceinline__ void vtile_read(unsigned addr, Vec2<double> (&o)[8]) {
  asm volatile(
      "ds_read2st64_b64 %0, %8 offset0:0 offset1:1\n\tds_read2st64_b64 %1, %8 offset0:2 offset1:3\n\t"
      "ds_read2st64_b64 %2, %8 offset0:4 offset1:5\n\tds_read2st64_b64 %3, %8 offset0:6 offset1:7\n\t"
      "ds_read2st64_b64 %4, %8 offset0:8 offset1:9\n\tds_read2st64_b64 %5, %8 offset0:10 offset1:11\n\t"
      "ds_read2st64_b64 %6, %8 offset0:12 offset1:13\n\tds_read2st64_b64 %7, %8 offset0:14 offset1:15\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(addr)
      : "memory");
}
__device__ __forceinline__ void vtile_read(unsigned addr, Vec2<double> (&o)[6]) {
  asm volatile(
      "ds_read2st64_b64 %0, %6 offset0:0 offset1:1\n\tds_read2st64_b64 %1, %6 offset0:2 offset1:3\n\t"
      "ds_read2st64_b64 %2, %6 offset0:4 offset1:5\n\tds_read2st64_b64 %3, %6 offset0:6 offset1:7\n\t"
      "ds_read2st64_b64 %4, %6 offset0:8 offset1:9\n\tds_read2st64_b64 %5, %6 offset0:10 offset1:11\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
      : "v"(addr)
      : "memory");
}
__device__ __forceinline__ void vtile_read(unsigned addr, Vec2<double> (&o)[4]) {
  asm volatile(
      "ds_read2st64_b64 %0, %4 offset0:0 offset1:1\n\tds_read2st64_b64 %1, %4 offset0:2 offset1:3\n\t"
      "ds_read2st64_b64 %2, %4 offset0:4 offset1:5\n\tds_read2st64_b64 %3, %4 offset0:6 offset1:7\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(addr)
      : "memory");
}
__device__ __forceinline__ void vtile_read(unsigned addr, Vec2<float> (&o)[8]) {
  asm volatile(
      "ds_read2st64_b32 %0, %8 offset0:0 offset1:1\n\tds_read2st64_b32 %1, %8 offset0:2 offset1:3\n\t"
      "ds_read2st64_b32 %2, %8 offset0:4 offset1:5\n\tds_read2st64_b32 %3, %8 offset0:6 offset1:7\n\t"
      "ds_read2st64_b32 %4, %8 offset0:8 offset1:9\n\tds_read2st64_b32 %5, %8 offset0:10 offset1:11\n\t"
      "ds_read2st64_b32 %6, %8 offset0:12 offset1:13\n\tds_read2st64_b32 %7, %8 offset0:14 offset1:15\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(addr)
      : "memory");
}
__device__ __forceinline__ void vtile_read(unsigned addr, Vec2<float> (&o)[6]) {
  asm volatile(
      "ds_read2st64_b32 %0, %6 offset0:0 offset1:1\n\tds_read2st64_b32 %1, %6 offset0:2 offset1:3\n\t"
      "ds_read2st64_b32 %2, %6 offset0:4 offset1:5\n\tds_read2st64_b32 %3, %6 offset0:6 offset1:7\n\t"
      "ds_read2st64_b32 %4, %6 offset0:8 offset1:9\n\tds_read2st64_b32 %5, %6 offset0:10 offset1:11\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
      : "v"(addr)
      : "memory");
}
__device__ __forceinline__ void vtile_read(unsigned addr, Vec2<float> (&o)[4]) {
  asm volatile(
      "ds_read2st64_b32 %0, %4 offset0:0 offset1:1\n\tds_read2st64_b32 %1, %4 offset0:2 offset1:3\n\t"
      "ds_read2st64_b32 %2, %4 offset0:4 offset1:5\n\tds_read2st64_b32 %3, %4 offset0:6 offset1:7\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(addr)
      : "memory");
}

constexpr int VDMA_SLOTS = 2;  // LDS ring depth of the V tile (8 waves x 2 x 8 KiB = 128 KiB of the CU's 160 KiB)

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
#ifndef STREAM_TRACE
#define STREAM_TRACE 0  // 1: entry / first block / exit of every workgroup of cov_stream_kernel on the 100 MHz clock (tools/probes/stream_trace.py)
#endif
#if STREAM_TRACE && !defined(ASSX_PROBE_BUILD)
#error "STREAM_TRACE adds a debug entry point and stamps: build it with -DASSX_PROBE_BUILD into a probe library, never into libassx.so"
#endif
#if STREAM_TRACE
__device__ unsigned long long g_stream_trace[8 * 4096];
#endif

template <typename R, int M, int WK, bool K4, bool D2, int LS, int DXT, int DWT, int MINW = 1, bool VDMA = false>
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
  constexpr bool VD = VDMA && WK == WK_TV && K4;  // V tile through the LDS-direct ring (see VTileDma)
  static_assert(!VD || (LS == 1 && DXT >= 2), "the LDS ring path walks whole 64-frame blocks");
  using VT = VTileDma<R, (VD ? SPL * KU : 16)>;
  extern __shared__ __attribute__((aligned(16))) unsigned char vlds[];  // VD: VDMA_SLOTS tiles, wave-private
  const int lane = threadIdx.x & (WAVE - 1);
  const int fl = lane & (FB - 1);              // frame within the block
  const int s0 = (lane / FB) * SPL;            // first source of this lane group
  const int F = a.d.F, T = a.d.T, K = a.d.K, TBk = a.fp.len;
  const size_t FT = (size_t)F * T;
  const int g = workgroup_range((int)blockIdx.x, (int)gridDim.x, a.fp);
  if (g < 0) return;
  Cursor cc;                             // consume cursor
  int nblk;
  if (!flat_start(a.fp, g, cc.b, cc.f, cc.tb, nblk)) return;
#if STREAM_TRACE
  if (lane == 0 && g < 4096) {
    g_stream_trace[8 * g + 0] = __builtin_amdgcn_s_memrealtime();
    g_stream_trace[8 * g + 2] = (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) | ((unsigned long long)nblk << 8) |
                                ((unsigned long long)blockIdx.x << 32);
    g_stream_trace[8 * g + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID
    g_stream_trace[8 * g + 5] = __builtin_readcyclecounter();
  }
#endif
  const int bf_first = cc.b * F + cc.f;
  Cursor px = cc, pw = cc;               // prefetch cursors (X ring, weight ring)
  const bool ragged = (T % FB) != 0;     // only then can a lane fall beyond the last frame

  R acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0;

  // Buffer addressing: one descriptor per array (rebased per utterance), per-row strides as wave-uniform byte
  // offsets (SGPRs, constant for the whole kernel), ONE per-lane byte offset per block.
  const unsigned x_row = (unsigned)FT * (unsigned)sizeof(Cx<R>);  // X row m starts m * F * T complex in
  const unsigned t_row = (unsigned)T * (unsigned)sizeof(R);       // one (source, component) row of V / r(n,t)

  // `after`: a value the refill must not overtake (see order_after) -- the last accumulator fed from the slot
  auto issue_x = [&](const Cursor& c, Vec2<R>(&x)[M], R after) {
    const int t = c.tb * FB + fl;
    const unsigned tc = (unsigned)(t < T ? t : T - 1);
    const BufRsrc rx = make_rsrc(X + (size_t)c.b * M * FT);
    const unsigned voff = order_after(((unsigned)c.f * (unsigned)T + tc) * (unsigned)sizeof(Cx<R>), after);
    unsigned so = 0;
    const unsigned step = sgpr_opaque(x_row);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      x[m] = buf_ldv<R>(rx, voff, so);
      so += step;
    }
  };
  // VD path: the refill lands in the registers the slot already occupies (buf_ldv_tied); waits are explicit
  auto issue_x_tied = [&](const Cursor& c, Vec2<R>(&x)[M]) {
    const int t = c.tb * FB + fl;
    const unsigned tc = (unsigned)(t < T ? t : T - 1);
    const buf_u4 rx = make_rsrc_words(X + (size_t)c.b * M * FT, ~(size_t)0);
    const unsigned voff = ((unsigned)c.f * (unsigned)T + tc) * (unsigned)sizeof(Cx<R>);
    unsigned so = 0;
    const unsigned step = sgpr_opaque(x_row);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      buf_ldv_tied(x[m], rx, voff, so);
      so += step;
    }
  };
  auto issue_w = [&](const Cursor& c, R(&wv)[SPL][NWV]) {
    const int t = c.tb * FB + fl;
    const unsigned tc = (unsigned)(t < T ? t : T - 1);
    if (WK == WK_NT) {
      const BufRsrc rr = make_rsrc(rw + ((size_t)c.b * N + s0) * T);
      unsigned so = 0;
      const unsigned step = sgpr_opaque(t_row);
#pragma unroll
      for (int j = 0; j < SPL; ++j) {
        wv[j][0] = buf_ld<R>(rr, tc * (unsigned)sizeof(R), so);
        so += step;
      }
    } else if (WK == WK_NFT) {
      const BufRsrc rr = make_rsrc(rw + ((size_t)c.b * N + s0) * FT);
      const unsigned voff = ((unsigned)c.f * (unsigned)T + tc) * (unsigned)sizeof(R);
      unsigned so = 0;
      const unsigned step = sgpr_opaque((unsigned)FT * (unsigned)sizeof(R));
#pragma unroll
      for (int j = 0; j < SPL; ++j) {
        wv[j][0] = buf_ld<R>(rr, voff, so);
        so += step;
      }
    } else if (WK == WK_TV && K4) {
      const BufRsrc rv = make_rsrc(V + ((size_t)c.b * N + s0) * K * T);
      unsigned so = 0;
      const unsigned step = sgpr_opaque(t_row);
#pragma unroll
      for (int j = 0; j < SPL; ++j)
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) {
          wv[j][kk] = buf_ld<R>(rv, tc * (unsigned)sizeof(R), so);
          if (kk < K - 1 || kk == KU - 1) so += step;  // rows beyond K re-read row K-1 (their basis entry is 0)
        }
    }
  };
  VT vt;
  if (VD) vt.init(lane, K, t_row);
  // LDS byte address of this lane's column of slot 0
  const unsigned vlds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)vlds + (unsigned)lane * (unsigned)sizeof(R);
  auto issue_vtile = [&](const Cursor& c, int slot) {
    // exact size in the descriptor: a tile reaching past the end of V reads zeros
    const BufRsrc rv = make_rsrc_sized(V + (size_t)c.b * N * K * T, (size_t)(a.d.B - c.b) * N * K * T * sizeof(R));
#ifdef COV_VTILE_FIXED  // knock-out probe (tools/probes): every tile request reads frame block 0 -- the L2 traffic of the tiles goes, everything else stays
    vt.issue(rv, vlds + slot * VT::TILE_BYTES, 0u);
#else
    vt.issue(rv, vlds + slot * VT::TILE_BYTES, (unsigned)c.tb * (unsigned)VT::ROW_BYTES);
#endif
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
  if (VD) {  // the tiles go first: they must be OLDER than the X loads issued with them (in-order vmcnt)
#pragma unroll
    for (int j = 0; j < VDMA_SLOTS; ++j) {
      if (j < nblk) {
        issue_vtile(pw, j);
        advance(pw, TBk, F);
      }
    }
  }
  if (WK != WK_NONE && !VD) {  // weights before X, as in the steady state (the loop-header merge of the compiler's
                               // vmcnt model takes the stricter of the two orders -- forever)
#pragma unroll
    for (int j = 0; j < DWT; ++j) {
      if (j < nblk) {
        issue_w(pw, wq[j]);
        advance(pw, TBk, F);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < DXT; ++j) {
    if (j < nblk) {
      if (VD) {
#pragma unroll
        for (int m = 0; m < M; ++m) xq[j][m] = Vec2<R>{0, 0};
        issue_x_tied(px, xq[j]);
      } else {
        issue_x(px, xq[j], (R)0);
      }
      advance(px, TBk, F);
    }
  }
  load_basis_row(cc);

  // One block: consume ring slot J, refill it.  STEADY: every refill is known to be in range, so the block is
  // straight-line code -- the compiler's vmcnt bookkeeping stays exact (with conditional refills it merges the
  // "not issued" path in and drains the whole queue at every wait, which silently collapses the prefetch ring).
  auto block = [&](auto jc, auto mode, const int it) {
    constexpr int j = decltype(jc)::value;
    constexpr bool STEADY = decltype(mode)::value != 0;  // 0: drain, 1: steady state, 2: first trip (after the prologue)
    constexpr bool FIRST = decltype(mode)::value == 2;
    const Cursor cur = cc;
    advance(cc, TBk, F);
    const bool more = STEADY || it + 1 < nblk;

    // ---- consume block `cur` straight out of its ring slots: weights first, then the slot is refilled (no
    //      register copies); Hermitian products once, one weighted accumulate per source, then the X slot is
    //      refilled
    const int t = cur.tb * FB + fl;
    if (VD) {
      // tile `it` was requested two blocks ago; younger than it are at most the next tile and the X loads of
      // two blocks (fewer near the end of the range, where everything is drained instead)
      if (STEADY || it + DXT - 1 < nblk) wait_vmcnt<VT::NI + 2 * M>();
      else wait_vmcnt<0>();
      Vec2<R> vv[VT::NP];
      vtile_read(vlds0 + (unsigned)(it & (VDMA_SLOTS - 1)) * (unsigned)VT::TILE_BYTES, vv);
#pragma unroll
      for (int q = 0; q < SPL; ++q)
#pragma unroll
        for (int kk = 0; kk < KU; kk += 2) {
          wq[0][q][kk] = vv[(q * KU + kk) / 2].x;
          wq[0][q][kk + 1] = vv[(q * KU + kk) / 2].y;
        }
    }
    R wgt[SPL];
#if ASSX_BATCH_RCP
    R rfl[SPL];
#endif
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
            for (int kk = 0; kk < KU; ++kk) tv = fma(tbr[q][kk], wq[VD ? 0 : j % DWT][q][kk], tv);
          } else {
            const R* tbn = Tb + (((size_t)cur.b * N + s0 + q) * F + cur.f) * K;
            const R* vb = V + ((size_t)cur.b * N + s0 + q) * K * T + (t < T ? t : T - 1);
            for (int k = 0; k < K; ++k) tv = fma(tbn[k], vb[(size_t)k * T], tv);
          }
          r = D2 ? tv : powspec<R>(tv, a.p2d);  // R = (T V)^(2/domain), floored AFTER the power (ilrma.py:499-509)
        } else {
          r = wq[j % DWT][q][0];
        }
#if ASSX_BATCH_RCP
        rfl[q] = floor_eps<R>(r, a.eps);
#else
        wgt[q] = fast_rcp(floor_eps<R>(r, a.eps));
#endif
      }
    }
#if ASSX_BATCH_RCP
    if (WK != WK_NONE) batch_rcp<SPL>(rfl, wgt);
#endif
    if (ragged && cur.tb == TBk - 1) {  // wave-uniform: only the last block of a bin can hold idle lanes
#pragma unroll
      for (int q = 0; q < SPL; ++q)
        if (t >= T) wgt[q] = 0;
    }
    if (VD) {
      if (STEADY || it + VDMA_SLOTS < nblk) {  // this wave has consumed tile `it`: its slot takes tile it + 2
        issue_vtile(pw, it & (VDMA_SLOTS - 1));
        advance(pw, TBk, F);
      }
    } else if (WK != WK_NONE && (STEADY || it + DWT < nblk)) {
      issue_w(pw, wq[j % DWT]);
      advance(pw, TBk, F);
    }
    // each Hermitian product is formed once and fanned into every source's accumulator right away, so the
    // M*M products are never all live
    if (VD) {
      // slot `it` was refilled DXT blocks ago; younger: the X blocks of the DXT - 1 slots behind it and the tiles still
      // in flight -- tile it + 1 and the tile just requested, whatever DXT is (tiles up to `it` have been waited for)
      // (first trip: the slot was filled by the prologue, followed only by the other slots' X blocks)
      if (FIRST) wait_slot<(DXT - 1) * M + VT::NI>(xq[j]);
      else if (STEADY) wait_slot<(DXT - 1) * M + VDMA_SLOTS * VT::NI>(xq[j]);
      else wait_slot<0>(xq[j]);
    }
    {
      Cx<R> x[M];
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = tocx<R>(xq[j][m]);
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
    }
    // the refill must stay BELOW the last use of the slot: hoisted above it, the new block lands in other registers
    // and comes back as a copy at the loop back-edge -- behind a full vmcnt(0) drain
    __builtin_amdgcn_sched_barrier(0);
    if (VD) value_fence(acc);  // every accumulate that read the slot is above this line
    if (STEADY || it + DXT < nblk) {
      if (VD) issue_x_tied(px, xq[j]);
      else issue_x(px, xq[j], acc[NACC - 1]);
      advance(px, TBk, F);
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
  };

  // steady state: whole trips of DXT blocks whose refills (X: it + DXT; weights / tile: it + 2 at most) all exist
  int it0 = 0;
  if (2 * DXT <= nblk) {
    static_for<DXT>([&](auto jc) { block(jc, IntC<2>(), decltype(jc)::value); });
    it0 = DXT;
  }
#if STREAM_TRACE
  if (lane == 0 && g < 4096) g_stream_trace[8 * g + 3] = __builtin_amdgcn_s_memrealtime();
#endif
  for (; it0 + 2 * DXT <= nblk; it0 += DXT)
    static_for<DXT>([&](auto jc) { block(jc, IntC<1>(), it0 + decltype(jc)::value); });
  // drain: the last blocks, refills guarded
  for (; it0 < nblk; it0 += DXT)
    static_for<DXT>([&](auto jc) {
      if (it0 + decltype(jc)::value < nblk) block(jc, IntC<0>(), it0 + decltype(jc)::value);
    });
#if STREAM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && g < 4096) {
    g_stream_trace[8 * g + 1] = __builtin_amdgcn_s_memrealtime();
    g_stream_trace[8 * g + 6] = __builtin_readcyclecounter();
  }
#endif
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
  const int g = workgroup_range((int)blockIdx.x, (int)gridDim.x, a.fp);
  if (g < 0) return;
  Cursor c0;
  int nblk;
  if (!flat_start(a.fp, g, c0.b, c0.f, c0.tb, nblk)) return;
  const int bf_first = c0.b * F + c0.f;
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
            for (int m = 0; m < M; ++m) demix_mac(y, w[n][m], x[m]);
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

// K <= KU form of basis_stream_kernel with the activation tile through the LDS-direct ring (VTileDma) and the X
// slots refilled in place (buf_ldv_tied): same partition, same records, same arithmetic.  All VMEM waits are
// explicit; per block the issue order is [tile it+2] ... [X it+DXT], which fixes the vmcnt distances used below.
// LOSS (domain 2, Gaussian model): the pass also accumulates the data term of the negative log-likelihood of the
// model it reads -- sum_{n,t} P/R + log R with the very y = W x and R = max(T V, eps) it forms anyway (ilrma.py:672-675)
// -- so recording the loss of iteration i costs ~16 instructions per block of iteration i+1's basis pass instead of
// a pass over X of its own.  One partial per (utterance, covering workgroup): lpart[b][g - first workgroup of b].
template <typename R, int M, bool D2, int DXT, int MINW, bool TD, bool LOSS = false>
__global__ void __launch_bounds__(64, MINW)
    basis_stream_vd_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W, const R* __restrict__ Tb,
                           const R* __restrict__ V, R* __restrict__ part, NmfArgs<R> a, double* __restrict__ lpart,
                           int lstride) {
  static_assert(!LOSS || (D2 && !TD), "the fused loss is the domain-2 Gaussian one");
  constexpr int N = M;
  constexpr int NACC = N * KU * 2;
  constexpr int NV = next_pow2_c(NACC);
  static_assert(NV <= WAVE && DXT >= 2, "accumulators / ring depth");
  using VT = VTileDma<R, N * KU>;
  extern __shared__ __attribute__((aligned(16))) unsigned char vlds[];
  const int lane = threadIdx.x & (WAVE - 1);
  const int F = a.d.F, T = a.d.T, K = a.d.K, TBk = a.fp.len;
  const size_t FT = (size_t)F * T;
  const int g = workgroup_range((int)blockIdx.x, (int)gridDim.x, a.fp);
  if (g < 0) return;
  Cursor cc;
  int nblk;
  if (!flat_start(a.fp, g, cc.b, cc.f, cc.tb, nblk)) return;
  const int bf_first = cc.b * F + cc.f;
  Cursor px = cc, pw = cc;
  const unsigned x_row = (unsigned)FT * (unsigned)sizeof(Cx<R>);
  const unsigned t_row = (unsigned)T * (unsigned)sizeof(R);

  R acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0;
  double lacc = 0.0, lm = 1.0;  // LOSS: sum P/R, and sum log R carried as mantissa product + exponent
  int le = 0;

  auto issue_x_tied = [&](const Cursor& c, Vec2<R>(&x)[M]) {
    const int t = c.tb * WAVE + lane;
    const unsigned tc = (unsigned)(t < T ? t : T - 1);
    const buf_u4 rx = make_rsrc_words(X + (size_t)c.b * M * FT, ~(size_t)0);
    const unsigned voff = ((unsigned)c.f * (unsigned)T + tc) * (unsigned)sizeof(Cx<R>);
    unsigned so = 0;
    const unsigned step = sgpr_opaque(x_row);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      buf_ldv_tied(x[m], rx, voff, so);
      so += step;
    }
  };
  VT vt;
  vt.init(lane, K, t_row);
  const unsigned vlds0 =
      (unsigned)(size_t)(__attribute__((address_space(3))) void*)vlds + (unsigned)lane * (unsigned)sizeof(R);
  auto issue_vtile = [&](const Cursor& c, int slot) {
    const BufRsrc rv = make_rsrc_sized(V + (size_t)c.b * N * K * T, (size_t)(a.d.B - c.b) * N * K * T * sizeof(R));
    vt.issue(rv, vlds + slot * VT::TILE_BYTES, (unsigned)c.tb * (unsigned)VT::ROW_BYTES);
  };
  Cx<R> w[N][M];  // demixing rows of the bin: wave-uniform, SGPRs (scalar loads)
  R tbr[N][KU];   // basis rows of the bin: wave-uniform too, but held in VGPRs (broadcast vector loads) -- together
                  // with w they would overflow the scalar file and come back as two v_readlane per use
  // ASSX_W_VGPR_ROWS: the last so many rows of w are held in VGPRs as well (float64 only: a row is 16 SGPRs).  The kernel
  // sits at the 106-SGPR ceiling and the spills come back as v_readlane_b32 -- ~45 per trip of the steady loop with all of
  // w in SGPRs, a third of that with two rows in VGPRs: 295 / 210 / 88 / 40 v_readlane for 0 / 1 / 2 / 3 rows at 202 / 225 /
  // 249 / 256 VGPRs, kernel 50.6 / 50.7 / 49.1 / 49.3 us (profiles/r04_sched_flags.txt).  Same values, same arithmetic.
#ifndef ASSX_W_VGPR_ROWS
#define ASSX_W_VGPR_ROWS 2
#endif
  // (the plain domain-2 variant only: with the loss or the t-ILRMA statistic on board the extra 32 VGPRs do not fit)
  constexpr int WVN = (sizeof(R) == 8 && D2 && !TD && !LOSS && MINW == 2) ? (ASSX_W_VGPR_ROWS < N ? ASSX_W_VGPR_ROWS : N) : 0;
  const unsigned zero_v = order_after(0u, lane);  // a zero the compiler cannot prove uniform
  auto load_rows = [&](const Cursor& cu) {  // once per bin
    const Cx<R>* wp = W + ((size_t)cu.b * F + cu.f) * (N * M);
#pragma unroll
    for (int n = 0; n < N - WVN; ++n)
#pragma unroll
      for (int m = 0; m < M; ++m) w[n][m] = wp[n * M + m];
    if (WVN > 0) {
      const BufRsrc rw = make_rsrc(reinterpret_cast<const R*>(W + (size_t)cu.b * F * (N * M)));
      const unsigned so = (unsigned)(cu.f * (N * M)) * (unsigned)sizeof(Cx<R>);
#pragma unroll
      for (int n = N - WVN; n < N; ++n)
#pragma unroll
        for (int m = 0; m < M; ++m) {
          w[n][m].x = buf_ld<R>(rw, zero_v + (unsigned)((n * M + m) * 2 * (int)sizeof(R)), so);
          w[n][m].y = buf_ld<R>(rw, zero_v + (unsigned)(((n * M + m) * 2 + 1) * (int)sizeof(R)), so);
        }
    }
    const BufRsrc rt = make_rsrc(Tb + (size_t)cu.b * N * F * K);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const unsigned so = (unsigned)((n * F + cu.f) * K) * (unsigned)sizeof(R);
#pragma unroll
      for (int kk = 0; kk < KU; ++kk) {
        const R v = buf_ld<R>(rt, zero_v + (unsigned)((kk < K ? kk : 0) * (int)sizeof(R)), so);
        tbr[n][kk] = (kk < K) ? v : (R)0;
      }
    }
  };

  Vec2<R> xq[DXT][M];
#pragma unroll
  for (int j = 0; j < VDMA_SLOTS; ++j) {  // tiles first: older than every X load issued with them
    if (j < nblk) {
      issue_vtile(pw, j);
      advance(pw, TBk, F);
    }
  }
#pragma unroll
  for (int j = 0; j < DXT; ++j) {
    if (j < nblk) {
#pragma unroll
      for (int m = 0; m < M; ++m) xq[j][m] = Vec2<R>{0, 0};
      issue_x_tied(px, xq[j]);
      advance(px, TBk, F);
    }
  }
  load_rows(cc);

  auto block = [&](auto jc, auto mode, const int it) {
    constexpr int j = decltype(jc)::value;
    constexpr bool STEADY = decltype(mode)::value != 0;  // 0: drain, 1: steady state, 2: first trip
    constexpr bool FIRST = decltype(mode)::value == 2;
    const Cursor cur = cc;
    advance(cc, TBk, F);
    const bool more = STEADY || it + 1 < nblk;
    const int t = cur.tb * WAVE + lane;

    // tile `it` (requested two blocks ago): younger are at most the next tile and two X blocks
    if (STEADY || it + DXT - 1 < nblk) wait_vmcnt<VT::NI + 2 * M>();
    else wait_vmcnt<0>();
    Vec2<R> vv[VT::NP];
    vtile_read(vlds0 + (unsigned)(it & (VDMA_SLOTS - 1)) * (unsigned)VT::TILE_BYTES, vv);
    if (STEADY || it + VDMA_SLOTS < nblk) {  // the slot is free again: request tile it + 2
      issue_vtile(pw, it & (VDMA_SLOTS - 1));
      advance(pw, TBk, F);
    }
    // X slot `it` (refilled DXT blocks ago)
    if (FIRST) wait_slot<(DXT - 1) * M + VT::NI>(xq[j]);
    else if (STEADY) wait_slot<(DXT - 1) * M + VDMA_SLOTS * VT::NI>(xq[j]);
    else wait_slot<0>(xq[j]);

    Cx<R> x[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = tocx<R>(xq[j][m]);
    const R live = (t < T) ? (R)1 : (R)0;
    double lterm = 0.0, lprod = 1.0;
#if ASSX_BATCH_RCP
    R tvn[N], invn[N];  // every source's floored variance first: their reciprocals share one hardware reciprocal (batch_rcp)
#pragma unroll
    for (int n = 0; n < N; ++n) {
      R tv = 0;
#pragma unroll
      for (int kk = 0; kk < KU; kk += 2) {
        tv = fma(tbr[n][kk], vv[(n * KU + kk) / 2].x, tv);
        tv = fma(tbr[n][kk + 1], vv[(n * KU + kk) / 2].y, tv);
      }
      tvn[n] = floor_eps<R>(tv, a.eps);
    }
    batch_rcp<N>(tvn, invn);
#endif
#pragma unroll
    for (int n = 0; n < N; ++n) {
      Cx<R> y = cmake<R>(0, 0);
#pragma unroll
      for (int m = 0; m < M; ++m) demix_mac(y, w[n][m], x[m]);
      R P = cabs2(y);
      R v[KU];
#pragma unroll
      for (int kk = 0; kk < KU; kk += 2) {
        v[kk] = vv[(n * KU + kk) / 2].x;
        v[kk + 1] = vv[(n * KU + kk) / 2].y;
      }
#if ASSX_BATCH_RCP
      const R tv = tvn[n];
      if (TD) P = t_harmonic<R>(P, tv, a.nu);
      R inv = invn[n];
#else
      R tv = 0;
#pragma unroll
      for (int kk = 0; kk < KU; ++kk) tv = fma(tbr[n][kk], v[kk], tv);
      tv = floor_eps<R>(tv, a.eps);
      if (TD) P = t_harmonic<R>(P, tv, a.nu);
      R inv = fast_rcp(tv);                                 // TV_inverse
#endif
      R D = D2 ? P * inv * inv : P / powspec<R>(tv, a.p1);   // division = P / TV**((d+2)/d)
      if (LOSS) {
        lterm += (double)(P * inv);
        lprod *= (double)tv;
      }
      inv *= live;  // frames past the end of the utterance contribute nothing (one multiply, not four selects)
      D *= live;
#pragma unroll
      for (int kk = 0; kk < KU; ++kk) {
        acc[(n * KU + kk) * 2 + 0] = fma(D, v[kk], acc[(n * KU + kk) * 2 + 0]);
        acc[(n * KU + kk) * 2 + 1] = fma(inv, v[kk], acc[(n * KU + kk) * 2 + 1]);
      }
    }
    if (LOSS) {
      if (t < T) {
        lacc += lterm;
        int e;
        lm = frexp(lm * lprod, &e);
        le += e;
      }
      asm volatile("" : "+v"(lacc), "+v"(lm));
    }
    value_fence(acc);  // every read of the slot is above this line
    if (STEADY || it + DXT < nblk) {
      issue_x_tied(px, xq[j]);
      advance(px, TBk, F);
    }

    if (cc.tb == 0 || !more) {
      const R tot = wave_reduce_scatter<R, NV>(acc);
      const int i = scatter_index<NV>();
      const int slot = cur.b * F + cur.f - bf_first;
      const int n = i / (2 * KU), k = (i >> 1) % KU;
      if (scatter_leader<NV>() && i < NACC && k < K)
        part[(((size_t)g * a.fp.S + slot) * N + n) * (size_t)(2 * K) + k * 2 + (i & 1)] = tot;
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = 0;
      if (LOSS && (cc.b != cur.b || !more)) {  // leaving utterance cur.b: publish this workgroup's share of its loss
        double tot = lacc + (double)le * 0.6931471805599453 + log(lm);
        tot = wave_allreduce_sum<double>(tot);
        if (lane == 0) lpart[(size_t)cur.b * lstride + (g - cur.b * a.fp.Gu)] = tot;
        lacc = 0.0;
        lm = 1.0;
        le = 0;
      }
      if (more) load_rows(cc);
    }
  };

  int it0 = 0;
  if (2 * DXT <= nblk) {
    static_for<DXT>([&](auto jc) { block(jc, IntC<2>(), decltype(jc)::value); });
    it0 = DXT;
  }
  for (; it0 + 2 * DXT <= nblk; it0 += DXT)
    static_for<DXT>([&](auto jc) { block(jc, IntC<1>(), it0 + decltype(jc)::value); });
  for (; it0 < nblk; it0 += DXT)
    static_for<DXT>([&](auto jc) {
      if (it0 + decltype(jc)::value < nblk) block(jc, IntC<0>(), it0 + decltype(jc)::value);
    });
}

template <int M, typename R>
__device__ __forceinline__ double neg2T_logabsdet(const Cx<R>* __restrict__ W, size_t bf, int T);  // assx_group_linalg.hpp

// s += p[threadIdx.x], p[threadIdx.x + 256], ... (a 256-thread workgroup; ascending order per thread, as the plain loop):
// eight loads in flight per trip instead of one round trip per element -- the loss sums are ~3000 doubles per utterance read
// by ONE workgroup, 13 dependent L2 latencies per thread in the plain form (8.7 against 5.2 us for the launch that hosts it)
__device__ __forceinline__ void strided_sum_256(const double* __restrict__ p, int n, double& s) {
  for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u;
      v[u] = i < n ? p[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
}

// T *= (num / max(den, eps)) ** (d/(d+2))      (ilrma.py:417-419)
// ML > 0 (round 6; the loss folded into the basis pass): workgroups past the `nb_main` that do the update write the F
// log-det terms -2 T log|det W_f| of the loss at lpart[b][ncov + f] (ilrma.py:675; ML = the channel count) -- what
// logdet_kernel did in a launch of its own ahead of the pass.  W does not move between the two places; same values.
template <typename R, int ML = 0>
__global__ void __launch_bounds__(256) basis_stream_finalize_kernel(const R* __restrict__ part, R* __restrict__ Tb,
                                                                   int B, int N, int F, int K, FlatPart fp, R eps,
                                                                   PowSpec p2, unsigned src_mask, int nb_main = 0,
                                                                   const Cx<R>* __restrict__ W = nullptr,
                                                                   double* __restrict__ lpart = nullptr, int lstride = 0,
                                                                   int ncov = 0, int T = 0) {
  if constexpr (ML > 0) {
    if ((int)blockIdx.x >= nb_main) {
      const int i = ((int)blockIdx.x - nb_main) * (int)blockDim.x + (int)threadIdx.x;
      if (i < B * F) {
        const int b = i / F, f = i - b * F;
        lpart[(size_t)b * lstride + ncov + f] = neg2T_logabsdet<ML, R>(W, (size_t)i, T);
      }
      return;
    }
  }
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * F * K;
  if (idx >= total) return;
  const int k = idx % K;
  const int f = (idx / K) % F;
  const int n = (idx / ((size_t)K * F)) % N;
  if (!((src_mask >> n) & 1u)) return;  // pairwise source-model update: only the selected sources move
  const int b = idx / ((size_t)K * F * N);
  const long long j = (long long)b * F + f;
  int g_lo, g_hi;
  flat_cover(fp, j, g_lo, g_hi);
  const R told = Tb[idx];  // requested with the records: one round trip for the whole kernel instead of one per load
  R num = 0, den = 0;
  constexpr int RC = 4;  // a bin is covered by 2-3 workgroups at benchmark size; same summation order
  for (int g0 = g_lo; g0 <= g_hi; g0 += RC) {
    R vn[RC], vd[RC];
#pragma unroll
    for (int c = 0; c < RC; ++c) {
      const int g = min(g0 + c, g_hi);
      const int slot = flat_slot(fp, j, g);
      const R* p = part + (((size_t)g * fp.S + slot) * N + n) * (size_t)(2 * K) + k * 2;
      vn[c] = p[0];
      vd[c] = p[1];
    }
#pragma unroll
    for (int c = 0; c < RC; ++c)
      if (g0 + c <= g_hi) {
        num += vn[c];
        den += vd[c];
      }
  }
  den = floor_eps<R>(den, eps);
  Tb[idx] = told * powspec<R>(num / den, p2);
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
  const int g = xcd_local_range((int)blockIdx.x, (int)gridDim.x), b = blockIdx.y;
  long long q0, q1;
  flat_range(a.fp, g, q0, q1);
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
            for (int m = 0; m < M; ++m) demix_mac(y, w[n][m], x[m]);
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

// K <= KU form of loss_stream_kernel on the LDS-direct activation ring (same pipeline as basis_stream_vd_kernel).
template <typename R, int M, bool D2, int DXT, int MINW, bool TD>
__global__ void __launch_bounds__(64, MINW)
    loss_stream_vd_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W, const R* __restrict__ Tb,
                          const R* __restrict__ V, double* __restrict__ lpart, int lstride, NmfArgs<R> a, PowSpec p2d) {
  constexpr int N = M;
  static_assert(DXT >= 2, "ring depth");
  using VT = VTileDma<R, N * KU>;
  extern __shared__ __attribute__((aligned(16))) unsigned char vlds[];
  const int lane = threadIdx.x & (WAVE - 1);
  const int F = a.d.F, T = a.d.T, K = a.d.K, TBk = a.fp.len;
  const size_t FT = (size_t)F * T;
  const int g = xcd_local_range((int)blockIdx.x, (int)gridDim.x), b = blockIdx.y;
  long long q0, q1;
  flat_range(a.fp, g, q0, q1);
  double acc = 0.0;
  double lm = 1.0, tm = 1.0;  // running mantissa products of R and of 1 + (2/nu) P/R (see loss_stream_kernel)
  int le = 0, te = 0;
  if (q0 < q1) {
    const int nblk = (int)(q1 - q0);
    Cursor cc;
    cc.f = (int)(q0 / TBk);
    cc.tb = (int)(q0 - (long long)cc.f * TBk);
    cc.b = b;
    Cursor px = cc, pw = cc;
    const unsigned x_row = (unsigned)FT * (unsigned)sizeof(Cx<R>);
    const unsigned t_row = (unsigned)T * (unsigned)sizeof(R);
    const buf_u4 rx = make_rsrc_words(X + (size_t)b * M * FT, ~(size_t)0);
    const BufRsrc rv = make_rsrc_sized(V + (size_t)b * N * K * T, (size_t)(a.d.B - b) * N * K * T * sizeof(R));
    auto issue_x_tied = [&](const Cursor& c, Vec2<R>(&x)[M]) {
      const int t = c.tb * WAVE + lane;
      const unsigned tc = (unsigned)(t < T ? t : T - 1);
      const unsigned voff = ((unsigned)c.f * (unsigned)T + tc) * (unsigned)sizeof(Cx<R>);
      unsigned so = 0;
      const unsigned step = sgpr_opaque(x_row);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        buf_ldv_tied(x[m], rx, voff, so);
        so += step;
      }
    };
    VT vt;
    vt.init(lane, K, t_row);
    const unsigned vlds0 =
        (unsigned)(size_t)(__attribute__((address_space(3))) void*)vlds + (unsigned)lane * (unsigned)sizeof(R);
    auto issue_vtile = [&](const Cursor& c, int slot) {
      vt.issue(rv, vlds + slot * VT::TILE_BYTES, (unsigned)c.tb * (unsigned)VT::ROW_BYTES);
    };
    Cx<R> w[N][M];
    R tbr[N][KU];
    const unsigned zero_v = order_after(0u, lane);
    auto load_rows = [&](const Cursor& cu) {  // once per bin: w -> SGPRs, basis rows -> VGPRs (see basis_stream_vd_kernel)
      const Cx<R>* wp = W + ((size_t)b * F + cu.f) * (N * M);
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int m = 0; m < M; ++m) w[n][m] = wp[n * M + m];
      const BufRsrc rt = make_rsrc(Tb + (size_t)b * N * F * K);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const unsigned so = (unsigned)((n * F + cu.f) * K) * (unsigned)sizeof(R);
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) {
          const R v = buf_ld<R>(rt, zero_v + (unsigned)((kk < K ? kk : 0) * (int)sizeof(R)), so);
          tbr[n][kk] = (kk < K) ? v : (R)0;
        }
      }
    };
    Vec2<R> xq[DXT][M];
#pragma unroll
    for (int j = 0; j < VDMA_SLOTS; ++j) {
      if (j < nblk) {
        issue_vtile(pw, j);
        advance(pw, TBk, F);
      }
    }
#pragma unroll
    for (int j = 0; j < DXT; ++j) {
      if (j < nblk) {
#pragma unroll
        for (int m = 0; m < M; ++m) xq[j][m] = Vec2<R>{0, 0};
        issue_x_tied(px, xq[j]);
        advance(px, TBk, F);
      }
    }
    load_rows(cc);

    auto block = [&](auto jc, auto mode, const int it) {
      constexpr int j = decltype(jc)::value;
      constexpr bool STEADY = decltype(mode)::value != 0;
      constexpr bool FIRST = decltype(mode)::value == 2;
      const Cursor cur = cc;
      advance(cc, TBk, F);
      const int t = cur.tb * WAVE + lane;
      if (STEADY || it + DXT - 1 < nblk) wait_vmcnt<VT::NI + 2 * M>();
      else wait_vmcnt<0>();
      Vec2<R> vv[VT::NP];
      vtile_read(vlds0 + (unsigned)(it & (VDMA_SLOTS - 1)) * (unsigned)VT::TILE_BYTES, vv);
      if (STEADY || it + VDMA_SLOTS < nblk) {
        issue_vtile(pw, it & (VDMA_SLOTS - 1));
        advance(pw, TBk, F);
      }
      if (FIRST) wait_slot<(DXT - 1) * M + VT::NI>(xq[j]);
      else if (STEADY) wait_slot<(DXT - 1) * M + VDMA_SLOTS * VT::NI>(xq[j]);
      else wait_slot<0>(xq[j]);
      Cx<R> x[M];
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = tocx<R>(xq[j][m]);
      double term = 0.0, rprod = 1.0, tprod = 1.0;
#pragma unroll
      for (int n = 0; n < N; ++n) {
        Cx<R> y = cmake<R>(0, 0);
#pragma unroll
        for (int m = 0; m < M; ++m) demix_mac(y, w[n][m], x[m]);
        R tv = 0;
#pragma unroll
        for (int kk = 0; kk < KU; kk += 2) {
          tv = fma(tbr[n][kk], vv[(n * KU + kk) / 2].x, tv);
          tv = fma(tbr[n][kk + 1], vv[(n * KU + kk) / 2].y, tv);
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
      asm volatile("" : "+v"(acc), "+v"(lm), "+v"(tm));  // the slot has been read: the refill may follow
      if (STEADY || it + DXT < nblk) {
        issue_x_tied(px, xq[j]);
        advance(px, TBk, F);
      }
      if (cc.tb == 0 && (STEADY || it + 1 < nblk)) load_rows(cc);
    };
    int it0 = 0;
    if (2 * DXT <= nblk) {
      static_for<DXT>([&](auto jc) { block(jc, IntC<2>(), decltype(jc)::value); });
      it0 = DXT;
    }
    for (; it0 + 2 * DXT <= nblk; it0 += DXT)
      static_for<DXT>([&](auto jc) { block(jc, IntC<1>(), it0 + decltype(jc)::value); });
    for (; it0 < nblk; it0 += DXT)
      static_for<DXT>([&](auto jc) {
        if (it0 + decltype(jc)::value < nblk) block(jc, IntC<0>(), it0 + decltype(jc)::value);
      });
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
constexpr int ACT_NH = 4;

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
  const int g = workgroup_range((int)blockIdx.x, (int)gridDim.x, a.fp);
  if (g < 0) return;
  long long q0, q1;
  flat_range(a.fp, g, q0, q1);
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

// K <= KU form of act_stream_kernel.  Every block is a new bin, so the bin's wave-uniform constants -- the N x M
// demixing rows and the N x K basis rows, 384 bytes in f64 -- change every block: as scalar loads they cannot be
// prefetched (two sets do not fit the scalar file; one set already spills) and their latency lands on every block.
// Here they ride an LDS ring: one dword-per-lane LDS-direct load per array and bin, requested DXT bins ahead together
// with the bin's X block, read back as broadcast LDS loads.  X slots are refilled in place; the only VMEM wait per
// block is explicit: per bin the issue order is [W row][T rows][X], so a slot is DXT-1 whole bins old when consumed.
constexpr int ACT_RING_SLOT = 512;  // bytes per bin: W rows at 0 (<= 256), T rows at 256 (<= 128)

template <typename R, int M, bool D2, int DXT, int MINW, bool TD>
__global__ void __launch_bounds__(64 * ACT_NH, MINW)
    act_stream_vd_kernel(const Cx<R>* __restrict__ X, const Cx<R>* __restrict__ W, const R* __restrict__ Tb,
                         const R* __restrict__ V, R* __restrict__ part, NmfArgs<R> a) {
  constexpr int N = M;
  constexpr int NACC = N * KU * 2;
  constexpr int NISS = 2 + M;  // VMEM instructions per bin
  static_assert(N * M * 2 * (int)sizeof(R) <= 256 && N * KU * (int)sizeof(R) <= 256, "ring slot layout");
  __shared__ R lds[(ACT_NH - 1) * NACC * WAVE];
  __shared__ __attribute__((aligned(16))) unsigned char ring[ACT_NH * DXT * ACT_RING_SLOT];
  const int lane = threadIdx.x & (WAVE - 1);
  const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int F = a.d.F, T = a.d.T, K = a.d.K;
  const size_t FT = (size_t)F * T;
  const int g = workgroup_range((int)blockIdx.x, (int)gridDim.x, a.fp);
  if (g < 0) return;
  // this workgroup's items (frame block, bin) in 32-bit arithmetic (flat_start): the 64-bit divisions of a global item
  // index were ~2 us at the head of the kernel, before its first load (in-kernel timestamps)
  int b, tb_first, f_first, nitems;
  if (!flat_start(a.fp, g, b, tb_first, f_first, nitems)) return;
  const int tb_last = tb_first + (f_first + nitems - 1) / F;
  const int f_end = f_first + nitems - (tb_last - tb_first) * F;  // one past the last bin, in the last frame block
  const unsigned x_row = (unsigned)FT * (unsigned)sizeof(Cx<R>);
  unsigned char* myring = ring + h * (DXT * ACT_RING_SLOT);
  const unsigned ring0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)myring;
  // per-lane source offsets of the two LDS-direct loads (lanes past the end re-read the last dword)
  const int w_dwords = N * M * 2 * (int)sizeof(R) / 4;
  const unsigned voff_w = (unsigned)(lane < w_dwords ? lane : w_dwords - 1) * 4u;
  // basis rows land as [n][KU]: lane = (n, kk, dword of the value); components kk >= K and sources n >= N point
  // past the end of the descriptor and read zeros, so the consumer needs no `kk < K` tests
  constexpr int DPV = (int)sizeof(R) / 4;  // dwords per value
  const int tn = lane / (KU * DPV), tkk = (lane / DPV) % KU;
  const unsigned voff_t = (tn < N && tkk < K)
                              ? ((unsigned)tn * (unsigned)F * (unsigned)K + (unsigned)tkk) * (unsigned)sizeof(R) +
                                    (unsigned)(lane % DPV) * 4u
                              : 0xfffffff0u;

  for (int tb = tb_first; tb <= tb_last; ++tb) {  // segments of the range, one frame block each
    const int fa = tb == tb_first ? f_first : 0;
    const int fb = tb == tb_last ? f_end : F;
    const int t = tb * WAVE + lane;
    const unsigned tc = (unsigned)(t < T ? t : T - 1);
    R v[N][KU];
    {
      const BufRsrc rv = make_rsrc(V + (size_t)b * N * K * T);
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int kk = 0; kk < KU; ++kk)
          v[n][kk] = buf_ld<R>(rv, tc * (unsigned)sizeof(R),
                               (unsigned)(n * K + (kk < K ? kk : K - 1)) * (unsigned)T * (unsigned)sizeof(R));
    }
    R acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0;

    const buf_u4 rx = make_rsrc_words(X + (size_t)b * M * FT, ~(size_t)0);
    const buf_u4 rw = make_rsrc_words(W + (size_t)b * F * N * M, ~(size_t)0);
    const buf_u4 rt = make_rsrc_words(Tb + (size_t)b * N * F * K, (size_t)N * F * K * sizeof(R));  // exact: OOB = 0
    // this wave's bins: fa + h, fa + h + ACT_NH, ...
    const int nmine = (fb - fa - h + ACT_NH - 1) / ACT_NH;
    auto issue_bin = [&](int f, int slot, Vec2<R>(&x)[M]) {
      const unsigned la = ring0 + (unsigned)slot * (unsigned)ACT_RING_SLOT;
      buf_dword_to_lds(la, rw, voff_w, (unsigned)f * (unsigned)(N * M * 2 * (int)sizeof(R)));
      buf_dword_to_lds(la + 256u, rt, voff_t, (unsigned)f * (unsigned)K * (unsigned)sizeof(R));
      const unsigned voff = ((unsigned)f * (unsigned)T + tc) * (unsigned)sizeof(Cx<R>);
      unsigned so = 0;
      const unsigned step = sgpr_opaque(x_row);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        buf_ldv_tied(x[m], rx, voff, so);
        so += step;
      }
    };
    Vec2<R> xq[DXT][M];
#pragma unroll
    for (int j = 0; j < DXT; ++j)
      if (j < nmine) {
#pragma unroll
        for (int m = 0; m < M; ++m) xq[j][m] = Vec2<R>{0, 0};
        issue_bin(fa + h + j * ACT_NH, j, xq[j]);
      }
    // the activation columns must have landed before the loop starts: the compiler would otherwise wait for them
    // with vmcnt(0) inside the loop -- on every trip, since its model merges the loop entry with the back-edge --
    // and drain the ring (whose loads it cannot see) each time.  The wait sits AFTER the ring's first requests, so the
    // columns and the first bins travel together: one memory round trip per segment instead of two in a row
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int n = 0; n < N; ++n) value_fence(v[n]);

    auto block = [&](auto jc, auto steady, const int it) {
      constexpr int j = decltype(jc)::value;
      constexpr bool STEADY = decltype(steady)::value;
      const int f = fa + h + it * ACT_NH;
      if (STEADY) wait_slot<(DXT - 1) * NISS>(xq[j]);  // the bin's W / T rows are older than its X block
      else wait_slot<0>(xq[j]);
      Cx<R> x[M];
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = tocx<R>(xq[j][m]);
      const Cx<R>* lw = reinterpret_cast<const Cx<R>*>(myring + j * ACT_RING_SLOT);
      const R* lt = reinterpret_cast<const R*>(myring + j * ACT_RING_SLOT + 256);
#if ASSX_BATCH_RCP
      R tvn[N], invn[N];  // every source's floored variance first: their reciprocals share one hardware reciprocal
#pragma unroll
      for (int n = 0; n < N; ++n) {
        R tv = 0;
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) tv = fma(lt[n * KU + kk], v[n][kk], tv);
        tvn[n] = floor_eps<R>(tv, a.eps);
      }
      batch_rcp<N>(tvn, invn);
      asm volatile("" ::: "memory");  // the basis rows are read again below, one source at a time (not kept in 32 VGPRs)
#endif
#pragma unroll
      for (int n = 0; n < N; ++n) {
        Cx<R> y = cmake<R>(0, 0);
#pragma unroll
        for (int m = 0; m < M; ++m) cfma(y, lw[n * M + m], x[m]);
        R P = cabs2(y);
        R tk[KU];
#if ASSX_BATCH_RCP
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) tk[kk] = lt[n * KU + kk];
        const R tv = tvn[n];
        if (TD) P = t_harmonic<R>(P, tv, a.nu);
        const R inv = invn[n];
#else
        R tv = 0;
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) {
          tk[kk] = lt[n * KU + kk];
          tv = fma(tk[kk], v[n][kk], tv);
        }
        tv = floor_eps<R>(tv, a.eps);
        if (TD) P = t_harmonic<R>(P, tv, a.nu);
        const R inv = fast_rcp(tv);
#endif
        const R D = D2 ? P * inv * inv : P / powspec<R>(tv, a.p1);
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) {
          acc[(n * KU + kk) * 2 + 0] = fma(tk[kk], D, acc[(n * KU + kk) * 2 + 0]);
          acc[(n * KU + kk) * 2 + 1] = fma(tk[kk], inv, acc[(n * KU + kk) * 2 + 1]);
        }
        // one source's rows live at a time: hoisting every LDS read to the top of the block costs 96 VGPRs
        if (n + 1 < N && (n & 1)) __builtin_amdgcn_sched_barrier(0);
      }
      value_fence(acc);  // every read of the slot (registers and LDS) is above this line
      if (STEADY || it + DXT < nmine) issue_bin(f + DXT * ACT_NH, j, xq[j]);
    };
    int it0 = 0;
    for (; it0 + 2 * DXT <= nmine; it0 += DXT)
      static_for<DXT>([&](auto jc) { block(jc, BoolC<true>(), it0 + decltype(jc)::value); });
    for (; it0 < nmine; it0 += DXT)
      static_for<DXT>([&](auto jc) {
        if (it0 + decltype(jc)::value < nmine) block(jc, BoolC<false>(), it0 + decltype(jc)::value);
      });

    // combine the ACT_NH bin streams through LDS, then one coalesced partial record per (n, k, num|den)
    if (h > 0) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) lds[((h - 1) * NACC + i) * WAVE + lane] = acc[i];
    }
    __syncthreads();
    if (h == 0) {
      const int slot = tb - tb_first;
      R* out = part + ((size_t)g * a.fp.S + slot) * (size_t)(N * 2 * K) * WAVE + lane;
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        R tot = acc[i];
#pragma unroll
        for (int hh = 1; hh < ACT_NH; ++hh) tot += lds[((hh - 1) * NACC + i) * WAVE + lane];
        const int n = i / (2 * KU), k = (i >> 1) % KU;
        if (k < K) out[(size_t)(n * 2 * K + k * 2 + (i & 1)) * WAVE] = tot;
      }
    }
    __syncthreads();
  }
}

// V *= (num / max(den, eps)) ** (d/(d+2))      (ilrma.py:426-428)
// lpart != nullptr (round 6; the loss folded into the basis pass): workgroup nb_main + b completes utterance b's loss --
// the data-term partials the basis pass left at lpart[b][0 .. ncov) plus the F log-det terms behind them, in the order of
// ilrma_loss_finish_kernel (the launch this replaces: same bits).
template <typename R>
__global__ void __launch_bounds__(256) act_stream_finalize_kernel(const R* __restrict__ part, R* __restrict__ V, int B,
                                                                 int N, int F, int K, int T, FlatPart fp, R eps,
                                                                 PowSpec p2, unsigned src_mask, int nb_main = 0,
                                                                 const double* __restrict__ lpart = nullptr,
                                                                 double* __restrict__ loss = nullptr, int ncov = 0,
                                                                 int lstride = 0) {
  if (lpart != nullptr && (int)blockIdx.x >= nb_main) {
    __shared__ double sm[256];
    const int b = (int)blockIdx.x - nb_main;
    double s = 0.0;
    strided_sum_256(lpart + (size_t)b * lstride, ncov, s);
    strided_sum_256(lpart + (size_t)b * lstride + ncov, F, s);
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
      if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) loss[b] = sm[0];
    return;
  }
  // 64 outputs (one frame block of one (b, n, k)) per workgroup, 4 strands per output over the covering records
  // (independent loads instead of a chain of ~9 L2 latencies), combined in a fixed order
  __shared__ R sn[4][WAVE], sd[4][WAVE];
  const int lane = threadIdx.x & (WAVE - 1), q = threadIdx.x >> 6;
  const int TBk = (T + WAVE - 1) / WAVE;
  // blockIdx.x enumerates (b, n, k, tb)
  const int tb = blockIdx.x % TBk;
  const int k = (blockIdx.x / TBk) % K;
  const int n = (blockIdx.x / (TBk * K)) % N;
  const int b = blockIdx.x / (TBk * K * N);
  const int t = tb * WAVE + lane;
  const bool upd = (src_mask >> n) & 1u;
  const long long j = (long long)b * TBk + tb;
  int g_lo, g_hi;
  flat_cover(fp, j, g_lo, g_hi);
  R* v = V + (((size_t)b * N + n) * K + k) * T + (t < T ? t : T - 1);
  const R vold = (q == 0 && upd) ? *v : (R)0;  // requested with the records, not after the barrier
  R num = 0, den = 0;
  if (upd) {
    constexpr int RC = 4;  // records of a strand in chunks whose loads are all in flight together; same order
    for (int g0 = g_lo + q; g0 <= g_hi; g0 += 4 * RC) {
      R vn[RC], vd[RC];
#pragma unroll
      for (int c = 0; c < RC; ++c) {
        const int g = min(g0 + 4 * c, g_hi);
        const int slot = flat_slot(fp, j, g);
        const R* p = part + ((((size_t)g * fp.S + slot) * N + n) * (size_t)(2 * K) + k * 2) * WAVE + lane;
        vn[c] = p[0];
        vd[c] = p[WAVE];
      }
#pragma unroll
      for (int c = 0; c < RC; ++c)
        if (g0 + 4 * c <= g_hi) {
          num += vn[c];
          den += vd[c];
        }
    }
  }
  sn[q][lane] = num;
  sd[q][lane] = den;
  __syncthreads();
  if (q == 0 && upd && t < T) {
    num = (sn[0][lane] + sn[1][lane]) + (sn[2][lane] + sn[3][lane]);
    den = (sd[0][lane] + sd[1][lane]) + (sd[2][lane] + sd[3][lane]);
    den = floor_eps<R>(den, eps);
    *v = vold * powspec<R>(num / den, p2);
  }
}

}  // namespace assx
