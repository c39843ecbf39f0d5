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
  static constexpr __host__ __device__ __forceinline__ int crow(int r, int lane) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma16<float> {
  using acc_t = v4f_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static constexpr __host__ __device__ __forceinline__ int crow(int r, int lane) { return 4 * (lane >> 4) + r; }
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

// the multiplicative step of one element (nmf.py:317, 325 etc.); same arithmetic as nmf_finalize_kernel
template <typename R, int D2K>
__device__ __forceinline__ R nmf_apply(R old, R num, R den, R eps, PowSpec p) {
  if (D2K < 0 && p.mode == POW_CAUCHY_ME) {  // num = B, den = A  (nmf.py:550-552)
    const R d2 = floor_eps<R>(den + sqrt(fma(den, den, (R)2 * num * den)), eps);
    return old * (num / d2);
  }
  den = floor_eps<R>(den, eps);
  const R q = num / den;
  // domain 2, EUC / KL / IS: the exponent is 1 or 1/2 -- keeps pow() (and its ~70 VGPRs) out of those instantiations
  if (D2K >= 0) return old * (p.mode == POW_SQRT ? sqrt(q) : q);
  return old * powspec<R>(q, p);
}

// ---------------------------------------------------------------------------------------------------------
// Work partition of the two half kernels.  One matrix = nblk output blocks (basis half: 16 bins; activation half: 16
// frames) x nstep wave-steps (16-frame / 16-bin sub-tiles the block's sums run over); the 4 waves of a workgroup take the
// steps of its range in turn.  The workgroups whose ranges meet a block are its GROUP: member i writes slab i of the
// block's records, and (apply != 0) the member that takes the last ticket sums the slabs in ascending order and applies
// the multiplicative step (no finalize launches since round 4).  G, hence every summation order, is a function of one
// matrix's geometry and the group size (assx_ctx::nmf_group) only, never of the batch.
// Round 4 cut the nblk * nstep steps into G contiguous ranges of equal length (the "flat" form, still used where a block
// would get fewer than two workgroups); round 5's in-kernel stamps (tools/probes/nmf_trace.py) showed what that cost and
// that round 4's reading of the sweep ("a SIMD spends ~3700 cycles per step whoever issues it") had divided the whole
// kernel by the steps: the loop itself runs at ~1550 cycles per step and SIMD with two waves on it, the matrix cores' rate
// (24 MFMA x 64 cycles); everything else is per-workgroup and per-launch skeleton.
// ---------------------------------------------------------------------------------------------------------
struct NmfPart {
  int nblk, nstep, G, maxslots;
  int w;  // > 0: block-aligned partition, w workgroups per block; 0: the flat partition of round 4
};
// Block-aligned since round 5: in-kernel stamps of config 2 showed the workgroups whose flat range crossed a block
// boundary (64 of 512) paying two block prologues (cold operand loads, ~3 us) and two epilogues (combine, records,
// ticket, ~5 us): they left 8 us after the median workgroup, and they are the last members of their blocks, so every
// block's finalize waited for them.  With w workgroups per block every workgroup has one prologue and one epilogue,
// every block exactly w members (slabs), and a block's members are consecutive workgroups.
// 32-bit arithmetic: make_nmf_part guarantees (nblk * nstep + 1) * G < 2^32 (a matrix below 4 GiB has < 2^22 steps)
__host__ __device__ inline unsigned nmf_part_lo(const NmfPart& p, int g) {
  if (p.w > 0) {
    const unsigned blk = (unsigned)g / (unsigned)p.w, sl = (unsigned)g % (unsigned)p.w;
    return blk * (unsigned)p.nstep + sl * (unsigned)p.nstep / (unsigned)p.w;
  }
  return (unsigned)g * ((unsigned)p.nblk * (unsigned)p.nstep) / (unsigned)p.G;
}
// range [lo, hi) of workgroup g (aligned: three divisions by w instead of the eight of lo(g), lo(g + 1) and two owners --
// scalar division is a long dependent chain, and it sits in front of the first operand request of every workgroup)
__host__ __device__ inline void nmf_part_range(const NmfPart& p, int g, unsigned& lo, unsigned& hi) {
  if (p.w > 0) {
    const unsigned blk = (unsigned)g / (unsigned)p.w, sl = (unsigned)g - blk * (unsigned)p.w;
    lo = blk * (unsigned)p.nstep + sl * (unsigned)p.nstep / (unsigned)p.w;
    hi = blk * (unsigned)p.nstep + (sl + 1u) * (unsigned)p.nstep / (unsigned)p.w;
  } else {
    lo = nmf_part_lo(p, g);
    hi = nmf_part_lo(p, g + 1);
  }
}
// the workgroup whose range holds step x (inverse of nmf_part_lo)
__host__ __device__ inline int nmf_part_owner(const NmfPart& p, unsigned x) {
  if (p.w > 0) {
    const unsigned blk = x / (unsigned)p.nstep, r = x % (unsigned)p.nstep;
    return (int)(blk * (unsigned)p.w + ((r + 1u) * (unsigned)p.w - 1u) / (unsigned)p.nstep);
  }
  return (int)(((x + 1u) * (unsigned)p.G - 1u) / ((unsigned)p.nblk * (unsigned)p.nstep));
}
inline NmfPart make_nmf_part(int nblk, int nstep, int group, int target_wgs) {
  NmfPart p;
  p.nblk = nblk;
  p.nstep = nstep;
  const long long wt = (long long)nblk * nstep;
  long long G = target_wgs / (group < 1 ? 1 : group);
  static const int aligned = lab_int("ASSX_NMF_ALIGNED", 1);  // laboratory builds, 0: round 4's flat partition (A/B runs)
  // Aligned only where a block gets at least two workgroups: with fewer the flat partition wastes nothing (a range is
  // then whole blocks plus one boundary) and keeps G inside the budget -- one workgroup per block would launch nblk of
  // them, a few more than fit at once when nblk is just above the budget (wide-channel source model: 65 blocks x 8
  // matrices against 512: the 8 late workgroups were the tail of the kernel).
  if (aligned && G / nblk >= 2) {
    long long w = G / nblk;                // never more workgroups than the budget ...
    if (w > 16) w = 16;                    // ... at most 16 slabs for the holder of the last ticket to sum
    if (w > nstep / 4) w = nstep / 4;      // every wave gets a step
    if (w < 1) w = 1;                      // more blocks than budget: one workgroup per block, several rounds
    while (w > 1 && (wt + 1) * (nblk * w) >= (1ll << 32)) --w;
    p.w = (int)w;
    p.G = (int)(nblk * w);
    p.maxslots = (int)w;
    return p;
  }
  p.w = 0;
  if (G > (long long)nblk * 16) G = (long long)nblk * 16;  // at most ~16 slabs for the holder of the last ticket to sum
  if (G > wt / 4) G = wt / 4;  // every wave gets a step (a 513 x 256 matrix: 132 workgroups of one step per wave; one
                               // workgroup per block with 4 steps per wave and no tickets at all measured slower)
  if (G > wt) G = wt;
  while (G > 1 && (wt + 1) * G >= (1ll << 32)) G /= 2;  // 32-bit partition arithmetic in the kernels
  if (G < 1) G = 1;
  p.G = (int)G;
  const long long per = wt / G;  // shortest range
  p.maxslots = (int)((nstep + per - 1) / per) + 1;
  return p;
}

// (target + eps) / (input + eps) of the divergences (criterion/divergence.py:26-27, 39-40).  The hardware reciprocal + two
// Newton steps is the correctly rounded quotient for a normal denominator; outside that range (eps = 0 with a zero model
// entry: a denominator of 0) the Newton step would turn the reference's inf into NaN (fma(-0, inf, 1)), so those lanes take
// the IEEE division (round 5's advisor).
__device__ __forceinline__ double loss_ratio(double tg, double in) {
  if (__builtin_expect(in > 1e-290 && in < 1e290, 1)) return tg * fast_rcp(in);
  return tg / in;
}

#ifndef NMF_TRACE
#define NMF_TRACE 0  // 1: shader-clock stamps of the basis half (tools/probes/nmf_trace.py): every workgroup's entry / exit on the
                     // 100 MHz clock, every step of the waves of workgroup NMF_TRACE_WG on the shader clock
#endif
#if NMF_TRACE && !defined(ASSX_PROBE_BUILD)
#error "NMF_TRACE adds a debug entry point and stamps: build it with -DASSX_PROBE_BUILD into a probe library, never into libassx.so"
#endif
#if NMF_TRACE
#ifndef NMF_TRACE_WG
#define NMF_TRACE_WG 100
#endif
__device__ unsigned long long g_nmf_trace[8192 + 4 * 512];
#define NMF_STAMP(id)                                                                                              \
  do {                                                                                                             \
    if (g == NMF_TRACE_WG && b == 0 && lane == 0 && tr_n < 512)                                                    \
      g_nmf_trace[8192 + wv * 512 + tr_n++] = ((unsigned long long)(id) << 56) | (__builtin_readcyclecounter() & 0xffffffffffffffull); \
  } while (0)
#else
#define NMF_STAMP(id) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------
// basis half: num|den (F,K) = [A|Bm] (F,T) . V^T.   grid (G, 1, B), 4 waves; block = 16 bins, step = 16 frames.
//   part[slab][b*2 + s][f*K + k]
// ---------------------------------------------------------------------------------------------------------
// LOSS (domain 2, EUC / KL / IS: D2K >= 0): the half also accumulates criterion(Tb V, X) of the model it READS (nmf.py:
// 170-174, 229-233, 288-292; divergence.py:21-45) -- the loss the reference records after the PREVIOUS update -- from the
// very Tb V and X it holds, every (f, t) exactly once: one partial per (workgroup, wave) at lpart[b * lstride + 4 g + wave],
// summed by the caller.  IS: sum (ratio - 1) and sum log ratio as a mantissa product; KL needs a log per element.
template <typename R, int KT, int D2K = -1, bool LOSS = false>
__global__ void __launch_bounds__(256)
    nmf_basis_mfma_kernel(const R* __restrict__ X, R* Tb, const R* __restrict__ V, R* part, int* tickets, int apply,
                          NmfPart pt, int B, int F, int T, int K, R eps, TermSpec s, PowSpec pe,
                          double* __restrict__ lpart = nullptr, int lstride = 0, double leps = 0.0) {
  static_assert(!LOSS || D2K >= 0, "the fused loss is evaluated for domain 2 (EUC / KL / IS)");
  using MM = Mfma16<R>;
  using acc_t = typename MM::acc_t;
  constexpr int KS = KT * 4;      // k-slices of 4 in product (1)
  constexpr int KP = KT * 16;     // n_basis padded to the tile
  constexpr int LD = 17;          // padded row of the staged tile (bank-conflict-free column reads)
  constexpr int NLD = KP * 16 / 64;  // staged elements per lane and sub-tile
  // The V tile (KP x 16 frames) of a sub-tile is read in two layouts -- as A operand of (1) and as B operand of (3).
  // Each wave stages it once through its private LDS slice with coalesced 128-byte row reads.  At the end of a block
  // the same memory carries the cross-wave combine.
  constexpr int VT_ELEMS = 4 * KP * LD, RED_ELEMS = 4 * KT * 8 * 64;
  __shared__ R smem[VT_ELEMS > RED_ELEMS ? VT_ELEMS : RED_ELEMS];
  __shared__ int s_last;
  // wave index as a scalar: loop counters, base pointers and the fast-path tests below then live on the scalar unit
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int b = blockIdx.z, g = blockIdx.x;
#if NMF_TRACE
  int tr_n = 0;
  if (b == 0 && threadIdx.x == 0 && g < 1024) {
    g_nmf_trace[8 * g + 0] = __builtin_amdgcn_s_memrealtime();
    g_nmf_trace[8 * g + 4] = __builtin_readcyclecounter();
    g_nmf_trace[8 * g + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
    g_nmf_trace[8 * g + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
  }
  int tr_fin = 0;
  NMF_STAMP(1);
#endif
  R(*vt)[LD] = reinterpret_cast<R(*)[LD]>(smem + wv * KP * LD);
  R(*red)[KT * 8][64] = reinterpret_cast<R(*)[KT * 8][64]>(smem);
  const R* vb = V + (size_t)b * K * T;
  const R* xb = X + (size_t)b * F * T;
  const size_t FK = (size_t)F * K;
  // staged element e = i*64 + lane -> row k = 4 i + lk, frame t0 + li (16 lanes cover one 128-byte row segment).
  // Rows K .. KP-1 read row K-1 instead: they meet zeros of tb in (1) and output columns that are never written in
  // (3), so neither a select nor a branch is needed (until round 3 every k-slice sat behind a wave-uniform
  // `4 j < n_basis` test: a chain of scalar branches that serialised read - wait - MFMA per slice).  Addresses =
  // descriptor + wave-uniform offset (advances with t0) + a per-lane byte offset fixed for the whole kernel (raw buffer
  // loads): no address arithmetic on the vector ALU inside the loop.  One matrix stays below 4 GiB (checked by the host).
  unsigned voff[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) voff[i] = (unsigned)((min(4 * i + lk, K - 1) * (size_t)T + li) * sizeof(R));
  constexpr int XSTEP = MM::crow(1, 0) - MM::crow(0, 0);  // frames between consecutive accumulator registers
  const BufRsrc vrs = make_rsrc(vb), xrs = make_rsrc(xb);
  double ltot = 0.0;  // LOSS: this wave's share

  // ---- end of a block's share of this workgroup: records (or the direct update), ticket, finalize.  ns / ds: wave 0's
  // lanes hold the workgroup's sums in the accumulator layout D[row = f0 + crow][col = kb].
  auto block_tail = [&](int blk, int f0, int slot, int members, auto&& sums) {  // sums(c, r, n, d): one (c, r) pair of them
    R* pn = part + ((size_t)slot * B * 2 + (size_t)b * 2) * FK;
    const bool direct = apply && members == 1;
    // Every wave puts out its share of the block -- accumulator register r = wave index, every column tile c -- instead of
    // wave 0 all of it: the address arithmetic and the stores of a lone wave are one dependent chain (1.7 us in the
    // stamps), four waves do a quarter each.  Three straight-line forms: in one loop body the `direct` path's read of Tb
    // made the compiler wait for ALL memory operations between the record stores (3.5 us, round 4's form).
    {
      const int r = wv, fo = f0 + MM::crow(r, lane);
      if (direct) {
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const int kb = 16 * c + li;
          if (fo < F && kb < K) {
            R n, d;
            sums(c, r, n, d);
            R* tp = Tb + (size_t)b * FK + (size_t)fo * K + kb;
            *tp = nmf_apply<R, D2K>(*tp, n, d, eps, pe);
          }
        }
      } else if (apply) {
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const int kb = 16 * c + li;
          if (fo < F && kb < K) {
            R n, d;
            sums(c, r, n, d);
            const size_t o = (size_t)fo * K + kb;
            st_agent(pn + o, n);
            st_agent(pn + FK + o, d);
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const int kb = 16 * c + li;
          if (fo < F && kb < K) {
            R n, d;
            sums(c, r, n, d);
            const size_t o = (size_t)fo * K + kb;
            pn[o] = n;
            pn[FK + o] = d;
          }
        }
      }
    }
    NMF_STAMP(6);
    if (apply && !direct) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's records are out ...
      __syncthreads();                                   // ... and so are the other waves'
      if (wv == 0) {
        const bool last = take_ticket(tickets + (size_t)b * pt.nblk + blk, members);
        if (lane == 0) s_last = last;
      }
      NMF_STAMP(7);
#if NMF_TRACE
      if (b == 0 && threadIdx.x == 0 && g < 1024) g_nmf_trace[8 * g + 2] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    const size_t slab = (size_t)B * 2 * FK;
    if (!apply) {
      // callers that combine the slabs themselves sum maxslots of them: the last member clears the unused ones
      if (slot == members - 1)
        for (int sl = members; sl < pt.maxslots; ++sl)
          for (int o = threadIdx.x; o < 16 * K; o += 256) {
            const int fo = f0 + o / K;
            if (fo >= F) break;
            R* q = part + (size_t)sl * slab + (size_t)b * 2 * FK + (size_t)fo * K + o % K;
            q[0] = 0;
            q[FK] = 0;
          }
    } else if (!direct) {
      __syncthreads();
      if (s_last) {
        // ---- holder of the last ticket: sum the slabs of this 16 x K block and update Tb in place.  Every member
        // read its rows of Tb at the top of the block, i.e. before it took its ticket.
        // Two outputs per thread and trip, their four slab sums in flight together (slab_sum4_n).
        constexpr int NO = (D2K < 0 || KT >= 3) ? 1 : 2;  // the pow() / wide variants have no registers to spare for a second output
        const R* p0 = part + (size_t)b * 2 * FK;
        R* tbo = Tb + (size_t)b * FK;
        for (int o = threadIdx.x; o < 16 * K; o += 256 * NO) {
          unsigned idx[NO];
          bool on[NO];
          R old[NO];
#pragma unroll
          for (int u = 0; u < NO; ++u) {
            const int ou = o + 256 * u, fo = f0 + ou / K;
            on[u] = ou < 16 * K && fo < F;
            idx[u] = on[u] ? (unsigned)fo * (unsigned)K + (unsigned)(ou % K) : 0u;
            old[u] = on[u] ? tbo[idx[u]] : (R)0;  // requested with the slabs
          }
          R sum[2 * NO];
          slab_sum4_n<R, NO>(p0, p0 + FK, idx, on, slab, members, sum);
#pragma unroll
          for (int u = 0; u < NO; ++u)
            if (on[u]) tbo[idx[u]] = nmf_apply<R, D2K>(old[u], sum[2 * u], sum[2 * u + 1], eps, pe);
        }
      }
    }
    __syncthreads();  // the combine memory becomes staging memory again (next block of this range)
    NMF_STAMP(s_last ? 9 : 8);
#if NMF_TRACE
    tr_fin += (apply && members > 1 && s_last) ? 1 : 0;
#endif
  };

  unsigned lo, hi;
  nmf_part_range(pt, g, lo, hi);
  NMF_STAMP(11);
  for (int blk = (int)(lo / (unsigned)pt.nstep); (unsigned)blk * (unsigned)pt.nstep < hi; ++blk) {
    const unsigned base = (unsigned)blk * (unsigned)pt.nstep;
    const int s0 = lo > base ? (int)(lo - base) : 0;
    const int s1 = hi - base < (unsigned)pt.nstep ? (int)(hi - base) : pt.nstep;
    const int gf = pt.w > 0 ? blk * pt.w : nmf_part_owner(pt, base);  // aligned: the block's w workgroups
    const int members = pt.w > 0 ? pt.w : nmf_part_owner(pt, base + pt.nstep - 1) - gf + 1;
    const int slot = g - gf;
    const int f0 = blk * 16;
    const int f = min(f0 + li, F - 1);  // rows past F feed only output rows that are never written
    const unsigned xoff = (unsigned)(((size_t)f * T + MM::crow(0, lane)) * sizeof(R));
    double lacc = 0.0, lm = 1.0;  // LOSS: this lane's bin; sum log ratio (IS) as mantissa product + exponent
    int le = 0;

    R tb[KS];  // B operand of product (1): Tb^T[k = 4j + lk][f]
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int k = 4 * j + lk;
      tb[j] = (k < K) ? Tb[((size_t)b * F + f) * K + k] : (R)0;
    }
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
    R xq[4];  // X of the next sub-tile, in the accumulator layout (HBM latency hidden behind the current one)
    auto fetch = [&](int t0) {
      if (t0 + 16 <= T) {  // whole sub-tile inside the matrix (wave-uniform)
        const unsigned soff = (unsigned)t0 * (unsigned)sizeof(R);
#pragma unroll
        for (int i = 0; i < NLD; ++i) stage[i] = buf_ld<R>(vrs, voff[i], soff);
#pragma unroll
        for (int r = 0; r < 4; ++r) xq[r] = buf_ld<R>(xrs, xoff + r * XSTEP * (unsigned)sizeof(R), soff);
      } else {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          const int e = i * 64 + lane;
          const int k = min(e >> 4, K - 1), tt = min(t0 + (e & 15), T - 1);
          stage[i] = vb[(size_t)k * T + tt];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) xq[r] = xb[(size_t)f * T + min(t0 + MM::crow(r, lane), T - 1)];
      }
    };
    auto compute = [&](auto masked, int t0, const R (&xc)[4]) {
      // (1) TV^T sub-tile: rows = frames t0 + crow, columns = bins f0 + li
      acc_t tv;
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[r] = 0;
#pragma unroll
      for (int j = 0; j < KS; ++j) tv = MM::mma(vt[4 * j + lk][li], tb[j], tv);  // A operand: V^T[t = li][k]
      // (2) elementwise in the accumulator layout
      R a[4], bm[4];
      double lprod = 1.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        nmf_terms<R, D2K>(s, xc[r], tv[r], eps, a[r], bm[r]);
        const bool dead = decltype(masked)::value && t0 + MM::crow(r, lane) >= T;  // ragged last sub-tile of the matrix only
        if (LOSS && !dead) {  // criterion((Tb V)^(2/2), x): Tb V as the product gave it, NOT floored
          const double in = (double)tv[r], xx = (double)xc[r];
          if (D2K == ASSX_NMF_EUC) {
            lacc = fma(xx - in, xx - in, lacc);
          } else {
            const double in_ = in + leps, tg_ = xx + leps;  // divergence.py:26-27, 39-40
            const double ratio = loss_ratio(tg_, in_);
            if (D2K == ASSX_NMF_KL) {
              lacc += tg_ * log(ratio) + in_ - tg_;
            } else {
              lacc += ratio - 1.0;
              lprod *= ratio;
              if (r == 1) {  // mantissa / exponent after every PAIR of ratios: four of them beyond 1e+-77 each would leave the range
                int e2;
                lm = frexp(lm * lprod, &e2);
                le += e2;
                lprod = 1.0;
              }
            }
          }
        }
        if (dead) {
          a[r] = 0;
          bm[r] = 0;
        }
      }
      if (LOSS && D2K == ASSX_NMF_IS_MM) {
        int e;
        lm = frexp(lm * lprod, &e);
        le += e;
      }
      // (3) num[f, kb] += sum_t a[f,t] V[kb,t]: accumulator register r is k-slice r of the A operand
#pragma unroll
      for (int c = 0; c < KT; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const R vv = vt[16 * c + li][MM::crow(r, lane)];  // B operand: V^T[t-pos of slice r][kb]
          num[c] = MM::mma(a[r], vv, num[c]);
          den[c] = MM::mma(bm[r], vv, den[c]);
        }
      }
    };
    auto step = [&](auto masked, int t0) {
      R xc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) xc[r] = xq[r];
#pragma unroll
      for (int i = 0; i < NLD; ++i) vt[(i * 64 + lane) >> 4][lane & 15] = stage[i];
      if (t0 + 64 < te) fetch(t0 + 64);  // the next tile travels while this one is consumed
      __builtin_amdgcn_wave_barrier();
      NMF_STAMP(3);
      compute(masked, t0, xc);
      __builtin_amdgcn_wave_barrier();
    };
    // whole sub-tiles in the loop, the (at most one) ragged sub-tile of the matrix after it: one loop body with both
    // forms made the accumulators of the two paths meet in copies at the back-edge (16 v_mov_b64 behind the last MFMA)
    int t0 = (s0 + wv) * 16;
    NMF_STAMP(2);
    if (t0 < te) fetch(t0);
    for (; t0 + 16 <= te; t0 += 64) step(IntC<0>(), t0);
    if (t0 < te) step(IntC<1>(), t0);
    NMF_STAMP(4);

    if (LOSS && f0 + li < F)  // a lane's elements belong to one bin: rows past F (copies of row F-1) drop out here
      ltot += lacc - ((double)le * 0.6931471805599453 + log(lm));
    // ---- end of the block's share of this workgroup: the 4 waves in ascending order (wave 0 holds the total)
    __syncthreads();  // every wave is done with its staging slice: the memory now carries the combine
#pragma unroll
    for (int c = 0; c < KT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[wv][(c * 2 + 0) * 4 + r][lane] = num[c][r];
        red[wv][(c * 2 + 1) * 4 + r][lane] = den[c][r];
      }
    __syncthreads();
    NMF_STAMP(5);
    block_tail(blk, f0, slot, members, [&](int c, int r, R& n, R& d) {  // the four waves' sums, wave 0 first (as ever)
      n = red[0][(c * 2 + 0) * 4 + r][lane];
      d = red[0][(c * 2 + 1) * 4 + r][lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        n += red[w][(c * 2 + 0) * 4 + r][lane];
        d += red[w][(c * 2 + 1) * 4 + r][lane];
      }
    });
  }
  if (LOSS) {
    ltot = wave_allreduce_sum<double>(ltot);
    if (lane == 0) lpart[(size_t)b * lstride + (size_t)g * 4 + wv] = ltot;
  }
#if NMF_TRACE
  if (b == 0 && threadIdx.x == 0 && g < 1024) {
    g_nmf_trace[8 * g + 1] = __builtin_amdgcn_s_memrealtime();
    g_nmf_trace[8 * g + 3] = (unsigned long long)tr_fin;
    g_nmf_trace[8 * g + 5] = __builtin_readcyclecounter();
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// activation half: num|den (K,T) = Tb^T (K,F) . [A|Bm] (F,T).   grid (G, 1, B), 4 waves; block = 16 frames,
// step = 16 bins.   part[slab][b*2 + s][k*T + t]
// ---------------------------------------------------------------------------------------------------------
template <typename R, int KT, int D2K = -1>
__global__ void __launch_bounds__(256)
    nmf_act_mfma_kernel(const R* __restrict__ X, const R* __restrict__ Tb, R* V, R* part, int* tickets, int apply,
                        NmfPart pt, int B, int F, int T, int K, R eps, TermSpec s, PowSpec pe) {
  using MM = Mfma16<R>;
  using acc_t = typename MM::acc_t;
  constexpr int KS = KT * 4;
  constexpr int KP = KT * 16;
  constexpr int LD = KP + 4;          // padded row of the staged 16 x KP basis tile
  constexpr int NLD = KP * 16 / 64;   // staged elements per lane and sub-tile
  constexpr int TT_ELEMS = 4 * 16 * LD, RED_ELEMS = 4 * KT * 8 * 64;
  __shared__ R smem[TT_ELEMS > RED_ELEMS ? TT_ELEMS : RED_ELEMS];  // wave-private basis tiles, then the combine
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int b = blockIdx.z, g = blockIdx.x;
  const R* tbb = Tb + (size_t)b * F * K;
  const R* xb = X + (size_t)b * F * T;
  R(*tt_)[LD] = reinterpret_cast<R(*)[LD]>(smem + wv * 16 * LD);
  R(*red)[KT * 8][64] = reinterpret_cast<R(*)[KT * 8][64]>(smem);
  const size_t KTt = (size_t)K * T;
  // staged element e = i*64 + lane -> bin f0 + e / KP, basis e % KP: consecutive lanes read consecutive k of a row.
  // Columns K .. KP-1 read column K-1 instead (they meet zeros of vbr in (1) and output rows never written in (3)).
  // KP = 16 / 32 / 64: element i of a lane sits 64 / KP whole rows below element i - 1, i.e. a wave-uniform step that
  // rides in the scalar offset of the load (one VGPR of addresses instead of NLD); KP = 48 keeps the per-element table
  constexpr bool TUNI = (64 % KP) == 0;
  unsigned toff[TUNI ? 1 : NLD];
#pragma unroll
  for (int i = 0; i < (TUNI ? 1 : NLD); ++i) {
    const int e = i * 64 + lane;
    toff[i] = (unsigned)(((size_t)(e / KP) * K + min(e % KP, K - 1)) * sizeof(R));
  }
  const unsigned tstep = (unsigned)(64 / KP) * (unsigned)K * (unsigned)sizeof(R);
  const BufRsrc trs = make_rsrc(tbb), xrs = make_rsrc(xb);  // buffer addressing, as in the basis half

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
    const int t = min(t0 + li, T - 1);  // columns past T feed only output columns that are never written
    const bool tvalid = t0 + li < T;
    const unsigned xoff = (unsigned)(((size_t)MM::crow(0, lane) * T + t) * sizeof(R));
    const unsigned xstep = (unsigned)(MM::crow(1, 0) - MM::crow(0, 0)) * (unsigned)T * (unsigned)sizeof(R);  // rows between accumulator registers

    R vbr[KS];  // B operand of product (1): V[k = 4j + lk][t]
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int k = 4 * j + lk;
      vbr[j] = (k < K) ? V[((size_t)b * K + k) * T + t] : (R)0;
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
    R xq[4];  // X of the next sub-tile, in the accumulator layout
    auto fetch = [&](int f0) {
      if (f0 + 16 <= fe) {  // whole sub-tile inside the bin range (wave-uniform)
        const unsigned tso = (unsigned)f0 * (unsigned)K * (unsigned)sizeof(R), xso = (unsigned)f0 * (unsigned)T * (unsigned)sizeof(R);
#pragma unroll
        for (int i = 0; i < NLD; ++i) stage[i] = TUNI ? buf_ld<R>(trs, toff[0], tso + i * tstep) : buf_ld<R>(trs, toff[TUNI ? 0 : i], tso);
#pragma unroll
        for (int r = 0; r < 4; ++r) xq[r] = buf_ld<R>(xrs, xoff, xso + r * xstep);
      } else {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          const int e = i * 64 + lane;
          const int fr = f0 + e / KP, k = min(e % KP, K - 1);
          stage[i] = (fr < fe) ? tbb[(size_t)fr * K + k] : (R)0;  // bins beyond the range contribute 0
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) xq[r] = xb[(size_t)min(f0 + MM::crow(r, lane), F - 1) * T + t];
      }
    };
    auto compute = [&](auto masked, int f0, const R (&xc)[4]) {
      // (1) TV sub-tile: rows = bins f0 + crow, columns = frames t0 + li
      acc_t tv;
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[r] = 0;
#pragma unroll
      for (int j = 0; j < KS; ++j) tv = MM::mma(tt_[li][4 * j + lk], vbr[j], tv);  // A operand: Tb[f = li][k]
      // (2)
      R a[4], bm[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        nmf_terms<R, D2K>(s, xc[r], tv[r], eps, a[r], bm[r]);
        if (decltype(masked)::value && f0 + MM::crow(r, lane) >= fe) {  // ragged last sub-tile of the matrix only
          a[r] = 0;
          bm[r] = 0;
        }
      }
      // (3) num[kb, t] += sum_f Tb[f,kb] a[f,t]: accumulator register r is k-slice r of the B operand
#pragma unroll
      for (int c = 0; c < KT; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const R tv3 = tt_[MM::crow(r, lane)][16 * c + li];  // A operand: Tb^T[kb][f-pos of slice r]
          num[c] = MM::mma(tv3, a[r], num[c]);
          den[c] = MM::mma(tv3, bm[r], den[c]);
        }
      }
    };
    auto step = [&](auto masked, int f0) {
      R xc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) xc[r] = xq[r];
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int e = i * 64 + lane;
        tt_[e / KP][e % KP] = stage[i];
      }
      if (f0 + 64 < fe) fetch(f0 + 64);
      __builtin_amdgcn_wave_barrier();
      compute(masked, f0, xc);
      __builtin_amdgcn_wave_barrier();
    };
    int f0 = (s0 + wv) * 16;
    if (f0 < fe) fetch(f0);
    for (; f0 + 16 <= fe; f0 += 64) step(IntC<0>(), f0);  // whole sub-tiles; the ragged one (at most) after the loop
    if (f0 < fe) step(IntC<1>(), f0);

    // ---- end of the block's share of this workgroup: the 4 waves' sums in ascending order, every wave puts out its
    // quarter of the block (accumulator register r = wave index; see the basis half)
    __syncthreads();
#pragma unroll
    for (int c = 0; c < KT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[wv][(c * 2 + 0) * 4 + r][lane] = num[c][r];
        red[wv][(c * 2 + 1) * 4 + r][lane] = den[c][r];
      }
    __syncthreads();
    R* pn = part + ((size_t)slot * B * 2 + (size_t)b * 2) * KTt;
    const bool direct = apply && members == 1;
    {
      const int r = wv;
      auto sums = [&](int c, R& n, R& d) {
        n = red[0][(c * 2 + 0) * 4 + r][lane];
        d = red[0][(c * 2 + 1) * 4 + r][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          n += red[w][(c * 2 + 0) * 4 + r][lane];
          d += red[w][(c * 2 + 1) * 4 + r][lane];
        }
      };
      if (direct) {
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const int kb = 16 * c + MM::crow(r, lane);  // D[row = kb][col = t]
          if (kb < K && tvalid) {
            R n, d;
            sums(c, n, d);
            R* vp = V + (size_t)b * KTt + (size_t)kb * T + t;
            *vp = nmf_apply<R, D2K>(*vp, n, d, eps, pe);
          }
        }
      } else if (apply) {
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const int kb = 16 * c + MM::crow(r, lane);
          if (kb < K && tvalid) {
            R n, d;
            sums(c, n, d);
            const size_t o = (size_t)kb * T + t;
            st_agent(pn + o, n);
            st_agent(pn + KTt + o, d);
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const int kb = 16 * c + MM::crow(r, lane);
          if (kb < K && tvalid) {
            R n, d;
            sums(c, n, d);
            const size_t o = (size_t)kb * T + t;
            pn[o] = n;
            pn[KTt + o] = d;
          }
        }
      }
    }
    if (apply && !direct) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's records are out ...
      __syncthreads();                                   // ... and so are the other waves'
      if (wv == 0) {
        const bool last = take_ticket(tickets + (size_t)b * pt.nblk + blk, members);
        if (lane == 0) s_last = last;
      }
    }
    const size_t slab = (size_t)B * 2 * KTt;
    if (!apply) {
      if (slot == members - 1)
        for (int sl = members; sl < pt.maxslots; ++sl)
          for (int o = threadIdx.x; o < 16 * K; o += 256) {
            const int tc = t0 + (o & 15);
            if (tc >= T) continue;
            R* q = part + (size_t)sl * slab + (size_t)b * 2 * KTt + (size_t)(o >> 4) * T + tc;
            q[0] = 0;
            q[KTt] = 0;
          }
    } else if (!direct) {
      __syncthreads();
      if (s_last) {
        // ---- holder of the last ticket: sum the slabs of this K x 16 block and update V in place (every member read
        // its columns of V at the top of the block, before it took its ticket)
        const R* p0 = part + (size_t)b * 2 * KTt;
        R* vo = V + (size_t)b * KTt;
        constexpr int NO = (D2K < 0 || KT >= 3) ? 1 : 2;
        for (int o = threadIdx.x; o < 16 * K; o += 256 * NO) {
          unsigned idx[NO];
          bool on[NO];
          R old[NO];
#pragma unroll
          for (int u = 0; u < NO; ++u) {
            const int ou = o + 256 * u, tc = t0 + (ou & 15);
            on[u] = ou < 16 * K && tc < T;
            idx[u] = on[u] ? (unsigned)(ou >> 4) * (unsigned)T + (unsigned)tc : 0u;
            old[u] = on[u] ? vo[idx[u]] : (R)0;
          }
          R sum[2 * NO];
          slab_sum4_n<R, NO>(p0, p0 + KTt, idx, on, slab, members, sum);
#pragma unroll
          for (int u = 0; u < NO; ++u)
            if (on[u]) vo[idx[u]] = nmf_apply<R, D2K>(old[u], sum[2 * u], sum[2 * u + 1], eps, pe);
        }
      }
    }
    __syncthreads();
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
// D2K >= 0: domain 2 with the EUC / KL / IS criterion -- no pow(), and for IS no logarithm per element either: sum (ratio - 1)
// and the product of the ratios (mantissa + exponent), as the criterion fused into the basis half does.  The operands of the
// next 16 rows travel while the current ones are consumed (round 5: the kernel was load - wait - use per trip, 43 us at config 2).
template <typename R, int KT, int D2K = -1>
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
  double acc = 0.0, lm = 1.0;
  int le = 0;
  // rows past the matrix repeat its last row, basis columns past n_basis its last column (they meet zeros of vbr): no branches
  auto fetch = [&](int f0, R(&xn)[4], R(&tn)[KS]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) xn[r] = xb[(size_t)min(f0 + MM::crow(r, lane), F - 1) * T + t];
    const R* row = tbb + (size_t)min(f0 + li, F - 1) * K;
#pragma unroll
    for (int j = 0; j < KS; ++j) tn[j] = row[min(4 * j + lk, K - 1)];
  };
  R xc[4], ta[KS];
  int f0 = fa + 16 * wv;
  if (f0 < fe) fetch(f0, xc, ta);
  for (; f0 < fe; f0 += 64) {
    R xn[4], tn[KS];
    const bool more = f0 + 64 < fe;  // wave-uniform
    if (more) fetch(f0 + 64, xn, tn);
    acc_t tv;
#pragma unroll
    for (int r = 0; r < 4; ++r) tv[r] = 0;
#pragma unroll
    for (int j = 0; j < KS; ++j) tv = MM::mma(ta[j], vbr[j], tv);
    double lprod = 1.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool live = tvalid && f0 + MM::crow(r, lane) < fe;
      if (D2K < 0) {
        if (live) {
          const double in = (double)powspec<R>(tv[r], p2d);  // (T V) ** (2 / domain), not floored
          acc += nmf_criterion(kind, in, (double)xc[r], eps, p0);
        }
      } else {
        const double in = (double)tv[r], xx = (double)xc[r];
        if (D2K == ASSX_NMF_EUC) {
          if (live) acc = fma(xx - in, xx - in, acc);
        } else {
          const double in_ = in + eps, tg_ = xx + eps;  // divergence.py:26-27, 39-40
          const double ratio = live ? loss_ratio(tg_, in_) : 1.0;
          if (D2K == ASSX_NMF_KL) {
            if (live) acc += tg_ * log(ratio) + in_ - tg_;
          } else {
            acc += ratio - 1.0;
            lprod *= ratio;
            if (r == 1) {  // see the fused form in nmf_basis_mfma_kernel: a pair of ratios per mantissa / exponent step
              int e2;
              lm = frexp(lm * lprod, &e2);
              le += e2;
              lprod = 1.0;
            }
          }
        }
      }
    }
    if (D2K == ASSX_NMF_IS_MM) {
      int e;
      lm = frexp(lm * lprod, &e);
      le += e;
    }
    if (more) {
#pragma unroll
      for (int r = 0; r < 4; ++r) xc[r] = xn[r];
#pragma unroll
      for (int j = 0; j < KS; ++j) ta[j] = tn[j];
    }
  }
  if (D2K == ASSX_NMF_IS_MM) acc -= (double)le * 0.6931471805599453 + log(lm);
  acc = wave_allreduce_sum<double>(acc);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    lpart[(size_t)b * gridDim.y * gridDim.x + (size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace assx
