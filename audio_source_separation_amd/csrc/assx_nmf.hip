// NMF multiplicative updates (EUC / KL / IS-mm / IS-me) for gfx950 and their C-ABI entry points.
//
// Reference: src/algorithm/nmf.py:182-207 (EUC), 241-266 (KL), 302-327 (IS mm), 329-356 (IS me);
// loss: nmf.py:170-174, 229-233, 288-292 with src/criterion/divergence.py:21-45.
//
// One half-update =
//   (1) terms kernel : TV = max(Tb V, eps) on the fly, writes the two (F,T) operands
//                      A = numerator weights (X * g(TV)),  Bm = denominator weights (h(TV));
//   (2) batched GEMM : basis  num|den = [A|Bm] V^T   (reduce over t, split-K, no atomics)
//                      activ. num|den = Tb^T [A|Bm]  (reduce over f)
//   (3) finalize     : sum the split-K slabs, floor the denominator, multiply-update in place.
// The GEMM is an LDS-tiled 64x64x16 register-blocked kernel in the storage precision.
// That three-step form is the fallback for n_basis > 64; up to 64 a half-update is ONE matrix-core kernel
// (assx_nmf_mfma.hpp) that also finalizes its output behind a "last workgroup done" ticket.
#include <cstdlib>
#include "assx_common.hpp"
#include "assx_nmf_mfma.hpp"
#include "assx_nmf_small.hpp"
#include "assx_nmf_xfed.hpp"

#include "assx_nmf_internal.hpp"

using namespace assx;

namespace {

constexpr int BM = 64, BN = 64, BK = 16, TM = 4, TN = 4;

// kind/domain -> elementwise weights (only TV and the denominators are floored: nmf.py:312-316)
inline TermSpec make_terms(int kind, double d, double p0) {
  TermSpec s;
  s.kind = kind;
  s.p0 = p0;
  switch (kind) {
    case ASSX_NMF_EUC:  // num: X * TV^((2-d)/d)   den: TV^((4-d)/d)
      s.pa = make_pow((2.0 - d) / d);
      s.pb = make_pow((4.0 - d) / d);
      break;
    case ASSX_NMF_KL:  // num: X / TV             den: TV^((2-d)/d)
      s.pa = make_pow(1.0);
      s.pb = make_pow((2.0 - d) / d);
      break;
    default:  // IS: num: X / TV^((d+2)/d)  den: 1 / TV
      s.pa = make_pow((d + 2.0) / d);
      s.pb = make_pow(1.0);
  }
  return s;
}

inline PowSpec update_exponent(int kind, double d) {
  switch (kind) {
    case ASSX_NMF_EUC: return make_pow(d / (4.0 - d));
    case ASSX_NMF_KL: return make_pow(d / 2.0);
    case ASSX_NMF_IS_MM: return make_pow(d / (d + 2.0));
    case ASSX_NMF_T:
    case ASSX_NMF_T_RAW:
    case ASSX_NMF_CAUCHY_MM:
    case ASSX_NMF_CAUCHY_MM_FAST: return make_pow(0.5);  // np.sqrt (nmf.py:420,429,514,530,583,597)
    case ASSX_NMF_CAUCHY_ME: {
      PowSpec p = make_pow(1.0);
      p.mode = POW_CAUCHY_ME;  // T *= B / max(A + sqrt(A^2 + 2 B A), eps)  (nmf.py:550-552)
      return p;
    }
    default: return make_pow(1.0);  // IS me, Cauchy naive: no outer power (nmf.py:345,354,487)
  }
}

// AB (B,2,F,T): AB[b,0] = numerator weights, AB[b,1] = denominator weights
template <typename R>
__global__ void __launch_bounds__(256) nmf_terms_kernel(const R* __restrict__ X, const R* __restrict__ Tb,
                                                       const R* __restrict__ V, R* __restrict__ AB, int F, int T, int K,
                                                       R eps, TermSpec s) {
  const int f = blockIdx.y, b = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const R* tb = Tb + ((size_t)b * F + f) * K;
  const R* vb = V + (size_t)b * K * T + t;
  R tv = 0;
  for (int k = 0; k < K; ++k) tv = fma(tb[k], vb[(size_t)k * T], tv);
  const R x = X[((size_t)b * F + f) * T + t];
  R a, bm;
  nmf_terms<R, -1>(s, x, tv, eps, a, bm);  // floors T V where the kind's reference code does
  const size_t FT = (size_t)F * T;
  R* o = AB + (size_t)b * 2 * FT + (size_t)f * T + t;
  o[0] = a;
  o[FT] = bm;
}

// Batched tiled GEMM:  C[z][m][n] = sum_k A[z][m,k] * Bop[z][k,n]  over the k-range of this split.
//   A element (m,k)  at A  + z_a(z) + m*sam + k*sak      Bop element (k,n) at Bm + z_b(z) + k*sbk + n*sbn
//   Cpart[split][z][m][n]
struct GemmArgs {
  const void* A;
  const void* Bm;
  void* C;
  int Mg, Ng, Kg;
  long sam, sak, sbk, sbn;
  long za_outer, za_inner, zb_outer, zb_inner;  // batch z = outer*2 + inner
  int Z, splits, kchunk;
};

template <typename R, bool A_KCONTIG, bool B_KCONTIG>
__global__ void __launch_bounds__(256) gemm_tile_kernel(GemmArgs g) {
  __shared__ R As[BK][BM + 4];
  __shared__ R Bs[BK][BN + 4];
  const int z = blockIdx.z % g.Z, split = blockIdx.z / g.Z;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tid = threadIdx.x;
  const int tx = tid % (BN / TN), ty = tid / (BN / TN);  // 16 x 16 threads, each TM x TN outputs
  const R* A = (const R*)g.A + (z / 2) * g.za_outer + (z % 2) * g.za_inner;
  const R* Bp = (const R*)g.Bm + (z / 2) * g.zb_outer + (z % 2) * g.zb_inner;
  const int k_begin = split * g.kchunk;
  const int k_end = min(g.Kg, k_begin + g.kchunk);
  R acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = 0;

  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    // stage A tile (BM x BK) and B tile (BK x BN); 1024 elements each, 4 per thread
#pragma unroll
    for (int e = 0; e < (BM * BK) / 256; ++e) {
      const int i = tid + e * 256;
      int mm, kk;
      if (A_KCONTIG) {
        kk = i % BK;
        mm = i / BK;
      } else {
        mm = i % BM;
        kk = i / BM;
      }
      const int m = m0 + mm, k = k0 + kk;
      As[kk][mm] = (m < g.Mg && k < k_end) ? A[(long)m * g.sam + (long)k * g.sak] : (R)0;
    }
#pragma unroll
    for (int e = 0; e < (BN * BK) / 256; ++e) {
      const int i = tid + e * 256;
      int nn, kk;
      if (B_KCONTIG) {
        kk = i % BK;
        nn = i / BK;
      } else {
        nn = i % BN;
        kk = i / BN;
      }
      const int n = n0 + nn, k = k0 + kk;
      Bs[kk][nn] = (n < g.Ng && k < k_end) ? Bp[(long)k * g.sbk + (long)n * g.sbn] : (R)0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      R a[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[kk][ty * TM + i];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bs[kk][tx * TN + j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = fma(a[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  R* C = (R*)g.C + ((size_t)split * g.Z + z) * (size_t)g.Mg * g.Ng;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + ty * TM + i;
    if (m >= g.Mg) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + tx * TN + j;
      if (n < g.Ng) C[(size_t)m * g.Ng + n] = acc[i][j];
    }
  }
}

// out[b][i] *= (num / max(den, eps)) ** p ;  part[split][b*2 + s][i]
template <typename R>
__global__ void __launch_bounds__(256) nmf_finalize_kernel(const R* __restrict__ part, R* __restrict__ out, int B,
                                                          size_t count, int splits, R eps, PowSpec p) {
  // 64 outputs per workgroup, 4 strands per output (independent loads; the single-thread walk over ~30 slabs was a
  // chain of L2 latencies), combined in a fixed order
  __shared__ R sn[4][64], sd[4][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const size_t idx = (size_t)blockIdx.x * 64 + o;
  const bool ok = idx < (size_t)B * count;
  const size_t b = ok ? idx / count : 0, i = ok ? idx % count : 0;
  const R old = (q == 0 && ok) ? out[idx] : (R)0;  // requested with the slabs, not after the barrier
  R num = 0, den = 0;
  if (ok) {
#pragma unroll 4
    for (int s = q; s < splits; s += 4) {
      const R* pq = part + ((size_t)s * B * 2 + b * 2) * count + i;
      num += pq[0];
      den += pq[count];
    }
  }
  sn[q][o] = num;
  sd[q][o] = den;
  __syncthreads();
  if (q == 0 && ok) {
    num = (sn[0][o] + sn[1][o]) + (sn[2][o] + sn[3][o]);
    den = (sd[0][o] + sd[1][o]) + (sd[2][o] + sd[3][o]);
    if (p.mode == POW_CAUCHY_ME) {  // num = B, den = A
      const R d2 = floor_eps<R>(den + sqrt(fma(den, den, (R)2 * num * den)), eps);
      out[idx] = old * (num / d2);
    } else {
      den = floor_eps<R>(den, eps);
      out[idx] = old * powspec<R>(num / den, p);
    }
  }
}

// The two halves of nmf_finalize_kernel as separate steps (F-sharded mode: the sums of a bin shard travel through an
// all-reduce before they are applied).  sums[s][b][i], s = 0 numerator, 1 denominator; same strand order.
template <typename R>
__global__ void __launch_bounds__(256) nmf_sum_slabs_kernel(const R* __restrict__ part, R* __restrict__ sums, int B,
                                                           size_t count, int splits) {
  __shared__ R sn[4][64], sd[4][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const size_t idx = (size_t)blockIdx.x * 64 + o;
  const bool ok = idx < (size_t)B * count;
  const size_t b = ok ? idx / count : 0, i = ok ? idx % count : 0;
  R num = 0, den = 0;
  if (ok) {
#pragma unroll 4
    for (int s = q; s < splits; s += 4) {
      const R* pq = part + ((size_t)s * B * 2 + b * 2) * count + i;
      num += pq[0];
      den += pq[count];
    }
  }
  sn[q][o] = num;
  sd[q][o] = den;
  __syncthreads();
  if (q == 0 && ok) {
    sums[idx] = (sn[0][o] + sn[1][o]) + (sn[2][o] + sn[3][o]);
    sums[(size_t)B * count + idx] = (sd[0][o] + sd[1][o]) + (sd[2][o] + sd[3][o]);
  }
}

template <typename R>
__global__ void __launch_bounds__(256) nmf_apply_sums_kernel(R* __restrict__ out, const R* __restrict__ sums, size_t total,
                                                            R eps, PowSpec p) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const R num = sums[idx];
  R den = sums[total + idx];
  if (p.mode == POW_CAUCHY_ME) {
    const R d2 = floor_eps<R>(den + sqrt(fma(den, den, (R)2 * num * den)), eps);
    out[idx] = out[idx] * (num / d2);
  } else {
    den = floor_eps<R>(den, eps);
    out[idx] = out[idx] * powspec<R>(num / den, p);
  }
}

// loss partials: lpart[b][f*nblk_t + blk]
template <typename R>
__global__ void __launch_bounds__(256) nmf_loss_kernel(const R* __restrict__ X, const R* __restrict__ Tb,
                                                      const R* __restrict__ V, double* __restrict__ lpart, int F, int T,
                                                      int K, int kind, double eps, PowSpec p2d, double p0) {
  __shared__ double sm[256];
  const int f = blockIdx.y, b = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double term = 0.0;
  if (t < T) {
    const R* tb = Tb + ((size_t)b * F + f) * K;
    const R* vb = V + (size_t)b * K * T + t;
    R tv = 0;
    for (int k = 0; k < K; ++k) tv = fma(tb[k], vb[(size_t)k * T], tv);
    const double in = (double)powspec<R>(tv, p2d);      // (T V) ** (2 / domain), not floored
    term = nmf_criterion(kind, in, (double)X[((size_t)b * F + f) * T + t], eps, p0);
  }
  sm[threadIdx.x] = term;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) lpart[((size_t)b * F + f) * gridDim.x + blockIdx.x] = sm[0];
}

__global__ void __launch_bounds__(256) sum_reduce_f64_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                            size_t L) {
  __shared__ double sm[256];
  const double* p = in + (size_t)blockIdx.x * L;
  double s = 0.0;
  for (size_t i = threadIdx.x; i < L; i += 256) s += p[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

inline unsigned nblocks(size_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

inline int pick_splits(int tiles, int Kg) {
  int s = (768 + tiles - 1) / tiles;  // aim at ~3 workgroups per CU
  int max_s = (Kg + 4 * BK - 1) / (4 * BK);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}

struct NmfWs {
  size_t ab, part, lpart, total;
  size_t part_elems;  // elements reserved at `part`: every launcher checks its slab count against it
};

constexpr int NMF_MFMA_MAX_K = 64;


// split counts of the MFMA path (deterministic: no device query)
inline int nmf_group(const assx_ctx* ctx) { return (ctx && ctx->nmf_group > 0) ? ctx->nmf_group : 1; }

// Work partitions of the two matrix-core half kernels (assx_nmf_mfma.hpp: NmfPart).  `group` = matrices per independent
// problem (assx_ctx::nmf_group), NOT the batch size: the partition must not depend on how many problems share the
// launch.  The workgroup budget is a constant of the MI355X geometry (256 CUs), so that no summation order depends
// on a runtime occupancy query.
constexpr int MFMA_WG_BUDGET = 512;  // two workgroups per CU (profiles/r04_nmf_wgs_sweep.txt)
template <typename R>
inline NmfPart mfma_basis_part(int group, int F, int T, int KT) {
  return make_nmf_part((F + 15) / 16, (T + 15) / 16, group, knob_int("ASSX_NMF_BASIS_WGS", MFMA_WG_BUDGET));
}
template <typename R>
inline NmfPart mfma_act_part(int group, int F, int T, int KT) {
  return make_nmf_part((T + 15) / 16, (F + 15) / 16, group, knob_int("ASSX_NMF_ACT_WGS", MFMA_WG_BUDGET));
}
// split-F slabs of the loss kernel
inline void mfma_loss_split(int group, int F, int T, int* FS, int* fchunk) {
  const int tg = (T + 15) / 16;
  const int wgs = 1024;
  int fs = (wgs + tg * group - 1) / (tg * group);
  const int max_fs = (F + 63) / 64;
  if (fs > max_fs) fs = max_fs;
  if (fs < 1) fs = 1;
  int chunk = ((F + fs - 1) / fs + 15) / 16 * 16;
  *fchunk = chunk;
  *FS = (F + chunk - 1) / chunk;
}

// slab counts of the small-rank kernels (assx_nmf_small.hpp): one resident round of workgroups / ~4096 waves for ONE
// problem group; a function of one problem's geometry and the group size only
inline void small_splits(int group, int KC, int F, int T, int* TS, int* tchunk, int* FS, int* fchunk) {
  const int fw = (F + 4 * small_bpw(KC) - 1) / (4 * small_bpw(KC));  // workgroups per problem and slab (4 waves x bins per wave)
  int ts = (1024 + fw * group - 1) / (fw * group);  // ~4096 waves: memory-level parallelism comes from occupancy here
  if (ts > T / 256) ts = T / 256;
  if (ts < 1) ts = 1;
  *tchunk = ((T + ts - 1) / ts + WAVE - 1) / WAVE * WAVE;
  *TS = (T + *tchunk - 1) / *tchunk;
  const int tbk = (T + WAVE - 1) / WAVE;
  int fs = (4096 + tbk * group - 1) / (tbk * group);
  if (fs > F / 16) fs = F / 16;
  if (fs < 1) fs = 1;
  *fchunk = ((F + fs - 1) / fs + 3) / 4 * 4;
  *FS = (F + *fchunk - 1) / *fchunk;
}

// partitions of the X-fed source-model halves (assx_nmf_xfed.hpp; launched by nmf_update_xfed_t below)
inline NmfPart xfed_basis_part(int F, int T, int KT) {
  // one utterance = one partition; every step is done by the M waves (sources) of a workgroup at once.
  // three workgroups per CU where their LDS (tile double buffer + staging, < 54 KB at n_basis <= 16) allows it, two
  // otherwise (profiles/r04_xfed_wgs_sweep.txt: 256 / 384 / 512 / 640 / 768 / 1024 -> 0.253 / 0.263 / 0.227 / 0.248 /
  // 0.224 / 0.247 ms per n_basis = 10 iteration)
  return make_nmf_part((F + 15) / 16, (T + 15) / 16, 1, knob_int("ASSX_NMF_XFED_WGS", KT == 1 ? 768 : MFMA_WG_BUDGET));
}
inline NmfPart xfed_act_part(int F, int T, int KT) {
  return make_nmf_part((T + 15) / 16, (F + 15) / 16, 1, knob_int("ASSX_NMF_XFED_WGS", KT == 1 ? 768 : MFMA_WG_BUDGET));
}
constexpr int XFED_MAX_K = 32;

inline NmfWs nmf_ws_compute(int B, int F, int T, int K, int dtype);
// The layout is a pure function of the shape and of the three partition knobs; it is asked for by every update / loss call,
// inside the one-call loops too, and enumerates ~70 partitions: the last answer of the calling thread is kept (round 5's
// advisor: ~3.5 us per call against ~15 us per update at config 1).  The knobs must not change between
// assx_nmf_workspace_bytes and the calls that use the buffer it sized -- the launchers compare their slab counts with the
// layout, not with the caller's real allocation.
inline NmfWs nmf_ws(int B, int F, int T, int K, int dtype) {
  struct Key {
    int B, F, T, K, dtype, kb, ka, kx;
  };
  const Key key{B, F, T, K, dtype, knob_int("ASSX_NMF_BASIS_WGS", -1), knob_int("ASSX_NMF_ACT_WGS", -1),
                knob_int("ASSX_NMF_XFED_WGS", -1)};
  thread_local Key last{};
  thread_local NmfWs last_ws{};
  thread_local bool have = false;
  if (have && memcmp(&last, &key, sizeof(Key)) == 0) return last_ws;
  last_ws = nmf_ws_compute(B, F, T, K, dtype);
  last = key;
  have = true;
  return last_ws;
}
inline NmfWs nmf_ws_compute(int B, int F, int T, int K, int dtype) {
  const size_t r = dtype == ASSX_F64 ? 8 : 4;
  NmfWs w;
  w.ab = 0;
  size_t off = align_up((size_t)B * 2 * F * T * r, 256);
  w.part = off;
  // split-K slabs: basis (2B, F, K) x splits_t ; activation (2B, K, T) x splits_f  (bounded above)
  const size_t tiles_b = (size_t)((F + BM - 1) / BM) * ((K + BN - 1) / BN) * 2;  // group = 1: the most slabs
  const size_t tiles_a = (size_t)((K + BM - 1) / BM) * ((T + BN - 1) / BN) * 2;
  const size_t sb = pick_splits((int)tiles_b, T), sa = pick_splits((int)tiles_a, F);
  size_t pmax = sb * 2 * B * F * K;
  if (sa * 2 * B * K * T > pmax) pmax = sa * 2 * B * K * T;
  {
    // matrix-core halves: the slab bound of every group size a caller may set (assx_ctx::nmf_group: 1 for plain NMF, the
    // number of sources for the ILRMA source models).  Until round 4 group = 1 had the most slabs; since the partition is
    // block-aligned where a block gets two workgroups and flat otherwise, a larger group can fall back to the flat form
    // with a (loose) bound above the aligned one of group 1.
    const int KTv = (K + 15) / 16;
    for (int grp = 1; grp <= 32; ++grp) {
      const NmfPart pb = dtype == ASSX_F64 ? mfma_basis_part<double>(grp, F, T, KTv) : mfma_basis_part<float>(grp, F, T, KTv);
      const NmfPart pa = dtype == ASSX_F64 ? mfma_act_part<double>(grp, F, T, KTv) : mfma_act_part<float>(grp, F, T, KTv);
      if ((size_t)pb.maxslots * 2 * B * F * K > pmax) pmax = (size_t)pb.maxslots * 2 * B * F * K;
      if ((size_t)pa.maxslots * 2 * B * K * T > pmax) pmax = (size_t)pa.maxslots * 2 * B * K * T;
    }
    if (K <= XFED_MAX_K) {
      // the X-fed halves have their own (larger) workgroup budget: a shorter range per workgroup means MORE workgroups
      // meet one block, hence more slabs than the map-fed partitions above (F = 1025, T = 660, n_basis = 10: 12
      // against 11 -- round 4 sized the area without them and the 12th slab landed past it)
      const NmfPart xb = xfed_basis_part(F, T, KTv), xa = xfed_act_part(F, T, KTv);
      if ((size_t)xb.maxslots * 2 * B * F * K > pmax) pmax = (size_t)xb.maxslots * 2 * B * F * K;
      if ((size_t)xa.maxslots * 2 * B * K * T > pmax) pmax = (size_t)xa.maxslots * 2 * B * K * T;
    }
    if (K <= SMALL_K) {  // small-rank kernels; group = 1 gives the most slabs
      int TS, tchunk, FS, fchunk;
      small_splits(1, small_kc(K), F, T, &TS, &tchunk, &FS, &fchunk);
      if ((size_t)TS * 2 * B * F * K > pmax) pmax = (size_t)TS * 2 * B * F * K;
      if ((size_t)FS * 2 * B * K * T > pmax) pmax = (size_t)FS * 2 * B * K * T;
    }
  }
  w.part_elems = pmax;
  off += align_up(pmax * r, 256);
  w.lpart = off;
  {
    int FS, fchunk;
    mfma_loss_split(1, F, T, &FS, &fchunk);
    size_t nl = (size_t)B * F * ((T + 255) / 256);
    if ((size_t)B * FS * ((T + 15) / 16) > nl) nl = (size_t)B * FS * ((T + 15) / 16);
    // loss on the basis half: one partial per (workgroup, wave); group = 1 has the most workgroups
    const size_t fused = (size_t)B * 4 * (size_t)(dtype == ASSX_F64 ? mfma_basis_part<double>(1, F, T, (K + 15) / 16).G
                                                                     : mfma_basis_part<float>(1, F, T, (K + 15) / 16).G);
    if (fused > nl) nl = fused;
    off += align_up(nl * 8, 256);
  }
  w.total = off;
  return w;
}

// matrix-core path (n_basis <= 64): two chained-MFMA kernels, each finalizing its own output behind a ticket
// (assx_nmf_mfma.hpp) -- one update is 2 launches (4 until round 3)
template <typename R, int KT>
int nmf_update_mfma(assx_ctx* ctx, int kind, double domain, double param, double eps, const void* X, void* Tb, void* V, void* ws,
                    int B, int F, int T, int K, int dtype, hipStream_t st, double* loss_prev = nullptr) {
  const NmfWs L = nmf_ws(B, F, T, K, dtype);
  R* part = (R*)((char*)ws + L.part);
  double* lpart = (double*)((char*)ws + L.lpart);
  const TermSpec ts = make_terms(kind, domain, param);
  const PowSpec pe = update_exponent(kind, domain);
  const NmfPart pb = mfma_basis_part<R>(nmf_group(ctx), F, T, KT), pa = mfma_act_part<R>(nmf_group(ctx), F, T, KT);
  if ((size_t)pb.maxslots * 2 * B * F * K > L.part_elems || (size_t)pa.maxslots * 2 * B * K * T > L.part_elems)
    return fail(ctx, ASSX_E_UNSUPPORTED, "NMF halves: %d / %d slabs per block exceed the workspace's slab area", pb.maxslots,
                pa.maxslots);
  int* tickets = nullptr;
  const int trc = ensure_tickets(ctx, (size_t)B * (pb.nblk > pa.nblk ? pb.nblk : pa.nblk), st, &tickets);
  if (trc) return trc;  // the hipError_t of the allocation, message in ctx
  const bool d2 = domain == 2.0 && kind < ASSX_NMF_T;  // every exponent is 0, 1 or 2: pow()-free instantiations
#define NMF_BASIS(D2K)                                                                                         \
  hipLaunchKernelGGL((nmf_basis_mfma_kernel<R, KT, D2K>), dim3(pb.G, 1, B), dim3(256), 0, st, (const R*)X,      \
                     (R*)Tb, (const R*)V, part, tickets, 1, pb, B, F, T, K, (R)eps, ts, pe, (double*)nullptr, 0, 0.0)
#define NMF_BASIS_LOSS(D2K)                                                                                    \
  hipLaunchKernelGGL((nmf_basis_mfma_kernel<R, KT, D2K, true>), dim3(pb.G, 1, B), dim3(256), 0, st, (const R*)X, \
                     (R*)Tb, (const R*)V, part, tickets, 1, pb, B, F, T, K, (R)eps, ts, pe, lpart, 4 * pb.G, eps)
#define NMF_ACT(D2K)                                                                                           \
  hipLaunchKernelGGL((nmf_act_mfma_kernel<R, KT, D2K>), dim3(pa.G, 1, B), dim3(256), 0, st, (const R*)X,         \
                     (const R*)Tb, (R*)V, part, tickets, 1, pa, B, F, T, K, (R)eps, ts, pe)
  if (loss_prev && !d2) return fail(ctx, ASSX_E_UNSUPPORTED, "nmf_update_mfma: the fused loss needs domain 2 and EUC / KL / IS");
  if (loss_prev) {
    // the criterion of the model AT ENTRY rides on the basis half; its per-wave partials are summed right behind it
    if (kind == ASSX_NMF_EUC) NMF_BASIS_LOSS(ASSX_NMF_EUC);
    else if (kind == ASSX_NMF_KL) NMF_BASIS_LOSS(ASSX_NMF_KL);
    else NMF_BASIS_LOSS(ASSX_NMF_IS_MM);
    ASSX_LAUNCH_CHECK(ctx, "nmf_basis_mfma_kernel(loss)");
    hipLaunchKernelGGL(sum_reduce_f64_kernel, dim3(B), dim3(256), 0, st, (const double*)lpart, loss_prev, (size_t)4 * pb.G);
    ASSX_LAUNCH_CHECK(ctx, "sum_reduce_f64_kernel");
  } else if (d2 && kind == ASSX_NMF_EUC) NMF_BASIS(ASSX_NMF_EUC);
  else if (d2 && kind == ASSX_NMF_KL) NMF_BASIS(ASSX_NMF_KL);
  else if (d2) NMF_BASIS(ASSX_NMF_IS_MM);
  else NMF_BASIS(-1);
  ASSX_LAUNCH_CHECK(ctx, "nmf_basis_mfma_kernel");
  if (d2 && kind == ASSX_NMF_EUC) NMF_ACT(ASSX_NMF_EUC);
  else if (d2 && kind == ASSX_NMF_KL) NMF_ACT(ASSX_NMF_KL);
  else if (d2) NMF_ACT(ASSX_NMF_IS_MM);
  else NMF_ACT(-1);
#undef NMF_BASIS
#undef NMF_BASIS_LOSS
#undef NMF_ACT
  ASSX_LAUNCH_CHECK(ctx, "nmf_act_mfma_kernel");
  return 0;
}

// n_basis <= 4, IS rule, domain 2: the vector-ALU kernels of assx_nmf_small.hpp (the matrix-core kernels pad the rank to
// 16).  Slab counts: one resident round of workgroups for ONE problem group, never more slabs than the matrix-core
// path would use (the scratch is sized for those).
template <typename R, int KC>
int nmf_update_small_kc(assx_ctx* ctx, double eps, const void* X, void* Tb, void* V, void* ws, int B, int F, int T, int K,
                        int dtype, hipStream_t st) {
  const NmfWs L = nmf_ws(B, F, T, K, dtype);
  R* part = (R*)((char*)ws + L.part);
  const PowSpec pe = update_exponent(ASSX_NMF_IS_MM, 2.0);
  int TS, tchunk, FS, fchunk;
  small_splits(nmf_group(ctx), KC, F, T, &TS, &tchunk, &FS, &fchunk);
  const int fw = (F + 4 * small_bpw(KC) - 1) / (4 * small_bpw(KC)), tbk = (T + WAVE - 1) / WAVE;
  hipLaunchKernelGGL((nmf_basis_small_kernel<R, KC>), dim3(fw, TS, B), dim3(256), 0, st, (const R*)X, (const R*)Tb,
                     (const R*)V, part, B, F, T, K, tchunk, (R)eps);
  ASSX_LAUNCH_CHECK(ctx, "nmf_basis_small_kernel");
  hipLaunchKernelGGL((nmf_finalize_kernel<R>), dim3(nblocks((size_t)B * F * K, 64)), dim3(256), 0, st, (const R*)part,
                     (R*)Tb, B, (size_t)F * K, TS, (R)eps, pe);
  ASSX_LAUNCH_CHECK(ctx, "nmf_finalize_kernel(basis)");
  hipLaunchKernelGGL((nmf_act_small_kernel<R, KC>), dim3(tbk, FS, B), dim3(64), 0, st, (const R*)X, (const R*)Tb,
                     (const R*)V, part, B, F, T, K, fchunk, (R)eps);
  ASSX_LAUNCH_CHECK(ctx, "nmf_act_small_kernel");
  hipLaunchKernelGGL((nmf_finalize_kernel<R>), dim3(nblocks((size_t)B * K * T, 64)), dim3(256), 0, st, (const R*)part,
                     (R*)V, B, (size_t)K * T, FS, (R)eps, pe);
  ASSX_LAUNCH_CHECK(ctx, "nmf_finalize_kernel(activation)");
  return 0;
}
// n_basis <= 4 (ASSX_NMF_SMALL=0: never), IS rule, domain 2: the vector-ALU kernels of assx_nmf_small.hpp (the
// matrix-core kernels pad the rank to a multiple of 16)
template <typename R>
int nmf_update_small(assx_ctx* ctx, double eps, const void* X, void* Tb, void* V, void* ws, int B, int F, int T, int K,
                     int dtype, hipStream_t st) {
  return nmf_update_small_kc<R, 4>(ctx, eps, X, Tb, V, ws, B, F, T, K, dtype, st);
}
inline int small_rank_max() {  // largest n_basis routed to the vector-ALU kernels (0 = never)
  static const int v = lab_int("ASSX_NMF_SMALL", 4);
  return v > SMALL_K ? SMALL_K : v;
}

template <typename R>
int nmf_update_impl(assx_ctx* ctx, int kind, double domain, double param, double eps, const void* X, void* Tb, void* V, void* ws,
                    int B, int F, int T, int K, int dtype, hipStream_t st, double* loss_prev = nullptr) {
  static const int no_mfma = lab_int("ASSX_NMF_NO_MFMA", 0);
  if (loss_prev) {
    // loss of the model at entry: on the basis half where that kernel can (matrix-core path, domain 2, EUC / KL / IS), else
    // by the stand-alone pass first
    static const int fuse = lab_int("ASSX_NMF_FUSE_LOSS", 1);
    const bool fits = (unsigned long long)F * T * sizeof(R) < (1ull << 32) && (unsigned long long)K * T * sizeof(R) < (1ull << 32);
    const bool small = K <= small_rank_max() && kind == ASSX_NMF_IS_MM && domain == 2.0;
    const bool fusable = fuse && !no_mfma && fits && !small && K <= NMF_MFMA_MAX_K && domain == 2.0 && kind <= ASSX_NMF_IS_ME;
    if (!fusable) {
      int rc = assx_nmf_loss_ex(ctx, kind, domain, param, eps, X, Tb, V, loss_prev, ws, B, F, T, K, dtype, (void*)st);
      if (rc) return rc;
      loss_prev = nullptr;
    }
  }
  if (K <= small_rank_max() && kind == ASSX_NMF_IS_MM && domain == 2.0 && !no_mfma)
    return nmf_update_small<R>(ctx, eps, X, Tb, V, ws, B, F, T, K, dtype, st);
  // the matrix-core kernels address a matrix with 32-bit byte offsets
  const bool fits32 = (unsigned long long)F * T * sizeof(R) < (1ull << 32) && (unsigned long long)K * T * sizeof(R) < (1ull << 32);
  if (K <= NMF_MFMA_MAX_K && !no_mfma && fits32) {
    switch ((K + 15) / 16) {
      case 1: return nmf_update_mfma<R, 1>(ctx, kind, domain, param, eps, X, Tb, V, ws, B, F, T, K, dtype, st, loss_prev);
      case 2: return nmf_update_mfma<R, 2>(ctx, kind, domain, param, eps, X, Tb, V, ws, B, F, T, K, dtype, st, loss_prev);
      case 3: return nmf_update_mfma<R, 3>(ctx, kind, domain, param, eps, X, Tb, V, ws, B, F, T, K, dtype, st, loss_prev);
      default: return nmf_update_mfma<R, 4>(ctx, kind, domain, param, eps, X, Tb, V, ws, B, F, T, K, dtype, st, loss_prev);
    }
  }
  const NmfWs L = nmf_ws(B, F, T, K, dtype);
  R* AB = (R*)((char*)ws + L.ab);
  R* part = (R*)((char*)ws + L.part);
  const TermSpec ts = make_terms(kind, domain, param);
  const PowSpec pe = update_exponent(kind, domain);
  const size_t FT = (size_t)F * T;
  dim3 tgrid(nblocks(T, 256), F, B);

  // ---------------- basis: num|den (F,K) = [A|Bm] (F,T) . V^T (T,K)
  hipLaunchKernelGGL((nmf_terms_kernel<R>), tgrid, dim3(256), 0, st, (const R*)X, (const R*)Tb, (const R*)V, AB, F, T, K,
                     (R)eps, ts);
  ASSX_LAUNCH_CHECK(ctx, "nmf_terms_kernel");
  {
    GemmArgs g;
    g.A = AB;
    g.Bm = V;
    g.C = part;
    g.Mg = F;
    g.Ng = K;
    g.Kg = T;
    g.sam = T;
    g.sak = 1;        // A[m=f][k=t]
    g.sbk = 1;
    g.sbn = T;        // Bop[k=t][n=kb] = V[kb][t]
    g.za_outer = 2 * (long)FT;
    g.za_inner = (long)FT;
    g.zb_outer = (long)K * T;
    g.zb_inner = 0;
    g.Z = 2 * B;
    const int tiles = ((F + BM - 1) / BM) * ((K + BN - 1) / BN) * 2 * nmf_group(ctx);
    g.splits = pick_splits(tiles, T);
    g.kchunk = ((T + g.splits - 1) / g.splits + BK - 1) / BK * BK;
    g.splits = (T + g.kchunk - 1) / g.kchunk;
    dim3 grid((K + BN - 1) / BN, (F + BM - 1) / BM, g.Z * g.splits);
    hipLaunchKernelGGL((gemm_tile_kernel<R, true, true>), grid, dim3(256), 0, st, g);
    ASSX_LAUNCH_CHECK(ctx, "gemm_tile_kernel(basis)");
    hipLaunchKernelGGL((nmf_finalize_kernel<R>), dim3(nblocks((size_t)B * F * K, 64)), dim3(256), 0, st,
                       (const R*)part, (R*)Tb, B, (size_t)F * K, g.splits, (R)eps, pe);
    ASSX_LAUNCH_CHECK(ctx, "nmf_finalize_kernel(basis)");
  }
  // ---------------- activation (new basis): num|den (K,T) = Tb^T (K,F) . [A|Bm] (F,T)
  hipLaunchKernelGGL((nmf_terms_kernel<R>), tgrid, dim3(256), 0, st, (const R*)X, (const R*)Tb, (const R*)V, AB, F, T, K,
                     (R)eps, ts);
  ASSX_LAUNCH_CHECK(ctx, "nmf_terms_kernel");
  {
    GemmArgs g;
    g.A = Tb;
    g.Bm = AB;
    g.C = part;
    g.Mg = K;
    g.Ng = T;
    g.Kg = F;
    g.sam = 1;
    g.sak = K;        // A[m=kb][k=f] = Tb[f][kb]
    g.sbk = T;
    g.sbn = 1;        // Bop[k=f][n=t] = AB[f][t]
    g.za_outer = (long)F * K;
    g.za_inner = 0;
    g.zb_outer = 2 * (long)FT;
    g.zb_inner = (long)FT;
    g.Z = 2 * B;
    const int tiles = ((K + BM - 1) / BM) * ((T + BN - 1) / BN) * 2 * nmf_group(ctx);
    g.splits = pick_splits(tiles, F);
    g.kchunk = ((F + g.splits - 1) / g.splits + BK - 1) / BK * BK;
    g.splits = (F + g.kchunk - 1) / g.kchunk;
    dim3 grid((T + BN - 1) / BN, (K + BM - 1) / BM, g.Z * g.splits);
    hipLaunchKernelGGL((gemm_tile_kernel<R, false, false>), grid, dim3(256), 0, st, g);
    ASSX_LAUNCH_CHECK(ctx, "gemm_tile_kernel(activation)");
    hipLaunchKernelGGL((nmf_finalize_kernel<R>), dim3(nblocks((size_t)B * K * T, 64)), dim3(256), 0, st,
                       (const R*)part, (R*)V, B, (size_t)K * T, g.splits, (R)eps, pe);
    ASSX_LAUNCH_CHECK(ctx, "nmf_finalize_kernel(activation)");
  }
  return 0;
}

}  // namespace

namespace assx {

template <typename R, int KT>
static int nmf_half_launch(assx_ctx* ctx, const TermSpec& ts, int half, const void* X, const void* Tb, const void* V,
                           R* part, int B, int F, int T, int K, double eps, hipStream_t st, int* slabs) {
  if (half == NMF_HALF_BASIS) {
    const NmfPart pb = mfma_basis_part<R>(nmf_group(ctx), F, T, KT);
    hipLaunchKernelGGL((nmf_basis_mfma_kernel<R, KT, -1>), dim3(pb.G, 1, B), dim3(256), 0, st, (const R*)X,
                       (R*)const_cast<void*>(Tb), (const R*)V, part, (int*)nullptr, 0, pb, B, F, T, K, (R)eps, ts,
                       make_pow(1.0));
    *slabs = pb.maxslots;  // unused slabs of a block are cleared by the kernel
  } else {
    const NmfPart pa = mfma_act_part<R>(nmf_group(ctx), F, T, KT);
    hipLaunchKernelGGL((nmf_act_mfma_kernel<R, KT, -1>), dim3(pa.G, 1, B), dim3(256), 0, st, (const R*)X,
                       (const R*)Tb, (R*)const_cast<void*>(V), part, (int*)nullptr, 0, pa, B, F, T, K, (R)eps, ts,
                       make_pow(1.0));
    *slabs = pa.maxslots;
  }
  ASSX_LAUNCH_CHECK(ctx, "nmf half kernel");
  return 0;
}

static bool xfed_applies(int kind, int M, int K) {
  static const int on = lab_int("ASSX_NMF_XFED", 1);  // laboratory builds, 0: the power-map route of rounds 1-3 (A/B runs)
  return on && M >= 2 && M <= 4 && K <= XFED_MAX_K && (kind == ASSX_NMF_IS_MM || kind == ASSX_NMF_T_RAW);
}

template <typename R, int M, int KT>
static int nmf_update_xfed_t(assx_ctx* ctx, int kind, double domain, double param, double eps, const void* X,
                             const void* W, void* Tb, void* V, void* ws, int B, int F, int T, int K, int dtype,
                             hipStream_t st, double* lpart, int lstride) {
  const NmfWs L = nmf_ws(B * M, F, T, K, dtype);
  R* part = (R*)((char*)ws + L.part);
  const TermSpec ts = make_terms(kind, domain, param);
  const PowSpec pe = update_exponent(kind, domain);
  const NmfPart pb = xfed_basis_part(F, T, KT), pa = xfed_act_part(F, T, KT);
  // the slab area is sized by nmf_ws from these very partitions; refuse rather than write past it if the two ever disagree
  if ((size_t)pb.maxslots * 2 * B * M * F * K > L.part_elems || (size_t)pa.maxslots * 2 * B * M * K * T > L.part_elems)
    return fail(ctx, ASSX_E_UNSUPPORTED, "X-fed source model: %d / %d slabs per block exceed the workspace's slab area",
                pb.maxslots, pa.maxslots);
  int* tickets = nullptr;
  const int trc = ensure_tickets(ctx, (size_t)B * (pb.nblk > pa.nblk ? pb.nblk : pa.nblk), st, &tickets);
  if (trc) return trc;
  const bool d2 = domain == 2.0 && kind == ASSX_NMF_IS_MM;
#define XFED(D2K)                                                                                                        \
  do {                                                                                                                   \
    if (lpart)                                                                                                           \
      hipLaunchKernelGGL((nmf_basis_xfed_kernel<R, M, KT, ASSX_NMF_IS_MM, true>), dim3(pb.G, 1, B), dim3(64 * M), 0, st, \
                         (const Cx<R>*)X, (const Cx<R>*)W, (R*)Tb, (const R*)V, part, tickets, pb, B, F, T, K, (R)eps,   \
                         ts, pe, lpart, lstride);                                                                        \
    else                                                                                                                 \
      hipLaunchKernelGGL((nmf_basis_xfed_kernel<R, M, KT, D2K>), dim3(pb.G, 1, B), dim3(64 * M), 0, st, (const Cx<R>*)X, \
                         (const Cx<R>*)W, (R*)Tb, (const R*)V, part, tickets, pb, B, F, T, K, (R)eps, ts, pe,            \
                         (double*)nullptr, 0);                                                                           \
    ASSX_LAUNCH_CHECK(ctx, "nmf_basis_xfed_kernel");                                                                     \
    hipLaunchKernelGGL((nmf_act_xfed_kernel<R, M, KT, D2K>), dim3(pa.G, 1, B), dim3(64 * M), 0, st, (const Cx<R>*)X,     \
                       (const Cx<R>*)W, (const R*)Tb, (R*)V, part, tickets, pa, B, F, T, K, (R)eps, ts, pe);             \
    ASSX_LAUNCH_CHECK(ctx, "nmf_act_xfed_kernel");                                                                       \
  } while (0)
  if (d2) XFED(ASSX_NMF_IS_MM);
  else XFED(-1);
#undef XFED
  return 0;
}

int nmf_update_with_loss(assx_ctx* ctx, int kind, double domain, double param, double eps, const void* X, void* Tb,
                         void* V, double* loss_prev, void* ws, int B, int F, int T, int K, int dtype, hipStream_t st) {
  if (dtype == ASSX_F64) return nmf_update_impl<double>(ctx, kind, domain, param, eps, X, Tb, V, ws, B, F, T, K, dtype, st, loss_prev);
  if (dtype == ASSX_F32) return nmf_update_impl<float>(ctx, kind, domain, param, eps, X, Tb, V, ws, B, F, T, K, dtype, st, loss_prev);
  return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
}

int nmf_xfed_loss_partials(int M, int F, int T, int K) {
  if (!xfed_applies(ASSX_NMF_IS_MM, M, K)) return 0;
  return xfed_basis_part(F, T, (K + 15) / 16).G * M;
}

int nmf_update_xfed(assx_ctx* ctx, int kind, double domain, double param, double eps, const void* X, const void* W,
                    void* Tb, void* V, void* ws, int B, int M, int F, int T, int K, int dtype, hipStream_t st,
                    double* lpart, int lstride) {
  if (!xfed_applies(kind, M, K)) return ASSX_E_UNSUPPORTED;
  if (lpart && !(kind == ASSX_NMF_IS_MM && domain == 2.0)) return ASSX_E_UNSUPPORTED;
  if ((unsigned long long)M * F * T * 16 >= (1ull << 32) || (unsigned long long)K * T * 8 >= (1ull << 32)) return ASSX_E_UNSUPPORTED;
  // the partition arithmetic of the kernels is 32-bit and a workgroup holds < 64 KB of LDS
#define XFED_BY(RT)                                                                                                     \
  switch (M * 10 + (K + 15) / 16) {                                                                                     \
    case 21: return nmf_update_xfed_t<RT, 2, 1>(ctx, kind, domain, param, eps, X, W, Tb, V, ws, B, F, T, K, dtype, st, lpart, lstride); \
    case 22: return nmf_update_xfed_t<RT, 2, 2>(ctx, kind, domain, param, eps, X, W, Tb, V, ws, B, F, T, K, dtype, st, lpart, lstride); \
    case 31: return nmf_update_xfed_t<RT, 3, 1>(ctx, kind, domain, param, eps, X, W, Tb, V, ws, B, F, T, K, dtype, st, lpart, lstride); \
    case 32: return nmf_update_xfed_t<RT, 3, 2>(ctx, kind, domain, param, eps, X, W, Tb, V, ws, B, F, T, K, dtype, st, lpart, lstride); \
    case 41: return nmf_update_xfed_t<RT, 4, 1>(ctx, kind, domain, param, eps, X, W, Tb, V, ws, B, F, T, K, dtype, st, lpart, lstride); \
    case 42: return nmf_update_xfed_t<RT, 4, 2>(ctx, kind, domain, param, eps, X, W, Tb, V, ws, B, F, T, K, dtype, st, lpart, lstride); \
  }
  if (dtype == ASSX_F64) {
    XFED_BY(double)
  } else if (dtype == ASSX_F32) {
    XFED_BY(float)
  }
#undef XFED_BY
  return ASSX_E_UNSUPPORTED;
}

int nmf_half_partials(assx_ctx* ctx, int kind, double domain, double param, double eps, int half, const void* X,
                      const void* Tb, const void* V, void* ws, int B, int F, int T, int K, int dtype, hipStream_t st,
                      const void** part, int* slabs) {
  if (K > NMF_MFMA_MAX_K) return fail(ctx, ASSX_E_UNSUPPORTED, "nmf_half_partials: n_basis %d > %d", K, NMF_MFMA_MAX_K);
  if ((unsigned long long)F * T * 8 >= (1ull << 32) || (unsigned long long)K * T * 8 >= (1ull << 32))
    return fail(ctx, ASSX_E_UNSUPPORTED, "nmf_half_partials: one matrix must stay below 4 GiB (F = %d, T = %d)", F, T);
  const NmfWs L = nmf_ws(B, F, T, K, dtype);
  const TermSpec ts = make_terms(kind, domain, param);
  void* p = (char*)ws + L.part;
  *part = p;
  if (K <= small_rank_max() && kind == ASSX_NMF_IS_MM && domain == 2.0 && (dtype == ASSX_F64 || dtype == ASSX_F32)) {
    // small n_basis: the vector-ALU halves (assx_nmf_small.hpp), same record layout
    const int kc = small_kc(K);
    int TS, tchunk, FS, fchunk;
    small_splits(nmf_group(ctx), kc, F, T, &TS, &tchunk, &FS, &fchunk);
    const int fw = (F + 4 * small_bpw(kc) - 1) / (4 * small_bpw(kc)), tbk = (T + WAVE - 1) / WAVE;
#define SMALL_HALF(RT, KCV)                                                                                          \
  if (half == NMF_HALF_BASIS)                                                                                        \
    hipLaunchKernelGGL((nmf_basis_small_kernel<RT, KCV>), dim3(fw, TS, B), dim3(256), 0, st, (const RT*)X,            \
                       (const RT*)Tb, (const RT*)V, (RT*)p, B, F, T, K, tchunk, (RT)eps);                            \
  else                                                                                                               \
    hipLaunchKernelGGL((nmf_act_small_kernel<RT, KCV>), dim3(tbk, FS, B), dim3(64), 0, st, (const RT*)X,              \
                       (const RT*)Tb, (const RT*)V, (RT*)p, B, F, T, K, fchunk, (RT)eps)
#define SMALL_HALF_BY_KC(RT) SMALL_HALF(RT, 4);
    if (dtype == ASSX_F64) {
      SMALL_HALF_BY_KC(double)
    } else {
      SMALL_HALF_BY_KC(float)
    }
#undef SMALL_HALF_BY_KC
#undef SMALL_HALF
    *slabs = half == NMF_HALF_BASIS ? TS : FS;
    ASSX_LAUNCH_CHECK(ctx, "nmf small-rank half kernel");
    return 0;
  }
#define HALF_BY_K(RT)                                                                                              \
  switch ((K + 15) / 16) {                                                                                         \
    case 1: return nmf_half_launch<RT, 1>(ctx, ts, half, X, Tb, V, (RT*)p, B, F, T, K, eps, st, slabs);            \
    case 2: return nmf_half_launch<RT, 2>(ctx, ts, half, X, Tb, V, (RT*)p, B, F, T, K, eps, st, slabs);            \
    case 3: return nmf_half_launch<RT, 3>(ctx, ts, half, X, Tb, V, (RT*)p, B, F, T, K, eps, st, slabs);            \
    default: return nmf_half_launch<RT, 4>(ctx, ts, half, X, Tb, V, (RT*)p, B, F, T, K, eps, st, slabs);           \
  }
  if (dtype == ASSX_F64) {
    HALF_BY_K(double)
  } else if (dtype == ASSX_F32) {
    HALF_BY_K(float)
  }
#undef HALF_BY_K
  return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
}

}  // namespace assx

extern "C" {

size_t assx_nmf_workspace_bytes(int B, int F, int T, int K, int dtype) {
  if (B < 1 || F < 1 || T < 1 || K < 1) return 0;
  return nmf_ws(B, F, T, K, dtype).total;
}

int assx_nmf_partition_query(int feed, int half, int group, int F, int T, int K, int dtype, int32_t out[6]) {
  if (!out || F < 1 || T < 1 || K < 1 || K > NMF_MFMA_MAX_K || (dtype != ASSX_F64 && dtype != ASSX_F32)) return ASSX_E_ARG;
  if ((feed != 0 && feed != 1) || (half != 0 && half != 1) || group < 1) return ASSX_E_ARG;
  if (feed == 1 && K > XFED_MAX_K) return ASSX_E_ARG;
  const int KT = (K + 15) / 16;
  NmfPart p;
  if (feed == 1) p = half == 0 ? xfed_basis_part(F, T, KT) : xfed_act_part(F, T, KT);
  else if (dtype == ASSX_F64) p = half == 0 ? mfma_basis_part<double>(group, F, T, KT) : mfma_act_part<double>(group, F, T, KT);
  else p = half == 0 ? mfma_basis_part<float>(group, F, T, KT) : mfma_act_part<float>(group, F, T, KT);
  // workgroup g owns steps [lo(g), lo(g + 1)) of the flattened (block, step) space and writes slab g - (first workgroup
  // of the block) for every block its range meets
  int worst = 0;
  for (int blk = 0; blk < p.nblk; ++blk) {
    const unsigned first = (unsigned)blk * (unsigned)p.nstep, last = first + (unsigned)p.nstep - 1u;
    const int n = nmf_part_owner(p, last) - nmf_part_owner(p, first) + 1;
    if (n > worst) worst = n;
  }
  const NmfWs L = nmf_ws(1, F, T, K, dtype);
  const size_t per_slab = half == 0 ? (size_t)2 * F * K : (size_t)2 * K * T;
  out[0] = p.G;
  out[1] = p.nblk;
  out[2] = p.nstep;
  out[3] = p.maxslots;
  out[4] = worst;
  out[5] = (int32_t)(L.part_elems / per_slab);
  return 0;
}

int assx_nmf_update_ex(assx_ctx* ctx, int kind, double domain, double param, double eps, const void* X, void* Tb,
                       void* V, void* ws, int B, int F, int T, int K, int dtype, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, B >= 1 && F >= 1 && T >= 1 && K >= 1, ASSX_E_ARG, "invalid sizes B=%d F=%d T=%d K=%d", B, F, T, K);
  ASSX_REQUIRE(ctx, X && Tb && V && ws, ASSX_E_NULL, "assx_nmf_update: NULL array");
  ASSX_REQUIRE(ctx, kind >= ASSX_NMF_EUC && kind <= ASSX_NMF_T_RAW, ASSX_E_ARG, "bad NMF kind %d", kind);
  ASSX_REQUIRE(ctx, kind != ASSX_NMF_T_RAW || param >= 0.0, ASSX_E_ARG, "nu must be >= 0, got %g", param);
  ASSX_REQUIRE(ctx, domain >= 1.0 && domain <= 2.0, ASSX_E_ARG, "1 <= domain <= 2 is not satisfied (%g)", domain);
  ASSX_REQUIRE(ctx, kind != ASSX_NMF_IS_ME || domain == 2.0, ASSX_E_ARG, "Only domain = 2 is supported (IS me).");
  ASSX_REQUIRE(ctx, kind < ASSX_NMF_T || domain == 2.0, ASSX_E_ARG, "Only domain = 2 is supported (tNMF, CauchyNMF).");
  ASSX_REQUIRE(ctx, kind != ASSX_NMF_T || param > 0.0, ASSX_E_ARG, "tNMF: nu must be > 0, got %g", param);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ASSX_F64) return nmf_update_impl<double>(ctx, kind, domain, param, eps, X, Tb, V, ws, B, F, T, K, dtype, st);
  if (dtype == ASSX_F32) return nmf_update_impl<float>(ctx, kind, domain, param, eps, X, Tb, V, ws, B, F, T, K, dtype, st);
  return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
}

int assx_nmf_half_sums(assx_ctx* ctx, int kind, double domain, double param, double eps, int half, const void* X,
                       const void* Tb, const void* V, void* sums, void* ws, int B, int F, int T, int K, int dtype,
                       void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, B >= 1 && F >= 1 && T >= 1 && K >= 1, ASSX_E_ARG, "invalid sizes B=%d F=%d T=%d K=%d", B, F, T, K);
  ASSX_REQUIRE(ctx, X && Tb && V && sums && ws, ASSX_E_NULL, "assx_nmf_half_sums: NULL array");
  ASSX_REQUIRE(ctx, kind >= ASSX_NMF_EUC && kind <= ASSX_NMF_T_RAW, ASSX_E_ARG, "bad NMF kind %d", kind);
  ASSX_REQUIRE(ctx, half == 0 || half == 1, ASSX_E_ARG, "half must be 0 (basis) or 1 (activation), got %d", half);
  ASSX_REQUIRE(ctx, domain >= 1.0 && domain <= 2.0, ASSX_E_ARG, "1 <= domain <= 2 is not satisfied (%g)", domain);
  ASSX_REQUIRE(ctx, kind < ASSX_NMF_T || domain == 2.0, ASSX_E_ARG, "Only domain = 2 is supported (tNMF, CauchyNMF).");
  hipStream_t st = (hipStream_t)stream;
  const void* part = nullptr;
  int slabs = 0;
  int rc = nmf_half_partials(ctx, kind, domain, param, eps, half, X, Tb, V, ws, B, F, T, K, dtype, st, &part, &slabs);
  if (rc) return rc;
  const size_t count = half == NMF_HALF_BASIS ? (size_t)F * K : (size_t)K * T;
  if (dtype == ASSX_F64)
    hipLaunchKernelGGL((nmf_sum_slabs_kernel<double>), dim3(nblocks((size_t)B * count, 64)), dim3(256), 0, st,
                       (const double*)part, (double*)sums, B, count, slabs);
  else
    hipLaunchKernelGGL((nmf_sum_slabs_kernel<float>), dim3(nblocks((size_t)B * count, 64)), dim3(256), 0, st,
                       (const float*)part, (float*)sums, B, count, slabs);
  ASSX_LAUNCH_CHECK(ctx, "nmf_sum_slabs_kernel");
  return 0;
}

int assx_nmf_apply_sums(assx_ctx* ctx, int kind, double domain, double eps, void* A, const void* sums, int B,
                        long long count, int dtype, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, B >= 1 && count >= 1, ASSX_E_ARG, "invalid sizes B=%d count=%lld", B, count);
  ASSX_REQUIRE(ctx, A && sums, ASSX_E_NULL, "assx_nmf_apply_sums: NULL array");
  ASSX_REQUIRE(ctx, kind >= ASSX_NMF_EUC && kind <= ASSX_NMF_T_RAW, ASSX_E_ARG, "bad NMF kind %d", kind);
  hipStream_t st = (hipStream_t)stream;
  const PowSpec pe = update_exponent(kind, domain);
  const size_t total = (size_t)B * (size_t)count;
  if (dtype == ASSX_F64)
    hipLaunchKernelGGL((nmf_apply_sums_kernel<double>), dim3(nblocks(total, 256)), dim3(256), 0, st, (double*)A,
                       (const double*)sums, total, eps, pe);
  else if (dtype == ASSX_F32)
    hipLaunchKernelGGL((nmf_apply_sums_kernel<float>), dim3(nblocks(total, 256)), dim3(256), 0, st, (float*)A,
                       (const float*)sums, total, (float)eps, pe);
  else
    return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
  ASSX_LAUNCH_CHECK(ctx, "nmf_apply_sums_kernel");
  return 0;
}

int assx_nmf_update(assx_ctx* ctx, int kind, double domain, double eps, const void* X, void* Tb, void* V, void* ws,
                    int B, int F, int T, int K, int dtype, void* stream) {
  ASSX_REQUIRE(ctx, ctx == nullptr || (kind >= ASSX_NMF_EUC && kind <= ASSX_NMF_IS_ME), ASSX_E_ARG, "bad NMF kind %d", kind);
  return assx_nmf_update_ex(ctx, kind, domain, 0.0, eps, X, Tb, V, ws, B, F, T, K, dtype, stream);
}

int assx_nmf_loss(assx_ctx* ctx, int kind, double domain, double eps, const void* X, const void* Tb, const void* V,
                  double* loss, void* ws, int B, int F, int T, int K, int dtype, void* stream) {
  ASSX_REQUIRE(ctx, ctx == nullptr || (kind >= ASSX_NMF_EUC && kind <= ASSX_NMF_IS_ME), ASSX_E_ARG, "bad NMF kind %d", kind);
  return assx_nmf_loss_ex(ctx, kind, domain, 0.0, eps, X, Tb, V, loss, ws, B, F, T, K, dtype, stream);
}

int assx_nmf_loss_ex(assx_ctx* ctx, int kind, double domain, double param, double eps, const void* X, const void* Tb,
                     const void* V, double* loss, void* ws, int B, int F, int T, int K, int dtype, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, B >= 1 && F >= 1 && T >= 1 && K >= 1, ASSX_E_ARG, "invalid sizes B=%d F=%d T=%d K=%d", B, F, T, K);
  ASSX_REQUIRE(ctx, X && Tb && V && loss && ws, ASSX_E_NULL, "assx_nmf_loss: NULL array");
  ASSX_REQUIRE(ctx, kind >= ASSX_NMF_EUC && kind <= ASSX_NMF_CAUCHY_MM_FAST, ASSX_E_ARG, "bad NMF kind %d", kind);
  ASSX_REQUIRE(ctx, kind != ASSX_NMF_T || param > 0.0, ASSX_E_ARG, "tNMF: nu must be > 0, got %g", param);
  hipStream_t st = (hipStream_t)stream;
  const NmfWs L = nmf_ws(B, F, T, K, dtype);
  double* lpart = (double*)((char*)ws + L.lpart);
  const PowSpec p2d = make_pow(2.0 / domain);
  const int k2 = (kind == ASSX_NMF_IS_ME) ? ASSX_NMF_IS_MM : kind;
  static const int no_mfma = lab_int("ASSX_NMF_NO_MFMA", 0);
  if (K <= NMF_MFMA_MAX_K && !no_mfma) {
    int FS, fchunk;
    mfma_loss_split(nmf_group(ctx), F, T, &FS, &fchunk);
    const dim3 g2((T + 15) / 16, FS, B);
    const bool d2 = domain == 2.0 && k2 <= ASSX_NMF_IS_MM;  // EUC / KL / IS at domain 2: the pow()- and (IS) log-free forms
#define NMF_LOSS_LAUNCH_D(RT, KTV, D2KV)                                                                         \
  hipLaunchKernelGGL((nmf_loss_mfma_kernel<RT, KTV, D2KV>), g2, dim3(256), 0, st, (const RT*)X, (const RT*)Tb,    \
                     (const RT*)V, lpart, F, T, K, fchunk, k2, eps, p2d, param)
#define NMF_LOSS_LAUNCH(RT, KTV)                                         \
  do {                                                                   \
    if (!d2) NMF_LOSS_LAUNCH_D(RT, KTV, -1);                             \
    else if (k2 == ASSX_NMF_EUC) NMF_LOSS_LAUNCH_D(RT, KTV, ASSX_NMF_EUC); \
    else if (k2 == ASSX_NMF_KL) NMF_LOSS_LAUNCH_D(RT, KTV, ASSX_NMF_KL);   \
    else NMF_LOSS_LAUNCH_D(RT, KTV, ASSX_NMF_IS_MM);                     \
  } while (0)
#define NMF_LOSS_BY_K(RT)                     \
  switch ((K + 15) / 16) {                    \
    case 1: NMF_LOSS_LAUNCH(RT, 1); break;    \
    case 2: NMF_LOSS_LAUNCH(RT, 2); break;    \
    case 3: NMF_LOSS_LAUNCH(RT, 3); break;    \
    default: NMF_LOSS_LAUNCH(RT, 4);          \
  }
    if (dtype == ASSX_F64) {
      NMF_LOSS_BY_K(double)
    } else if (dtype == ASSX_F32) {
      NMF_LOSS_BY_K(float)
    } else {
      return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
    }
#undef NMF_LOSS_BY_K
#undef NMF_LOSS_LAUNCH
#undef NMF_LOSS_LAUNCH_D
    ASSX_LAUNCH_CHECK(ctx, "nmf_loss_mfma_kernel");
    hipLaunchKernelGGL(sum_reduce_f64_kernel, dim3(B), dim3(256), 0, st, (const double*)lpart, loss,
                       (size_t)g2.x * g2.y);
    ASSX_LAUNCH_CHECK(ctx, "sum_reduce_f64_kernel");
    return 0;
  }
  dim3 grid(nblocks(T, 256), F, B);
  if (dtype == ASSX_F64)
    hipLaunchKernelGGL((nmf_loss_kernel<double>), grid, dim3(256), 0, st, (const double*)X, (const double*)Tb,
                       (const double*)V, lpart, F, T, K, k2, eps, p2d, param);
  else if (dtype == ASSX_F32)
    hipLaunchKernelGGL((nmf_loss_kernel<float>), grid, dim3(256), 0, st, (const float*)X, (const float*)Tb,
                       (const float*)V, lpart, F, T, K, k2, eps, p2d, param);
  else
    return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
  ASSX_LAUNCH_CHECK(ctx, "nmf_loss_kernel");
  hipLaunchKernelGGL(sum_reduce_f64_kernel, dim3(B), dim3(256), 0, st, (const double*)lpart, loss,
                     (size_t)F * grid.x);
  ASSX_LAUNCH_CHECK(ctx, "sum_reduce_f64_kernel");
  return 0;
}

}  // extern "C"

#if NMF_TRACE
// timing-experiment builds only (tools/probes/nmf_trace.py); not part of include/assx.h
extern "C" int assx_debug_nmf_trace(unsigned long long* host, int clear) {
  using assx::g_nmf_trace;
  constexpr size_t N = 8192 + 4 * 512;
  if (clear) {
    static unsigned long long zeros[N];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_nmf_trace), zeros, sizeof(zeros));
  }
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_nmf_trace), sizeof(unsigned long long) * N);
}
#endif
