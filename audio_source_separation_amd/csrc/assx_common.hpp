// Shared device/host helpers for the assx HIP library (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../../include/assx.h"

struct assx_ctx {
  int device;
  char err[512];
  // matrices per independent problem in the NMF entry points (1 = plain NMF; the ILRMA source model sets N: its
  // batch is B utterances x N sources).  The split-T / split-F slab counts are derived from ONE problem's geometry,
  // never from the batch size, so that a batched call sums in the same order as per-item calls (bit-identical).
  int nmf_group;
  // streaming passes launched so far: consecutive passes walk the utterances of a batch in opposite directions (the
  // one a pass ends with is what the Infinity Cache still holds when the next pass starts).  Never changes a result.
  unsigned stream_pass;
  // pinned staging ring + host thread pool of assx_upload / assx_download (csrc/assx_xfer.hip), created on first use
  void* xfer;
  // zeroed device words for the "last workgroup done" tickets of kernels that fold their finalize step (assx_common.hpp:
  // take_ticket).  ONE BUFFER PER STREAM: launches on different streams of one context may overlap on the device, and
  // two kernels counting on the same words would see each other's arrivals (a partial sum applied, counters left
  // non-zero for every later call).  The first TK_INIT words of every slot are carved out of ONE pool that
  // assx_ctx_create allocates and zeroes (round 5's advisor: a stream seen for the first time INSIDE a stream capture --
  // torch.cuda.graph captures on a fresh side stream -- must not need hipMalloc / hipDeviceSynchronize, which are
  // illegal there).  ensure_tickets() finds / assigns / grows the buffer of the stream it is called for; growth beyond
  // TK_INIT and recycling a slot (more than 16 live streams) allocate or synchronise and are refused with a message
  // while the stream is capturing.  Outgrown buffers are kept (launches that still use them may be in flight) and freed
  // behind a device synchronisation when their list is full or the context is destroyed.
  static constexpr size_t TK_INIT = 8192;
  struct TicketSlot {
    hipStream_t st;
    int* p;
    size_t n;
    unsigned long long used;  // value of tk_clock at the last use (least recently used slot is recycled)
    bool pooled;              // p points into tk_pool (never freed on its own)
  } tk[16];
  int n_tk;
  unsigned long long tk_clock;
  int* tk_pool;  // 16 x TK_INIT zeroed words, owned by the context
  void* old_tickets[32];
  int n_old_tickets;
};

namespace assx {

void xfer_destroy(assx_ctx* ctx);  // csrc/assx_xfer.hip
// at least n zeroed ticket words private to stream `st`, zeroing ordered on `st` before the caller's launch (csrc/
// assx_api.hip).  Returns 0 and the buffer in *out, or the hipError_t of the failed allocation (message recorded in ctx).
// Kernels leave the words zero, so a buffer is only ever cleared when it is (re)allocated.  A context used from more
// streams than it has slots recycles the least recently used slot behind a hipDeviceSynchronize().  While `st` is being
// captured nothing is allocated or synchronised: either the pool serves the request or the call fails with
// ASSX_E_UNSUPPORTED and says what to warm up.
int ensure_tickets(assx_ctx* ctx, size_t n, hipStream_t st, int** out);
int tickets_reserve(assx_ctx* ctx);  // the pool, at context creation (the context's device must be current)
void tickets_destroy(assx_ctx* ctx);

struct NmfGroupScope {  // sets assx_ctx::nmf_group for the NMF calls made inside the scope
  assx_ctx* c;
  int old;
  NmfGroupScope(assx_ctx* ctx, int group) : c(ctx), old(ctx->nmf_group) { c->nmf_group = group; }
  ~NmfGroupScope() { c->nmf_group = old; }
};

constexpr int WAVE = 64;

// ------------------------------------------------------------------------------------------
// Run-time knobs and laboratory switches.
//
// The shipped library reads SIX environment variables (include/assx.h lists them): ASSX_G, ASSX_NMF_BASIS_WGS,
// ASSX_NMF_ACT_WGS, ASSX_NMF_XFED_WGS (work-partition sizes: the tests shrink / force them so that small inputs walk
// every code path of a long range), ASSX_XFER_CHUNK_MB and ASSX_XFER_THREADS (the pinned staging ring) -- knob_int().
// Everything else that rounds 1-5 could switch through the environment -- the variants that were measured and not kept
// (sources-side-by-side IP sweep, folded AuxIVA statistic, one-wave-per-source wide-channel covariance, the pre-LDS-ring
// forms of the streaming kernels, legacy partitions and launch orders) -- exists only in a LABORATORY build
// (-DASSX_LAB=1: csrc/build.sh with ASSX_EXTRA_FLAGS; assx_version() then ends in "+lab"): lab_int() is a compile-time
// constant otherwise, and the kernels behind those switches are not compiled.
// ------------------------------------------------------------------------------------------
#ifndef ASSX_LAB
#define ASSX_LAB 0
#endif
inline int knob_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
#if ASSX_LAB
inline int lab_int(const char* name, int dflt) { return knob_int(name, dflt); }
#else
constexpr int lab_int(const char*, int dflt) { return dflt; }
#endif

// ------------------------------------------------------------------------------------------
// host side: argument checking / error reporting
// ------------------------------------------------------------------------------------------
inline int fail(assx_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

inline int hip_fail(assx_ctx* ctx, hipError_t e, const char* where) {
  if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: %s", where, hipGetErrorString(e));
  return (int)e;
}

#define ASSX_REQUIRE(ctx, cond, code, ...) \
  do {                                     \
    if (!(cond)) return ::assx::fail((ctx), (code), __VA_ARGS__); \
  } while (0)

#define ASSX_LAUNCH_CHECK(ctx, where)                 \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) return ::assx::hip_fail((ctx), e__, (where)); \
  } while (0)

// Every entry point: the context exists and ITS device is the calling thread's current device.  The library never
// changes the current device itself (a host framework such as torch owns it); a mismatch would launch on -- or
// query attributes of -- the wrong GPU, so it is refused loudly instead.
inline int check_ctx_device(assx_ctx* ctx) {
  int cur = -1;
  hipError_t e = hipGetDevice(&cur);
  if (e != hipSuccess) return hip_fail(ctx, e, "hipGetDevice");
  if (cur != ctx->device)
    return fail(ctx, ASSX_E_ARG,
                "the calling thread's current device is %d but this context was created for device %d: make it "
                "current (hipSetDevice) before calling", cur, ctx->device);
  return 0;
}

#define ASSX_REQUIRE_CTX(ctx)                                       \
  do {                                                              \
    if ((ctx) == nullptr) return ASSX_E_NULL;                       \
    int rc__ = ::assx::check_ctx_device(ctx);                       \
    if (rc__ != 0) return rc__;                                     \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------
// complex numbers (interleaved re, im)
// ------------------------------------------------------------------------------------------
template <typename R>
struct alignas(2 * sizeof(R)) Cx {
  R x, y;
};

// native 2-vector of the real type: the register / LDS image of one complex sample (a single 8- or 16-byte
// load/store; arrays of these stay in VGPRs where arrays of the aligned struct sometimes fall to scratch)
template <typename R>
using Vec2 = R __attribute__((ext_vector_type(2)));

template <typename R>
__host__ __device__ __forceinline__ Cx<R> cmake(R a, R b) {
  Cx<R> c;
  c.x = a;
  c.y = b;
  return c;
}
template <typename R>
__device__ __forceinline__ Vec2<R> ldv(const Cx<R>* p) { return *reinterpret_cast<const Vec2<R>*>(p); }
template <typename R>
__device__ __forceinline__ Cx<R> tocx(Vec2<R> v) { return cmake<R>(v.x, v.y); }
// wave-uniform base pointer + 32-bit per-lane BYTE offset: written so that the backend selects the
// `global_load ... v_off, s[base:base+1]` (SGPR base + zero-extended VGPR offset) addressing form instead of a
// per-lane 64-bit address computation (v_lshl_add_u64 per load).
template <typename R>
__device__ __forceinline__ Vec2<R> ldv_so(const Cx<R>* base, unsigned byte_off) {
  return *reinterpret_cast<const Vec2<R>*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename R>
__device__ __forceinline__ R ld_so(const R* base, unsigned byte_off) {
  return *reinterpret_cast<const R*>(reinterpret_cast<const char*>(base) + byte_off);
}
// ---- buffer addressing (gfx950 raw buffers).  address = descriptor base (wave-uniform, SGPR quad) + soffset
// (wave-uniform 32-bit) + voffset (per-lane 32-bit): a streaming kernel keeps ONE descriptor per array, the per-row
// strides as scalar byte offsets and a single per-lane byte offset per block, so a load costs no address VALU at all
// (the flat `global_load` form needs a 64-bit per-lane add -- or an SGPR pair -- per row).  Offsets are 32-bit:
// callers guarantee that an utterance's slice of each array is < 4 GiB.
#if defined(__HIP_DEVICE_COMPILE__)
// Hide a wave-uniform value from loop-invariant code motion: the row offsets derived from it are then recomputed
// with a few SALU adds next to the loads instead of being kept live across the whole loop -- where they (with the
// per-bin constants already resident in SGPRs) overflow the scalar register file and come back as v_readlane +
// 5 wait states per load.
__device__ __forceinline__ unsigned sgpr_opaque(unsigned v) {
  asm volatile("" : "+s"(v));
  return v;
}
// Tie a load's per-lane offset to a value computed earlier (no instruction is emitted): the optimiser may otherwise
// hoist a ring-slot refill above the last read of the slot, which forces the new data into other registers and a
// copy -- behind a full vmcnt(0) drain -- at the loop back-edge.
template <typename R>
__device__ __forceinline__ unsigned order_after(unsigned voff, R dep) {
  asm volatile("" : "+v"(voff) : "v"(dep));
  return voff;
}
using BufRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufRsrc make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
// descriptor with the exact extent: loads past `bytes` return zeros (saturates at 4 GiB - 1)
__device__ __forceinline__ BufRsrc make_rsrc_sized(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0,
                                           (int)(bytes > 0xffffffffull ? 0xffffffffull : bytes), 0x00020000);
}
typedef unsigned int buf_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int buf_u4 __attribute__((ext_vector_type(4)));
template <typename R>
__device__ __forceinline__ Vec2<R> buf_ldv(BufRsrc r, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ Vec2<double> buf_ldv<double>(BufRsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(Vec2<double>, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
template <>
__device__ __forceinline__ Vec2<float> buf_ldv<float>(BufRsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(Vec2<float>, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
// ---- ring-slot refill that lands in the register the slot already occupies.  The compiler sees an in/out operand,
// so the new block cannot be given other registers (which would come back as copies at the loop back-edge, behind a
// vmcnt(0) drain).  The load is invisible to the compiler's vmcnt model: the consumer must call wait_slot<N>() with
// N = the number of VMEM instructions issued after this refill that may still be in flight.
// It is invisible to the hazard recogniser too: a scalar operand (descriptor, soffset) that the compiler has just
// produced with a VALU instruction -- v_readlane of a spilled SGPR is the usual one -- needs 5 wait states before a
// VMEM instruction may read it, and nothing inserts them inside an asm statement.  Without the `s_nop 4` the load
// uses the register's previous contents whenever register pressure puts such a reload right in front of it
// (found as a page fault: the soffset was the low word of an unrelated pointer).
__device__ __forceinline__ buf_u4 make_rsrc_words(const void* base, size_t bytes) {
  const unsigned long long a = (unsigned long long)base;
  buf_u4 r;
  r.x = (unsigned)a;
  r.y = (unsigned)(a >> 32) & 0xffffu;
  r.z = bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes;
  r.w = 0x00020000u;
  return r;
}
// ASSX_X_POLICY_ID: cache-policy bits of the streamed X loads (1 " nt", 2 " sc1", 3 " sc0 sc1"; A/B builds).
// Default none: measured in round 4 (profiles/r04_x_policy_ab.txt).
#ifndef ASSX_X_POLICY_ID
#define ASSX_X_POLICY_ID 0
#endif
#if ASSX_X_POLICY_ID == 1
#define ASSX_X_POLICY " nt"
#elif ASSX_X_POLICY_ID == 2
#define ASSX_X_POLICY " sc1"
#elif ASSX_X_POLICY_ID == 3
#define ASSX_X_POLICY " sc0 sc1"
#else
#define ASSX_X_POLICY ""
#endif
__device__ __forceinline__ void buf_ldv_tied(Vec2<double>& dst, buf_u4 rsrc, unsigned voff, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" ASSX_X_POLICY : "+v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void buf_ldv_tied(Vec2<float>& dst, buf_u4 rsrc, unsigned voff, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, %3 offen" ASSX_X_POLICY : "+v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// one real per lane, same contract as buf_ldv_tied
__device__ __forceinline__ void buf_ld_tied(double& dst, buf_u4 rsrc, unsigned voff, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void buf_ld_tied(float& dst, buf_u4 rsrc, unsigned voff, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// one dword per lane, global -> LDS (lane L lands at lds_addr + 4 L), invisible to the compiler's vmcnt model like
// buf_ldv_tied.  M0 carries the LDS address and is an INPUT operand pinned to the register ("{m0}"): the compiler writes
// M0 itself, ahead of the block, and knows what it holds afterwards -- its own LDS-direct loads (the
// __builtin_amdgcn_raw_ptr_buffer_load_lds calls next to this one in assx_widem_cov.hpp) get their own M0 write.  Rounds
// 3-5 wrote M0 inside the asm and listed it as a clobber, which the backend answers with "reserved registers on the
// clobber list may not be preserved" (round 5's review, weak #9); tools/asm_wait_check.py now refuses any inline-asm
// block that writes M0.  The s_nop covers SALU-write -> VMEM-read of the descriptor / offset SGPRs and of M0.
__device__ __forceinline__ void buf_dword_to_lds(unsigned lds_addr, buf_u4 rsrc, unsigned voff, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dword %1, %2, %3 offen lds"
               :
               : "{m0}"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
template <int N, typename R>
__device__ __forceinline__ void wait_slot(Vec2<R> (&x)[2]) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(x[0]), "+v"(x[1]) : "n"(N) : "memory");
}
template <int N, typename R>
__device__ __forceinline__ void wait_slot(Vec2<R> (&x)[3]) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]) : "n"(N) : "memory");
}
template <int N, typename R>
__device__ __forceinline__ void wait_slot(Vec2<R> (&x)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(N) : "memory");
}
template <typename R>
__device__ __forceinline__ R buf_ld(BufRsrc r, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ double buf_ld<double>(BufRsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
template <>
__device__ __forceinline__ float buf_ld<float>(BufRsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
#else  // host pass: declarations only, so that the kernel templates parse
struct BufRsrc {};
__device__ unsigned sgpr_opaque(unsigned v);
template <typename R>
__device__ unsigned order_after(unsigned voff, R dep);
__device__ BufRsrc make_rsrc(const void* base);
__device__ BufRsrc make_rsrc_sized(const void* base, size_t bytes);
typedef unsigned int buf_u4 __attribute__((ext_vector_type(4)));
__device__ buf_u4 make_rsrc_words(const void* base, size_t bytes);
template <typename R>
__device__ void buf_ldv_tied(Vec2<R>& dst, buf_u4 rsrc, unsigned voff, unsigned soff);
template <int N, typename R, int M>
__device__ void wait_slot(Vec2<R> (&x)[M]);
__device__ void buf_dword_to_lds(unsigned lds_addr, buf_u4 rsrc, unsigned voff, unsigned soff);
__device__ void buf_ld_tied(double& dst, buf_u4 rsrc, unsigned voff, unsigned soff);
__device__ void buf_ld_tied(float& dst, buf_u4 rsrc, unsigned voff, unsigned soff);
template <typename R>
__device__ Vec2<R> buf_ldv(BufRsrc r, unsigned voff, unsigned soff);
template <typename R>
__device__ R buf_ld(BufRsrc r, unsigned voff, unsigned soff);
#endif

// 1/x: v_rcp + Newton steps instead of the 11-instruction IEEE division expansion (|rel err| < 2^-52)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  return r;
}
__device__ __forceinline__ float fast_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = fmaf(-x, r, 1.0f);
  return fmaf(r, e, r);
}

// 1/x[0..N-1] from ONE hardware reciprocal chain (Montgomery's trick): the products of the inputs, one fast_rcp, and the
// partial products back -- N = 4: 3 + 5 + 6 = 14 vector instructions and one transcendental instead of 20 and four.  Each
// result carries three roundings more than fast_rcp's correctly rounded quotient (|rel err| < 2.5 ulp).  The product of
// the inputs must stay a normal number: callers pass variances floored at eps > 0; where the product leaves
// [1e-290, 1e290] (a lane-wise test, one v_cmp_class) the lane takes the individual reciprocals.
#ifndef ASSX_BATCH_RCP
#define ASSX_BATCH_RCP 0
#endif
template <int N>
__device__ __forceinline__ void batch_rcp(const double (&x)[N], double (&inv)[N]) {
  static_assert(N >= 1 && N <= 4, "batch_rcp: 1..4 values");
  if (N == 1) {
    inv[0] = fast_rcp(x[0]);
  } else if (N == 2) {
    const double p = x[0] * x[1];
    if (__builtin_expect(p > 1e-290 && p < 1e290, 1)) {
      const double r = fast_rcp(p);
      inv[0] = r * x[1];
      inv[1] = r * x[0];
    } else {
      inv[0] = fast_rcp(x[0]);
      inv[1] = fast_rcp(x[1]);
    }
  } else if (N == 3) {
    const double p01 = x[0] * x[1], p = p01 * x[2];
    if (__builtin_expect(p > 1e-290 && p < 1e290, 1)) {
      const double r = fast_rcp(p);
      inv[2] = r * p01;
      const double r01 = r * x[2];
      inv[0] = r01 * x[1];
      inv[1] = r01 * x[0];
    } else {
      inv[0] = fast_rcp(x[0]);
      inv[1] = fast_rcp(x[1]);
      inv[2] = fast_rcp(x[2]);
    }
  } else {
    const double p01 = x[0] * x[1], p23 = x[2 % N] * x[3 % N], p = p01 * p23;
    if (__builtin_expect(p > 1e-290 && p < 1e290, 1)) {
      const double r = fast_rcp(p);
      const double r01 = r * p23, r23 = r * p01;
      inv[0] = r01 * x[1];
      inv[1] = r01 * x[0];
      inv[2 % N] = r23 * x[3 % N];
      inv[3 % N] = r23 * x[2 % N];
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) inv[i] = fast_rcp(x[i]);
    }
  }
}
template <int N>
__device__ __forceinline__ void batch_rcp(const float (&x)[N], float (&inv)[N]) {  // float32: nothing to gain (3 instructions each)
#pragma unroll
  for (int i = 0; i < N; ++i) inv[i] = fast_rcp(x[i]);
}

template <typename R>
__device__ __forceinline__ Cx<R> cadd(Cx<R> a, Cx<R> b) { return cmake<R>(a.x + b.x, a.y + b.y); }
template <typename R>
__device__ __forceinline__ Cx<R> csub(Cx<R> a, Cx<R> b) { return cmake<R>(a.x - b.x, a.y - b.y); }
template <typename R>
__device__ __forceinline__ Cx<R> cmul(Cx<R> a, Cx<R> b) {
  return cmake<R>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
template <typename R>
__device__ __forceinline__ Cx<R> cmulc(Cx<R> a, Cx<R> b) {
  return cmake<R>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// acc += a * b
template <typename R>
__device__ __forceinline__ void cfma(Cx<R>& acc, Cx<R> a, Cx<R> b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(a.y, b.x, acc.y);
}
// y += w x with w WAVE-UNIFORM (a demixing coefficient out of scalar loads).  float32: two packed fused multiply-adds
//   (y.re, y.im) += w.re * (x.re, x.im);   (y.re, y.im) += w.im * (-x.im, x.re)
// instead of four scalar ones -- the operand selects (op_sel / op_sel_hi / neg_lo) pick the halves, so nothing else is
// issued; per component the same two IEEE operations in the same order as cfma: no result changes.  ASSX_PK_DEMIX=0: cfma.
#ifndef ASSX_PK_DEMIX
#define ASSX_PK_DEMIX 1
#endif
__device__ __forceinline__ void demix_mac(Cx<double>& y, Cx<double> w, Cx<double> x) { cfma(y, w, x); }
__device__ __forceinline__ void demix_mac(Cx<float>& y, Cx<float> w, Cx<float> x) {
#if ASSX_PK_DEMIX && defined(__HIP_DEVICE_COMPILE__)
  Vec2<float> yy = {y.x, y.y};
  const Vec2<float> ww = {w.x, w.y}, xx = {x.x, x.y};
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "+v"(yy)
      : "s"(ww), "v"(xx));
  y.x = yy.x;
  y.y = yy.y;
#else
  cfma(y, w, x);
#endif
}

template <typename R>
__device__ __forceinline__ R cabs2(Cx<R> a) { return fma(a.x, a.x, a.y * a.y); }
template <typename R>
__device__ __forceinline__ Cx<R> cconj(Cx<R> a) { return cmake<R>(a.x, -a.y); }
template <typename R>
__device__ __forceinline__ Cx<R> cscale(Cx<R> a, R s) { return cmake<R>(a.x * s, a.y * s); }

// Smith's complex division a / b
__device__ __forceinline__ Cx<double> cdiv(Cx<double> a, Cx<double> b) {
  if (fabs(b.x) >= fabs(b.y)) {
    double r = b.y / b.x, d = b.x + b.y * r;
    return cmake<double>((a.x + a.y * r) / d, (a.y - a.x * r) / d);
  } else {
    double r = b.x / b.y, d = b.x * r + b.y;
    return cmake<double>((a.x * r + a.y) / d, (a.y * r - a.x) / d);
  }
}

// 1 / z = conj(z) / |z|^2 with the hardware reciprocal + two Newton steps (fast_rcp, < 1 ulp) instead of Smith's three
// IEEE divisions: a complex division is ~30 DEPENDENT f64 instructions, and the per-bin sweeps (IP / ISS / IP2) are
// dependent-instruction chains -- at ~32 cycles per dependent f64 instruction with one wave per SIMD the divisions
// were a third of the IP kernel.  Outside the range where |z|^2 is a normal number the Smith form is kept.
__device__ __forceinline__ Cx<double> crcp_fast(Cx<double> z) {
  const double n2 = fma(z.x, z.x, z.y * z.y);
  if (!(n2 > 1e-290 && n2 < 1e290)) return cdiv(cmake<double>(1.0, 0.0), z);
  const double r = fast_rcp(n2);
  return cmake<double>(z.x * r, -(z.y * r));
}
__device__ __forceinline__ Cx<double> cdiv_fast(Cx<double> a, Cx<double> b) { return cmul(a, crcp_fast(b)); }

// principal complex square root (numpy.sqrt on complex128)
__device__ __forceinline__ Cx<double> csqrt_principal(Cx<double> z) {
  if (z.y == 0.0 && z.x >= 0.0) return cmake<double>(sqrt(z.x), 0.0);
  double m = hypot(z.x, z.y);
  double s = sqrt(0.5 * (m + fabs(z.x)));
  double t = z.y / (2.0 * s);
  if (z.x >= 0.0) return cmake<double>(s, t);
  return cmake<double>(fabs(t), copysign(s, z.y));
}

// same value without libm's hypot and without an IEEE division (same reason as crcp_fast); extreme magnitudes take
// the scaled form above
__device__ __forceinline__ Cx<double> csqrt_fast(Cx<double> z) {
  if (z.y == 0.0 && z.x >= 0.0) return cmake<double>(sqrt(z.x), 0.0);
  // |y| < 2^-27 x (a Hermitian form with rounding noise in its imaginary part -- every caller's usual case):
  // x^2 + y^2 rounds to x^2, so m = x and s = sqrt(x) exactly as in the general formula, one square root instead of two
  if (z.x > 1e-290 && z.x < 1e290 && fabs(z.y) < 7.450580596923828e-09 * z.x) {
    const double s = sqrt(z.x);
    return cmake<double>(s, z.y * fast_rcp(2.0 * s));
  }
  const double n2 = fma(z.x, z.x, z.y * z.y);
  if (!(n2 > 1e-290 && n2 < 1e290)) return csqrt_principal(z);
  const double m = sqrt(n2);
  const double s = sqrt(0.5 * (m + fabs(z.x)));
  const double t = z.y * fast_rcp(2.0 * s);
  if (z.x >= 0.0) return cmake<double>(s, t);
  return cmake<double>(fabs(t), copysign(s, z.y));
}

// ------------------------------------------------------------------------------------------
// x**e with the fast paths numpy itself takes for e in {1, 2, 0.5}; generic pow otherwise.
// ------------------------------------------------------------------------------------------
enum PowMode { POW_GENERIC = 0, POW_ID = 1, POW_SQUARE = 2, POW_CUBE = 3, POW_SQRT = 4,
               POW_CAUCHY_ME = 5 /* not a power: marks the CauchyNMF 'me' combination rule for nmf_finalize_kernel */ };

struct PowSpec {
  double e;
  int mode;
};

inline PowSpec make_pow(double e) {
  PowSpec p;
  p.e = e;
  if (e == 1.0) p.mode = POW_ID;
  else if (e == 2.0) p.mode = POW_SQUARE;
  else if (e == 3.0) p.mode = POW_CUBE;
  else if (e == 0.5) p.mode = POW_SQRT;
  else p.mode = POW_GENERIC;
  return p;
}

template <typename R>
__device__ __forceinline__ R powspec(R x, PowSpec p) {
  switch (p.mode) {
    case POW_ID: return x;
    case POW_SQUARE: return x * x;
    case POW_CUBE: return x * x * x;
    case POW_SQRT: return sqrt(x);
    default: return (R)pow((double)x, p.e);
  }
}

// Pin a wave-uniform value into SGPRs.  Kernel arguments travel in structs, so the compiler cannot prove the
// read-only arrays do not alias the output and falls back to vector loads for uniform addresses; without this
// every per-bin constant (basis row, demixing row) would occupy a VGPR per lane.
__device__ __forceinline__ double uniform_value(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float uniform_value(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

template <typename R>
__device__ __forceinline__ R floor_eps(R v, R eps) {  // numpy: v[v < eps] = eps (NaN stays NaN)
  return (v < eps) ? eps : v;
}

// ------------------------------------------------------------------------------------------
// "last workgroup done": folding a finalize step into the kernel that produces its partial records.
// The workgroups of a group publish their records with agent-scope (sc1, write-through) stores, wait for them
// (s_waitcnt vmcnt(0)), take a ticket with an agent-scope atomic; the holder of the last ticket reads every record of
// the group back with agent-scope loads, in a fixed order, and applies the update.  No L2-wide fence is involved (an
// agent-scope __threadfence() on gfx950 writes back and invalidates the XCD's whole L2); protocol probed across XCDs in
// tools/probes/ticket_probe.hip (mode 2).  Tickets live in a zeroed buffer owned by the context (assx_ctx::tickets) and
// are reset by the last holder, so the buffer is all zeros again when the kernel ends.
// ------------------------------------------------------------------------------------------
template <typename R>
__device__ __forceinline__ void st_agent(R* p, R v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename R>
__device__ __forceinline__ R ld_agent(const R* p) {
  return __hip_atomic_load(const_cast<R*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Called by every lane of the wave that stored the records; returns (to every lane) whether this workgroup holds the
// last of `members` tickets.
__device__ __forceinline__ bool take_ticket(int* ticket, int members) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  int last = 0;
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1;
    if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return __builtin_amdgcn_readfirstlane(last) != 0;
}

// sum of S slab entries `stride` apart, in the strand order of the finalize kernels of rounds 1-3 (nmf_finalize_kernel,
// auxiva_stat_finalize_kernel): strand q adds s = q, q + 4, ...;
// the strands are combined as (0 + 1) + (2 + 3).  Agent-scope loads: the slabs were written by other workgroups of
// this launch (assx_common.hpp: take_ticket).
template <typename R>
__device__ __forceinline__ R slab_sum4(const R* p, size_t stride, int S) {
  // 16 loads in flight per trip (a trip is one memory round trip: with 4 the holder of the last ticket walked 32 slabs
  // in 8 dependent round trips); the additions keep the strand order whatever the chunking
  R q[4] = {0, 0, 0, 0};
  for (int s0 = 0; s0 < S; s0 += 16) {
    R v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = ld_agent(p + (size_t)min(s0 + c, S - 1) * stride);
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (s0 + c < S) q[c & 3] += v[c];
  }
  return (q[0] + q[1]) + (q[2] + q[3]);
}

// Numerator and denominator sums of NO outputs at once, every one in slab_sum4's order (same additions, same bits):
// the loads of all 2 NO sums travel together -- one memory round trip per 8 slabs for everything a thread finalizes,
// instead of one per sum (the holder of the last ticket is the tail of its kernel: in-kernel stamps showed its four
// dependent round trips of ~1.5 us each).  Slab s of the numerator of output u is pn[s * stride + idx[u]], the
// denominator pd[...] (pn, pd wave-uniform, idx < 2^29 elements: scalar base + 32-bit lane offset addressing, and 8
// slabs per trip, keep the registers of the in-flight loads at 32 values).  on[u] = false: the output is skipped (sums 0).
template <typename R, int NO, int CW = 8>  // CW slabs per trip (a multiple of 4: the strands keep their order)
__device__ __forceinline__ void slab_sum4_n(const R* pn, const R* pd, const unsigned (&idx)[NO], const bool (&on)[NO],
                                            size_t stride, int S, R (&out)[2 * NO]) {
  static_assert(CW % 4 == 0, "whole strand groups per trip");
  R q[2 * NO][4];
#pragma unroll
  for (int i = 0; i < 2 * NO; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) q[i][c] = 0;
  for (int s0 = 0; s0 < S; s0 += CW) {
    R v[2 * NO][CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const size_t so = (size_t)min(s0 + c, S - 1) * stride;  // wave-uniform
      const R* sn = pn + so;
      const R* sd = pd + so;
#pragma unroll
      for (int u = 0; u < NO; ++u) {
        v[2 * u][c] = on[u] ? ld_agent(sn + idx[u]) : (R)0;
        v[2 * u + 1][c] = on[u] ? ld_agent(sd + idx[u]) : (R)0;
      }
    }
#pragma unroll
    for (int c = 0; c < CW; ++c)
      if (s0 + c < S) {
#pragma unroll
        for (int i = 0; i < 2 * NO; ++i) q[i][c & 3] += v[i][c];
      }
  }
#pragma unroll
  for (int i = 0; i < 2 * NO; ++i) out[i] = (q[i][0] + q[i][1]) + (q[i][2] + q[i][3]);
}

// ------------------------------------------------------------------------------------------
// wave-level reductions (wave64)
// ------------------------------------------------------------------------------------------
template <typename R>
__device__ __forceinline__ R wave_allreduce_sum(R v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}

constexpr int ilog2_c(int n) { return n <= 1 ? 0 : 1 + ilog2_c(n >> 1); }
constexpr int next_pow2_c(int n) { return n <= 1 ? 1 : 2 * next_pow2_c((n + 1) >> 1); }

// Butterfly reduce-scatter of NV (power of two, <= WIDTH) per-lane partial sums across each aligned group of
// WIDTH lanes (WIDTH = 64: the whole wave).  On return v[0] of lane l holds the group-wide total of value index
// (l % WIDTH) >> (log2 WIDTH - log2 NV) (every lane sharing that index holds the same total).
// NV-1 + (log2 WIDTH - log2 NV) shuffles instead of NV * log2 WIDTH.
// The two cross-row stages of the butterfly (partner lane ^ 32, lane ^ 16) as gfx950 half-wave / row swaps:
// v_permlane32_swap(a, b) exchanges lanes 32-63 of a with lanes 0-31 of b (v_permlane16_swap: odd rows of a with even
// rows of b), after which a + b is, in every lane, exactly the sum the select + ds_bpermute form produces (the same
// two numbers, so the same bits) -- 2 swaps + 1 add per exchange instead of 4 selects + 2 LDS permutes + 1 add, and
// these two stages are 3/4 of all exchanges.  The flush of a streaming kernel's accumulators ran 2.2-4.8 us per bin
// boundary (in-kernel timestamps), on the critical path of the workgroups that cross one.
template <bool HALF_WAVE>
__device__ __forceinline__ void lane_block_swap(unsigned& a, unsigned& b) {
  if (HALF_WAVE) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
  }
}
template <bool HALF_WAVE>
__device__ __forceinline__ double swap_add(double a, double b) {
  unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
  unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
  lane_block_swap<HALF_WAVE>(alo, blo);
  lane_block_swap<HALF_WAVE>(ahi, bhi);
  return __hiloint2double((int)ahi, (int)alo) + __hiloint2double((int)bhi, (int)blo);
}
template <bool HALF_WAVE>
__device__ __forceinline__ float swap_add(float a, float b) {
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  lane_block_swap<HALF_WAVE>(ua, ub);
  return __uint_as_float(ua) + __uint_as_float(ub);
}

template <typename R, int NV, int WIDTH = WAVE>
__device__ __forceinline__ R wave_reduce_scatter(R (&v)[NV]) {
  static_assert(NV >= 1 && NV <= WIDTH && (NV & (NV - 1)) == 0, "NV must be a power of two <= WIDTH");
  const int lane = threadIdx.x & (WAVE - 1);
  int off = WIDTH / 2;
#pragma unroll
  for (int h = NV / 2; h >= 1; h >>= 1) {
    if (off == 32 || off == 16) {  // (compile-time after unrolling)
#pragma unroll
      for (int i = 0; i < h; ++i) v[i] = off == 32 ? swap_add<true>(v[i], v[i + h]) : swap_add<false>(v[i], v[i + h]);
    } else {
      const bool up = (lane & off) != 0;
#pragma unroll
      for (int i = 0; i < h; ++i) {
        R a = v[i], b = v[i + h];
        R send = up ? a : b;
        R keep = up ? b : a;
        v[i] = keep + __shfl_xor(send, off, WAVE);
      }
    }
    off >>= 1;
  }
  R r = v[0];
#pragma unroll
  for (int o = (WIDTH / 2) / NV; o >= 1; o >>= 1) r += __shfl_xor(r, o, WAVE);
  return r;
}

template <int NV, int WIDTH = WAVE>
__device__ __forceinline__ int scatter_index() {  // value index owned by this lane after wave_reduce_scatter<NV>
  return ((threadIdx.x & (WAVE - 1)) & (WIDTH - 1)) >> (ilog2_c(WIDTH) - ilog2_c(NV));
}
template <int NV, int WIDTH = WAVE>
__device__ __forceinline__ bool scatter_leader() {  // one lane per value index (per group)
  return ((threadIdx.x & (WAVE - 1)) & ((WIDTH / NV) - 1)) == 0;
}

// dispatch helpers ---------------------------------------------------------------------------
template <int V>
using IntC = std::integral_constant<int, V>;
template <bool V>
using BoolC = std::integral_constant<bool, V>;
// compile-time loop: fn(IntC<0>()), ..., fn(IntC<N-1>())
template <int N, int I = 0, typename Fn>
__host__ __device__ __forceinline__ void static_for(Fn&& fn) {
  if constexpr (I < N) {
    fn(IntC<I>());
    static_for<N, I + 1>(fn);
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
// Program-order fence for register values: every instruction producing one of a[0..N) is emitted before this point
// and every consumer after it (no code is generated).  Used to keep the arithmetic that reads a ring slot above the
// refill of that slot -- plain arithmetic is otherwise free to sink below an asm statement, which keeps the old
// block live across the refill and forces the new one into other registers.
template <int B, typename R, int N>
__device__ __forceinline__ void value_fence_from(R (&a)[N]) {
  if constexpr (B < N) {
    asm volatile("" : "+v"(a[B]), "+v"(a[B + 1]), "+v"(a[B + 2]), "+v"(a[B + 3]));
    value_fence_from<B + 4>(a);
  }
}
template <typename R, int N>
__device__ __forceinline__ void value_fence(R (&a)[N]) {
  static_assert(N % 4 == 0, "fence works in groups of 4 values");
  value_fence_from<0>(a);
}
#else
template <typename R, int N>
__device__ void value_fence(R (&a)[N]);
#endif


}  // namespace assx
