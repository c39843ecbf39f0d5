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
};

namespace assx {

constexpr int WAVE = 64;

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

// principal complex square root (numpy.sqrt on complex128)
__device__ __forceinline__ Cx<double> csqrt_principal(Cx<double> z) {
  if (z.y == 0.0 && z.x >= 0.0) return cmake<double>(sqrt(z.x), 0.0);
  double m = hypot(z.x, z.y);
  double s = sqrt(0.5 * (m + fabs(z.x)));
  double t = z.y / (2.0 * s);
  if (z.x >= 0.0) return cmake<double>(s, t);
  return cmake<double>(fabs(t), copysign(s, z.y));
}

// ------------------------------------------------------------------------------------------
// x**e with the fast paths numpy itself takes for e in {1, 2, 0.5}; generic pow otherwise.
// ------------------------------------------------------------------------------------------
enum PowMode { POW_GENERIC = 0, POW_ID = 1, POW_SQUARE = 2, POW_CUBE = 3, POW_SQRT = 4 };

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
template <typename R, int NV, int WIDTH = WAVE>
__device__ __forceinline__ R wave_reduce_scatter(R (&v)[NV]) {
  static_assert(NV >= 1 && NV <= WIDTH && (NV & (NV - 1)) == 0, "NV must be a power of two <= WIDTH");
  const int lane = threadIdx.x & (WAVE - 1);
  int off = WIDTH / 2;
#pragma unroll
  for (int h = NV / 2; h >= 1; h >>= 1) {
    const bool up = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      R a = v[i], b = v[i + h];
      R send = up ? a : b;
      R keep = up ? b : a;
      v[i] = keep + __shfl_xor(send, off, WAVE);
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

}  // namespace assx
