// STFT / iSTFT either side of the separation loop, gfx950.
//
// Semantics: scipy.signal.stft / istft exactly as the reference calls them (src/transform/stft.py:4-17):
// window of nperseg samples, hop = nperseg - noverlap, boundary='zeros' (nperseg//2 zeros either side),
// padded=True (zeros up to a whole number of hops), no detrending, scaling='spectrum' (X = rfft(w * seg) / sum(w));
// the inverse is the weighted overlap-add  y = sum_t w * irfft(X_t) * sum(w) / sum_t w^2  with the boundary removed.
//
// Layout: the separation loop wants X as (C, F, T) with T fastest, an FFT wants a frame contiguous.  One workgroup
// transforms one frame in LDS and writes it frame-major into scratch (coalesced); a tiled transpose produces
// (C, F, T).  The inverse runs the same two steps backwards and finishes with a gather-form overlap-add (each output
// sample sums the frames that cover it in ascending frame order: deterministic, no atomics).
//   * power-of-two nperseg <= 8192: radix-2 Stockham passes on one LDS buffer (register-staged, in place);
//   * anything else: direct DFT against the same twiddle table (exact index arithmetic, O(N^2) per frame).
// Twiddles exp(-2 pi i m / N) are built per call with sincospi in float64.
#include "assx_common.hpp"

using namespace assx;

namespace {

constexpr int FFT_THREADS = 256;
constexpr int FFT_MAX_POW2 = 8192;                        // N complex values of one frame in LDS (128 KB in float64)
constexpr int FFT_MAXB = FFT_MAX_POW2 / 2 / FFT_THREADS;  // butterflies per thread and pass

template <typename R>
__global__ void __launch_bounds__(256) twiddle_kernel(Cx<R>* __restrict__ tw, int N) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= N) return;
  double s, c;
  sincospi(-2.0 * (double)m / (double)N, &s, &c);
  tw[m] = cmake<R>((R)c, (R)s);
}

// In-place FFT of buf[0..N) (LDS), N a power of two.  Stockham DIT: natural order in, natural order out; every pass
// reads its butterflies into registers, then writes them to their auto-sorted places.
template <typename R, bool INV>
__device__ __forceinline__ void fft_pow2_lds(Cx<R>* buf, const Cx<R>* __restrict__ tw, int N) {
  const int half = N >> 1, tid = threadIdx.x;
  for (int p = 1; p < N; p <<= 1) {
    Cx<R> a[FFT_MAXB], b[FFT_MAXB];
#pragma unroll
    for (int q = 0; q < FFT_MAXB; ++q) {
      const int i = tid + q * FFT_THREADS;
      if (i < half) {
        a[q] = buf[i];
        b[q] = buf[i + half];
      }
    }
    __syncthreads();
    const int tstride = half / p;  // exp(-i pi k / p) = tw[k * N / (2 p)]
#pragma unroll
    for (int q = 0; q < FFT_MAXB; ++q) {
      const int i = tid + q * FFT_THREADS;
      if (i < half) {
        const int k = i & (p - 1), j = ((i - k) << 1) + k;
        Cx<R> w = tw[k * tstride];
        if (INV) w.y = -w.y;
        const Cx<R> u = cmul(w, b[q]);
        buf[j] = cadd(a[q], u);
        buf[j + p] = csub(a[q], u);
      }
    }
    __syncthreads();
  }
}

// ---- forward: one frame per workgroup -> tmp (C, T, F) ----------------------------------------------------
template <typename R, bool POW2>
__global__ void __launch_bounds__(FFT_THREADS) stft_frame_kernel(const R* __restrict__ x, const R* __restrict__ win,
                                                                const Cx<R>* __restrict__ tw, Cx<R>* __restrict__ tmp,
                                                                long long L, int N, int hop, int T, R scale) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int t = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
  const int F = N / 2 + 1;
  const long long start = (long long)t * hop - N / 2;
  const R* xc = x + (size_t)c * L;
  Cx<R>* out = tmp + ((size_t)c * T + t) * F;
  if (POW2) {
    Cx<R>* buf = reinterpret_cast<Cx<R>*>(smem);
    for (int n = tid; n < N; n += FFT_THREADS) {
      const long long i = start + n;
      buf[n] = cmake<R>((i >= 0 && i < L) ? xc[i] * win[n] : (R)0, (R)0);
    }
    __syncthreads();
    fft_pow2_lds<R, false>(buf, tw, N);
    for (int f = tid; f < F; f += FFT_THREADS) out[f] = cmake<R>(buf[f].x * scale, buf[f].y * scale);
  } else {
    R* seg = reinterpret_cast<R*>(smem);
    for (int n = tid; n < N; n += FFT_THREADS) {
      const long long i = start + n;
      seg[n] = (i >= 0 && i < L) ? xc[i] * win[n] : (R)0;
    }
    __syncthreads();
    for (int f = tid; f < F; f += FFT_THREADS) {
      R re = 0, im = 0;
      int m = 0;  // f * n mod N
      for (int n = 0; n < N; ++n) {
        const Cx<R> w = tw[m];
        re = fma(seg[n], w.x, re);
        im = fma(seg[n], w.y, im);
        m += f;
        if (m >= N) m -= N;
      }
      out[f] = cmake<R>(re * scale, im * scale);
    }
  }
}

// ---- inverse: one frame per workgroup, tmp (C, T, F) -> seg (C, T, N) = w * irfft(X_t) ----------------------
template <typename R, bool POW2>
__global__ void __launch_bounds__(FFT_THREADS) istft_frame_kernel(const Cx<R>* __restrict__ tmp, const R* __restrict__ win,
                                                                 const Cx<R>* __restrict__ tw, R* __restrict__ seg, int N,
                                                                 int T) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int t = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
  const int F = N / 2 + 1;
  const Cx<R>* in = tmp + ((size_t)c * T + t) * F;
  R* out = seg + ((size_t)c * T + t) * N;
  const R invN = (R)(1.0 / (double)N);
  Cx<R>* buf = reinterpret_cast<Cx<R>*>(smem);
  if (POW2) {
    // Hermitian extension; the imaginary parts of the DC and Nyquist bins are ignored, as irfft does
    for (int k = tid; k < N; k += FFT_THREADS) {
      Cx<R> v;
      if (k < F) {
        v = in[k];
        if (k == 0 || 2 * k == N) v.y = 0;
      } else {
        v = in[N - k];
        v.y = -v.y;
      }
      buf[k] = v;
    }
    __syncthreads();
    fft_pow2_lds<R, true>(buf, tw, N);
    for (int n = tid; n < N; n += FFT_THREADS) out[n] = buf[n].x * invN * win[n];
  } else {
    for (int k = tid; k < F; k += FFT_THREADS) buf[k] = in[k];
    __syncthreads();
    const int kmax = (N - 1) / 2;  // bins with a distinct conjugate partner
    for (int n = tid; n < N; n += FFT_THREADS) {
      R acc = 0;
      int m = n % N;  // k * n mod N, starting at k = 1
      for (int k = 1; k <= kmax; ++k) {
        const Cx<R> w = tw[m];  // exp(-2 pi i k n / N); the inverse uses its conjugate
        acc += buf[k].x * w.x + buf[k].y * w.y;
        m += n;
        if (m >= N) m -= N;
      }
      R y = buf[0].x + (R)2 * acc;
      if ((N & 1) == 0) y += ((n & 1) ? -buf[N / 2].x : buf[N / 2].x);
      out[n] = y * invN * win[n];
    }
  }
}

// ---- overlap-add, gather form: y[c, i] = sum(w) * sum_t seg[c, t, p - t hop] / sum_t w[p - t hop]^2 ----------
template <typename R>
__global__ void __launch_bounds__(256) overlap_add_kernel(const R* __restrict__ seg, const R* __restrict__ win,
                                                         R* __restrict__ y, int N, int hop, int T, long long Lout,
                                                         R winsum) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Lout) return;
  const int c = blockIdx.y;
  const long long p = i + N / 2;
  long long t_lo = p >= N ? (p - N) / hop + 1 : 0;
  long long t_hi = p / hop;
  if (t_hi > T - 1) t_hi = T - 1;
  R acc = 0, nrm = 0;
  for (long long t = t_lo; t <= t_hi; ++t) {
    const int n = (int)(p - t * hop);
    acc += seg[((size_t)c * T + t) * N + n];
    nrm = fma(win[n], win[n], nrm);
  }
  y[(size_t)c * Lout + i] = acc * winsum / (nrm > (R)1e-10 ? nrm : (R)1);
}

// ---- (C, A, Bd) -> (C, Bd, A), complex, 32 x 32 tiles through LDS ---------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) transpose_kernel(const Cx<R>* __restrict__ in, Cx<R>* __restrict__ out, int A,
                                                       int Bd) {
  __shared__ Cx<R> tile[32][33];
  const int c = blockIdx.z, a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const Cx<R>* ic = in + (size_t)c * A * Bd;
  Cx<R>* oc = out + (size_t)c * A * Bd;
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (a0 + r < A && b0 + tx < Bd) tile[r][tx] = ic[(size_t)(a0 + r) * Bd + b0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (b0 + r < Bd && a0 + tx < A) oc[(size_t)(b0 + r) * A + a0 + tx] = tile[tx][r];
}

struct StftWs {
  size_t tw, tmp, seg, total;
};
inline StftWs stft_ws(int C, int N, int T, int dtype) {
  const size_t r = dtype == ASSX_F64 ? 8 : 4;
  StftWs L;
  size_t off = 0;
  L.tw = off;
  off += align_up((size_t)N * 2 * r, 256);
  L.tmp = off;
  off += align_up((size_t)C * T * (N / 2 + 1) * 2 * r, 256);
  L.seg = off;
  off += align_up((size_t)C * T * N * r, 256);
  L.total = off;
  return L;
}

inline bool is_pow2(int n) { return n >= 2 && (n & (n - 1)) == 0; }

template <typename K>
int allow_lds(assx_ctx* ctx, K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes);
  if (e != hipSuccess) return hip_fail(ctx, e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  return 0;
}

template <typename R>
int stft_impl(assx_ctx* ctx, const void* x, const void* window, double window_sum, void* X, void* ws, int C,
              long long L, int N, int hop, int T, int dtype, hipStream_t st) {
  const StftWs W = stft_ws(C, N, T, dtype);
  Cx<R>* tw = (Cx<R>*)((char*)ws + W.tw);
  Cx<R>* tmp = (Cx<R>*)((char*)ws + W.tmp);
  const int F = N / 2 + 1;
  hipLaunchKernelGGL((twiddle_kernel<R>), dim3((N + 255) / 256), dim3(256), 0, st, tw, N);
  ASSX_LAUNCH_CHECK(ctx, "twiddle_kernel");
  const R scale = (R)(1.0 / window_sum);
  int rc;
  if (is_pow2(N) && N <= FFT_MAX_POW2) {
    const size_t lds = (size_t)N * sizeof(Cx<R>);
    if ((rc = allow_lds(ctx, stft_frame_kernel<R, true>, lds))) return rc;
    hipLaunchKernelGGL((stft_frame_kernel<R, true>), dim3(T, C), dim3(FFT_THREADS), lds, st, (const R*)x,
                       (const R*)window, (const Cx<R>*)tw, tmp, L, N, hop, T, scale);
  } else {
    const size_t lds = (size_t)N * sizeof(R);
    ASSX_REQUIRE(ctx, lds <= 160 * 1024, ASSX_E_UNSUPPORTED, "fft_size %d does not fit one frame in LDS", N);
    if ((rc = allow_lds(ctx, stft_frame_kernel<R, false>, lds))) return rc;
    hipLaunchKernelGGL((stft_frame_kernel<R, false>), dim3(T, C), dim3(FFT_THREADS), lds, st, (const R*)x,
                       (const R*)window, (const Cx<R>*)tw, tmp, L, N, hop, T, scale);
  }
  ASSX_LAUNCH_CHECK(ctx, "stft_frame_kernel");
  hipLaunchKernelGGL((transpose_kernel<R>), dim3((F + 31) / 32, (T + 31) / 32, C), dim3(256), 0, st,
                     (const Cx<R>*)tmp, (Cx<R>*)X, T, F);
  ASSX_LAUNCH_CHECK(ctx, "transpose_kernel");
  return 0;
}

template <typename R>
int istft_impl(assx_ctx* ctx, const void* X, const void* window, double window_sum, void* y, void* ws, int C, int N,
               int hop, int T, int dtype, hipStream_t st) {
  const StftWs W = stft_ws(C, N, T, dtype);
  Cx<R>* tw = (Cx<R>*)((char*)ws + W.tw);
  Cx<R>* tmp = (Cx<R>*)((char*)ws + W.tmp);
  R* seg = (R*)((char*)ws + W.seg);
  const int F = N / 2 + 1;
  hipLaunchKernelGGL((twiddle_kernel<R>), dim3((N + 255) / 256), dim3(256), 0, st, tw, N);
  ASSX_LAUNCH_CHECK(ctx, "twiddle_kernel");
  hipLaunchKernelGGL((transpose_kernel<R>), dim3((T + 31) / 32, (F + 31) / 32, C), dim3(256), 0, st, (const Cx<R>*)X,
                     tmp, F, T);
  ASSX_LAUNCH_CHECK(ctx, "transpose_kernel");
  int rc;
  if (is_pow2(N) && N <= FFT_MAX_POW2) {
    const size_t lds = (size_t)N * sizeof(Cx<R>);
    if ((rc = allow_lds(ctx, istft_frame_kernel<R, true>, lds))) return rc;
    hipLaunchKernelGGL((istft_frame_kernel<R, true>), dim3(T, C), dim3(FFT_THREADS), lds, st, (const Cx<R>*)tmp,
                       (const R*)window, (const Cx<R>*)tw, seg, N, T);
  } else {
    const size_t lds = (size_t)F * sizeof(Cx<R>);
    ASSX_REQUIRE(ctx, lds <= 160 * 1024, ASSX_E_UNSUPPORTED, "fft_size %d does not fit one frame in LDS", N);
    if ((rc = allow_lds(ctx, istft_frame_kernel<R, false>, lds))) return rc;
    hipLaunchKernelGGL((istft_frame_kernel<R, false>), dim3(T, C), dim3(FFT_THREADS), lds, st, (const Cx<R>*)tmp,
                       (const R*)window, (const Cx<R>*)tw, seg, N, T);
  }
  ASSX_LAUNCH_CHECK(ctx, "istft_frame_kernel");
  const long long Lout = (long long)N + (long long)(T - 1) * hop - 2 * (N / 2);
  if (Lout > 0) {
    hipLaunchKernelGGL((overlap_add_kernel<R>), dim3((unsigned)((Lout + 255) / 256), C), dim3(256), 0, st,
                       (const R*)seg, (const R*)window, (R*)y, N, hop, T, Lout, (R)window_sum);
    ASSX_LAUNCH_CHECK(ctx, "overlap_add_kernel");
  }
  return 0;
}

}  // namespace

extern "C" {

long long assx_stft_num_frames(long long n_samples, int fft_size, int hop) {
  if (n_samples < 0 || fft_size < 1 || hop < 1) return -1;
  const long long Lb = n_samples + 2 * (long long)(fft_size / 2);
  long long rem = (Lb - fft_size) % hop;  // Python's (-(Lb - N)) % hop
  if (rem < 0) rem += hop;
  long long nadd = rem == 0 ? 0 : hop - rem;
  nadd %= fft_size;
  const long long Lp = Lb + nadd;
  if (Lp < fft_size) return 0;
  return (Lp - fft_size) / hop + 1;
}

long long assx_istft_num_samples(int fft_size, int hop, int n_frames) {
  if (fft_size < 1 || hop < 1 || n_frames < 1) return -1;
  return (long long)fft_size + (long long)(n_frames - 1) * hop - 2 * (long long)(fft_size / 2);
}

size_t assx_stft_workspace_bytes(int C, int fft_size, int n_frames, int dtype) {
  if (C < 1 || fft_size < 1 || n_frames < 1) return 0;
  return stft_ws(C, fft_size, n_frames, dtype).total;
}

int assx_stft(assx_ctx* ctx, const void* x, const void* window, double window_sum, void* X, void* ws, int C,
              long long n_samples, int fft_size, int hop, int n_frames, int dtype, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, x && window && X && ws, ASSX_E_NULL, "assx_stft: NULL array");
  ASSX_REQUIRE(ctx, C >= 1 && n_samples >= 1 && fft_size >= 2 && hop >= 1, ASSX_E_ARG,
               "assx_stft: invalid sizes C=%d n_samples=%lld fft_size=%d hop=%d", C, n_samples, fft_size, hop);
  ASSX_REQUIRE(ctx, n_samples >= fft_size, ASSX_E_UNSUPPORTED,
               "assx_stft: the signal (%lld samples) is shorter than fft_size=%d (scipy would shrink the window)", n_samples,
               fft_size);
  ASSX_REQUIRE(ctx, n_frames == assx_stft_num_frames(n_samples, fft_size, hop), ASSX_E_ARG,
               "assx_stft: n_frames=%d but %lld samples give %lld frames", n_frames, n_samples,
               assx_stft_num_frames(n_samples, fft_size, hop));
  ASSX_REQUIRE(ctx, window_sum != 0.0, ASSX_E_ARG, "assx_stft: the window sums to zero");
  ASSX_REQUIRE(ctx, C <= 65535, ASSX_E_UNSUPPORTED, "assx_stft: at most 65535 channels per call");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ASSX_F64)
    return stft_impl<double>(ctx, x, window, window_sum, X, ws, C, n_samples, fft_size, hop, n_frames, dtype, st);
  if (dtype == ASSX_F32)
    return stft_impl<float>(ctx, x, window, window_sum, X, ws, C, n_samples, fft_size, hop, n_frames, dtype, st);
  return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
}

int assx_istft(assx_ctx* ctx, const void* X, const void* window, double window_sum, void* y, void* ws, int C,
               int fft_size, int hop, int n_frames, int dtype, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, X && window && y && ws, ASSX_E_NULL, "assx_istft: NULL array");
  ASSX_REQUIRE(ctx, C >= 1 && n_frames >= 1 && fft_size >= 2 && hop >= 1, ASSX_E_ARG,
               "assx_istft: invalid sizes C=%d n_frames=%d fft_size=%d hop=%d", C, n_frames, fft_size, hop);
  ASSX_REQUIRE(ctx, C <= 65535, ASSX_E_UNSUPPORTED, "assx_istft: at most 65535 channels per call");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ASSX_F64) return istft_impl<double>(ctx, X, window, window_sum, y, ws, C, fft_size, hop, n_frames, dtype, st);
  if (dtype == ASSX_F32) return istft_impl<float>(ctx, X, window, window_sum, y, ws, C, fft_size, hop, n_frames, dtype, st);
  return fail(ctx, ASSX_E_ARG, "bad dtype %d", dtype);
}

}  // extern "C"
