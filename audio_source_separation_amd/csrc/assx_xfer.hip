// Host <-> HBM staging for the NumPy-in / NumPy-out call (include/assx.h: assx_upload / assx_download).
//
// The reference's `__call__` takes and returns pageable NumPy arrays (src/bss/ilrma.py:203-273); at config 4 that is
// 268.7 MB each way, more time than the 100 iterations in between when it goes through a pageable hipMemcpy
// (measured in round 2: ~12 GB/s, 44 ms of a 62 ms call).  Here the array is cut into chunks that ride a small ring
// of PINNED staging buffers: a pool of host threads copies (and, when the host and device precisions differ, converts)
// chunk i+1 into / out of its buffer while the DMA engine moves chunk i over PCIe on a stream of its own.  The
// precision conversion therefore costs no extra pass and the narrower of the two types crosses the bus.  Fresh
// destination pages of a download are first touched by the pool as well (single-threaded first touch alone would cost
// ~10 ms for 268 MB).
//
// No kernels in this file: it is host code next to the HIP runtime.
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "assx_common.hpp"

namespace assx {

namespace {

enum { XF_COPY = 0, XF_D2F = 1, XF_F2D = 2 };

inline void convert_slice(int mode, const void* src, void* dst, size_t n0, size_t n1, size_t src_es, size_t dst_es) {
  const char* s = (const char*)src + n0 * src_es;
  char* d = (char*)dst + n0 * dst_es;
  const size_t n = n1 - n0;
  if (mode == XF_COPY) {
    memcpy(d, s, n * src_es);
  } else if (mode == XF_D2F) {
    const double* sp = (const double*)s;
    float* dp = (float*)d;
    for (size_t i = 0; i < n; ++i) dp[i] = (float)sp[i];
  } else {
    const float* sp = (const float*)s;
    double* dp = (double*)d;
    for (size_t i = 0; i < n; ++i) dp[i] = (double)sp[i];
  }
}

// A fixed pool; one job at a time (the pipeline below is driven by a single host thread per context).
class Pool {
 public:
  explicit Pool(int n) : n_(n) {
    for (int i = 1; i < n_; ++i) th_.emplace_back([this, i] { loop(i); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int size() const { return n_; }
  // elements [0, n) of src -> dst, split into n_ contiguous slices on 64-element boundaries
  void run(int mode, const void* src, void* dst, size_t n, size_t src_es, size_t dst_es) {
    if (n_ == 1 || n < 65536) {
      convert_slice(mode, src, dst, 0, n, src_es, dst_es);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      mode_ = mode, src_ = src, dst_ = dst, n_elems_ = n, ses_ = src_es, des_ = dst_es;
      pending_.store(n_ - 1, std::memory_order_relaxed);
      ++gen_;
    }
    cv_.notify_all();
    slice(0);
    // the slices are a few hundred microseconds each: spin briefly, then sleep
    for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; ++spin) {
      if (spin > 2000) {
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_.load(std::memory_order_acquire) == 0; });
        break;
      }
      std::this_thread::yield();
    }
  }

 private:
  void slice(int i) {
    const size_t per = ((n_elems_ + (size_t)n_ - 1) / (size_t)n_ + 63) / 64 * 64;
    const size_t a = per * (size_t)i < n_elems_ ? per * (size_t)i : n_elems_;
    const size_t b = a + per < n_elems_ ? a + per : n_elems_;
    if (b > a) convert_slice(mode_, src_, dst_, a, b, ses_, des_);
  }
  void loop(int i) {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      slice(i);
      if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> lk(m_);
        done_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  unsigned long gen_ = 0;
  bool stop_ = false;
  std::atomic<int> pending_{0};
  int mode_ = 0;
  const void* src_ = nullptr;
  void* dst_ = nullptr;
  size_t n_elems_ = 0, ses_ = 0, des_ = 0;
};


}  // namespace

struct Xfer {
  static constexpr int NBUF = 4;
  size_t chunk_bytes = 0;  // staged bytes per chunk
  void* pin[NBUF] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev[NBUF] = {nullptr, nullptr, nullptr, nullptr};
  bool busy[NBUF] = {false, false, false, false};
  hipEvent_t fence = nullptr;
  hipStream_t copy = nullptr;
  Pool* pool = nullptr;

  ~Xfer() {
    for (int i = 0; i < NBUF; ++i) {
      if (busy[i]) (void)hipEventSynchronize(ev[i]);
      if (ev[i]) (void)hipEventDestroy(ev[i]);
      if (pin[i]) (void)hipHostFree(pin[i]);
    }
    if (fence) (void)hipEventDestroy(fence);
    if (copy) (void)hipStreamDestroy(copy);
    delete pool;
  }
};

static int xfer_get(assx_ctx* ctx, Xfer** out) {
  if (ctx->xfer) {
    *out = (Xfer*)ctx->xfer;
    return 0;
  }
  Xfer* x = new Xfer();
  // ASSX_XFER_CHUNK_MB: staged bytes per chunk; ASSX_XFER_THREADS: host threads per transfer (default: a quarter of the
  // hardware threads, at most 16 -- a copy stops scaling once it saturates the memory controllers it can reach)
  x->chunk_bytes = (size_t)(knob_int("ASSX_XFER_CHUNK_MB", 16) > 0 ? knob_int("ASSX_XFER_CHUNK_MB", 16) : 16) << 20;
  unsigned hw = std::thread::hardware_concurrency();
  int nt = knob_int("ASSX_XFER_THREADS", (int)(hw / 4 > 16 ? 16 : (hw / 4 < 1 ? 1 : hw / 4)));
  if (nt < 1) nt = 1;
  if (nt > 64) nt = 64;
  hipError_t e = hipStreamCreateWithFlags(&x->copy, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&x->fence, hipEventDisableTiming);
  for (int i = 0; i < Xfer::NBUF && e == hipSuccess; ++i) {
    e = hipHostMalloc(&x->pin[i], x->chunk_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev[i], hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    delete x;
    return hip_fail(ctx, e, "assx staging buffers");
  }
  x->pool = new Pool(nt);
  ctx->xfer = x;
  *out = x;
  return 0;
}

void xfer_destroy(assx_ctx* ctx) {
  if (ctx && ctx->xfer) {
    delete (Xfer*)ctx->xfer;
    ctx->xfer = nullptr;
  }
}

static inline size_t esize(int dtype) { return dtype == ASSX_F64 ? 8 : 4; }

}  // namespace assx

using namespace assx;

#define XF_HIP(ctx, call, what)                                  \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) return hip_fail((ctx), e__, (what));  \
  } while (0)

extern "C" {

int assx_upload(assx_ctx* ctx, const void* host, int host_dtype, void* dev, int dev_dtype, size_t count, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, (host && dev) || count == 0, ASSX_E_NULL, "assx_upload: NULL array");
  ASSX_REQUIRE(ctx, (host_dtype == ASSX_F32 || host_dtype == ASSX_F64) && (dev_dtype == ASSX_F32 || dev_dtype == ASSX_F64),
               ASSX_E_ARG, "assx_upload: dtype must be ASSX_F32 or ASSX_F64");
  if (count == 0) return 0;
  Xfer* x = nullptr;
  int rc = xfer_get(ctx, &x);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t hes = esize(host_dtype), des = esize(dev_dtype);
  const int mode = hes == des ? XF_COPY : (hes == 8 ? XF_D2F : XF_F2D);
  const size_t per = x->chunk_bytes / des;  // the DEVICE type is what sits in the staging buffer and crosses the bus
  // the destination may still be in use by work queued on `stream` (a recycled allocation): order the copies after it
  XF_HIP(ctx, hipEventRecord(x->fence, st), "assx_upload: hipEventRecord");
  XF_HIP(ctx, hipStreamWaitEvent(x->copy, x->fence, 0), "assx_upload: hipStreamWaitEvent");
  int last = -1;
  size_t c = 0;
  for (size_t off = 0; off < count; off += per, ++c) {
    const int s = (int)(c % Xfer::NBUF);
    const size_t n = count - off < per ? count - off : per;
    if (x->busy[s]) XF_HIP(ctx, hipEventSynchronize(x->ev[s]), "assx_upload: hipEventSynchronize");
    x->pool->run(mode, (const char*)host + off * hes, x->pin[s], n, hes, des);
    XF_HIP(ctx, hipMemcpyAsync((char*)dev + off * des, x->pin[s], n * des, hipMemcpyHostToDevice, x->copy),
           "assx_upload: hipMemcpyAsync");
    XF_HIP(ctx, hipEventRecord(x->ev[s], x->copy), "assx_upload: hipEventRecord");
    x->busy[s] = true;
    last = s;
  }
  // `host` has been consumed entirely; the tail of the DMA is still in flight from the staging buffers (owned by the
  // context): consumers on `stream` wait for it on the device, the host does not
  if (last >= 0) XF_HIP(ctx, hipStreamWaitEvent(st, x->ev[last], 0), "assx_upload: hipStreamWaitEvent");
  return 0;
}

int assx_download(assx_ctx* ctx, const void* dev, int dev_dtype, void* host, int host_dtype, size_t count, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, (host && dev) || count == 0, ASSX_E_NULL, "assx_download: NULL array");
  ASSX_REQUIRE(ctx, (host_dtype == ASSX_F32 || host_dtype == ASSX_F64) && (dev_dtype == ASSX_F32 || dev_dtype == ASSX_F64),
               ASSX_E_ARG, "assx_download: dtype must be ASSX_F32 or ASSX_F64");
  if (count == 0) return 0;
  Xfer* x = nullptr;
  int rc = xfer_get(ctx, &x);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t hes = esize(host_dtype), des = esize(dev_dtype);
  const int mode = hes == des ? XF_COPY : (des == 8 ? XF_D2F : XF_F2D);
  const size_t per = x->chunk_bytes / des;
  const size_t nchunks = (count + per - 1) / per;
  XF_HIP(ctx, hipEventRecord(x->fence, st), "assx_download: hipEventRecord");  // the producers of `dev` run on `stream`
  XF_HIP(ctx, hipStreamWaitEvent(x->copy, x->fence, 0), "assx_download: hipStreamWaitEvent");
  auto issue = [&](size_t c) -> hipError_t {
    const int s = (int)(c % Xfer::NBUF);
    const size_t off = c * per, n = count - off < per ? count - off : per;
    hipError_t e = hipSuccess;
    if (x->busy[s]) e = hipEventSynchronize(x->ev[s]);  // an upload's tail still reading this buffer
    if (e == hipSuccess) e = hipMemcpyAsync(x->pin[s], (const char*)dev + off * des, n * des, hipMemcpyDeviceToHost, x->copy);
    if (e == hipSuccess) e = hipEventRecord(x->ev[s], x->copy);
    x->busy[s] = e == hipSuccess;
    return e;
  };
  for (size_t c = 0; c < nchunks && c < (size_t)Xfer::NBUF - 1; ++c) XF_HIP(ctx, issue(c), "assx_download: hipMemcpyAsync");
  for (size_t c = 0; c < nchunks; ++c) {
    const int s = (int)(c % Xfer::NBUF);
    const size_t off = c * per, n = count - off < per ? count - off : per;
    // keep NBUF - 1 chunks in flight: the slot freed by the previous trip takes chunk c + NBUF - 1
    if (c + Xfer::NBUF - 1 < nchunks) XF_HIP(ctx, issue(c + Xfer::NBUF - 1), "assx_download: hipMemcpyAsync");
    XF_HIP(ctx, hipEventSynchronize(x->ev[s]), "assx_download: hipEventSynchronize");
    x->busy[s] = false;
    x->pool->run(mode, x->pin[s], (char*)host + off * hes, n, des, hes);
  }
  return 0;
}

}  // extern "C"
