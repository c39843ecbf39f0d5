// A HOST emulation of the part of the HIP runtime that csrc/assx_api.hip and csrc/assx_xfer.hip call, so that the
// library's host-side threaded code -- the pinned staging ring, its thread pool, the event hand-overs, the per-stream
// ticket slots -- can run under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer in a container without a
// GPU (tools/sanitize/run.sh; round 5's review, weak #11).  Test infrastructure: never linked into libassx.so.
//
// What is modelled, because the code under test depends on it:
//   * a stream is an in-order queue served by its own thread: hipMemcpyAsync returns at once and the copy happens later,
//     concurrently with the caller (a staging buffer reused too early IS a data race here, as it is with a DMA engine);
//   * hipEventRecord marks a point of a stream, hipStreamWaitEvent makes another stream wait for it, hipEventSynchronize
//     makes the host wait; hipStreamSynchronize / hipDeviceSynchronize drain;
//   * "device" and pinned memory are plain heap blocks (so ASan sees every out-of-bounds byte of a copy).
// Everything else returns hipSuccess or a fixed answer (one device, device 0 current, no stream is ever capturing).
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

namespace {

struct Event {
  std::mutex m;
  std::condition_variable cv;
  unsigned long recorded = 0, done = 0;  // hipEventRecord calls issued / completed
};

struct Stream {
  std::mutex m;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  bool stop = false, busy = false;
  std::thread th;
  Stream() : th([this] { loop(); }) {}
  ~Stream() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    th.join();
  }
  void push(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lk(m);
      q.push_back(std::move(f));
    }
    cv.notify_all();
  }
  void drain() {
    std::unique_lock<std::mutex> lk(m);
    idle.wait(lk, [this] { return q.empty() && !busy; });
  }
  void loop() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return stop || !q.empty(); });
        if (q.empty()) return;
        f = std::move(q.front());
        q.pop_front();
        busy = true;
      }
      f();
      {
        std::lock_guard<std::mutex> lk(m);
        busy = false;
      }
      idle.notify_all();
    }
  }
};

std::mutex g_m;
std::set<Stream*> g_streams;
Stream* g_null = nullptr;
int g_device = 0;

Stream* S(hipStream_t st) {
  if (st) return reinterpret_cast<Stream*>(st);
  std::lock_guard<std::mutex> lk(g_m);
  if (!g_null) {
    g_null = new Stream;
    g_streams.insert(g_null);
  }
  return g_null;
}
bool known(hipStream_t st) {  // the ticket tests hand in made-up stream handles that are never dereferenced
  std::lock_guard<std::mutex> lk(g_m);
  return st == nullptr || g_streams.count(reinterpret_cast<Stream*>(st)) != 0;
}

}  // namespace

extern "C" {

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = g_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d != 0) return hipErrorInvalidDevice; g_device = d; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }

hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemset(void* p, int v, size_t n) { S(nullptr)->drain(); memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t st) {
  if (!known(st)) { memset(p, v, n); return hipSuccess; }
  S(st)->push([=] { memset(p, v, n); });
  return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { S(nullptr)->drain(); memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
  S(st)->push([=] { memcpy(d, s, n); });
  return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) {
  Stream* s = new Stream;
  {
    std::lock_guard<std::mutex> lk(g_m);
    g_streams.insert(s);
  }
  *st = reinterpret_cast<hipStream_t>(s);
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t st) {
  Stream* s = reinterpret_cast<Stream*>(st);
  s->drain();
  {
    std::lock_guard<std::mutex> lk(g_m);
    g_streams.erase(s);
  }
  delete s;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t st) { if (known(st)) S(st)->drain(); return hipSuccess; }
hipError_t hipDeviceSynchronize(void) {
  std::vector<Stream*> all;
  {
    std::lock_guard<std::mutex> lk(g_m);
    all.assign(g_streams.begin(), g_streams.end());
  }
  for (Stream* s : all) s->drain();
  return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* cs) { *cs = hipStreamCaptureStatusNone; return hipSuccess; }

hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new Event); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<Event*>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) {
  Event* ev = reinterpret_cast<Event*>(e);
  unsigned long gen;
  {
    std::lock_guard<std::mutex> lk(ev->m);
    gen = ++ev->recorded;
  }
  S(st)->push([ev, gen] {
    std::lock_guard<std::mutex> lk(ev->m);  // notified under the lock: a waiter may destroy the event as soon as it wakes
    if (ev->done < gen) ev->done = gen;
    ev->cv.notify_all();
  });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  Event* ev = reinterpret_cast<Event*>(e);
  std::unique_lock<std::mutex> lk(ev->m);
  const unsigned long want = ev->recorded;
  ev->cv.wait(lk, [&] { return ev->done >= want; });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned) {
  Event* ev = reinterpret_cast<Event*>(e);
  unsigned long want;
  {
    std::lock_guard<std::mutex> lk(ev->m);
    want = ev->recorded;  // the record call(s) made so far, as in HIP
  }
  S(st)->push([ev, want] {
    std::unique_lock<std::mutex> lk(ev->m);
    ev->cv.wait(lk, [&] { return ev->done >= want; });
  });
  return hipSuccess;
}

}  // extern "C"
