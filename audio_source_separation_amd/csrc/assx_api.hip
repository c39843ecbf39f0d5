// Context management and version for the assx C-ABI (include/assx.h).
#include "assx_common.hpp"

namespace assx {

static bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return cs != hipStreamCaptureStatusNone;
}

static int capture_refusal(assx_ctx* ctx, const char* what) {
  return fail(ctx, ASSX_E_UNSUPPORTED,
              "ticket words: %s is not possible while the stream is being captured (it needs hipMalloc / "
              "hipDeviceSynchronize).  Run the same call once on this stream BEFORE the capture begins (graph users: warm "
              "up on the capture stream), then capture.", what);
}

static int tickets_alloc(assx_ctx* ctx, size_t want, hipStream_t st, int** out) {
  int* p = nullptr;
  hipError_t e = hipMalloc((void**)&p, want * sizeof(int));
  if (e == hipSuccess) e = hipMemsetAsync(p, 0, want * sizeof(int), st);
  if (e != hipSuccess) {
    if (p) (void)hipFree(p);
    *out = nullptr;
    return hip_fail(ctx, e, "ensure_tickets");
  }
  *out = p;
  return 0;
}

// every launch that could still count on a retired buffer has finished once the device is idle
static int tickets_drain_old(assx_ctx* ctx) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return hip_fail(ctx, e, "ensure_tickets (synchronise before recycling ticket buffers)");
  for (int i = 0; i < ctx->n_old_tickets; ++i) (void)hipFree(ctx->old_tickets[i]);
  ctx->n_old_tickets = 0;
  return 0;
}

int tickets_reserve(assx_ctx* ctx) {
  constexpr size_t NSLOT = sizeof(ctx->tk) / sizeof(ctx->tk[0]);
  const size_t bytes = NSLOT * assx_ctx::TK_INIT * sizeof(int);
  int* p = nullptr;
  hipError_t e = hipMalloc((void**)&p, bytes);
  if (e == hipSuccess) e = hipMemset(p, 0, bytes);
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // zero before any stream of the context can count on them
  if (e != hipSuccess) {
    if (p) (void)hipFree(p);
    return (int)e;
  }
  ctx->tk_pool = p;
  return 0;
}

int ensure_tickets(assx_ctx* ctx, size_t n, hipStream_t st, int** out) {
  *out = nullptr;
  constexpr int NSLOT = (int)(sizeof(ctx->tk) / sizeof(ctx->tk[0]));
  constexpr int NOLD = (int)(sizeof(ctx->old_tickets) / sizeof(ctx->old_tickets[0]));
  ++ctx->tk_clock;
  assx_ctx::TicketSlot* s = nullptr;
  for (int i = 0; i < ctx->n_tk; ++i)
    if (ctx->tk[i].st == st) s = &ctx->tk[i];
  if (s && s->n >= n) {
    s->used = ctx->tk_clock;
    *out = s->p;
    return 0;
  }
  if (!s) {
    if (ctx->n_tk < NSLOT) {
      // a stream this context has not seen: its range of the pool -- no allocation, legal inside a stream capture
      const int i = ctx->n_tk++;
      s = &ctx->tk[i];
      s->p = ctx->tk_pool + (size_t)i * assx_ctx::TK_INIT;
      s->n = assx_ctx::TK_INIT;
      s->pooled = true;
    } else {
      // more streams than slots: the least recently used slot changes hands.  Its words are zero and unused once the
      // device is idle (a rare, slow path: a context serves one host thread and normally one or two streams).
      if (stream_is_capturing(st)) return capture_refusal(ctx, "recycling the ticket slot of another stream (more than 16 streams on one context)");
      int rc = tickets_drain_old(ctx);
      if (rc) return rc;
      s = &ctx->tk[0];
      for (int i = 1; i < NSLOT; ++i)
        if (ctx->tk[i].used < s->used) s = &ctx->tk[i];
    }
    s->st = st;
    s->used = ctx->tk_clock;
    if (s->n >= n) {
      *out = s->p;
      return 0;
    }
  }
  if (stream_is_capturing(st)) return capture_refusal(ctx, "growing the ticket buffer of this stream");
  size_t want = s->n ? s->n * 2 : assx_ctx::TK_INIT;  // geometric growth: a handful of buffers per stream at most
  while (want < n) want *= 2;
  if (s->p && !s->pooled && ctx->n_old_tickets >= NOLD) {
    int rc = tickets_drain_old(ctx);
    if (rc) return rc;
  }
  int* p = nullptr;
  int rc = tickets_alloc(ctx, want, st, &p);
  if (rc) return rc;
  // launches in flight on `st` may still use the old buffer: kept until the device has been idle (a range of the pool
  // simply stays where it is, zero and unused from now on)
  if (s->p && !s->pooled) ctx->old_tickets[ctx->n_old_tickets++] = s->p;
  s->p = p;
  s->n = want;
  s->pooled = false;
  s->used = ctx->tk_clock;
  *out = p;
  return 0;
}

void tickets_destroy(assx_ctx* ctx) {
  // Kernels of this context may still be counting on these words (a worker thread that ends right after queueing its
  // launches: its context is destroyed by the garbage collector, on whatever thread and current device that runs --
  // round 5's advisor).  Wait for the CONTEXT's device, whichever device is current here, before anything is freed.
  int prev = -1;
  (void)hipGetDevice(&prev);
  const bool switched = prev != ctx->device && hipSetDevice(ctx->device) == hipSuccess;
  (void)hipDeviceSynchronize();
  for (int i = 0; i < ctx->n_tk; ++i)
    if (ctx->tk[i].p && !ctx->tk[i].pooled) (void)hipFree(ctx->tk[i].p);
  for (int i = 0; i < ctx->n_old_tickets; ++i) (void)hipFree(ctx->old_tickets[i]);
  if (ctx->tk_pool) (void)hipFree(ctx->tk_pool);
  ctx->tk_pool = nullptr;
  ctx->n_tk = 0;
  ctx->n_old_tickets = 0;
  if (switched && prev >= 0) (void)hipSetDevice(prev);
  (void)hipGetLastError();
}

}  // namespace assx

extern "C" {

int assx_ctx_create(int device, assx_ctx** out) {
  if (!out) return ASSX_E_NULL;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess) return (int)e;
  if (device < 0 || device >= count) return ASSX_E_ARG;
  // the current device of the calling thread is left alone (it belongs to the host framework); entry points check
  // that it equals `device` when they are called (assx_common.hpp: check_ctx_device)
  assx_ctx* c = (assx_ctx*)calloc(1, sizeof(assx_ctx));
  if (!c) return ASSX_E_ARG;
  c->device = device;
  c->err[0] = 0;
  // the ticket pool lives on `device`: make it current for the allocation only
  int prev = -1;
  e = hipGetDevice(&prev);
  if (e == hipSuccess && prev != device) e = hipSetDevice(device);
  int rc = e == hipSuccess ? assx::tickets_reserve(c) : (int)e;
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (rc) {
    free(c);
    return rc;
  }
  *out = c;
  return 0;
}

int assx_ctx_destroy(assx_ctx* ctx) {
  if (!ctx) return ASSX_E_NULL;
  assx::xfer_destroy(ctx);
  assx::tickets_destroy(ctx);
  free(ctx);
  return 0;
}

const char* assx_last_error(const assx_ctx* ctx) { return ctx ? ctx->err : "ctx is NULL"; }

// a laboratory build (-DASSX_LAB=1: the environment switches of the measured-and-not-kept variants are live) says so
const char* assx_version(void) { return ASSX_LAB ? ASSX_VERSION_STRING "+lab" : ASSX_VERSION_STRING; }

}  // extern "C"
