// Context management and version for the assx C-ABI (include/assx.h).
#include "assx_common.hpp"

namespace assx {

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
      s = &ctx->tk[ctx->n_tk++];
      s->p = nullptr;
      s->n = 0;
    } else {
      // more streams than slots: the least recently used slot changes hands.  Its words are zero and unused once the
      // device is idle (a rare, slow path: a context serves one host thread and normally one or two streams).
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
  size_t want = s->n ? s->n * 2 : 8192;  // geometric growth: a handful of buffers per stream at most
  while (want < n) want *= 2;
  if (s->p && ctx->n_old_tickets >= NOLD) {
    int rc = tickets_drain_old(ctx);
    if (rc) return rc;
  }
  int* p = nullptr;
  int rc = tickets_alloc(ctx, want, st, &p);
  if (rc) return rc;
  if (s->p) ctx->old_tickets[ctx->n_old_tickets++] = s->p;  // launches in flight on `st` may still use it
  s->p = p;
  s->n = want;
  s->used = ctx->tk_clock;
  *out = p;
  return 0;
}

void tickets_destroy(assx_ctx* ctx) {
  for (int i = 0; i < ctx->n_tk; ++i)
    if (ctx->tk[i].p) (void)hipFree(ctx->tk[i].p);
  for (int i = 0; i < ctx->n_old_tickets; ++i) (void)hipFree(ctx->old_tickets[i]);
  ctx->n_tk = 0;
  ctx->n_old_tickets = 0;
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

const char* assx_version(void) { return ASSX_VERSION_STRING; }

}  // extern "C"
