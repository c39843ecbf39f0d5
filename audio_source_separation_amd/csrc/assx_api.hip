// Context management and version for the assx C-ABI (include/assx.h).
#include "assx_common.hpp"

namespace assx {

int* ensure_tickets(assx_ctx* ctx, size_t n, hipStream_t st) {
  if (ctx->tickets && ctx->n_tickets >= n) return ctx->tickets;
  if (ctx->n_old_tickets >= (int)(sizeof(ctx->old_tickets) / sizeof(ctx->old_tickets[0]))) {
    fail(ctx, ASSX_E_UNSUPPORTED, "ticket buffer regrown too often");
    return nullptr;
  }
  size_t want = ctx->n_tickets ? ctx->n_tickets * 2 : 8192;  // geometric growth: at most a handful of buffers per context
  while (want < n) want *= 2;
  int* p = nullptr;
  hipError_t e = hipMalloc((void**)&p, want * sizeof(int));
  if (e == hipSuccess) e = hipMemsetAsync(p, 0, want * sizeof(int), st);
  if (e != hipSuccess) {
    if (p) (void)hipFree(p);
    hip_fail(ctx, e, "ensure_tickets");
    return nullptr;
  }
  if (ctx->tickets) ctx->old_tickets[ctx->n_old_tickets++] = ctx->tickets;  // launches in flight may still use it
  ctx->tickets = p;
  ctx->n_tickets = want;
  return p;
}

void tickets_destroy(assx_ctx* ctx) {
  if (ctx->tickets) (void)hipFree(ctx->tickets);
  for (int i = 0; i < ctx->n_old_tickets; ++i) (void)hipFree(ctx->old_tickets[i]);
  ctx->tickets = nullptr;
  ctx->n_tickets = 0;
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
