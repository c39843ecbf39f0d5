// Context management and version for the assx C-ABI (include/assx.h).
#include "assx_common.hpp"

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
  free(ctx);
  return 0;
}

const char* assx_last_error(const assx_ctx* ctx) { return ctx ? ctx->err : "ctx is NULL"; }

const char* assx_version(void) { return ASSX_VERSION_STRING; }

}  // extern "C"
