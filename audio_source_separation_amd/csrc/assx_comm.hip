// Multi-GPU edges behind the C-ABI (include/assx.h: assx_comm_*, assx_scatter, assx_gather; SURVEY.md section 8b lists them
// in the boundary's minimum set).  The hot path shards by utterance and needs no data-path collective (ref
// src/bss/ilrma.py:203-273: every __call__ owns all of its state): what crosses GPUs is the scatter of the mixtures from a
// root and the gather of the separated outputs, point to point -- root <-> 7 peers are 7 concurrent xGMI links, so each
// edge is ONE grouped batch of ncclSend / ncclRecv on contiguous row blocks of the root's array, the static block
// partition of audio_source_separation_amd/distributed.py (shard_range).  The Python classes keep using torch.distributed
// for the same edges; these entry points are the route of a host that has no torch (C, Go, Java over FFI).
//
// RCCL is loaded on first use (dlopen "librccl.so"): libassx.so itself does not link it, so the library loads -- and every
// single-GPU entry point works -- on a machine without RCCL; assx_comm_* then fail with ASSX_E_UNSUPPORTED and a message.
// No kernels in this file.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "assx_common.hpp"

struct assx_comm {
  assx_ctx* ctx;
  ncclComm_t nccl;
  int world, rank;
};

namespace assx {
namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char why[256] = {0};
};

Rccl* rccl() {
  static Rccl r;
  static const bool ok = [] {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
    if (!r.so) {
      snprintf(r.why, sizeof(r.why), "librccl.so could not be loaded (%s)", dlerror());
      return false;
    }
#define ASSX_SYM(field, sym)                                                        \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.so, sym));                   \
  if (!r.field) {                                                                    \
    snprintf(r.why, sizeof(r.why), "librccl.so has no symbol %s", sym);              \
    return false;                                                                    \
  }
    ASSX_SYM(GetUniqueId, "ncclGetUniqueId")
    ASSX_SYM(CommInitRank, "ncclCommInitRank")
    ASSX_SYM(CommDestroy, "ncclCommDestroy")
    ASSX_SYM(Send, "ncclSend")
    ASSX_SYM(Recv, "ncclRecv")
    ASSX_SYM(GroupStart, "ncclGroupStart")
    ASSX_SYM(GroupEnd, "ncclGroupEnd")
    ASSX_SYM(GetErrorString, "ncclGetErrorString")
#undef ASSX_SYM
    return true;
  }();
  return ok ? &r : nullptr;
}

int nccl_fail(assx_ctx* ctx, ncclResult_t e, const char* where) {
  Rccl* r = rccl();
  return fail(ctx, 1000 + (int)e, "%s: RCCL error %d (%s)", where, (int)e, r ? r->GetErrorString(e) : "?");
}

}  // namespace
}  // namespace assx

using namespace assx;

#define ASSX_NCCL(ctx, call, where)                                  \
  do {                                                               \
    ncclResult_t e__ = (call);                                       \
    if (e__ != ncclSuccess) return nccl_fail((ctx), e__, (where));   \
  } while (0)
// inside ncclGroupStart / ncclGroupEnd: the group is closed before the error is returned
#define ASSX_NCCL_G(ctx, call, where)                                \
  do {                                                               \
    ncclResult_t e__ = (call);                                       \
    if (e__ != ncclSuccess) {                                        \
      (void)r->GroupEnd();                                           \
      return nccl_fail((ctx), e__, (where));                         \
    }                                                                \
  } while (0)

extern "C" {

void assx_shard_range(size_t n_items, int world, int rank, size_t* lo, size_t* hi) {
  // distributed.py: shard_range -- contiguous blocks, sizes differ by at most one, the first n % world ranks hold one more
  if (world < 1 || rank < 0 || rank >= world) {
    if (lo) *lo = 0;
    if (hi) *hi = 0;
    return;
  }
  const size_t base = n_items / (size_t)world, extra = n_items % (size_t)world;
  const size_t start = (size_t)rank * base + ((size_t)rank < extra ? (size_t)rank : extra);
  if (lo) *lo = start;
  if (hi) *hi = start + base + ((size_t)rank < extra ? 1 : 0);
}

int assx_comm_unique_id(void* id) {
  if (!id) return ASSX_E_NULL;
  Rccl* r = rccl();
  if (!r) return ASSX_E_UNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == ASSX_COMM_ID_BYTES, "ASSX_COMM_ID_BYTES must be RCCL's unique-id size");
  ncclUniqueId u;
  const ncclResult_t e = r->GetUniqueId(&u);
  if (e != ncclSuccess) return 1000 + (int)e;
  memcpy(id, &u, sizeof(u));
  return 0;
}

int assx_comm_init(assx_ctx* ctx, int world, int rank, const void* id, assx_comm** out) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, out && id, ASSX_E_NULL, "assx_comm_init: NULL argument");
  *out = nullptr;
  ASSX_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, ASSX_E_ARG, "assx_comm_init: bad world / rank %d / %d", world, rank);
  Rccl* r = rccl();
  if (!r) return fail(ctx, ASSX_E_UNSUPPORTED, "assx_comm_init: RCCL is not available on this machine (librccl.so could not be loaded)");
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t c = nullptr;
  ASSX_NCCL(ctx, r->CommInitRank(&c, world, u, rank), "ncclCommInitRank");  // on the current device = the context's
  assx_comm* cm = (assx_comm*)calloc(1, sizeof(assx_comm));
  if (!cm) {
    (void)r->CommDestroy(c);
    return fail(ctx, ASSX_E_ARG, "assx_comm_init: out of host memory");
  }
  cm->ctx = ctx;
  cm->nccl = c;
  cm->world = world;
  cm->rank = rank;
  *out = cm;
  return 0;
}

int assx_comm_destroy(assx_comm* comm) {
  if (!comm) return ASSX_E_NULL;
  Rccl* r = rccl();
  if (r && comm->nccl) (void)r->CommDestroy(comm->nccl);
  free(comm);
  return 0;
}

int assx_scatter(assx_comm* comm, int root, const void* all, void* local, size_t n_items, size_t item_bytes, void* stream) {
  if (!comm) return ASSX_E_NULL;
  assx_ctx* ctx = comm->ctx;
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, root >= 0 && root < comm->world, ASSX_E_ARG, "assx_scatter: root %d out of range", root);
  Rccl* r = rccl();
  if (!r) return fail(ctx, ASSX_E_UNSUPPORTED, "assx_scatter: RCCL is not available");
  hipStream_t st = (hipStream_t)stream;
  size_t lo, hi;
  assx_shard_range(n_items, comm->world, comm->rank, &lo, &hi);
  ASSX_REQUIRE(ctx, hi == lo || local, ASSX_E_NULL, "assx_scatter: NULL receive block");
  if (comm->rank != root) {
    if (hi > lo) {
      ASSX_NCCL(ctx, r->GroupStart(), "ncclGroupStart");
      ASSX_NCCL_G(ctx, r->Recv(local, (hi - lo) * item_bytes, ncclChar, root, comm->nccl, st), "ncclRecv");
      ASSX_NCCL(ctx, r->GroupEnd(), "ncclGroupEnd");
    }
    return 0;
  }
  ASSX_REQUIRE(ctx, all || n_items == 0, ASSX_E_NULL, "assx_scatter: the root needs the whole array");
  // one grouped batch: a send per peer with a non-empty block, straight from the root's array (no staging copy); the
  // root's own block travels the same way (a send to / receive from itself inside the group) unless it is already in place
  const char* mine = (const char*)all + lo * item_bytes;
  ASSX_NCCL(ctx, r->GroupStart(), "ncclGroupStart");
  for (int p = 0; p < comm->world; ++p) {
    size_t a, b;
    assx_shard_range(n_items, comm->world, p, &a, &b);
    if (b == a) continue;
    if (p == root && (const void*)mine == (const void*)local) continue;
    ASSX_NCCL_G(ctx, r->Send((const char*)all + a * item_bytes, (b - a) * item_bytes, ncclChar, p, comm->nccl, st), "ncclSend");
    if (p == root) ASSX_NCCL_G(ctx, r->Recv(local, (b - a) * item_bytes, ncclChar, root, comm->nccl, st), "ncclRecv");
  }
  ASSX_NCCL(ctx, r->GroupEnd(), "ncclGroupEnd");
  return 0;
}

int assx_gather(assx_comm* comm, int root, const void* local, void* all, size_t n_items, size_t item_bytes, void* stream) {
  if (!comm) return ASSX_E_NULL;
  assx_ctx* ctx = comm->ctx;
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, root >= 0 && root < comm->world, ASSX_E_ARG, "assx_gather: root %d out of range", root);
  Rccl* r = rccl();
  if (!r) return fail(ctx, ASSX_E_UNSUPPORTED, "assx_gather: RCCL is not available");
  hipStream_t st = (hipStream_t)stream;
  size_t lo, hi;
  assx_shard_range(n_items, comm->world, comm->rank, &lo, &hi);
  ASSX_REQUIRE(ctx, hi == lo || local, ASSX_E_NULL, "assx_gather: NULL send block");
  if (comm->rank != root) {
    if (hi > lo) {
      ASSX_NCCL(ctx, r->GroupStart(), "ncclGroupStart");
      ASSX_NCCL_G(ctx, r->Send(local, (hi - lo) * item_bytes, ncclChar, root, comm->nccl, st), "ncclSend");
      ASSX_NCCL(ctx, r->GroupEnd(), "ncclGroupEnd");
    }
    return 0;
  }
  ASSX_REQUIRE(ctx, all || n_items == 0, ASSX_E_NULL, "assx_gather: the root needs the destination array");
  char* mine = (char*)all + lo * item_bytes;
  ASSX_NCCL(ctx, r->GroupStart(), "ncclGroupStart");
  for (int p = 0; p < comm->world; ++p) {
    size_t a, b;
    assx_shard_range(n_items, comm->world, p, &a, &b);
    if (b == a) continue;
    if (p == root && (const void*)mine == local) continue;
    if (p == root) ASSX_NCCL_G(ctx, r->Send(local, (b - a) * item_bytes, ncclChar, root, comm->nccl, st), "ncclSend");
    ASSX_NCCL_G(ctx, r->Recv((char*)all + a * item_bytes, (b - a) * item_bytes, ncclChar, p, comm->nccl, st), "ncclRecv");
  }
  ASSX_NCCL(ctx, r->GroupEnd(), "ncclGroupEnd");
  return 0;
}

}  // extern "C"
