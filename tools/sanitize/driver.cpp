// Drives the host-side threaded code of libassx -- assx_upload / assx_download (pinned ring + thread pool + events) and the
// per-stream ticket slots of the context -- against tools/sanitize/hip_stub.cpp, for the sanitizer builds of
// tools/sanitize/run.sh.  Every transfer is verified element by element; a sanitizer report or a mismatch fails the run.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/assx.h"

namespace assx {
int ensure_tickets(assx_ctx* ctx, size_t n, hipStream_t st, int** out);
}

#define CHECK(cond)                                                                  \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);              \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

template <typename H, typename D>
static void round_trip(assx_ctx* ctx, size_t count, hipStream_t st, unsigned seed) {
  const int hd = sizeof(H) == 8 ? ASSX_F64 : ASSX_F32, dd = sizeof(D) == 8 ? ASSX_F64 : ASSX_F32;
  std::vector<H> src(count), back(count, (H)-1);
  for (size_t i = 0; i < count; ++i) src[i] = (H)((double)((i * 2654435761u + seed) % 100003) / 7.0 - 5000.0);
  void* dev = nullptr;
  CHECK(hipMalloc(&dev, count * sizeof(D) + 16) == hipSuccess);
  memset(dev, 0xAB, count * sizeof(D) + 16);
  CHECK(assx_upload(ctx, src.data(), hd, dev, dd, count, st) == 0);
  // no host wait in between: the download's copies are ordered behind the upload's tail on the device side only
  CHECK(assx_download(ctx, dev, dd, back.data(), hd, count, st) == 0);
  CHECK(hipStreamSynchronize(st) == hipSuccess);
  for (size_t i = 0; i < count; ++i) {
    const H want = (H)(D)src[i];
    CHECK(back[i] == want);
  }
  for (int i = 0; i < 16; ++i) CHECK(((unsigned char*)dev)[count * sizeof(D) + i] == 0xAB);  // nothing past the end
  CHECK(hipFree(dev) == hipSuccess);
}

static void transfers(int threads) {
  char buf[16];
  snprintf(buf, sizeof buf, "%d", threads);
  setenv("ASSX_XFER_THREADS", buf, 1);
  setenv("ASSX_XFER_CHUNK_MB", "1", 1);  // 131072 doubles per chunk: the sizes below go round the four-buffer ring twice
  assx_ctx* ctx = nullptr;
  CHECK(assx_ctx_create(0, &ctx) == 0 && ctx);
  hipStream_t user = nullptr;
  CHECK(hipStreamCreateWithFlags(&user, hipStreamNonBlocking) == hipSuccess);
  const size_t sizes[] = {0, 1, 63, 65535, 65536 + 7, 131072, 131072 * 3 + 5, 131072 * 9 + 11};
  unsigned seed = 1;
  for (hipStream_t st : {(hipStream_t) nullptr, user})
    for (size_t n : sizes) {
      round_trip<double, double>(ctx, n, st, seed++);
      round_trip<double, float>(ctx, n, st, seed++);
      round_trip<float, double>(ctx, n, st, seed++);
      round_trip<float, float>(ctx, n, st, seed++);
    }
  CHECK(assx_upload(ctx, nullptr, ASSX_F64, nullptr, ASSX_F64, 5, nullptr) == ASSX_E_NULL);
  CHECK(assx_upload(ctx, &seed, 7, &seed, ASSX_F64, 1, nullptr) == ASSX_E_ARG);
  CHECK(hipStreamDestroy(user) == hipSuccess);
  CHECK(assx_ctx_destroy(ctx) == 0);
  printf("transfers with %d host thread(s): ok\n", threads);
}

static void tickets() {
  assx_ctx* ctx = nullptr;
  CHECK(assx_ctx_create(0, &ctx) == 0 && ctx);
  static char handles[24];  // made-up stream handles: the slot table only compares them
  int* first[24];
  for (int round = 0; round < 3; ++round)
    for (int s = 0; s < 24; ++s) {  // 24 > the 16 slots: the least recently used slot changes hands behind a device wait
      int* p = nullptr;
      CHECK(assx::ensure_tickets(ctx, 100 + (size_t)s, (hipStream_t)&handles[s], &p) == 0 && p);
      for (int i = 0; i < 100 + s; ++i) CHECK(p[i] == 0);
      if (round == 0) first[s] = p;
      (void)first;
    }
  for (int grow = 0; grow < 40; ++grow) {  // outgrown buffers pile up on the retired list and are drained when it is full
    int* p = nullptr;
    const size_t n = (size_t)8192 << (grow % 6 + 1);
    CHECK(assx::ensure_tickets(ctx, n, (hipStream_t)&handles[grow % 24], &p) == 0 && p);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    for (size_t i = 0; i < n; i += 997) CHECK(p[i] == 0);
  }
  int *a = nullptr, *b = nullptr;  // two streams never share words
  CHECK(assx::ensure_tickets(ctx, 64, (hipStream_t)&handles[0], &a) == 0);
  CHECK(assx::ensure_tickets(ctx, 64, (hipStream_t)&handles[1], &b) == 0);
  CHECK(a && b && (a + 64 <= b || b + 64 <= a));
  CHECK(assx_ctx_destroy(ctx) == 0);
  printf("ticket slots: ok\n");
}

int main() {
  for (int t : {1, 3, 8}) transfers(t);
  tickets();
  printf("%s\n", assx_version());
  return 0;
}
