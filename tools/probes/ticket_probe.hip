// Probe: "last workgroup done" across XCDs -- plain stores + __threadfence + agent-scope ticket + agent-scope loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef STAGGER
#define STAGGER 1
#endif
template <int MODE>
__global__ void __launch_bounds__(64) k(double* rec, int* ticket, double* out, int per, int* err) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int grp = g / per;
  // a little work so that workgroups finish at different times
  double s = 0;
  for (int i = 0; i < STAGGER * (g % 7) * 200 + 50; ++i) s += sin((double)(i + lane));
  if (lane < 32) {
    if (MODE == 0) rec[(size_t)g * 32 + lane] = (double)(g * 100 + lane) + (s * 0.0);
    else __hip_atomic_store(&rec[(size_t)g * 32 + lane], (double)(g * 100 + lane) + (s * 0.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // narrow protocol: write-through stores, no L2-wide fence
  else __threadfence();
  int last = 0;
  if (lane == 0) {
    last = atomicAdd(&ticket[grp], 1) == per - 1;
    if (last) ticket[grp] = 0;
  }
  last = __builtin_amdgcn_readfirstlane(last);
  if (last) {
    if (MODE != 2) __threadfence();
    if (lane < 32) {
      double sum = 0;
      int bad = 0;
      for (int gg = grp * per; gg < (grp + 1) * per; ++gg) {
        const double v = __hip_atomic_load(&rec[(size_t)gg * 32 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != (double)(gg * 100 + lane)) bad = 1;
        sum += v;
      }
      out[grp * 32 + lane] = sum;
      if (bad) atomicAdd(err, 1);
    }
  }
}
int main() {
  const int per = 3, groups = 680, G = per * groups;
  double *rec, *out; int *ticket, *err;
  hipMalloc(&rec, (size_t)G * 32 * 8); hipMalloc(&out, groups * 32 * 8); hipMalloc(&ticket, groups * 4); hipMalloc(&err, 4);
  for (int mode = 0; mode < 3; ++mode) {
    int total = 0;
    for (int rep = 0; rep < 200; ++rep) {
      hipMemset(rec, 0xff, (size_t)G * 32 * 8); hipMemset(ticket, 0, groups * 4); hipMemset(err, 0, 4);
      if (mode == 0) k<0><<<G, 64>>>(rec, ticket, out, per, err);
      else if (mode == 1) k<1><<<G, 64>>>(rec, ticket, out, per, err);
      else k<2><<<G, 64>>>(rec, ticket, out, per, err);
      int h; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost); total += h;
    }
    printf("ticket_probe mode %d (%s stores): %d stale lanes in 200 launches\n", mode, mode == 0 ? "plain + fences" : mode == 1 ? "agent-scope atomic + fences" : "agent-scope atomic, no fences", total);
  }
  return 0;
}
