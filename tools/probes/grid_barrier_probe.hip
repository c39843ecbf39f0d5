// What does a GRID-WIDE barrier cost inside one kernel on MI355X (8 XCDs, 256 CUs)?  The alternative to a dependent launch
// (2.6-2.9 us, launch_floor_probe.hip) for a persistent multi-pass kernel.  All workgroups co-resident (grid <= 2 x 256).
//   variant 0: one agent-scope counter, every workgroup arrives (atomic add) and spins on it (agent-scope loads)
//   variant 1: one counter per XCD-sized group of workgroups (blockIdx % 8) + a top counter: 2-level
//   hipcc --offload-arch=gfx950 -O3 grid_barrier_probe.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int VARIANT>
__global__ void __launch_bounds__(256) k(int* ctr, int nbar, int* sink) {
  const int nwg = gridDim.x;
  int phase = 0;
  for (int it = 0; it < nbar; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      ++phase;
      if (VARIANT == 0) {
        __hip_atomic_fetch_add(&ctr[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * nwg) __builtin_amdgcn_s_sleep(1);
      } else {
        const int x = blockIdx.x & 7, per = (nwg + 7 - x) / 8;  // workgroups dealt to this XCD (round robin)
        const int old = __hip_atomic_fetch_add(&ctr[16 + 16 * x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == phase * per - 1) __hip_atomic_fetch_add(&ctr[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * 8) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = phase;
}

int main() {
  int *ctr, *sink;
  hipMalloc(&ctr, 4096);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int variant = 0; variant < 2; ++variant)
    for (int nwg : {256, 512}) {
      for (int nbar : {1, 201}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          hipMemset(ctr, 0, 4096);
          hipDeviceSynchronize();
          hipEventRecord(e0, 0);
          if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 0, 0, ctr, nbar, sink);
          else hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 0, 0, ctr, nbar, sink);
          hipEventRecord(e1, 0);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        printf("variant %d  %3d workgroups  %3d barriers: %8.2f us\n", variant, nwg, nbar, best * 1e3f);
      }
    }
  printf("(per barrier = (time of 201 - time of 1) / 200)\n");
  return 0;
}
