// What does one workgroup barrier cost on gfx950?  256 workgroups (one per CU) of NW waves loop over
// { CH dependent-free f64 FMAs; [LDS round trip]; s_barrier } -- the skeleton of cov_wide_kernel's item loop.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/barrier_cost_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NW, int NFMA, bool BAR, bool LDS>
__global__ void __launch_bounds__(64 * NW) probe(double* out, int iters, double a) {
  extern __shared__ double sm[];
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < NFMA / 16; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, 1.0);
    if (LDS) {
      sm[threadIdx.x] = acc[0];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (BAR) asm volatile("s_barrier" ::: "memory");
    if (LDS) acc[1] += sm[(threadIdx.x + 64) % (64 * NW)];
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NW, int NFMA, bool BAR, bool LDS>
void run(const char* name, size_t lds_bytes) {
  double* out;
  hipMalloc(&out, 256 * 64 * NW * sizeof(double));
  const int iters = 2000;
  auto k = probe<NW, NFMA, BAR, LDS>;
  if (lds_bytes > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * NW), lds_bytes, 0, out, iters, 0.999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s waves %d  fma/iter %3d  lds %6zu B : %7.1f ns per iteration\n", name, NW, NFMA, lds_bytes, ms * 1e6 / iters);
  fflush(stdout);
  hipFree(out);
}

int main() {
  run<8, 64, false, false>("no barrier", 8192);
  run<8, 64, true, false>("s_barrier", 8192);
  run<8, 64, true, true>("LDS write + wait + s_barrier + LDS read", 8192);
  run<8, 64, true, false>("s_barrier, 84 KB LDS", 84 * 1024);
  run<8, 256, false, false>("no barrier", 8192);
  run<8, 256, true, false>("s_barrier", 8192);
  run<4, 64, true, false>("s_barrier", 8192);
  run<4, 256, true, false>("s_barrier", 8192);
  run<16, 64, true, false>("s_barrier", 8192);
  run<1, 64, false, false>("one wave, no barrier", 8192);
  return 0;
}
