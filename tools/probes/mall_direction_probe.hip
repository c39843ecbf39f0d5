// Does the 256 MiB Infinity Cache reward re-reading a just-streamed buffer in the OPPOSITE direction?
// A streaming iteration of ILRMA reads X (268.7 MB at config 4) three times; if the memory-side cache replaces
// LRU-like, re-reading in the same order thrashes it while alternating directions hits on what is still resident.
//   hipcc --offload-arch=gfx950 -O3 mall_direction_probe.hip -o /tmp/mall && /tmp/mall
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

// every workgroup streams its own contiguous chunk, forward (dir = 0) or backward (dir = 1)
__global__ void __launch_bounds__(256) stream_sum(const double2* __restrict__ x, size_t n16, double* out, int dir) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
  double s = 0;
  const size_t steps = (hi > lo) ? (hi - lo + 255) / 256 : 0;
  for (size_t i = 0; i < steps; ++i) {
    const size_t j = dir ? steps - 1 - i : i;
    const size_t idx = lo + j * 256 + threadIdx.x;
    if (idx < hi) {
      const double2 v = x[idx];
      s += v.x + v.y;
    }
  }
  if (s == 123.456) out[0] = s;
}

static double run(const double2* x, size_t bytes, double* out, int passes, bool alternate) {
  const size_t n16 = bytes / 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int p = 0; p < 4; ++p) hipLaunchKernelGGL(stream_sum, dim3(2048), dim3(256), 0, 0, x, n16, out, alternate ? (p & 1) : 0);
  hipEventRecord(e0);
  for (int p = 0; p < passes; ++p) hipLaunchKernelGGL(stream_sum, dim3(2048), dim3(256), 0, 0, x, n16, out, alternate ? (p & 1) : 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)bytes * passes / (ms * 1e-3) / 1e12;
}

int main() {
  const size_t maxb = (size_t)1200 << 20;
  double2* x;
  double* out;
  hipMalloc(&x, maxb);
  hipMalloc(&out, 64);
  hipMemset(x, 0, maxb);
  printf("%10s %14s %14s\n", "MB", "same dir TB/s", "alternate TB/s");
  for (double mb : {64.0, 128.0, 192.0, 240.0, 268.7, 300.0, 350.0, 400.0, 537.4, 1074.8}) {
    const size_t bytes = ((size_t)(mb * 1e6) / 4096) * 4096;
    const double a = run(x, bytes, out, 30, false), b = run(x, bytes, out, 30, true);
    printf("%10.1f %14.2f %14.2f\n", mb, a, b);
  }
  return 0;
}
