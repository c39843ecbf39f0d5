// Does the ACCESS PATTERN of the streaming kernels cap their HBM rate?  A wave of cov / basis / act_stream_kernel reads,
// per 64-frame block, one 1 KB piece (64 lanes x 16 B) from each of M = 4 channel planes that lie |plane| = F*T*16 B
// apart, then moves on by 1 KB in each plane: 2048 waves x 4 planes = 8192 concurrent 1 KB-granular streams.
// This probe reads the same bytes with no arithmetic in three shapes, one wave per workgroup, 2048 workgroups:
//   planes  : the kernels' shape (4 planes, 1 KB per plane per step)
//   planes2 : 2 KB contiguous per plane per step (two consecutive blocks requested together)
//   flat    : 4 KB contiguous per step (what a plain copy does)
//   hipcc --offload-arch=gfx950 -O3 stream_pattern_probe.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void __launch_bounds__(64) rd(const double2* __restrict__ x, size_t plane16, int steps, double* out) {
  // workgroup g owns `steps` consecutive 1 KB pieces of every plane (MODE 0/1) or steps*4 KB contiguous (MODE 2)
  const size_t lane = threadIdx.x;
  double s = 0;
  if (MODE == 2) {
    const double2* p = x + (size_t)blockIdx.x * steps * 256 + lane;
    for (int i = 0; i < steps; ++i) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double2 v = p[(size_t)i * 256 + m * 64];
        s += v.x + v.y;
      }
    }
  } else if (MODE == 0) {
    const double2* p = x + (size_t)blockIdx.x * steps * 64 + lane;
    for (int i = 0; i < steps; ++i) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double2 v = p[(size_t)m * plane16 + (size_t)i * 64];
        s += v.x + v.y;
      }
    }
  } else {
    const double2* p = x + (size_t)blockIdx.x * steps * 64 + lane;
    for (int i = 0; i < steps; i += 2) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double2 v = p[(size_t)m * plane16 + (size_t)i * 64];
        const double2 w = p[(size_t)m * plane16 + (size_t)i * 64 + 64];
        s += v.x + v.y + w.x + w.y;
      }
    }
  }
  if (s == 123.456) out[0] = s;
}

template <int MODE>
static double run(const double2* x, size_t bytes, double* out) {
  const int G = 2048;
  const size_t plane16 = bytes / 4 / 16;            // 16-byte elements per plane
  const int steps = (int)(plane16 / 64 / G) & ~1;   // 1 KB pieces per workgroup and plane
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(rd<MODE>, dim3(G), dim3(64), 0, 0, x, plane16, steps, out);
  hipEventRecord(e0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(rd<MODE>, dim3(G), dim3(64), 0, 0, x, plane16, steps, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)G * steps * 4096.0 * reps / (ms * 1e-3) / 1e12;
}

int main() {
  const size_t maxb = (size_t)2300 << 20;
  double2* x;
  double* out;
  hipMalloc(&x, maxb);
  hipMalloc(&out, 64);
  hipMemset(x, 0, maxb);
  printf("%10s %12s %12s %12s   (TB/s, 2048 single-wave workgroups)\n", "MB", "planes", "planes2", "flat");
  for (double mb : {268.7, 537.4, 1074.8, 2149.6}) {
    const size_t bytes = ((size_t)(mb * 1e6) / 65536) * 65536;
    printf("%10.1f %12.2f %12.2f %12.2f\n", mb, run<0>(x, bytes, out), run<1>(x, bytes, out), run<2>(x, bytes, out));
  }
  return 0;
}
