// What does a dependent kernel boundary cost on this part?  A chain of N tiny kernels in one stream (each waits for
// the previous one, as the 7 launches of an ILRMA iteration do), timed with events: plain launches vs the same chain
// captured once into a hipGraph and replayed.
//   hipcc --offload-arch=gfx950 -O3 launch_floor_probe.hip -o /tmp/lf && /tmp/lf
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void tiny(double* p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0;
}
__global__ void touch(double* p, int n) {  // 256 workgroups read-modify-write 64 KB: a "finalize"-sized kernel
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}

static float chain_ms(hipStream_t st, double* p, int n, int kind, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&] {
    for (int i = 0; i < n; ++i) {
      if (kind == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, p);
      else hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, st, p, 65536);
    }
  };
  run();
  hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) run();
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

static float graph_ms(hipStream_t st, double* p, int n, int kind, int reps) {
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; ++i) {
    if (kind == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, st, p, 65536);
  }
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  double* p;
  hipMalloc(&p, 65536 * 8);
  hipMemset(p, 0, 65536 * 8);
  const int n = 200;
  for (int kind = 0; kind < 2; ++kind) {
    const float a = chain_ms(st, p, n, kind, 20), b = graph_ms(st, p, n, kind, 20);
    printf("%-28s plain stream %.2f us per kernel   hipGraph replay %.2f us per kernel\n",
           kind == 0 ? "1 workgroup, 1 store" : "256 workgroups, 64 KB r/w", a * 1e3 / n, b * 1e3 / n);
  }
  return 0;
}
