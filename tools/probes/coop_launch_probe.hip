// What a cooperative launch with one grid-wide synchronisation costs against two dependent plain launches
// (MI355X; round 5: would ONE launch for both halves of an NMF update pay?).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/coop_launch_probe.hip -o /tmp/coop_probe && /tmp/coop_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void __launch_bounds__(256) k_plain(double* p, int phase) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  p[i] = p[i] * 1.0000001 + phase;
}
__global__ void __launch_bounds__(256) k_coop(double* p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  p[i] = p[i] * 1.0000001 + 1;
  cg::this_grid().sync();
  const int j = (i + 256 * 17) % (gridDim.x * blockDim.x);  // read what another workgroup wrote
  p[i] = p[j] * 1.0000001 + 2;
}

int main() {
  for (int wgs : {128, 256, 512}) {
    double* p;
    hipMalloc(&p, (size_t)wgs * 256 * sizeof(double));
    hipMemset(p, 0, (size_t)wgs * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n = 500;
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(k_plain, dim3(wgs), dim3(256), 0, 0, p, 1);
        hipLaunchKernelGGL(k_plain, dim3(wgs), dim3(256), 0, 0, p, 2);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double plain = ms * 1e3 / n;
    void* args[] = {&p};
    hipError_t err = hipSuccess;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < n; ++i) err = hipLaunchCooperativeKernel((void*)k_coop, dim3(wgs), dim3(256), args, 0, 0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%4d workgroups: two dependent plain launches %.2f us | one cooperative launch with a grid sync %.2f us (%s)\n", wgs, plain,
           ms * 1e3 / n, hipGetErrorString(err));
    hipFree(p);
  }
  return 0;
}
