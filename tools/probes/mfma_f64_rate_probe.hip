// Measured ceiling of v_mfma_f64_16x16x4_f64 (and v_mfma_f32_16x16x4_f32, v_mfma_f32_32x32x2_f32) on MI355X: the
// microarchitecture guide lists no FP64 matrix rate, so the roofline of the NMF matrix-core kernels is priced against
// THIS number (profiles/r02_mfma_rate_probe.txt), not against a data-sheet figure.
//   hipcc --offload-arch=gfx950 -O3 mfma_f64_rate_probe.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// CH = independent accumulator chains per wave (1 = every MFMA depends on the previous one), WPS = waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int CH>
__global__ void __launch_bounds__(256) k_f64(double* out, int iters, double a0) {
  v4d c[CH];
  for (int i = 0; i < CH; ++i) c[i] = v4d{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < CH; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
__global__ void __launch_bounds__(256) k_f32(float* out, int iters, float a0) {
  v4f c[CH];
  for (int i = 0; i < CH; ++i) c[i] = v4f{0, 0, 0, 0};
  float a = a0 + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < CH; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
__global__ void __launch_bounds__(256) k_f32_32(float* out, int iters, float a0) {
  v16f c[CH];
  for (int i = 0; i < CH; ++i)
    for (int j = 0; j < 16; ++j) c[i][j] = 0;
  float a = a0 + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < CH; ++i)
    for (int j = 0; j < 16; ++j) s += c[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  void* out;
  hipMalloc(&out, 256 * 4 * 64 * 8 * 8 * sizeof(double));
  const int iters = 4000, CUS = 256;
  printf("%-34s %4s %4s %10s %12s\n", "instruction", "CH", "WPS", "ms", "TFLOP/s");
#define RUN(NAME, KERN, T, CH, WPS, FLOPS_PER)                                                       \
  {                                                                                                  \
    const int wgs = CUS * (WPS);                                                                     \
    double ms = time_ms([&] { hipLaunchKernelGGL((KERN<CH>), dim3(wgs), dim3(256), 0, 0, (T*)out, iters, (T)1.0); }); \
    double fl = (double)wgs * 4 * iters * (CH) * (FLOPS_PER);                                        \
    printf("%-34s %4d %4d %10.3f %12.1f\n", NAME, CH, WPS, ms, fl / (ms * 1e-3) / 1e12);             \
  }
  RUN("v_mfma_f64_16x16x4_f64", k_f64, double, 1, 1, 2048.0)
  RUN("v_mfma_f64_16x16x4_f64", k_f64, double, 2, 1, 2048.0)
  RUN("v_mfma_f64_16x16x4_f64", k_f64, double, 4, 1, 2048.0)
  RUN("v_mfma_f64_16x16x4_f64", k_f64, double, 1, 2, 2048.0)
  RUN("v_mfma_f64_16x16x4_f64", k_f64, double, 1, 4, 2048.0)
  RUN("v_mfma_f64_16x16x4_f64", k_f64, double, 4, 2, 2048.0)
  RUN("v_mfma_f32_16x16x4_f32", k_f32, float, 1, 1, 2048.0)
  RUN("v_mfma_f32_16x16x4_f32", k_f32, float, 4, 1, 2048.0)
  RUN("v_mfma_f32_16x16x4_f32", k_f32, float, 4, 2, 2048.0)
  RUN("v_mfma_f32_32x32x2_f32", k_f32_32, float, 1, 1, 4096.0)
  RUN("v_mfma_f32_32x32x2_f32", k_f32_32, float, 4, 1, 4096.0)
  RUN("v_mfma_f32_32x32x2_f32", k_f32_32, float, 4, 2, 4096.0)
  return 0;
}
