// What the shader clock does under load, and whether the f64 ceilings measured in round 2 (45.6 TFLOP/s for
// v_mfma_f64_16x16x4, ~60 T for v_fma_f64) are ISSUE limits or CLOCK limits: every kernel stamps s_memtime (counts at
// the shader clock) and s_memrealtime (constant 100 MHz) at entry and exit of one wave per workgroup; the ratio is the
// clock the CU actually ran at during the kernel, and instructions / cycle follows.
//   hipcc --offload-arch=gfx950 -O3 clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

struct Stamp {
  unsigned long long c0, c1, r0, r1;
};
#define STAMP_IN()                                                  \
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define STAMP_OUT()                                                                          \
  if ((threadIdx.x & 63) == 0) {                                                              \
    Stamp s{c0, __builtin_amdgcn_s_memtime(), r0, __builtin_amdgcn_s_memrealtime()};          \
    st[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;                                \
  }

template <int CH>
__global__ void __launch_bounds__(256) k_mfma64(double* out, Stamp* st, int iters) {
  STAMP_IN();
  v4d c[CH];
  for (int i = 0; i < CH; ++i) c[i] = v4d{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < CH; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  STAMP_OUT();
}
template <int CH>
__global__ void __launch_bounds__(256) k_mfma32(double* out, Stamp* st, int iters) {
  STAMP_IN();
  v16f c[CH];
  for (int i = 0; i < CH; ++i)
    for (int j = 0; j < 16; ++j) c[i][j] = 0;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < CH; ++i)
    for (int j = 0; j < 16; ++j) s += c[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  STAMP_OUT();
}
template <typename R, int CH>
__global__ void __launch_bounds__(256) k_fma(double* out, Stamp* st, int iters) {
  STAMP_IN();
  R c[CH];
  for (int i = 0; i < CH; ++i) c[i] = (R)(i + threadIdx.x);
  const R a = (R)(1.0 + threadIdx.x * 1e-9), b = (R)1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if constexpr (sizeof(R) == 8) c[i] = __builtin_fma(c[i], a, b);
      else c[i] = __builtin_fmaf(c[i], a, b);
    }
  }
  R s = 0;
  for (int i = 0; i < CH; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (double)s;
  STAMP_OUT();
}
typedef float pk2_t __attribute__((ext_vector_type(2)));
template <int CH>
__global__ void __launch_bounds__(256) k_pkfma(double* out, Stamp* st, int iters) {  // v_pk_fma_f32: two multiply-adds per lane
  STAMP_IN();
  pk2_t c[CH];
  for (int i = 0; i < CH; ++i) c[i] = pk2_t{(float)(i + threadIdx.x), (float)(i + 1)};
  const pk2_t a = {1.0f + threadIdx.x * 1e-9f, 1.0f}, b = {1e-9f, 2e-9f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(c[i]) : "v"(a), "v"(b));
  }
  float s = 0;
  for (int i = 0; i < CH; ++i) s += c[i].x + c[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (double)s;
  STAMP_OUT();
}
__global__ void __launch_bounds__(256) k_sleep(double* out, Stamp* st, int iters) {
  STAMP_IN();
  for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(64);
  out[blockIdx.x * blockDim.x + threadIdx.x] = 0;
  STAMP_OUT();
}

int main() {
  const int CUS = 256;
  double* out;
  Stamp* st;
  hipMalloc(&out, (size_t)CUS * 8 * 256 * sizeof(double));
  hipMalloc(&st, (size_t)CUS * 8 * 4 * sizeof(Stamp));
  std::vector<Stamp> h((size_t)CUS * 8 * 4);
  printf("%-28s %3s %3s %9s %9s %8s %10s %12s\n", "kernel", "CH", "WPS", "ms", "clock GHz", "cyc/inst", "T inst/s", "TFLOP/s");
  auto report = [&](const char* name, int ch, int wps, int wgs, double ms, double insts_per_wave, double flops_per_inst) {
    hipMemcpy(h.data(), st, (size_t)wgs * 4 * sizeof(Stamp), hipMemcpyDeviceToHost);
    double clk = 0, cyc = 0;
    for (int i = 0; i < wgs * 4; ++i) {
      clk += (double)(h[i].c1 - h[i].c0) / ((double)(h[i].r1 - h[i].r0) * 10.0);  // cycles per ns = GHz
      cyc += (double)(h[i].c1 - h[i].c0);
    }
    clk /= wgs * 4;
    cyc /= wgs * 4;
    const double total = (double)wgs * 4 * insts_per_wave;
    // cycles per instruction PER SIMD: a SIMD hosts wps waves, each issuing insts_per_wave in `cyc` cycles
    printf("%-28s %3d %3d %9.3f %9.3f %8.1f %10.2f %12.1f\n", name, ch, wps, ms, clk, cyc / (insts_per_wave * wps),
           total / (ms * 1e-3) / 1e12, total * flops_per_inst / (ms * 1e-3) / 1e12);
  };
#define RUN(NAME, KERN, CH, WPS, ITERS, FLOPS_PER)                                                   \
  {                                                                                                  \
    const int wgs = CUS * (WPS);                                                                     \
    hipEvent_t e0, e1;                                                                               \
    hipEventCreate(&e0);                                                                             \
    hipEventCreate(&e1);                                                                             \
    hipLaunchKernelGGL(KERN, dim3(wgs), dim3(256), 0, 0, out, st, ITERS);                            \
    hipDeviceSynchronize();                                                                          \
    hipEventRecord(e0);                                                                              \
    hipLaunchKernelGGL(KERN, dim3(wgs), dim3(256), 0, 0, out, st, ITERS);                            \
    hipEventRecord(e1);                                                                              \
    hipEventSynchronize(e1);                                                                         \
    float ms;                                                                                        \
    hipEventElapsedTime(&ms, e0, e1);                                                                \
    report(NAME, CH, WPS, wgs, ms, (double)(ITERS) * (CH), FLOPS_PER);                               \
  }
  RUN("s_sleep (idle chip)", k_sleep, 1, 1, 20000, 0.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<1>), 1, 1, 40000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<1>), 1, 2, 40000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<1>), 1, 3, 40000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<1>), 1, 4, 40000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<1>), 1, 6, 40000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<1>), 1, 8, 40000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<2>), 2, 1, 20000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<2>), 2, 2, 20000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<4>), 4, 1, 20000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<4>), 4, 2, 20000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<2>), 2, 3, 20000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<4>), 4, 3, 20000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<2>), 2, 4, 20000, 2048.0)
  RUN("v_mfma_f64_16x16x4", (k_mfma64<4>), 4, 4, 20000, 2048.0)
  RUN("v_fma_f64", (k_fma<double, 1>), 1, 1, 200000, 128.0)
  RUN("v_fma_f64", (k_fma<double, 8>), 8, 1, 100000, 128.0)
  RUN("v_fma_f64", (k_fma<double, 8>), 8, 2, 100000, 128.0)
  RUN("v_fma_f64", (k_fma<double, 8>), 8, 4, 100000, 128.0)
  RUN("v_fma_f32", (k_fma<float, 8>), 8, 2, 100000, 128.0)
  RUN("v_fma_f32", (k_fma<float, 8>), 8, 4, 100000, 128.0)
  RUN("v_pk_fma_f32", (k_pkfma<8>), 8, 1, 100000, 256.0)
  RUN("v_pk_fma_f32", (k_pkfma<8>), 8, 2, 100000, 256.0)
  RUN("v_pk_fma_f32", (k_pkfma<8>), 8, 4, 100000, 256.0)
  RUN("v_fma_f32", (k_fma<float, 8>), 8, 1, 100000, 128.0)
  RUN("v_mfma_f32_32x32x2", (k_mfma32<4>), 4, 2, 20000, 4096.0)
  RUN("s_sleep (idle chip)", k_sleep, 1, 1, 20000, 0.0)
  return 0;
}
