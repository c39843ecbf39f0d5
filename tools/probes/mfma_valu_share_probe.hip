// Do v_mfma_f64_16x16x4 and v_fma_f64 run on the SAME arithmetic units of a gfx950 SIMD?  (The data sheet gives both
// 78.6 TFLOP/s; the NMF kernels' phases "add up" instead of overlapping, profiles/r03_nmf_parts.txt.)
// One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) run a matrix-core chain, waves 4-7 (the second wave of each
// SIMD) a vector chain -- alone (the other half exits at once) and together.  Independent units: together == max(alone);
// shared units: together == sum.  The same with f32 vector work beside the f64 matrix chain, and f32 matrix beside f64 vector.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_share_probe.hip -o /tmp/share_probe && /tmp/share_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// mode bit 0: matrix half runs, bit 1: vector half runs.  MK: 0 = mfma f64 16x16x4 (4 chains), 1 = mfma f32 32x32x2 (4 chains), 2 = mfma f64, one dependent chain.  VK: 0 = fma f64, 1 = fma f32
template <int MK, int VK>
__global__ void __launch_bounds__(512) k(double* out, int mode, int im, int iv) {
  const int half = threadIdx.x >> 8;
  double s = 0;
  if (half == 0) {
    if (!(mode & 1)) return;
    if (MK == 0) {
      v4d c[4];
      for (int i = 0; i < 4; ++i) c[i] = v4d{0, 0, 0, 0};
      const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
      for (int it = 0; it < im; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
      for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else if (MK == 2) {  // ONE dependent chain: the pipe idles ~100 of every 184 cycles
      v4d c = v4d{0, 0, 0, 0};
      const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
      for (int it = 0; it < im; ++it) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
      s += c[0] + c[1] + c[2] + c[3];
    } else {
      v16f c[4];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) c[i][j] = 0;
      const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
      for (int it = 0; it < im; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += c[i][j];
    }
  } else {
    if (!(mode & 2)) return;
    if (VK == 0) {
      double c[8];
      for (int i = 0; i < 8; ++i) c[i] = i + threadIdx.x;
      const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
      for (int it = 0; it < iv; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_fma(c[i], a, b);
      for (int i = 0; i < 8; ++i) s += c[i];
    } else {
      float c[8];
      for (int i = 0; i < 8; ++i) c[i] = i + threadIdx.x;
      const float a = 1.0f + threadIdx.x * 1e-6f, b = 1e-6f;
      for (int it = 0; it < iv; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_fmaf(c[i], a, b);
      for (int i = 0; i < 8; ++i) s += c[i];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MK, int VK>
static float run(double* out, int mode, int im, int iv) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MK, VK>), dim3(256), dim3(512), 0, 0, out, mode, im, iv);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MK, VK>), dim3(256), dim3(512), 0, 0, out, mode, im, iv);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* out;
  hipMalloc(&out, (size_t)256 * 512 * sizeof(double));
  printf("one workgroup of 8 waves per CU: waves 0-3 matrix chain (4 independent accumulators), waves 4-7 vector chain (8 accumulators)\n");
  printf("%-44s %10s %10s %10s %8s\n", "matrix | vector", "matrix ms", "vector ms", "both ms", "both/max");
#define CASE(NAME, MK, VK, IM, IV)                                  \
  {                                                                 \
    const float a = run<MK, VK>(out, 1, IM, IV), b = run<MK, VK>(out, 2, IM, IV), c = run<MK, VK>(out, 3, IM, IV); \
    printf("%-44s %10.3f %10.3f %10.3f %8.2f\n", NAME, a, b, c, c / (a > b ? a : b));                            \
  }
  CASE("v_mfma_f64_16x16x4 | v_fma_f64", 0, 0, 20000, 150000)
  CASE("v_mfma_f64_16x16x4 | v_fma_f32", 0, 1, 20000, 300000)
  CASE("v_mfma_f64 (one dependent chain) | v_fma_f64", 2, 0, 40000, 150000)
  CASE("v_mfma_f64 (one dependent chain) | v_fma_f32", 2, 1, 40000, 300000)
  CASE("v_mfma_f32_32x32x2 | v_fma_f64", 1, 0, 20000, 150000)
  CASE("v_mfma_f32_32x32x2 | v_fma_f32", 1, 1, 20000, 300000)
  return 0;
}
