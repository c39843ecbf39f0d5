// Measured issue rate of the f64 vector instructions the streaming kernels are made of (v_fma_f64, v_mul_f64,
// v_add_f64, v_rcp_f64) against v_fma_f32, on MI355X.  The microarchitecture guide quotes the f32 VALU rate only
// (a wave64 instruction issues in 2 cycles); every "instruction-issue bound" statement in DESIGN.md about an f64 kernel
// is priced with THIS measurement (profiles/r02_valu_f64_rate_probe.txt).
//   hipcc --offload-arch=gfx950 -O3 valu_f64_rate_probe.hip -o /tmp/valu_rate && /tmp/valu_rate
// CH = independent dependency chains per lane, WPS = waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>

enum { OP_FMA = 0, OP_MUL = 1, OP_ADD = 2, OP_RCP = 3 };

template <typename R, int CH, int OP>
__global__ void __launch_bounds__(256) k(R* out, int iters, R a0) {
  R c[CH];
  for (int i = 0; i < CH; ++i) c[i] = (R)i + (R)threadIdx.x * (R)1e-3;
  const R a = a0 + (R)threadIdx.x * (R)1e-6, b = (R)1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (OP == OP_FMA) {
        if constexpr (sizeof(R) == 8) c[i] = __builtin_fma(c[i], a, b);
        else c[i] = __builtin_fmaf(c[i], a, b);
      }
      else if (OP == OP_MUL) c[i] = c[i] * a;
      else if (OP == OP_ADD) c[i] = c[i] + b;
      else {
        if constexpr (sizeof(R) == 8) c[i] = __builtin_amdgcn_rcp(c[i]);
        else c[i] = __builtin_amdgcn_rcpf(c[i]);
      }
    }
  }
  R s = 0;
  for (int i = 0; i < CH; ++i) s += c[i];
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

template <typename R, int CH, int OP>
static void run(const char* name, int wps, void* out) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, iters = 1 << 17;
  const int blocks = cus * wps;  // 4 waves per block -> wps waves per SIMD
  const double ms = time_ms([&] { hipLaunchKernelGGL((k<R, CH, OP>), dim3(blocks), dim3(256), 0, 0, (R*)out, iters, (R)1.0000001); });
  const double wave_instr = (double)blocks * 4 * iters * CH;
  const double clk = p.clockRate * 1e3;  // Hz
  const double per_simd = wave_instr / (cus * 4.0);
  printf("%-14s CH=%d WPS=%d : %8.3f ms  %6.2f cycles / wave-instruction / SIMD   %7.2f T lane-op/s\n", name, CH, wps, ms,
         ms * 1e-3 * clk / per_simd, wave_instr * 64 / (ms * 1e-3) / 1e12);
}

int main() {
  void* out;
  hipMalloc(&out, 256 * 256 * 8 * 8 * 8);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<double, 8, OP_FMA>), dim3(2048), dim3(256), 0, 0, (double*)out, 1 << 15, 1.0);  // clock ramp
  hipDeviceSynchronize();
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s  CUs %d  clock %.0f MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1e3);
  for (int wps : {1, 2, 4}) {
    run<double, 1, OP_FMA>("v_fma_f64", wps, out);
    run<double, 4, OP_FMA>("v_fma_f64", wps, out);
    run<double, 8, OP_FMA>("v_fma_f64", wps, out);
  }
  run<double, 8, OP_MUL>("v_mul_f64", 2, out);
  run<double, 8, OP_ADD>("v_add_f64", 2, out);
  run<double, 8, OP_RCP>("v_rcp_f64", 2, out);
  run<float, 1, OP_FMA>("v_fma_f32", 2, out);
  run<float, 8, OP_FMA>("v_fma_f32", 2, out);
  run<float, 8, OP_FMA>("v_fma_f32", 4, out);
  run<float, 8, OP_RCP>("v_rcp_f32", 2, out);
  return 0;
}
