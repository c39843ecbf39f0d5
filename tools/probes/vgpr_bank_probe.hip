// Does the VGPR bank (register index mod 4) of the three 64-bit operands of v_fma_f64 change its issue rate on
// gfx950?  16 independent accumulators v[0:1] .. v[30:31] (so no dependent-issue stalls), multiplier operands
// chosen per variant; explicit registers through inline asm.   One workgroup of 8 waves per CU (2 waves / SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>

#define FMA16(S0, S1)                                                                                  \
  "v_fma_f64 v[0:1], " S0 ", " S1 ", v[0:1]\n\tv_fma_f64 v[4:5], " S0 ", " S1 ", v[4:5]\n\t"            \
  "v_fma_f64 v[8:9], " S0 ", " S1 ", v[8:9]\n\tv_fma_f64 v[12:13], " S0 ", " S1 ", v[12:13]\n\t"        \
  "v_fma_f64 v[16:17], " S0 ", " S1 ", v[16:17]\n\tv_fma_f64 v[20:21], " S0 ", " S1 ", v[20:21]\n\t"    \
  "v_fma_f64 v[24:25], " S0 ", " S1 ", v[24:25]\n\tv_fma_f64 v[28:29], " S0 ", " S1 ", v[28:29]\n\t"    \
  "v_fma_f64 v[2:3], " S0 ", " S1 ", v[2:3]\n\tv_fma_f64 v[6:7], " S0 ", " S1 ", v[6:7]\n\t"            \
  "v_fma_f64 v[10:11], " S0 ", " S1 ", v[10:11]\n\tv_fma_f64 v[14:15], " S0 ", " S1 ", v[14:15]\n\t"    \
  "v_fma_f64 v[18:19], " S0 ", " S1 ", v[18:19]\n\tv_fma_f64 v[22:23], " S0 ", " S1 ", v[22:23]\n\t"    \
  "v_fma_f64 v[26:27], " S0 ", " S1 ", v[26:27]\n\tv_fma_f64 v[30:31], " S0 ", " S1 ", v[30:31]\n\t"
// accumulators alternate bank pairs (0,1) / (2,3) above; the _LO form keeps them all on banks (0,1)
#define FMA16_LO(S0, S1)                                                                               \
  "v_fma_f64 v[0:1], " S0 ", " S1 ", v[0:1]\n\tv_fma_f64 v[4:5], " S0 ", " S1 ", v[4:5]\n\t"            \
  "v_fma_f64 v[8:9], " S0 ", " S1 ", v[8:9]\n\tv_fma_f64 v[12:13], " S0 ", " S1 ", v[12:13]\n\t"        \
  "v_fma_f64 v[16:17], " S0 ", " S1 ", v[16:17]\n\tv_fma_f64 v[20:21], " S0 ", " S1 ", v[20:21]\n\t"    \
  "v_fma_f64 v[24:25], " S0 ", " S1 ", v[24:25]\n\tv_fma_f64 v[28:29], " S0 ", " S1 ", v[28:29]\n\t"    \
  "v_fma_f64 v[32:33], " S0 ", " S1 ", v[32:33]\n\tv_fma_f64 v[36:37], " S0 ", " S1 ", v[36:37]\n\t"    \
  "v_fma_f64 v[40:41], " S0 ", " S1 ", v[40:41]\n\tv_fma_f64 v[44:45], " S0 ", " S1 ", v[44:45]\n\t"    \
  "v_fma_f64 v[48:49], " S0 ", " S1 ", v[48:49]\n\tv_fma_f64 v[52:53], " S0 ", " S1 ", v[52:53]\n\t"    \
  "v_fma_f64 v[56:57], " S0 ", " S1 ", v[56:57]\n\tv_fma_f64 v[60:61], " S0 ", " S1 ", v[60:61]\n\t"

#define CLOB                                                                                                     \
  "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16",     \
      "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31",     \
      "v32", "v33", "v36", "v37", "v40", "v41", "v44", "v45", "v48", "v49", "v52", "v53", "v56", "v57", "v60",     \
      "v61", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71"

template <int VAR>
__global__ void __launch_bounds__(512) probe(double* out, int iters) {
  asm volatile(
      "v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0x3ff00000\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0x3ff00000\n\t"
      "v_mov_b32 v68, 0\n\tv_mov_b32 v69, 0x3ff00000\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0x3ff00000\n\t" ::
          : CLOB);
  for (int it = 0; it < iters; ++it) {
    if (VAR == 0) asm volatile(FMA16_LO("v[64:65]", "v[68:69]") ::: CLOB);       // acc (0,1), s0 (0,1), s1 (0,1)
    if (VAR == 1) asm volatile(FMA16_LO("v[66:67]", "v[68:69]") ::: CLOB);       // acc (0,1), s0 (2,3), s1 (0,1)
    if (VAR == 2) asm volatile(FMA16_LO("v[66:67]", "v[70:71]") ::: CLOB);       // acc (0,1), s0 (2,3), s1 (2,3)
    if (VAR == 3) asm volatile(FMA16("v[64:65]", "v[68:69]") ::: CLOB);          // acc mixed, s0 (0,1), s1 (0,1)
    if (VAR == 4) asm volatile(FMA16("v[66:67]", "v[68:69]") ::: CLOB);          // acc mixed, s0 (2,3), s1 (0,1)
    if (VAR == 5) asm volatile(FMA16_LO("v[64:65]", "v[64:65]") ::: CLOB);       // same register twice
    if (VAR == 6) asm volatile(FMA16_LO("1.0", "v[68:69]") ::: CLOB);            // inline constant + one VGPR pair
    if (VAR == 7) asm volatile(FMA16_LO("s[2:3]", "v[70:71]") ::: CLOB);         // SGPR pair + VGPR pair on (2,3)
  }
  double r;
  asm volatile("v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1" : "=v"(((unsigned*)&r)[0]), "=v"(((unsigned*)&r)[1])::CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int VAR>
void run(const char* name, double* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  // 2 waves per SIMD, 16 FMAs per wave and iteration
  printf("%-52s %6.2f ns per v_fma_f64 and SIMD\n", name, ms * 1e6 / iters / 32.0);
  fflush(stdout);
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 512 * sizeof(double));
  run<6>("acc (0,1) | const, VGPR (0,1)", out);
  run<7>("acc (0,1) | SGPR, VGPR (2,3)", out);
  run<5>("acc (0,1) | same VGPR pair twice (0,1)", out);
  run<0>("acc (0,1) | s0 (0,1), s1 (0,1)", out);
  run<1>("acc (0,1) | s0 (2,3), s1 (0,1)", out);
  run<2>("acc (0,1) | s0 (2,3), s1 (2,3)", out);
  run<3>("acc alternating | s0 (0,1), s1 (0,1)", out);
  run<4>("acc alternating | s0 (2,3), s1 (0,1)", out);
  return 0;
}
