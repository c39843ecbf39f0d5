// Probe of v_mfma_f64_16x16x4_f64 operand / accumulator layout on gfx950.
//   hipcc --offload-arch=gfx950 -O2 mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Expected (cdna_hip_programming.md section 3): A lane l -> A[i = l&15][k = l>>4], B lane l -> B[k = l>>4][j = l&15],
// C/D reg r of lane l -> D[row = (l>>4) + 4r][col = l&15].
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A /*16x4*/, const double* B /*4x16*/, double* D /*16x16*/) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];
  const double b = B[(l >> 4) * 16 + (l & 15)];
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = 1 + 0.37 * i + 0.01 * i * i; hB[i] = 2 - 0.11 * i + 0.003 * i * i * i; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(hD[i] - ref[i]) / fabs(ref[i]));
  printf("max rel err vs reference with the documented layout: %.3e %s\n", err, err < 1e-13 ? "LAYOUT OK" : "LAYOUT MISMATCH");
  return err < 1e-13 ? 0 : 1;
}
