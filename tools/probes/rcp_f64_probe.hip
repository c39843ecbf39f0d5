// Accuracy of v_rcp_f64 and of 1 / 2 Newton steps on top of it (fast_rcp, csrc/assx_common.hpp), against the correctly
// rounded 1.0 / x:  hipcc --offload-arch=gfx950 -O3 tools/probes/rcp_f64_probe.hip -o /tmp/rcp_probe && /tmp/rcp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  double e = fma(-v, r, 1.0);
  r = fma(r, e, r);
  r1[i] = r;
  e = fma(-v, r, 1.0);
  r = fma(r, e, r);
  r2[i] = r;
}

int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), a(n), b(n), c(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    double m = 1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX * 1e-9;
    int ex = (rand() % 200) - 100;
    x[i] = ldexp(m, ex) * ((i & 1) ? 1 : -1);
  }
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0;
  long exact1 = 0, exact2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)x[i];
    m0 = fmax(m0, (double)fabsl(((long double)a[i] - t) / t));
    m1 = fmax(m1, (double)fabsl(((long double)b[i] - t) / t));
    m2 = fmax(m2, (double)fabsl(((long double)c[i] - t) / t));
    exact1 += b[i] == 1.0 / x[i];
    exact2 += c[i] == 1.0 / x[i];
  }
  printf("max relative error of 1/x over %d values: v_rcp_f64 %.3e (2^%.1f) | + 1 Newton step %.3e (%.2f ulp of 2^-53) | + 2 steps %.3e (%.2f)\n", n, m0,
         log2(m0), m1, m1 / 1.1102230246251565e-16, m2, m2 / 1.1102230246251565e-16);
  printf("equal to the correctly rounded quotient: 1 step %.2f %%, 2 steps %.2f %%\n", 100.0 * exact1 / n, 100.0 * exact2 / n);
  return 0;
}
