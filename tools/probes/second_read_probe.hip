// What does the SECOND read of a bin cost?  (round 5's review, item 2)
//
// The covariance pass of iteration i and the basis pass of iteration i+1 read the same X.  Fused per bin -- covariance of
// bin f, its IP sweep, then the basis sums of bin f with the new filter -- the second read would find the bin's
// 64 blocks x 4 planes x 1 KB = 262 KB wherever the first read left them.  This probe prices exactly that, with no
// arithmetic: W waves of one workgroup own a bin (block j of the bin goes to wave j % W), walk it ONCE, meet at a
// barrier, and -- in the "twice" form -- walk it again.  Against it: the shipped kernels' shape (2048 single-wave ranges
// per utterance, each plane piece read once).  Working sets: one utterance (268.7 MB, within 0.1 % of the 256 MiB
// Infinity Cache) and eight (2.15 GB, config 5's per-GPU batch, far beyond it).
//
//   hipcc --offload-arch=gfx950 -O3 second_read_probe.hip -o /tmp/sr && /tmp/sr
//
// Output: TB/s counted on the bytes REQUESTED (twice = 2 x the array), and the cost of the second walk as a fraction of
// the first: (t_twice - t_once) / t_once.  1.0 = the second read costs what the first did (nothing survived); 0 = free.
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int M = 4, F = 1025, T = 4096, TBK = T / 64;

// the shipped shape: range g of an utterance = L consecutive (bin, block) items, one wave
__global__ void __launch_bounds__(64) flat_once(const double2* __restrict__ x, int L, int Gu, double* out) {
  const int b = blockIdx.x / Gu, g = blockIdx.x % Gu;
  const size_t plane = (size_t)F * T;
  const double2* p = x + (size_t)b * M * plane + threadIdx.x;
  const int lo = g * L, hi = min(lo + L, F * TBK);
  double s = 0;
  for (int it = lo; it < hi; ++it) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const double2 v = p[(size_t)m * plane + (size_t)it * 64];
      s += v.x + v.y;
    }
  }
  if (s == 123.456) out[0] = s;
}

// a workgroup of W waves owns bin (b, f): PASSES walks over its 64 blocks, a barrier between them
template <int W, int PASSES>
__global__ void __launch_bounds__(64 * W) bin_owned(const double2* __restrict__ x, double* out) {
  const int b = blockIdx.x / F, f = blockIdx.x % F;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t plane = (size_t)F * T;
  const double2* p = x + (size_t)b * M * plane + (size_t)f * T + lane;
  double s = 0;
#pragma unroll 1
  for (int pass = 0; pass < PASSES; ++pass) {
    for (int tb = w; tb < TBK; tb += W) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const double2 v = p[(size_t)m * plane + (size_t)tb * 64];
        s += v.x + v.y;
      }
    }
    __syncthreads();
    asm volatile("" : "+v"(s));  // the second walk is not folded into the first
  }
  if (s == 123.456) out[0] = s;
}

template <typename Fn>
static double time_ms(Fn&& launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) launch();
  hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <int W>
static void bin_rows(const double2* x, int B, double bytes, double* out) {
  const double t1 = time_ms([&] { hipLaunchKernelGGL((bin_owned<W, 1>), dim3(B * F), dim3(64 * W), 0, 0, x, out); });
  const double t2 = time_ms([&] { hipLaunchKernelGGL((bin_owned<W, 2>), dim3(B * F), dim3(64 * W), 0, 0, x, out); });
  printf("  bin-owned, %2d waves per bin : once %7.1f us (%5.2f TB/s)   twice %7.1f us (%5.2f TB/s requested)   second walk costs %.2f of the first\n",
         W, t1 * 1e3, bytes / (t1 * 1e-3) / 1e12, t2 * 1e3, 2 * bytes / (t2 * 1e-3) / 1e12, (t2 - t1) / t1);
}

int main() {
  const size_t per = (size_t)M * F * T * 16;
  const int Bmax = 8;
  double2* x;
  double* out;
  hipMalloc(&x, per * Bmax);
  hipMalloc(&out, 64);
  hipMemset(x, 0, per * Bmax);
  for (int B : {1, 8}) {
    const double bytes = (double)per * B;
    printf("%d utterance(s), %.1f MB of X\n", B, bytes / 1e6);
    const int Gu = 2048, L = (F * TBK + Gu - 1) / Gu;
    const double tf = time_ms([&] { hipLaunchKernelGGL(flat_once, dim3(B * Gu), dim3(64), 0, 0, x, L, Gu, out); });
    printf("  flat ranges (shipped shape) : once %7.1f us (%5.2f TB/s)\n", tf * 1e3, bytes / (tf * 1e-3) / 1e12);
    bin_rows<1>(x, B, bytes, out);
    bin_rows<2>(x, B, bytes, out);
    bin_rows<4>(x, B, bytes, out);
    bin_rows<8>(x, B, bytes, out);
    bin_rows<16>(x, B, bytes, out);
  }
  return 0;
}
