// Probe: buffer_load_dwordx4 ... lds (LDS-direct load, gfx950) -- destination layout, vmcnt completion, OOB zeros.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef TT
#define TT 200
#endif

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(64) probe(const double* __restrict__ V, double* __restrict__ out, int T, int t0,
                                            unsigned nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16 * 512];
  __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (int)nbytes, 0x00020000);
  const int lane = threadIdx.x;
  // lanes 0-31: row r, lanes 32-63: row r+1; 16 bytes (2 frames) per lane
  const unsigned voff = (unsigned)(lane & 31) * 16u + (unsigned)(lane >> 5) * (unsigned)T * 8u + (unsigned)t0 * 8u;
  unsigned so = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, voff,
                                             so, 0, 0);
    so += 2u * (unsigned)T * 8u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned a = (unsigned)lane * 8u;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    d2 v;
    asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(v)
                 : "v"(a), "n"(r), "n"(r + 1)
                 : "memory");
    out[r * 64 + lane] = v.x;
    out[(r + 1) * 64 + lane] = v.y;
  }
}

int main() {
  const int T = TT, rows = 16;
  std::vector<double> h(rows * T);
  for (int i = 0; i < rows * T; ++i) h[i] = 1000.0 * (i / T) + (i % T);
  double *dV, *dO;
  hipMalloc(&dV, h.size() * 8);
  hipMalloc(&dO, 16 * 64 * 8);
  hipMemcpy(dV, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  int bad = 0;
  for (int t0 : {0, 64, 128, 192, 7}) {   // 192: frames 200..255 fall off the row (next row / OOB at the very end)
    hipMemset(dO, 0xff, 16 * 64 * 8);
    probe<<<1, 64>>>(dV, dO, T, t0, (unsigned)(h.size() * 8));
    std::vector<double> o(16 * 64);
    hipMemcpy(o.data(), dO, o.size() * 8, hipMemcpyDeviceToHost);
    for (int r = 0; r < 16; ++r)
      for (int l = 0; l < 64; ++l) {
        const long idx = (long)r * T + t0 + l;
        const double want = idx < (long)h.size() ? h[idx] : 0.0;  // past the buffer: zeros
        if (o[r * 64 + l] != want) {
          if (bad < 10) printf("t0=%d r=%d l=%d got %g want %g\n", t0, r, l, o[r * 64 + l], want);
          ++bad;
        }
      }
  }
  printf("lds_dma_probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
