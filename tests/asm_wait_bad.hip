// A kernel with the round-3 bug on purpose (tests/test_asm_waits.py): an inline-asm LDS read whose destination registers are
// consumed BEFORE the s_waitcnt that covers them.  tools/asm_wait_check.py must report it and csrc/build.sh must fail on it.
#include <hip/hip_runtime.h>

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void asm_wait_bad_kernel(const double* __restrict__ in, double* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) double tile[256];
  tile[threadIdx.x] = in[threadIdx.x];
  tile[threadIdx.x + 64] = in[threadIdx.x + 64];
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)tile + (threadIdx.x & 63) * 16u;
  v2d r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  double early;
  asm volatile("v_add_f64 %0, %1, %2" : "=v"(early) : "v"(r.x), "v"(r.y));  // touches r before the wait below
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[threadIdx.x] = early + r.x;
}
