// Two small LDS-direct rings (tests/test_asm_waits.py): rows of 64 doubles travel global -> LDS with buffer_load ... lds,
// DEPTH trips ahead of the inline-asm LDS read that consumes them.  ring_good_kernel waits for its row with a counted
// s_waitcnt before every read; ring_bad_kernel has lost the wait at the end of the trip (what an edit of the real kernels'
// pipelines could do) -- the reads then run with ever more loads in flight.  tools/asm_wait_check.py counts LDS-direct loads
// in flight at inline-asm LDS reads (going round every loop three times): it must pass the first and name the second.
#include <hip/hip_runtime.h>

typedef int buf_i4 __attribute__((ext_vector_type(4)));
constexpr int DEPTH = 4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

template <bool WAIT_PER_TRIP>
__device__ __forceinline__ void ring_body(const double* __restrict__ in, double* __restrict__ out, int rows) {
  __shared__ __attribute__((aligned(16))) double ring[DEPTH][64];
  const unsigned lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(in, (unsigned)rows * 512u);
  auto request = [&](int row, int slot) {  // lanes 0..31 carry 16 bytes each: one row of 64 doubles
    if (lane < 32)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&ring[slot][0], 16,
                                               (int)(lane * 16u), row * 512, 0, 0);
  };
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) request(i, i);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");  // row 0 has landed
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&ring[0][0] + lane * 8u;
  double acc = 0.0;
  int slot = 0;
  for (int it = 0; it < rows; ++it) {
    double v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(base + (unsigned)slot * 512u) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    acc += v;
    request(it + DEPTH < rows ? it + DEPTH : it, slot);  // refill the slot just read
    if (WAIT_PER_TRIP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");  // the next row has landed
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[threadIdx.x] = acc;
}

__global__ void __launch_bounds__(64) ring_good_kernel(const double* __restrict__ in, double* __restrict__ out, int rows) {
  ring_body<true>(in, out, rows);
}
__global__ void __launch_bounds__(64) ring_bad_kernel(const double* __restrict__ in, double* __restrict__ out, int rows) {
  ring_body<false>(in, out, rows);
}
