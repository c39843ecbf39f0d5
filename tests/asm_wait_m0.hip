// Two kernels for the M0 rule of tools/asm_wait_check.py (tests/test_asm_waits.py): an LDS-direct load whose inline asm
// WRITES M0 itself (rounds 3-5's form: reported) right in front of a compiler-managed LDS-direct load, and the same load
// with the LDS address handed in through an input operand pinned to M0 (round 6's form: clean).
#include <hip/hip_runtime.h>

typedef unsigned int buf_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ buf_u4 words(const void* p) {
  buf_u4 r;
  r.x = (unsigned)(size_t)p;
  r.y = (unsigned)((size_t)p >> 32) & 0xffffu;
  r.z = 0xffffffffu;
  r.w = 0x00020000u;
  return r;
}

__global__ void m0_bad_kernel(const float* __restrict__ in, float* __restrict__ out) {
  __shared__ float s[256];
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) void*)s;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dword %1, %2, 0 offen lds" : : "s"(la), "v"(threadIdx.x * 4u), "s"(words(in)) : "memory", "m0");
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000),
                                           (__attribute__((address_space(3))) void*)(s + 64), 4, (int)threadIdx.x * 4, 256, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + 64];
}

__global__ void m0_good_kernel(const float* __restrict__ in, float* __restrict__ out) {
  __shared__ float s[256];
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) void*)s;
  asm volatile("s_nop 4\n\tbuffer_load_dword %1, %2, 0 offen lds" : : "{m0}"(la), "v"(threadIdx.x * 4u), "s"(words(in)) : "memory");
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)in, 0, -1, 0x00020000),
                                           (__attribute__((address_space(3))) void*)(s + 64), 4, (int)threadIdx.x * 4, 256, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + 64];
}
