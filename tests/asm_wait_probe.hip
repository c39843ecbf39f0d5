// Explicit instantiations of the kernels whose inline-asm loads tests/test_asm_waits.py checks (compiled to assembly only).
#include "../audio_source_separation_amd/csrc/assx_widem_cov.hpp"

namespace assx {
namespace widem {
#define INST(R, M, WK)                                                                                              \
  template __global__ void pair_cov_kernel<R, M, WK>(const Cx<R>*, const R*, const R*, R*, Dims, FlatPart, R);       \
  template __global__ void src_cov_kernel<R, M, WK>(const Cx<R>*, const R*, const R*, R*, Dims, FlatPart, R);
INST(double, 5, WK_TV)
INST(double, 6, WK_TV)
INST(double, 7, WK_TV)
INST(double, 8, WK_TV)
INST(float, 5, WK_TV)
INST(float, 8, WK_TV)
INST(double, 8, WK_NFT)
INST(double, 5, WK_NT)
}  // namespace widem
}  // namespace assx
