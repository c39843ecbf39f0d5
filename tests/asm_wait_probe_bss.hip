// The M <= 4 streaming kernels in the instantiations of BASELINE configs 3 and 4 (and n_basis 10), referenced so that the
// compiler emits them: tests/test_asm_waits.py checks their inline-asm loads against their waits in the assembly.
#include "../audio_source_separation_amd/csrc/assx_stream.hpp"
#include "../audio_source_separation_amd/csrc/assx_cov_mfma.hpp"

namespace {
template <typename K>
void keep(K k) {
  static volatile const void* sink;
  sink = reinterpret_cast<const void*>(k);
}
}  // namespace

void asm_wait_probe_bss_instantiate() {
  using namespace assx;
  keep(&cov_stream_kernel<double, 4, 3, true, true, 1, 2, 1, 2, true>);
  keep(&cov_stream_kernel<float, 4, 3, true, true, 1, 2, 1, 2, true>);
  keep(&cov_stream_kernel<double, 2, 1, true, true, 1, 3, 1, 2, false>);
  keep(&basis_stream_vd_kernel<double, 4, true, 2, 2, false, false>);
  keep(&basis_stream_vd_kernel<double, 4, true, 2, 2, false, true>);
  keep(&basis_stream_vd_kernel<float, 4, true, 2, 2, false, false>);
  keep(&act_stream_vd_kernel<double, 4, true, 3, 2, false>);
  keep(&act_stream_vd_kernel<float, 4, true, 3, 2, false>);
  keep(&cov_mfma_kernel<double, 4, true, 3>);
}
