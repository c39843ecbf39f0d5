// Library-internal door into the NMF matrix-core kernels (not part of the C-ABI): one half of an update, stopped
// before the combination step, for callers that combine the per-element sums differently (the partitioning-function
// source model of ILRMA, ilrma.py:368-408).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/assx.h"

namespace assx {

constexpr int NMF_HALF_BASIS = 0, NMF_HALF_ACT = 1;

// Launches the basis (reduce over t) or activation (reduce over f) kernel of `kind` on X (B,F,T), Tb (B,F,K),
// V (B,K,T) and leaves the split partial sums in `ws` (assx_nmf_workspace_bytes):
//   basis:      part[slab][b*2 + s][f*K + k]        activation: part[slab][b*2 + s][k*T + t]     (s = 0 num, 1 den)
// Returns 0 and sets *part / *slabs; ASSX_E_UNSUPPORTED when K exceeds the matrix-core path (n_basis <= 64).
int nmf_half_partials(assx_ctx* ctx, int kind, double domain, double param, double eps, int half, const void* X,
                      const void* Tb, const void* V, void* ws, int B, int F, int T, int K, int dtype, hipStream_t st,
                      const void** part, int* slabs);

// The ILRMA source model for n_basis > 4 straight from the mixture (assx_nmf_xfed.hpp): the IS-type update `kind` with
// target |w_n^H x|^2 formed on the fly, Tb (B,N,F,K) and V (B,N,K,T) updated in place; `ws` = the NMF scratch of
// assx_nmf_workspace_bytes(B * M, F, T, K).  Returns ASSX_E_UNSUPPORTED (and launches nothing) outside its range
// (2 <= M <= 4, n_basis <= 32): the caller then takes the power-map route.
// lpart != nullptr (IS_MM, domain 2 only -- otherwise ASSX_E_UNSUPPORTED): the basis half also leaves the data term of
// the model's negative log-likelihood AT ENTRY behind, one partial per (workgroup, source) at lpart[b * lstride + i],
// i < nmf_xfed_loss_partials(...); the caller adds the log-det terms and sums.
int nmf_update_xfed(assx_ctx* ctx, int kind, double domain, double param, double eps, const void* X, const void* W,
                    void* Tb, void* V, void* ws, int B, int M, int F, int T, int K, int dtype, hipStream_t st,
                    double* lpart = nullptr, int lstride = 0);
int nmf_xfed_loss_partials(int M, int F, int T, int K);  // partials per utterance the fused loss writes; 0: not applicable

// One multiplicative update (the checks of assx_nmf_update_ex are the caller's) that also returns, in loss_prev (B,)
// float64 or nullptr, the criterion of the model AT ENTRY -- i.e. the loss the reference records after the previous
// update (nmf.py:48-53): accumulated inside the basis half where that is a matrix-core kernel with domain 2 and an EUC /
// KL / IS rule, by the stand-alone pass otherwise.  assx_nmf_iterate strings these together.
int nmf_update_with_loss(assx_ctx* ctx, int kind, double domain, double param, double eps, const void* X, void* Tb,
                         void* V, double* loss_prev, void* ws, int B, int F, int T, int K, int dtype, hipStream_t st);

}  // namespace assx
