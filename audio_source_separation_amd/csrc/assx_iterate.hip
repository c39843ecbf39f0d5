// Whole iteration loops behind the C-ABI (include/assx.h: assx_nmf_iterate, assx_auxiva_iterate, assx_ilrma_iterate).
//
// Reference: the Python loops of NMFbase.update (src/algorithm/nmf.py:45-53), AuxIVAbase.__call__ (src/bss/iva.py:
// 420-441) and GaussILRMA.__call__ (src/bss/ilrma.py:233-256).  Nothing is computed here: every iteration is the same
// sequence of the public entry points that the host classes issue from their own loops, enqueued from C so that an
// iteration costs a handful of launches instead of 4-7 interpreter round trips (the small BASELINE configurations were
// bounded by the Python loop, profiles/r03_small_cfgs.txt).
#include "assx_common.hpp"
#include "assx_nmf_internal.hpp"

using namespace assx;

extern "C" {

int assx_nmf_iterate(assx_ctx* ctx, int n_iter, int kind, double domain, double param, double eps, const void* X,
                     void* Tb, void* V, double* loss, void* ws, int B, int F, int T, int K, int dtype, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, n_iter >= 0, ASSX_E_ARG, "n_iter must be >= 0, got %d", n_iter);
  // loss[i] = criterion of the model after update i (nmf.py:48-53).  It is a function of the model update i + 1 reads, so
  // it rides on that update's basis half (nmf_update_with_loss); only the last one costs a pass of its own.
  for (int i = 0; i < n_iter; ++i) {
    int rc;
    if (i == 0 || !loss) {
      rc = assx_nmf_update_ex(ctx, kind, domain, param, eps, X, Tb, V, ws, B, F, T, K, dtype, stream);  // validates
    } else {
      rc = nmf_update_with_loss(ctx, kind, domain, param, eps, X, Tb, V, loss + (size_t)(i - 1) * B, ws, B, F, T, K, dtype,
                                (hipStream_t)stream);
    }
    if (rc) return rc;
  }
  if (loss && n_iter > 0)
    return assx_nmf_loss_ex(ctx, kind, domain, param, eps, X, Tb, V, loss + (size_t)(n_iter - 1) * B, ws, B, F, T, K, dtype,
                            stream);
  return 0;
}

int assx_auxiva_iterate(assx_ctx* ctx, int n_iter, int kind, int spatial, int pair_m, int pair_n, const void* X,
                        void* W, double eps, double threshold, void* r, double* loss, int32_t* status, void* ws,
                        int B, int M, int F, int T, int dtype, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, n_iter >= 0, ASSX_E_ARG, "n_iter must be >= 0, got %d", n_iter);
  ASSX_REQUIRE(ctx, r, ASSX_E_NULL, "assx_auxiva_iterate: NULL weight scratch");
  ASSX_REQUIRE(ctx, spatial >= ASSX_SPATIAL_IP && spatial <= ASSX_SPATIAL_IP2, ASSX_E_ARG, "bad spatial algorithm %d",
               spatial);
  ASSX_REQUIRE(ctx, spatial != ASSX_SPATIAL_IP2 || (M >= 2 && pair_m >= 0 && pair_m < M && pair_n >= 0 && pair_n < M && pair_m != pair_n),
               ASSX_E_ARG, "IP2 needs a pair of two different sources in [0, %d), got (%d, %d)", M, pair_m, pair_n);
  for (int i = 0; i < n_iter; ++i) {
    int rc = assx_auxiva_weights(ctx, X, W, kind, eps, r, loss ? loss + (size_t)i * B : nullptr, ws, B, M, F, T, dtype,
                                 stream);
    if (rc) return rc;
    rc = assx_auxiva_spatial_update(ctx, spatial, pair_m, pair_n, X, W, r, eps, threshold, nullptr, status, ws, B, M, F,
                                    T, dtype, stream);
    if (rc) return rc;
    if (spatial == ASSX_SPATIAL_IP2 && M > 0) {
      pair_m = (pair_m + 1) % M;
      pair_n = (pair_n + 1) % M;
    }
  }
  if (loss) return assx_auxiva_weights(ctx, X, W, kind, eps, r, loss + (size_t)n_iter * B, ws, B, M, F, T, dtype, stream);
  return 0;
}

int assx_ilrma_iterate(assx_ctx* ctx, int n_iter, int spatial, int pair_m, int pair_n, int normalize, int ref,
                       double pb_exponent, const void* X, void* W, void* Tb, void* V, double domain, double eps,
                       double threshold, const void* C, double* power_bins, void* scale, double* loss,
                       int32_t* status, void* ws, int B, int M, int F, int T, int K, int dtype, void* stream) {
  ASSX_REQUIRE_CTX(ctx);
  ASSX_REQUIRE(ctx, n_iter >= 0, ASSX_E_ARG, "n_iter must be >= 0, got %d", n_iter);
  ASSX_REQUIRE(ctx, normalize >= 0 && normalize <= 2, ASSX_E_ARG, "normalize must be 0 (none), 1 (power) or 2 (projection-back), got %d", normalize);
  ASSX_REQUIRE(ctx, normalize != 1 || (C && power_bins), ASSX_E_NULL, "assx_ilrma_iterate: 'power' normalisation needs C and power_bins");
  ASSX_REQUIRE(ctx, normalize != 2 || scale, ASSX_E_NULL, "assx_ilrma_iterate: 'projection-back' normalisation needs the scale scratch");
  ASSX_REQUIRE(ctx, spatial >= ASSX_SPATIAL_IP && spatial <= ASSX_SPATIAL_IP2, ASSX_E_ARG, "bad spatial algorithm %d",
               spatial);
  ASSX_REQUIRE(ctx, M >= 2 && M <= 32, ASSX_E_UNSUPPORTED, "2 <= M <= 32 channels are supported, got %d", M);
  // the pair is turned into a source mask below (a shift) BEFORE the spatial update would reject it
  ASSX_REQUIRE(ctx, spatial != ASSX_SPATIAL_IP2 || (pair_m >= 0 && pair_m < M && pair_n >= 0 && pair_n < M && pair_m != pair_n),
               ASSX_E_ARG, "IP2 needs a pair of two different sources in [0, %d), got (%d, %d)", M, pair_m, pair_n);
  const bool with_stat = normalize == 1;
  for (int i = 0; i < n_iter; ++i) {
    const unsigned mask = spatial == ASSX_SPATIAL_IP2 ? ((1u << pair_m) | (1u << pair_n)) : ~0u;
    int rc = assx_ilrma_source_update(ctx, X, W, Tb, V, domain, eps, mask, loss ? loss + (size_t)i * B : nullptr, ws, B, M, F,
                                      T, K, dtype, stream);
    if (rc) return rc;
    rc = assx_ilrma_spatial_update(ctx, spatial, pair_m, pair_n, X, W, Tb, V, domain, eps, threshold, nullptr,
                                   with_stat ? C : nullptr, with_stat ? power_bins : nullptr, status, ws, B, M, F, T, K,
                                   dtype, stream);
    if (rc) return rc;
    if (normalize == 1) {
      rc = assx_ilrma_normalize_power_bins(ctx, W, Tb, power_bins, domain, eps, B, M, F, K, dtype, stream);
    } else if (normalize == 2) {
      rc = assx_projection_back_scale(ctx, X, W, ref, scale, status, ws, B, M, F, T, dtype, stream);
      if (rc) return rc;
      rc = assx_ilrma_normalize_pb(ctx, W, Tb, scale, pb_exponent, B, M, F, K, dtype, stream);
    }
    if (rc) return rc;
    if (spatial == ASSX_SPATIAL_IP2) {
      pair_m = (pair_m + 1) % M;
      pair_n = (pair_n + 1) % M;
    }
  }
  if (loss) return assx_ilrma_loss(ctx, X, W, Tb, V, domain, eps, loss + (size_t)n_iter * B, ws, B, M, F, T, K, dtype, stream);
  return 0;
}

}  // extern "C"
