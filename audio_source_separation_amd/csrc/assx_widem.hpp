// Wide-channel path: 5 <= M <= 8 channels (= sources; the reference is determined and generic in M,
// src/bss/ilrma.py:61-62, src/bss/iva.py:39-59).
//
// The streaming kernels of assx_stream.hpp keep N*M*M Hermitian accumulators and the demixing rows of a bin in one
// wave's registers, which stops at M = 4.  Beyond that the same entry points run on MATERIALISED maps instead of
// recomputing y = W x in every pass:
//     P = |W x|^2 (B,N,F,T), R = (Tb V)^(2/domain) (B,N,F,T), Y = W x where it is needed (projection back),
// with one workgroup per (utterance, bin) accumulating a source's Hermitian covariance per wave, and the SAME
// lane-group IP / ISS / IP2 kernels (assx_group_linalg.hpp: 64 lanes = one 8 x 8 matrix).  The source model is the
// batched IS-NMF update on the matrix cores (the n_basis > 4 route of the M <= 4 path, any n_basis here).
// Correct and deterministic first; it moves ~3x the bytes of the streaming design and is not tuned.
//
// Every function mirrors the entry point of include/assx.h named in its comment and is called from there when M > 4.
#pragma once
#include "assx_common.hpp"

namespace assx {
namespace widem {

constexpr int MMIN = 5, MMAX = 8;  // compile-time channel counts (tuned kernels)
constexpr int RT_MMAX = 32;         // run-time channel counts above MMAX (assx_widem_rt.hpp: functional, not tuned)
inline bool handles(int M) { return M >= MMIN && M <= RT_MMAX; }

size_t workspace_bytes(int B, int M, int F, int T, int K, int dtype);

int demix(assx_ctx* ctx, const void* X, const void* W, const void* scale, void* Y, int B, int M, int F, int T, int dtype,
          hipStream_t st);                                                                    // assx_demix
int power_map(assx_ctx* ctx, const void* X, const void* W, void* P, int B, int M, int F, int T, int dtype,
              hipStream_t st);                                                                // assx_ilrma_power_map
int cov_accumulate(assx_ctx* ctx, const void* X, const void* r, int r_kind, double eps, void* U, void* ws, int B, int M,
                   int N, int F, int T, int dtype, hipStream_t st);                           // assx_cov_accumulate
int ip_update(assx_ctx* ctx, const void* U, void* W, double thr, int32_t* status, int B, int M, int F, int dtype,
              hipStream_t st);                                                                // assx_ip_update
int iss_update(assx_ctx* ctx, const void* U, void* W, int n_frames, int B, int M, int F, int dtype, hipStream_t st);
int ip2_update(assx_ctx* ctx, const void* U, void* W, double thr, int32_t* status, int pm, int pn, int B, int M, int F,
               int dtype, hipStream_t st);
int ilrma_source_update(assx_ctx* ctx, const void* X, const void* W, void* Tb, void* V, double domain, double eps,
                        unsigned source_mask, double* loss_prev, void* ws, int B, int M, int F, int T, int K, int dtype,
                        hipStream_t st);                                                      // assx_ilrma_source_update
int ilrma_spatial_update(assx_ctx* ctx, int spatial, int pm, int pn, const void* X, void* W, const void* Tb,
                         const void* V, double domain, double eps, double thr, void* U_out, const void* C,
                         double* power_bins, int32_t* status, void* ws, int B, int M, int F, int T, int K, int dtype,
                         hipStream_t st);                                                     // assx_ilrma_spatial_update
int ilrma_source_update_partitioned(assx_ctx* ctx, const void* X, const void* W, void* Z, void* Tb, void* V, void* Teff,
                                    void* Veff, double eps, void* ws, int B, int M, int F, int T, int K, int dtype,
                                    hipStream_t st);                          // assx_ilrma_source_update_partitioned
int normalize_power_bins_partitioned(assx_ctx* ctx, void* W, void* Z, void* Tb, const double* power_bins, double eps,
                                     void* ws, int B, int M, int F, int K, int dtype,
                                     hipStream_t st);                         // assx_ilrma_normalize_power_bins_partitioned
int weighted_ip(assx_ctx* ctx, const void* X, const void* r, double eps, double thr, double den_floor, void* W,
                int32_t* status, void* ws, int B, int M, int F, int T, int dtype,
                hipStream_t st);                              // assx_idlma_space_update / assx_fastmnmf_update_diagonalizer
int demix_power(assx_ctx* ctx, const void* X, const void* W, void* power, void* ws, int B, int M, int F, int T,
                int dtype, hipStream_t st);                                                   // assx_demix_power
int power_from_cov(assx_ctx* ctx, const void* C, const void* W, void* power, void* ws, int B, int M, int F, int dtype,
                   hipStream_t st);                                                           // assx_power_from_cov
int ilrma_loss(assx_ctx* ctx, const void* X, const void* W, const void* Tb, const void* V, double domain, double eps,
               double* loss, void* ws, int B, int M, int F, int T, int K, int dtype, hipStream_t st,
               double nu = -1.0);                          // assx_ilrma_loss; nu >= 0: assx_tilrma_loss
int tilrma_source_update(assx_ctx* ctx, const void* X, const void* W, void* Tb, void* V, double nu, double eps, void* ws,
                         int B, int M, int F, int T, int K, int dtype, hipStream_t st);       // assx_tilrma_source_update
int tilrma_spatial_update(assx_ctx* ctx, const void* X, void* W, const void* Tb, const void* V, double nu, double eps,
                          void* Xi, const void* C, double* power_bins, int32_t* status, void* ws, int B, int M, int F,
                          int T, int K, int dtype, hipStream_t st);                           // assx_tilrma_spatial_update
int auxiva_weights(assx_ctx* ctx, const void* X, const void* W, int kind, double eps, void* r, double* loss, void* ws,
                   int B, int M, int F, int T, int dtype, hipStream_t st);                     // assx_auxiva_weights
int auxiva_spatial_update(assx_ctx* ctx, int spatial, int pm, int pn, const void* X, void* W, const void* r, double eps,
                          double thr, void* U_out, int32_t* status, void* ws, int B, int M, int F, int T, int dtype,
                          hipStream_t st);                                                    // assx_auxiva_spatial_update
int projection_back_scale(assx_ctx* ctx, const void* X, const void* W, int ref, void* scale, int32_t* status, void* ws,
                          int B, int M, int F, int T, int dtype, hipStream_t st);             // assx_projection_back_scale
int projection_back(assx_ctx* ctx, const void* Y, const void* reference, void* scale, int32_t* status, int B, int N,
                    int F, int T, int dtype, hipStream_t st);                                 // assx_projection_back

}  // namespace widem

// (A B^H)(B B^H)^{-1} per bin: A = `na` rows, B = `M` rows of (.., F, T) complex planes; out[b, f] is na x M, stored
// at out + b*ob + f*of + i*oi + j*oj.  compute_demix_filter: A = Y, B = X; projection back: A = reference, B = Y.
// 1 <= na <= 8, 2 <= M <= 8.  Defined in assx_generic.hip.
int stack_gram_solve(assx_ctx* ctx, const void* A, size_t a_batch_stride, int na, const void* Bm, int M, void* out,
                     size_t ob, size_t of, size_t oi, size_t oj, int32_t* status, int B, int F, int T, int dtype,
                     hipStream_t st);

}  // namespace assx
