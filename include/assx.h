/*
 * assx.h -- C-ABI of the MI355X-native iterative source-separation hot path.
 *
 * Scope of this boundary: the DEVICE side of ONE rank.  Every compute entry point is communication-free; multi-GPU runs
 * are one process per GPU, each with its own context, and the only inter-GPU traffic is at the edges (scatter of the
 * mixtures, gather of the separated outputs -- utterances are independent, src/bss/ilrma.py:203-273).  The package's own
 * Python classes issue those edges through torch.distributed (audio_source_separation_amd/distributed.py; INTEGRATION.md
 * section 3); a host without torch uses assx_comm_init / assx_scatter / assx_gather at the end of this header (round 6:
 * the same grouped RCCL send / recv, RCCL loaded on first use).
 *
 * The reference (tky823/audio_source_separation) has NO plugin / FFI / operator API: its
 * boundary is the Python class surface.  This header is therefore the
 * NEW device boundary that the package's own Python classes (and any other host language)
 * bind with ctypes / cgo / JNI: plain pointers and sizes, no torch types, `extern "C"`.
 * Each entry point cites the reference code it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - All array arguments are DEVICE pointers owned by the caller, contiguous, row-major, laid
 *     out exactly like the reference's NumPy arrays (leading batch axis B of independent
 *     utterances added):
 *         X  (B, M, F, T) complex      mixture STFT, T fastest      ilrma.py:61
 *         W  (B, F, N, M) complex      demixing filters             ilrma.py:67-68
 *         Tb (B, N, F, K) real         NMF basis                    ilrma.py:97
 *         V  (B, N, K, T) real         NMF activation               ilrma.py:101
 *         U  (B, N, F, M, M) complex   weighted spatial covariance  ilrma.py:511
 *         Y  (B, N, F, T) complex      separated estimate           ilrma.py:153-165
 *     complex = interleaved (re, im) of the real type selected by `dtype`.
 *     The reference is determined: N == M (ilrma.py:61-62).  Supported: 2 <= M <= 32 -- M <= 4 on the streaming
 *     kernels (every entry point), 5 <= M <= 8 on the wide-channel path (csrc/assx_widem.hpp: |W x|^2 map for the
 *     source model, streaming covariance; every Gauss-ILRMA / AuxIVA / t-ILRMA / projection-back entry point, IP,
 *     ISS and IP2, the partitioning function with n_basis <= 64), 9 <= M <= 32 on the same path with a run-time
 *     channel count (csrc/assx_widem_rt.hpp: functional, not tuned; IP, ISS and IP2 sweeps and the partitioning
 *     function -- the last three since round 6).  One utterance must stay below
 *     4 GiB in complex128 (M*F*T < 2^28: in-kernel buffer offsets are 32-bit), any number of utterances.
 *   - dtype: ASSX_F32 (float / complex64) or ASSX_F64 (double / complex128 = the reference's).
 *   - `ws` is caller-owned device scratch of at least assx_workspace_bytes() bytes.
 *   - `status` is a device int32[B]; kernels OR flags into it (ASSX_STATUS_*), never clear it.
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default).
 *   - Return: 0 ok; <0 invalid argument (ASSX_E_*); >0 a hipError_t.  Nothing throws across the
 *     boundary; assx_last_error(ctx) returns the message of the last failure on that context.
 *   - One context per (device, host thread); contexts are not thread-safe.  One thread may drive several streams through
 *     its context at once (independent problems, each with its own `ws`): the only device state a context owns -- the
 *     words the matrix-core NMF / X-fed source-model kernels count their "last workgroup done" tickets on -- is kept
 *     per stream.  Two calls that share a `ws` must be ordered by the caller (same stream or an event), as ever.
 *   - The library never changes the calling thread's current device.  The caller makes the context's device current
 *     (hipSetDevice) before every call; a call made with another device current is refused with ASSX_E_ARG.
 *     (assx_ctx_create / assx_ctx_destroy switch to the context's device for their own allocation / wait and switch back.)
 *   - Stream capture (hipStreamBeginCapture, torch.cuda.graph): every entry point may be captured.  The ticket words of a
 *     stream the context has not seen come out of a pool made by assx_ctx_create, so a capture on a fresh stream needs no
 *     allocation; a problem whose kernels need more than 8192 ticket words per stream (more than ~30 batched matrices
 *     of 4096 frames) must run once on the capture stream before the capture begins, or the call is refused with
 *     ASSX_E_UNSUPPORTED and a message saying so.
 *
 * Run-time knobs (environment variables read by the shipped library -- all of them; everything else that earlier rounds
 * could switch through the environment exists only in laboratory builds, -DASSX_LAB=1, whose assx_version() ends in "+lab"):
 *     ASSX_G               ranges of the flat work partition of the streaming kernels (default: 8 per CU x 256 CUs; the
 *                          tests shrink it so that small inputs walk every code path of a long range).  Changes the
 *                          summation order, never the semantics.  Read on every call.
 *     ASSX_NMF_BASIS_WGS   workgroup budget of the matrix-core NMF basis half (default 512)   } the tests force
 *     ASSX_NMF_ACT_WGS     ... of the activation half (default 512)                            } many-slab and
 *     ASSX_NMF_XFED_WGS    ... of the X-fed source-model halves (default 768 / 512)            } ragged partitions
 *     ASSX_XFER_CHUNK_MB   chunk size of the pinned staging ring of assx_upload / assx_download (default 16)
 *     ASSX_XFER_THREADS    host threads of that ring (default: a quarter of the hardware threads, at most 16)
 */
#ifndef ASSX_H
#define ASSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASSX_VERSION_STRING "assx 0.1.0 (gfx950)"

enum { ASSX_F32 = 0, ASSX_F64 = 1 };

enum { ASSX_E_ARG = -1, ASSX_E_UNSUPPORTED = -2, ASSX_E_NULL = -3 };

/* status flags (device int32 per utterance) */
enum {
  ASSX_STATUS_SINGULAR = 1, /* an exactly singular W U_n met in IP: numpy.linalg.solve would raise LinAlgError */
  ASSX_STATUS_COND_REJECT = 2 /* informational: at least one bin kept its old row (cond >= threshold) */
};

/* weight kinds for assx_cov_accumulate */
enum { ASSX_W_NONE = 0, ASSX_W_NT = 1, ASSX_W_NFT = 2 };

/* AuxIVA contrast */
enum { ASSX_IVA_LAPLACE = 0, ASSX_IVA_GAUSS = 1 };

/* spatial update algorithm (algorithm_spatial of the reference classes) */
enum { ASSX_SPATIAL_IP = 0, ASSX_SPATIAL_ISS = 1, ASSX_SPATIAL_IP2 = 2 };

/* NMF divergence / algorithm */
enum { ASSX_NMF_EUC = 0, ASSX_NMF_KL = 1, ASSX_NMF_IS_MM = 2, ASSX_NMF_IS_ME = 3,
       /* row f4, same skeleton (assx_nmf_update_ex / assx_nmf_loss_ex only; domain 2 only): */
       ASSX_NMF_T = 4,              /* tNMF.update_once_mm, param = nu      (nmf.py:400-429) */
       ASSX_NMF_CAUCHY_NAIVE = 5,   /* CauchyNMF.update_once_naive          (nmf.py:468-502) */
       ASSX_NMF_CAUCHY_MM = 6,      /* CauchyNMF.update_once_mm             (nmf.py:504-534) */
       ASSX_NMF_CAUCHY_ME = 7,      /* CauchyNMF.update_once_me             (nmf.py:536-565) */
       ASSX_NMF_CAUCHY_MM_FAST = 8, /* CauchyNMF.update_once_mm_fast        (nmf.py:567-600) */
       ASSX_NMF_T_RAW = 9           /* tILRMA source model on a demixed power: ASSX_NMF_T with the target NOT floored,
                                       param = nu >= 0 (ilrma.py:899-922); update only */ };

typedef struct assx_ctx assx_ctx;

/* ---- context ------------------------------------------------------------------------- */
int assx_ctx_create(int device, assx_ctx** ctx);
int assx_ctx_destroy(assx_ctx* ctx);
const char* assx_last_error(const assx_ctx* ctx);
const char* assx_version(void);
/* scratch bytes sufficient for any call below at these sizes (n_basis > 4 adds one (B,N,F,T) real array and the
 * scratch of the batched NMF update: the source model then runs on the matrix cores) */
size_t assx_workspace_bytes(int B, int M, int F, int T, int K, int dtype);
/* Host-side query (no GPU): the launch order of a batched streaming pass over X (covariance / basis partition of
 * (B, F, T); csrc/assx_stream.hpp: workgroup_range).  ranges[w] = the range of the partition that workgroup w of the
 * grid takes, -1 for a padding slot; returns the grid size (pass ranges = NULL to size the array), < 0 on bad
 * arguments.  B == 1: the XCD-aware order over all ranges; B >= 2: utterance after utterance, each XCD (w % 8) on a
 * contiguous eighth of an utterance's ranges, everything backwards when `reverse` != 0 -- consecutive passes alternate,
 * so that a pass starts with what the previous one left in the Infinity Cache.  The order never changes a result;
 * tests/test_cabi_and_host.py checks that it is a permutation.  No reference counterpart (the reference is NumPy). */
int assx_launch_order(int B, int F, int T, int reverse, int* ranges, int capacity);

/* ---- (a3) demixing  y = W x ----------------------------------------------------------- */
/* ILRMAbase.separate / IVAbase.separate  (src/bss/ilrma.py:153-165, src/bss/iva.py:105-117).
 * scale: optional (B,N,F) complex multiplied into Y (the final projection-back scaling,
 * ilrma.py:270, iva.py:455-456); NULL = none. */
int assx_demix(assx_ctx* ctx, const void* X, const void* W, const void* scale, void* Y,
               int B, int M, int F, int T, int dtype, void* stream);

/* ---- (a4) weighted spatial covariance -------------------------------------------------- */
/* U[n,f] = (1/T) sum_t x(f,t) x(f,t)^H / max(r_n(.,t), eps)
 * (src/bss/ilrma.py:497-511, src/bss/iva.py:493-499, 726-732).
 * r: ASSX_W_NFT -> (B,N,F,T) real; ASSX_W_NT -> (B,N,T) real; ASSX_W_NONE -> NULL, N outputs
 * collapse to 1 (U is (B,1,F,M,M): the plain covariance).  N = number of weight sets. */
int assx_cov_accumulate(assx_ctx* ctx, const void* X, const void* r, int r_kind, double eps, void* U,
                        void* ws, int B, int M, int N, int F, int T, int dtype, void* stream);

/* ---- (a5) iterative projection --------------------------------------------------------- */
/* Gauss-Seidel IP sweep over the N sources, in place on W
 * (src/bss/ilrma.py:512-530, src/bss/iva.py:500-518, 733-751): WU = W U_n; keep the old row
 * unless cond_2(WU) < threshold; w = (WU)^{-1} e_n; W[n,:] = conj(w) / sqrt(w^H U_n w). */
int assx_ip_update(assx_ctx* ctx, const void* U, void* W, double threshold, int32_t* status,
                   int B, int M, int F, int dtype, void* stream);

/* ---- (f1) iterative source steering ------------------------------------------------------ */
/* One ISS sweep over the N sources, in place on W (src/bss/ilrma.py:537-564, src/bss/iva.py:525-542, 758-775),
 * expressed on the weighted covariances U (B,N,F,M,M) instead of on Y:  for n: v_s = w_s U_s w_n^H / w_n U_s w_n^H
 * (s != n), v_n = 1 - 1/sqrt(n_frames * w_n U_n w_n^H);  W[s,:] -= v_s W[n,:]. */
int assx_iss_update(assx_ctx* ctx, const void* U, void* W, int n_frames, int B, int M, int F, int dtype, void* stream);
/* IP2 / pairwise update of rows (pair_m, pair_n), in place on W (src/bss/ilrma.py:599-631, src/bss/iva.py:567-597):
 * P_x = (W U_x)^{-1}[e_m e_n], V_x = P_x^H U_x P_x, generalised 2x2 eigenproblem V_n^{-1} V_m, eigenvectors by descending
 * eigenvalue (zgeev phase convention), cond guard per row. */
int assx_ip2_update(assx_ctx* ctx, const void* U, void* W, double threshold, int32_t* status, int pair_m, int pair_n,
                    int B, int M, int F, int dtype, void* stream);

/* ---- (a2) ILRMA source model ----------------------------------------------------------- */
/* GaussILRMA.update_source_model_basic, non-partitioned (src/bss/ilrma.py:356-366, 409-430):
 * P = |W x|^2 recomputed on the fly; IS-NMF (mm) basis update, then activation update with
 * the new basis.  Tb, V updated in place.  source_mask: bit n set = source n is updated (all ones = every source;
 * two bits = update_source_model_pairwise, src/bss/ilrma.py:432-481).
 * loss_prev: optional (B,) float64 receiving compute_negative_loglikelihood (src/bss/ilrma.py:648-677) of the model
 * AT ENTRY, i.e. the loss the reference records at the end of the previous iteration.  For domain 2 and K <= 4 it
 * is accumulated inside the basis pass (which forms the same y = W x and T V) instead of costing a pass over X. */
int assx_ilrma_source_update(assx_ctx* ctx, const void* X, const void* W, void* Tb, void* V,
                             double domain, double eps, unsigned source_mask, double* loss_prev, void* ws,
                             int B, int M, int F, int T, int K, int dtype, void* stream);

/* ---- (f1) partitioning function: shared bases Tb (B,F,K), activations V (B,K,T), latent Z (B,N,K) ----------
 * GaussILRMA(partitioning=True), domain == 2 (src/bss/ilrma.py:79-95 init, 368-408 updates, 490-495 variance).
 * The model enters every other entry point through its per-source expansion
 *     Teff (B,N,F,K) = Z[n,k] Tb[f,k],   Veff (B,N,K,T) = V[k,t]
 * (pass Teff/Veff as Tb/V to assx_ilrma_spatial_update and assx_ilrma_loss).  Either output may be NULL. */
int assx_ilrma_expand_partitioned(assx_ctx* ctx, const void* Z, const void* Tb, const void* V, void* Teff, void* Veff,
                                  int B, int M, int F, int T, int K, int dtype, void* stream);
/* update_source_model_basic, partitioning branch (src/bss/ilrma.py:368-408): Z (then Z /= Z.sum(axis=0)), Tb, V in
 * place, in that order, each from the model as updated so far.  Teff/Veff are caller-owned scratch of the shapes above
 * and hold the expansion of the UPDATED model on return. */
int assx_ilrma_source_update_partitioned(assx_ctx* ctx, const void* X, const void* W, void* Z, void* Tb, void* V,
                                         void* Teff, void* Veff, double eps, void* ws,
                                         int B, int M, int F, int T, int K, int dtype, void* stream);
/* 'power' normalisation with latent variables (src/bss/ilrma.py:313-320): W[:,n,:] /= a_n; Z' = Z / a_n^2;
 * Tb *= sum_n Z'; Z = Z' / sum_n Z'.  power_bins as for assx_ilrma_normalize_power_bins. */
int assx_ilrma_normalize_power_bins_partitioned(assx_ctx* ctx, void* W, void* Z, void* Tb, const double* power_bins,
                                                double eps, void* ws, int B, int M, int F, int K, int dtype,
                                                void* stream);

/* ---- (a4+a5) ILRMA spatial model ------------------------------------------------------- */
/* GaussILRMA.update_spatial_model_ip (src/bss/ilrma.py:483-535): r = max((Tb V)^(2/domain), eps)
 * rebuilt in-kernel from Tb, V (never materialised), covariance, then the IP sweep (spatial = ASSX_SPATIAL_IP) or
 * the ISS sweep (ASSX_SPATIAL_ISS, src/bss/ilrma.py:537-564) or the pairwise update of rows (pair_m, pair_n)
 * (ASSX_SPATIAL_IP2, src/bss/ilrma.py:566-633; pair_* ignored otherwise).  W in place.
 * U_out: optional (B,N,F,M,M) complex receiving the covariance; NULL = not materialised.
 * C, power_bins: optional pair.  C (B,F,M,M) = plain covariance of X; power_bins (B,N,F) float64 receives
 * w_n^H C_f w_n of the UPDATED filters, the per-bin share of the power-normalisation statistic
 * (src/bss/ilrma.py:298-306) -- emitted by the IP kernel so normalisation needs no further pass or launch. */
int assx_ilrma_spatial_update(assx_ctx* ctx, int spatial, int pair_m, int pair_n, const void* X, void* W, const void* Tb, const void* V,
                              double domain, double eps, double threshold, void* U_out,
                              const void* C, double* power_bins,
                              int32_t* status, void* ws,
                              int B, int M, int F, int T, int K, int dtype, void* stream);

/* Stage 1 of assx_ilrma_spatial_update alone: ONE launch of the covariance-accumulate kernel (packed
 * Hermitian partial sums into ws).  Exposed so a harness can time exactly that kernel with HIP events.
 * n_basis > 4: the launch is cov_wide_kernel + its small finalize (dense U inside ws), or, when the activation
 * tile does not fit LDS, the source-variance map + the (N,F,T)-weights form of the streaming kernel. */
int assx_ilrma_cov_partials(assx_ctx* ctx, const void* X, const void* Tb, const void* V, double domain, double eps,
                            void* ws, int B, int M, int F, int T, int K, int dtype, void* stream);

/* ---- (a6) normalisation ---------------------------------------------------------------- */
/* power[b,n] = mean_{f,t} |(W x)_n|^2  (src/bss/ilrma.py:298-306), one pass over X. */
int assx_demix_power(assx_ctx* ctx, const void* X, const void* W, void* power /* (B,N) real */, void* ws,
                     int B, int M, int F, int T, int dtype, void* stream);
/* Same statistic from the plain covariance C (B,F,M,M): sum_t|y_n|^2 = T w_n^H C w_n; no pass over X. */
int assx_power_from_cov(assx_ctx* ctx, const void* C, const void* W, void* power, void* ws,
                        int B, int M, int F, int dtype, void* stream);
/* 'power' normalisation (src/bss/ilrma.py:304-322): a = max(sqrt(power), eps);
 * W[:,n,:] /= a_n;  Tb[n] /= a_n^domain. */
int assx_ilrma_normalize_power(assx_ctx* ctx, void* W, void* Tb, const void* power, double domain, double eps,
                               int B, int M, int F, int K, int dtype, void* stream);
/* Same normalisation with the statistic still per bin: power_bins (B,N,F) float64 from
 * assx_ilrma_spatial_update; mean power = (1/F) sum_f power_bins[b,n,f].  Reduction fused into the rescale. */
int assx_ilrma_normalize_power_bins(assx_ctx* ctx, void* W, void* Tb, const double* power_bins, double domain,
                                    double eps, int B, int M, int F, int K, int dtype, void* stream);
/* 'projection-back' normalisation (src/bss/ilrma.py:323-330): W[f,n,:] *= s[n,f]; Tb[n,f,:] *= |s[n,f]|^domain. */
int assx_ilrma_normalize_pb(assx_ctx* ctx, void* W, void* Tb, const void* scale /* (B,N,F) complex */, double domain,
                            int B, int M, int F, int K, int dtype, void* stream);

/* ---- (a7) negative log-likelihoods ------------------------------------------------------ */
/* GaussILRMA.compute_negative_loglikelihood (src/bss/ilrma.py:648-677). loss: (B,) float64. */
int assx_ilrma_loss(assx_ctx* ctx, const void* X, const void* W, const void* Tb, const void* V,
                    double domain, double eps, double* loss, void* ws,
                    int B, int M, int F, int T, int K, int dtype, void* stream);

/* ---- (f1) t-ILRMA (src/bss/ilrma.py:713-1020), domain 2, IP ------------------------------------------------
 * Source model (src/bss/ilrma.py:899-922): the IS-NMF updates of assx_ilrma_source_update with P replaced by the
 * harmonic statistic 1 / (2/((2+nu) TV) + nu/((2+nu) P)). */
int assx_tilrma_source_update(assx_ctx* ctx, const void* X, const void* W, void* Tb, void* V, double nu, double eps,
                              void* ws, int B, int M, int F, int T, int K, int dtype, void* stream);
/* Spatial model (src/bss/ilrma.py:926-983): Xi = (nu max(TV, eps) + 2 |W x|^2) / (nu + 2) with the filters before the
 * sweep, U_n = mean_t x x^H / Xi_n, then for every source w = (W U_n)^{-1} e_n (no condition-number guard),
 * W[n] = conj(w) / max(sqrt(w^H U_n w), eps).  Xi: caller-owned scratch (B,N,F,T) reals.  C / power_bins as for
 * assx_ilrma_spatial_update. */
int assx_tilrma_spatial_update(assx_ctx* ctx, const void* X, void* W, const void* Tb, const void* V, double nu,
                               double eps, void* Xi, const void* C, double* power_bins, int32_t* status, void* ws,
                               int B, int M, int F, int T, int K, int dtype, void* stream);
/* tILRMA.compute_negative_loglikelihood (src/bss/ilrma.py:991-1018):
 * sum (1 + nu/2) log(1 + (2/nu) P/R) + log R  -  2 T sum_f log|det W_f|.  loss: (B,) float64. */
int assx_tilrma_loss(assx_ctx* ctx, const void* X, const void* W, const void* Tb, const void* V, double nu, double eps,
                     double* loss, void* ws, int B, int M, int F, int T, int K, int dtype, void* stream);

/* ---- AuxIVA ----------------------------------------------------------------------------- */
/* r[b,n,t] from the current filters (src/bss/iva.py:489-491 Laplace sqrt(sum_f|y|^2),
 * 722-724 Gauss mean_f|y|^2; not floored), and in the same pass the data term of
 * compute_negative_loglikelihood for those filters (iva.py:604-619, 783-802) completed with
 * -2 T sum_f log|det W_f|.  loss may be NULL. */
int assx_auxiva_weights(assx_ctx* ctx, const void* X, const void* W, int kind, double eps,
                        void* r /* (B,N,T) real */, double* loss /* (B,) or NULL */, void* ws,
                        int B, int M, int F, int T, int dtype, void* stream);
/* update_once_ip / update_once_iss given r (src/bss/iva.py:493-518, 525-542, 726-751, 758-775): covariance with
 * (N,T) weights + IP or ISS sweep. */
int assx_auxiva_spatial_update(assx_ctx* ctx, int spatial, int pair_m, int pair_n, const void* X, void* W, const void* r, double eps, double threshold,
                               void* U_out, int32_t* status, void* ws,
                               int B, int M, int F, int T, int dtype, void* stream);

/* ---- (f4) the other callers of covariance-accumulate + IP --------------------------------------------------- */
/* GaussIDLMA.update_space_model (src/sss/idlma.py:175-210): R = dnn_output^(2/domain) floored at eps, U_n = mean_t
 * x x^H / R_n, then the IP sweep of assx_ip_update on W.  dnn_output (B,N,F,T) real is the caller's source-variance
 * estimate (in the reference: a DNN's output; the network itself is out of scope).  R_scratch: caller-owned
 * (B,N,F,T) reals, used only when domain != 2 (may be NULL otherwise). */
int assx_idlma_space_update(assx_ctx* ctx, const void* X, void* W, const void* dnn_output, double domain, double eps,
                            double threshold, void* R_scratch, int32_t* status, void* ws,
                            int B, int M, int F, int T, int dtype, void* stream);
/* FastMultichannelISNMF.update_diagonalizer (src/bss/mnmf.py:848-888): R[f,t,m] = sum_n Lambda[n,f,t] g[n,f,m]
 * floored at eps, V_m = mean_t x x^H / R[.,.,m], then for every channel m: q = (Q V_m)^{-1} e_m,
 * Q[m,:] = conj(q) / max(sqrt(q^H V_m q), eps) unless cond_2(Q V_m) >= threshold.  Q (B,F,M,M) complex in place;
 * Lambda (B,N,F,T) real = the sources' NMF variances (basis @ activation); g (B,N,F,M) real = spatial_covariance;
 * N = n_sources is free (it need not equal M).  R_scratch: caller-owned (B,M,F,T) reals. */
int assx_fastmnmf_update_diagonalizer(assx_ctx* ctx, const void* X, void* Q, const void* Lambda, const void* g,
                                      double eps, double threshold, void* R_scratch, int32_t* status, void* ws,
                                      int B, int M, int N, int F, int T, int dtype, void* stream);

/* ---- (a8) projection back --------------------------------------------------------------- */
/* projection_back(Y, reference) for a 2-D reference (src/algorithm/projection_back.py:13-21) with
 * Y = W X formed on the fly and reference = X[ref]:  scale[b,n,f] = (x_ref Y^H (Y Y^H)^{-1})[n]. */
int assx_projection_back_scale(assx_ctx* ctx, const void* X, const void* W, int ref, void* scale /* (B,N,F) complex */,
                               int32_t* status, void* ws, int B, int M, int F, int T, int dtype, void* stream);
/* General form on a materialised Y (B,N,F,T) and an explicit reference (B,F,T). */
int assx_projection_back(assx_ctx* ctx, const void* Y, const void* reference, void* scale,
                         int32_t* status, void* ws, int B, int N, int F, int T, int dtype, void* stream);

/* ---- (a9) least-squares demixing filter --------------------------------------------------- */
/* ILRMAbase.compute_demix_filter / IVAbase.compute_demix_filter (src/bss/ilrma.py:167-173, src/bss/iva.py:119-125):
 * W[b,f] = (Y X^H)(X X^H)^{-1} per bin, Y (B,M,F,T) an estimate, X (B,M,F,T) the mixture, W (B,F,M,M).  This is how
 * the reference rebuilds `demix_filter` from `estimation` for callbacks / the loss / the output of its ISS loop.
 * 2 <= M <= 32.  An exactly singular X X^H sets ASSX_STATUS_SINGULAR (numpy.linalg.inv would raise LinAlgError). */
int assx_compute_demix_filter(assx_ctx* ctx, const void* Y, const void* X, void* W, int32_t* status,
                              int B, int M, int F, int T, int dtype, void* stream);

/* ---- (a1) NMF multiplicative updates ----------------------------------------------------- */
/* EUCNMF/KLNMF/ISNMF.update_once_mm, ISNMF.update_once_me (src/algorithm/nmf.py:182-207,
 * 241-266, 302-356).  X (B,F,T) real >= 0, Tb (B,F,K), V (B,K,T); Tb, V updated in place. */
size_t assx_nmf_workspace_bytes(int B, int F, int T, int K, int dtype);
/* Host-side query (no GPU): the work partition of one matrix-core half update and the slab area the workspace holds
 * for it.  feed 0 = the map-fed halves of assx_nmf_update (group = matrices per independent problem, 1 for plain
 * NMF), 1 = the X-fed halves of the ILRMA source model (5 <= n_basis <= 32, M <= 4; `group` unused); half 0 = basis,
 * 1 = activation.  out[0] = workgroups G, out[1] = blocks, out[2] = steps per block, out[3] = the slab bound the
 * launchers use, out[4] = the largest number of workgroups that actually meet one block (enumerated: each writes
 * its own slab), out[5] = slabs per block that fit the area assx_nmf_workspace_bytes reserves.  out[4] <= out[3] <=
 * out[5] must hold for every shape (tests/test_cabi_and_host.py sweeps it: round 4 sized the area without the X-fed
 * partitions).  Returns 0, or ASSX_E_ARG.  No reference counterpart (the reference is NumPy). */
int assx_nmf_partition_query(int feed, int half, int group, int F, int T, int K, int dtype, int32_t out[6]);
int assx_nmf_update(assx_ctx* ctx, int kind, double domain, double eps, const void* X, void* Tb, void* V,
                    void* ws, int B, int F, int T, int K, int dtype, void* stream);
/* criterion((Tb V)^(2/domain), X).sum() (src/algorithm/nmf.py:170-174, 229-233, 288-292;
 * src/criterion/divergence.py:21-45).  kind IS_ME uses the IS criterion. loss: (B,) float64. */
int assx_nmf_loss(assx_ctx* ctx, int kind, double domain, double eps, const void* X, const void* Tb, const void* V,
                  double* loss, void* ws, int B, int F, int T, int K, int dtype, void* stream);
/* The same two calls for every kind of the enum, with the kind's extra parameter (ASSX_NMF_T: nu > 0; others: unused).
 * tNMF / CauchyNMF floor exactly where the reference does (mm_fast and me leave T V itself unfloored); their losses
 * are t_divergence (nmf.py:369-373) and cauchy_divergence (nmf.py:435-443) of (T V + eps, X + eps). */
int assx_nmf_update_ex(assx_ctx* ctx, int kind, double domain, double param, double eps, const void* X, void* Tb,
                       void* V, void* ws, int B, int F, int T, int K, int dtype, void* stream);
int assx_nmf_loss_ex(assx_ctx* ctx, int kind, double domain, double param, double eps, const void* X, const void* Tb,
                     const void* V, double* loss, void* ws, int B, int F, int T, int K, int dtype, void* stream);

/* ---- whole loops in ONE call ------------------------------------------------------------------------------------
 * The reference's drivers are Python loops around update_once() (src/algorithm/nmf.py:45-53, src/bss/iva.py:420-441,
 * src/bss/ilrma.py:233-256).  With kernels of 5-50 us a host-language loop -- one FFI call per stage, 4-7 per
 * iteration -- is the ceiling of the small configurations, so the loop itself is offered behind the boundary: these
 * calls enqueue every launch of `n_iter` iterations on `stream` and return; nothing is synchronised.  They run exactly
 * the entry points above in the order the classes call them, so model(X, iteration=k) equals k x update_once() bit
 * for bit.  The host classes use them when no callback has to run between iterations.
 *
 * assx_nmf_iterate: n_iter x { assx_nmf_update_ex ; loss[i] = criterion of the updated model } (NMFbase.update,
 *   nmf.py:45-53).  loss: device float64 (n_iter, B), or NULL = the criterion is not evaluated (an extension: the
 *   reference always records it).  The model is exactly that of n_iter x assx_nmf_update_ex; loss[i] equals
 *   assx_nmf_loss_ex up to the order of summation: for domain 2 and the EUC / KL / IS rules on the matrix-core path it is
 *   accumulated inside update i + 1 (which reads the very model it is the criterion of) instead of in a pass of its own. */
int assx_nmf_iterate(assx_ctx* ctx, int n_iter, int kind, double domain, double param, double eps, const void* X,
                     void* Tb, void* V, double* loss, void* ws, int B, int F, int T, int K, int dtype, void* stream);
/* assx_auxiva_iterate: the loop of AuxIVAbase.__call__ (iva.py:420-441) for one contrast `kind`:
 *   n_iter x { r, loss[i] = assx_auxiva_weights(W) ; assx_auxiva_spatial_update(spatial, pair, r) },
 *   then loss[n_iter] of the final filters.  loss: device float64 (n_iter + 1, B) -- entry 0 is the loss before the
 *   first iteration, as the reference records it -- or NULL.  r: caller-owned (B,N,T) reals; on return it holds the
 *   weights of the final filters when loss != NULL, those of the last iteration's input filters otherwise.
 *   spatial = ASSX_SPATIAL_IP2: (pair_m, pair_n) is the pair of the FIRST iteration; every iteration advances both
 *   by one modulo N (iva.py:370-382).  Ignored otherwise. */
int assx_auxiva_iterate(assx_ctx* ctx, int n_iter, int kind, int spatial, int pair_m, int pair_n, const void* X,
                        void* W, double eps, double threshold, void* r, double* loss, int32_t* status, void* ws,
                        int B, int M, int F, int T, int dtype, void* stream);
/* assx_ilrma_iterate: the loop of GaussILRMA.__call__ (ilrma.py:233-256) without a partitioning function:
 *   n_iter x { assx_ilrma_source_update (all sources; the two of the pair for IP2) ; assx_ilrma_spatial_update ;
 *              normalisation },  normalize = 0: none; 1: 'power' with the statistic from the plain covariance
 *   (C (B,F,M,M) given, power_bins (B,N,F) scratch: assx_ilrma_normalize_power_bins, ilrma.py:304-322); 2:
 *   'projection-back' (assx_projection_back_scale with reference channel `ref` into `scale` (B,N,F) complex scratch +
 *   assx_ilrma_normalize_pb with basis exponent `pb_exponent`, ilrma.py:323-330).
 *   loss: device float64 (n_iter + 1, B) or NULL: entry i is compute_negative_loglikelihood (ilrma.py:648-677) of the
 *   model after i iterations; entries 0 .. n_iter-1 ride on the basis pass of the following iteration (loss_prev of
 *   assx_ilrma_source_update), the last one costs a pass of its own. */
int assx_ilrma_iterate(assx_ctx* ctx, int n_iter, int spatial, int pair_m, int pair_n, int normalize, int ref,
                       double pb_exponent, const void* X, void* W, void* Tb, void* V, double domain, double eps,
                       double threshold, const void* C, double* power_bins, void* scale, double* loss,
                       int32_t* status, void* ws, int B, int M, int F, int T, int K, int dtype, void* stream);

/* ---- (f2) pieces of an iteration for the F-SHARDED single-utterance mode ------------------------------------------
 * Bins are independent in every step of GaussILRMA.update_once except the activation update, which reduces over f
 * (src/bss/ilrma.py:421-428), and the power / loss statistics, which reduce over (f, t) (ilrma.py:304-307, 648-677).
 * A rank that owns a contiguous block of bins runs the ordinary entry points on its block (X, W, Tb sliced to the
 * block; V whole) and needs only these extra pieces; the host all-reduces `sums` of the activation half (2 N K T
 * reals) and N power scalars per iteration (audio_source_separation_amd/bss/ilrma_fshard.py).
 *   assx_ilrma_power_map: P (B,N,F,T) real = |W x|^2 (ilrma.py:356-366) -- the target of the source-model NMF.
 *   assx_nmf_half_sums:   one half of a multiplicative update stopped before it is applied: half 0 = basis (reduce
 *                         over t), sums (2, B, F*K); half 1 = activation (reduce over f, with the CURRENT Tb), sums
 *                         (2, B, K*T); sums[0] numerators, sums[1] denominators (not floored).  n_basis <= 64.
 *   assx_nmf_apply_sums:  A[b][i] *= (num / max(den, eps))^e with the kind's exponent (nmf.py:317,325 etc.).
 * X (B,F,T) as for assx_nmf_update; for the ILRMA source model B = utterances x sources and X = P. */
int assx_ilrma_power_map(assx_ctx* ctx, const void* X, const void* W, void* P, int B, int M, int F, int T, int dtype,
                         void* stream);
int assx_nmf_half_sums(assx_ctx* ctx, int kind, double domain, double param, double eps, int half, const void* X,
                       const void* Tb, const void* V, void* sums, void* ws, int B, int F, int T, int K, int dtype,
                       void* stream);
int assx_nmf_apply_sums(assx_ctx* ctx, int kind, double domain, double eps, void* A, const void* sums, int B,
                        long long count, int dtype, void* stream);
/* out[i] = sum_{s=0}^{S-1} weights[s] * parts[s][i], added in ascending s (weights: device float64[S] or NULL = ones;
 * parts (S, count), out (count) of `dtype`).  The combination step after an all-gather of per-shard partials: every
 * rank gets the same bits, and the same bits as one process that holds all the shards. */
int assx_ordered_sum(assx_ctx* ctx, const void* parts, const double* weights, void* out, int S, long long count,
                     int dtype, void* stream);

/* ---- (f3) STFT / iSTFT either side of the loop ---------------------------------------------- */
/* stft / istft of src/transform/stft.py:4-17, i.e. scipy.signal.stft / istft with nperseg = fft_size,
 * noverlap = fft_size - hop, boundary='zeros', padded=True, detrend=False, scaling='spectrum', one-sided:
 *   X[c,f,t] = rfft(window * x_padded[c, t*hop : t*hop + fft_size])[f] / window_sum
 *   y        = weighted overlap-add of window * irfft(X[c,:,t]) * window_sum / sum_t window^2, boundary removed.
 * x (C, n_samples) real; window (fft_size,) real on the device, window_sum = its sum; X (C, fft_size/2+1, n_frames)
 * complex, frames fastest -- the layout the separation loop reads; y (C, assx_istft_num_samples(...)) real.
 * n_frames must equal assx_stft_num_frames(n_samples, fft_size, hop) (the count scipy produces); n_samples >=
 * fft_size (scipy shrinks the window for shorter signals: refused here).  Any fft_size whose frame fits LDS; powers
 * of two up to 8192 take the in-LDS FFT, the rest a direct DFT. */
long long assx_stft_num_frames(long long n_samples, int fft_size, int hop);
long long assx_istft_num_samples(int fft_size, int hop, int n_frames);
size_t assx_stft_workspace_bytes(int C, int fft_size, int n_frames, int dtype);
int assx_stft(assx_ctx* ctx, const void* x, const void* window, double window_sum, void* X, void* ws, int C,
              long long n_samples, int fft_size, int hop, int n_frames, int dtype, void* stream);
int assx_istft(assx_ctx* ctx, const void* X, const void* window, double window_sum, void* y, void* ws, int C,
               int fft_size, int hop, int n_frames, int dtype, void* stream);

/* ---- host <-> HBM staging of the call's input / output ---------------------------------------------------------- */
/* The reference's __call__ takes a pageable NumPy array and returns one (src/bss/ilrma.py:203-273, src/bss/iva.py:
 * 289-479, src/algorithm/nmf.py:22-53): these two calls are that edge.  `host` is ordinary (pageable) host memory,
 * `dev` a device array of `count` REAL elements (a complex array of n samples is 2n reals); host_dtype / dev_dtype are
 * ASSX_F32 / ASSX_F64 and may differ -- the conversion happens on the host inside the staging copy, so the float32
 * mode moves half the bytes over PCIe and the caller never makes a converted host copy.  The array rides a ring of
 * pinned buffers owned by the context (chunked, host copy threads overlap the DMA of the previous chunk).
 * assx_upload returns once `host` has been consumed (it may be freed / overwritten); the tail of the DMA is ordered
 * before later work on `stream`, not waited for by the host.  assx_download first orders itself after the work already
 * queued on `stream` and returns when `host` is complete.  Environment: ASSX_XFER_THREADS, ASSX_XFER_CHUNK_MB. */
int assx_upload(assx_ctx* ctx, const void* host, int host_dtype, void* dev, int dev_dtype, size_t count, void* stream);
int assx_download(assx_ctx* ctx, const void* dev, int dev_dtype, void* host, int host_dtype, size_t count, void* stream);

/* ---- (e) multi-GPU edges: utterance sharding over the GPUs of one node ------------------------------------------- */
/* No reference counterpart: the reference is a single-process NumPy program (src/bss/ilrma.py:203-273 -- every __call__ owns
 * all of its state, hence utterances shard with no data-path collective; SURVEY.md 8e, 8b "assx_comm_init(ndev),
 * assx_scatter/gather").  One process per GPU; the only traffic is root -> ranks (the mixtures) and ranks -> root (the
 * separated outputs), each ONE grouped batch of ncclSend / ncclRecv on contiguous row blocks of the root's array: root <->
 * 7 peers = 7 concurrent xGMI links, no ring, no staging copy, ragged blocks need no padding.  RCCL (librccl.so) is loaded
 * by the first of these calls, libassx.so does not link it: on a machine without RCCL they return ASSX_E_UNSUPPORTED and
 * everything else works.
 *   assx_shard_range    the static block partition (host-side, no GPU): item i of n_items belongs to exactly one rank, block
 *                       sizes differ by at most one, the first n_items % world ranks hold one more (64 over 8 -> 8 each);
 *                       the same partition as audio_source_separation_amd.distributed.shard_range.
 *   assx_comm_unique_id on ONE process: ASSX_COMM_ID_BYTES bytes (ncclGetUniqueId) that the host hands to every rank by its
 *                       own means (MPI_Bcast, a file, a socket) before
 *   assx_comm_init      on every rank, the context's device current: ncclCommInitRank.  A communicator belongs to the
 *                       context it was made with (errors are reported through assx_last_error(ctx)); world = 1 is valid.
 *   assx_scatter        root: `all` = (n_items, item_bytes) on its device; every rank (the root too) receives its own block
 *                       into `local` ((hi - lo) * item_bytes; may alias its place in `all` on the root: then nothing moves).
 *   assx_gather         the mirror image: every rank's `local` block lands at its place in the root's `all`.
 * Both are asynchronous on `stream` like every other entry point; a rank with an empty block posts nothing; RCCL errors
 * come back as 1000 + ncclResult_t with the message in the context. */
#define ASSX_COMM_ID_BYTES 128
typedef struct assx_comm assx_comm;
void assx_shard_range(size_t n_items, int world, int rank, size_t* lo, size_t* hi);
int assx_comm_unique_id(void* id);
int assx_comm_init(assx_ctx* ctx, int world, int rank, const void* id, assx_comm** comm);
int assx_comm_destroy(assx_comm* comm);
int assx_scatter(assx_comm* comm, int root, const void* all, void* local, size_t n_items, size_t item_bytes, void* stream);
int assx_gather(assx_comm* comm, int root, const void* local, void* all, size_t n_items, size_t item_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ASSX_H */
