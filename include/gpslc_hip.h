/*
 * gpslc_hip.h — C ABI of libgpslc_hip.so, the MI355X (gfx950) implementation of the
 * CausalGPSLC.jl GP-kernel + posterior-prediction hot path.
 *
 * Every entry point replaces the body of one Julia function of the reference (cited per
 * function, paths relative to the reference repository).  The Julia side keeps its
 * signatures and `ccall`s these (INTEGRATION.md shows the shim); in this repository a
 * Python ctypes harness (causalgpslc.jl_amd/_lib.py) binds exactly the same symbols.
 *
 * Conventions
 *   - all matrices are COLUMN-MAJOR (Julia layout), all reals are IEEE double;
 *   - Bool treatments are pre-converted by the caller to 0.0 / 1.0 (Julia promotes Bool on
 *     subtraction, src/kernel.jl:17);
 *   - return value: 0 = ok; < 0 = error (GPSLC_ERR_*; -1..-99 = "argument #k is invalid");
 *     > 0 = 1-based index of the pivot at which a Cholesky factorisation broke down
 *     (the Julia shim rethrows it as PosDefException(info), mirroring PDMats);
 *   - no C++ exception, abort or longjmp crosses this boundary;
 *   - the caller owns every host/device buffer it passes; the library keeps no pointer to
 *     caller memory after a call returns; device workspace belongs to the ctx;
 *   - calls on one ctx must be serialised by the caller; use one ctx per GPU / per thread.
 *   - "_dev" variants take DEVICE pointers (hipMalloc'ed / torch / AMDGPU.jl memory on the
 *     ctx's device) for every array argument; the plain variants take HOST pointers.
 *   - stream contract of the "_dev" variants: the library works on private non-blocking HIP streams of
 *     the ctx.  The caller must have completed (synchronised) whatever produced its device inputs
 *     before the call; when the call returns every output is complete and visible to any stream.
 *   - thread safety: distinct ctxs may be used concurrently from distinct threads (also on the same
 *     GPU); the library keeps no process-global mutable state besides per-device "attribute applied"
 *     bits, which are atomic.  The environment is never consulted by this library.
 */
#ifndef GPSLC_HIP_H
#define GPSLC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpslc_ctx gpslc_ctx;

#define GPSLC_OK                 0
#define GPSLC_ERR_HIP        (-1000)  /* a HIP runtime call failed; see gpslc_last_error  */
#define GPSLC_ERR_NOMEM      (-1001)  /* device workspace could not be allocated          */
#define GPSLC_ERR_NODATA     (-1002)  /* gpslc_set_data has not been called               */
#define GPSLC_ERR_NODEVICE   (-1003)  /* no usable gfx950 device                          */
#define GPSLC_ERR_INTERNAL   (-1004)
#define GPSLC_ERR_IO         (-1005)  /* posterior pack: file cannot be opened / read / written   */
#define GPSLC_ERR_FORMAT     (-1006)  /* posterior pack: bad magic, truncated or trailing bytes   */
#define GPSLC_ERR_UNSUPPORTED (-1007) /* the entry point does not cover this problem size / mode  */

/* flags for gpslc_create */
#define GPSLC_FLAG_DEFAULT            0u
#define GPSLC_FLAG_PROFILE            1u   /* record HIP events around the dominant kernels */
#define GPSLC_FLAG_FP32_KERNEL        2u   /* mixed precision (BASELINE config 5): RBF distances and exp in
                                              fp32, Gram matrix / Cholesky / solves in fp64                  */

/* ---- context ------------------------------------------------------------------------- */

/* One context per (GPU, data set).  n = instances, nX = covariate columns (0 = model without
 * X), nU = latent confounder columns (0 = model without U).  Plays the role of the data part
 * of GPSLCObject (src/types.jl:249-258). */
int gpslc_create(gpslc_ctx** out, int device, int64_t n, int32_t nX, int32_t nU, uint32_t flags);
int gpslc_destroy(gpslc_ctx* ctx);

/* g.X (n x nX, may be NULL when nX == 0), g.T (n), g.Y (n): src/types.jl:249-258. */
int gpslc_set_data(gpslc_ctx* ctx, const double* X, const double* T, const double* Y);
int gpslc_set_data_dev(gpslc_ctx* ctx, const double* X, const double* T, const double* Y);

/* Tuning knobs (0 = keep default): max posterior samples factorised concurrently, tile-panel
 * width of the blocked Cholesky, number of HIP streams chunks are spread over.  A panel width given here
 * also bounds the persistent task launch (gpslc_set_task_schedule below): matrices of more tiles per side
 * than panel_tiles then take the panel schedule (left-looking panels + trailing updates) instead of
 * being factorised as ONE panel of tasks. */
int gpslc_set_tuning(gpslc_ctx* ctx, int32_t max_batch, int32_t panel_tiles, int32_t n_streams);

/* Schedule of the factorisation of A at small tile counts (round 6; no reference counterpart: the reference factorises
 * one matrix at a time, src/likelihood.jl:42-43, src/estimation.jl:46).  Chunks of at least min_matrices matrices of
 * min_tiles .. max_tiles tiles of 128 per side (and at most panel_tiles wide: one left-looking panel) are factorised — and,
 * where MeanITE is wanted, back-substituted (src/estimation.jl:46's CovWWp \ Y) — by ONE persistent launch of tile tasks:
 * diagonal-tile, strip and back-substitution tasks of many matrices in flight at once, dependencies through per-matrix
 * progress words, instead of one launch per tile column.  Every output is bit-identical either way.
 * min_tiles: <= 0 = keep (default 2: N > 128); max_tiles: 0 = always one launch per column, negative = keep (default and
 * limit 32: N <= 4096 — beyond that, and beyond one panel of gpslc_set_tuning's panel_tiles when that was given, the panel
 * schedule is kept: left-looking panels of per-column launches + one trailing update per panel); min_matrices: <= 0 = keep (default 256: a persistent launch
 * over a few matrices is a chain of hand-offs — the single scores of an MH step keep the per-column launches); group:
 * matrices per group of the task order, <= 0 = keep (default 32).  Returns 0, or minus the number of the offending argument. */
int gpslc_set_task_schedule(gpslc_ctx* ctx, int32_t min_tiles, int32_t max_tiles, int32_t min_matrices, int32_t group);

/* Placement of a call's posterior samples inside a larger ensemble, for the library's own normals (z_or_null == NULL):
 * after gpslc_set_ensemble(ctx, s_off, S_total) sample s, level l of gpslc_predict[_dev] draws from the Philox stream
 * (s_off + s) + S_total * l instead of s + S * l.  A rank of a sharded prediction that holds the samples [s0, s1) of
 * S_total (the partition of the loop src/estimation.jl:78-84 under src/prediction.jl:30-33) sets (s0, S_total) and then
 * generates exactly the normals a single-process call over all S_total samples generates for them — the result does not
 * depend on the number of ranks.  S_total = 0 restores the default (offset 0, the call's own S). */
int gpslc_set_ensemble(gpslc_ctx* ctx, int64_t sample_offset, int64_t S_total);

const char* gpslc_last_error(const gpslc_ctx* ctx);

/* ---- src/kernel.jl ------------------------------------------------------------------- */

/* rbfKernelLog(X1, X2, LS) (src/kernel.jl:24-32; vectors are d = 1):
 * out[i + n*ip] = -sum_k (X1[i,k] - X2[ip,k])^2 / LS[k]^2, ls_len = 1 (scalar LS) or d. */
int gpslc_rbf_log(gpslc_ctx* ctx, const double* X1, const double* X2, int64_t n, int32_t d,
                  const double* ls, int32_t ls_len, double* out);
/* device pointers for X1, X2, ls and out: no allocation, no copy (src/inference.jl:225-227, 286-287, 343-344
 * call rbfKernelLog / processCov once per outer MCMC iteration) */
int gpslc_rbf_log_dev(gpslc_ctx* ctx, const double* X1, const double* X2, int64_t n, int32_t d,
                      const double* ls, int32_t ls_len, double* out);

/* processCov(logCov, scale[, noise]) (src/kernel.jl:53-55, 57-59): out = exp.(logcov)*scale
 * + noise*I.  The two-argument method is noise = 0.0. */
int gpslc_process_cov(gpslc_ctx* ctx, const double* logcov, int64_t n, double scale, double noise,
                      double* out);
int gpslc_process_cov_dev(gpslc_ctx* ctx, const double* logcov, int64_t n, double scale, double noise,
                          double* out);   /* device pointers; out may alias logcov */

/* ---- src/model_likelihood.jl :Y node -------------------------------------------------- */

/* logpdf(MvNormal(0, Symmetric(Ycov)), Y) with Ycov = processCov(uyCovLog .+ xyCovLog .+
 * tyCovLog, yScale, yNoise): generateYfromUXT / UT / XT / T (src/model_likelihood.jl:83-91,
 * 94-101, 104-111, 114-120).  U: n x nU (ignored when nU == 0).  X_or_null: overrides the ctx's
 * X for this call (the trace's :X => k => :X values; n x nX) or NULL to use gpslc_set_data's.
 * Y_or_null: the value being scored (n) — what Gen hands to a Distribution's logpdf — or NULL for
 * the ctx's Y (the constrained observation, src/model_likelihood.jl:89).  Uses the ctx's T.
 * S independent parameter sets are evaluated per call (S = 1 for one Gen `update`); U is
 * n x nU x S, uyLS nU x S, xyLS nX x S, the rest length S. */
int gpslc_y_logpdf(gpslc_ctx* ctx, int64_t S, const double* U, const double* X_or_null,
                   const double* Y_or_null, const double* uyLS, const double* xyLS, const double* tyLS,
                   const double* yScale, const double* yNoise, double* logpdf /* S */);

/* ---- the other Gaussian-process nodes of the Gen models -------------- */

/* log N(target; 0, scale * exp.(rbfKernelLog(F, F, ls)) + noise * I) for S parameter sets: the score of
 *   :X => k => :X   generateXfromU    F = U,       ls = uxLS[k, :]           (src/model_likelihood.jl:13-22)
 *   :T / :logitT    generate*TfromUX  F = [U | X], ls = [utLS ; xtLS]        (src/model_likelihood.jl:25-80)
 *   :Y              generateYfrom*    F = [U | X | T]                        (:83-120; gpslc_y_logpdf is the
 *                                                                             specialised form that shares X, T, Y)
 * F is n x nF x S (f_shared = 0) or n x nF shared by all sets (f_shared = 1); ls nF x S; scale, noise S;
 * target n x S (t_shared = 0) or n (t_shared = 1).  nF <= 32. */
int gpslc_gp_logpdf(gpslc_ctx* ctx, int64_t S, int32_t nF, const double* F, int32_t f_shared,
                    const double* ls, const double* scale, const double* noise, const double* target,
                    int32_t t_shared, double* logpdf /* S */);

/* The fused whole-model score: `count` independent Gaussian-process nodes, each with its OWN feature block, in one
 * call — what one Gen `update` re-scores (src/model.jl:11-131): the nX `:X => k => :X` nodes, `:T` / `:logitT` and
 * `:Y` (F = [U | X | T], ls = [uyLS ; xyLS ; tyLS]) of a proposal, or the nodes of several proposals at once.
 * logpdf[i] = log N(target_i; 0, scale_i * exp.(rbfKernelLog(F_i, F_i, ls_i)) + noise_i * I); host pointers.
 * While the n x n matrix fits one CU's LDS (n <= 160 for any nF <= 32, up to n = 176 for nF <= 16) ALL nodes are
 * scored by ONE kernel launch, one workgroup per node (Gram build, Cholesky, forward solve and reductions never
 * leave the CU); gpslc_gp_logpdf and gpslc_y_logpdf take the same path at those sizes.  Larger n: ONE batched pass of
 * the general tiled path over all nodes (feature blocks zero-padded to the widest node with lengthscale-1 columns, which
 * add +0.0 to every squared distance: scores bit-identical to node-by-node calls; 3 nodes cost 1.09 x one at n = 4096).
 * Return value and gpslc_last_info (count entries) as for gpslc_gp_logpdf. */
typedef struct gpslc_node {
    int32_t nF;            /* feature columns, 0..32 */
    int32_t reserved;
    const double* F;       /* n x nF, column-major */
    const double* ls;      /* nF */
    double scale, noise;
    const double* target;  /* n */
} gpslc_node;
int gpslc_nodes_logpdf(gpslc_ctx* ctx, int32_t count, const gpslc_node* nodes, double* logpdf /* count */);

/* Draws from the nodes' priors: draws[:, i] = chol(K_i) * target_i with K_i the node's covariance as above and
 * target_i a vector of standard normals supplied by the caller (the host keeps its random-number stream) — what Gen's
 * `mvnormal(zeros(n), cov)` does inside `elliptical_slice(trace, addr, mu, cov)` (src/inference.jl:48-54, 92-98, 232,
 * 348; covariances built at src/inference.jl:225-227, 286-287, 343-344) and inside `generate` for the prior draws.
 * Same single launch as gpslc_nodes_logpdf while the single-workgroup kernels cover the call (n <= 640, count <= 512: the
 * factor never leaves the CU / its L2 scratch); beyond — larger n, more nodes, the fp32 kernel mode — the batched tiled
 * factorisation of every node's covariance followed by L z on the predictive-draw kernel (round 6).  logpdf_or_null, when
 * given, receives log N(target_i; 0, K_i) as a by-product.  Return value / gpslc_last_info as above. */
int gpslc_nodes_draw(gpslc_ctx* ctx, int32_t count, const gpslc_node* nodes, double* draws /* n x count */,
                     double* logpdf_or_null /* count */);

/* log N(x_s; 0, covscale_s * cov) for S vectors and one dense n x n covariance: the :U => u => :U nodes
 * (generateUfromSigmaU, src/model_likelihood.jl:4-10 with uCov = SigmaU * uNoise; generateU,
 * src/model_prior.jl:27-30).  A non-NULL cov is handed over and cached in the ctx (SigmaU is constant for a data
 * set) — as the matrix itself for n <= 640 (every evaluation scales and factorises it inside one workgroup), as its
 * tiled factor beyond; cov = NULL re-uses it.  S = 0 with a non-NULL cov just hands over and validates (returns the
 * failing pivot if cov is not positive definite).  covscale may be NULL (= 1).  SigmaU is positive definite only by
 * its 1e-13 jitter (src/utils.jl:17-33): every factorisation and solve here is substitution-based (no products with
 * inverted blocks), i.e. backward stable like LAPACK's potrf / trsm. */
int gpslc_mvn_logpdf(gpslc_ctx* ctx, int64_t S, const double* cov, const double* covscale /* S */,
                     const double* x /* n x S */, double* logpdf /* S */);

/* draws[:, s] = chol(covscale_s * cov) z[:, s] for the host's standard normals z: Gen's `mvnormal(zeros(n), uCov)` inside
 * `elliptical_slice(trace, :U => k => :U, zeros(n), uCov)` (uCov = SigmaU * uNoise, src/inference.jl:48-54, 92-98, 233-239,
 * 293-299) and the prior draw of generateUfromSigmaU (src/model_likelihood.jl:4-10).  cov as for gpslc_mvn_logpdf: non-NULL =
 * handed over and cached, NULL = the cached one (the one gpslc_mvn_logpdf holds for the data set's SigmaU); chol(s C) =
 * sqrt(s) chol(C), so one factor serves every uNoise.  n <= 640: the node kernels' draw mode; beyond: the cached tiled
 * factor streamed once per vector by the predictive-draw kernel.  Returns 0, the failing pivot of cov, or a negative status. */
int gpslc_mvn_draw(gpslc_ctx* ctx, int64_t S, const double* cov, const double* covscale /* S */,
                   const double* z /* n x S */, double* draws /* n x S */);

/* ---- src/estimation.jl, src/driver.jl, src/prediction.jl ------------------------------ */

/* The ensemble driver: everything sampleITE / sampleSATE / predictCounterfactualEffects
 * (src/driver.jl:86-89, 108-111; src/prediction.jl:23-36) compute for S posterior samples
 * (the rows extractParameters, src/utils.jl:92-124, would return for nBurnIn:stepSize:nOuter)
 * and L intervention levels doT[0..L).
 *
 *   U       n x nU x S      uyLS  nU x S      xyLS  nX x S      tyLS, yScale, yNoise  S
 *   pred_noise  = hyperparams.predictionCovarianceNoise (src/estimation.jl:82)
 *
 * Outputs (any may be NULL):
 *   meanSATE, varSATE   S x L   (sample index fastest)   conditionalSATE, src/estimation.jl:116-121,
 *                               of Symmetric(CovITE) + pred_noise*I (src/estimation.jl:82, 127-140)
 *   meanITE             n x S x L                         conditionalITE mean, src/estimation.jl:46
 *   ite_draws           L x n x (S*spp), level fastest    predictCounterfactualEffects' `ite`
 *                               (src/prediction.jl:30-33); column order sample-outer/draw-inner
 *                               (src/estimation.jl:100-107)
 * Draws use z_or_null (n x spp x S x L standard normals: z[i + n*(d + spp*(s + S*l))]) when given,
 * else the library's Philox4x32-10 + Box-Muller stream seeded by `seed` (documented in DESIGN.md,
 * restated in oracle/gpslc_oracle.py: stream id = s + S*l, element = i + n*d). */
int gpslc_predict(gpslc_ctx* ctx, int64_t S, const double* U, const double* uyLS,
                  const double* xyLS, const double* tyLS, const double* yScale,
                  const double* yNoise, int32_t L, const double* doT, double pred_noise,
                  int32_t spp, uint64_t seed, const double* z_or_null,
                  double* meanSATE, double* varSATE, double* meanITE, double* ite_draws);
int gpslc_predict_dev(gpslc_ctx* ctx, int64_t S, const double* U, const double* uyLS,
                      const double* xyLS, const double* tyLS, const double* yScale,
                      const double* yNoise, int32_t L, const double* doT, double pred_noise,
                      int32_t spp, uint64_t seed, const double* z_or_null,
                      double* meanSATE, double* varSATE, double* meanITE, double* ite_draws);

/* The same call sharded over several GPUs of one node: what the loop of predictCounterfactualEffects (src/prediction.jl:30-33)
 * over the posterior samples (src/estimation.jl:78-84) becomes when the ensemble is partitioned.  ctxs[0..nctx) are
 * DISTINCT contexts created with the same (n, nX, nU), one per device (gpslc_create(&ctx_k, device_k, ...)), each holding the data
 * (gpslc_set_data on every one: X, T, Y are replicated, N (D + 2) doubles).  Context k computes the contiguous block
 * [k q + min(k, r), ...) of q = S / nctx (+ 1 for k < r = S mod nctx) posterior samples on its own host thread and its own device;
 * nothing moves between the devices while they compute, and every device copies its block of each result straight into the
 * caller's HOST arrays (the same arrays, layouts and NULL conventions as gpslc_predict).  The library's normals are placed with
 * gpslc_set_ensemble semantics, so seeded draws — like every other output — are bit-identical to ONE gpslc_predict call over all S
 * samples, whatever nctx (several contexts on one device are allowed: that is how the single-GPU tests check it).  A placement set on
 * ctxs[0] beforehand is honoured as the placement of the whole call (one node's share of a larger ensemble).
 * info_or_null (S): the 1-based failing pivot of every sample as gpslc_last_info reports it.  Returns the first negative status of
 * any shard, else the first failing pivot in sample order, else 0; gpslc_last_error(ctxs[0]) names the shard.  Host pointers only. */
int gpslc_predict_multi(int32_t nctx, gpslc_ctx* const* ctxs, int64_t S, const double* U, const double* uyLS,
                        const double* xyLS, const double* tyLS, const double* yScale, const double* yNoise,
                        int32_t L, const double* doT, double pred_noise, int32_t spp, uint64_t seed,
                        const double* z_or_null, double* meanSATE, double* varSATE, double* meanITE,
                        double* ite_draws, int32_t* info_or_null);

/* The partition gpslc_predict_multi uses, for callers who drive the contexts themselves (results LEFT on their devices:
 * gpslc_predict_dev per context after gpslc_set_ensemble(ctx_k, s0, S)): block k of nblocks is the posterior samples [*s0, *s1),
 * sizes differ by at most one, the first S mod nblocks blocks are the longer ones.  Host-only arithmetic (no ctx, no GPU); the
 * same partition as causalgpslc.jl_amd/sharded.py: shard_range. */
int gpslc_shard_range(int64_t S, int32_t nblocks, int32_t k, int64_t* s0, int64_t* s1);

/* ITEDistributions(g, doT) (src/estimation.jl:66-86) for one intervention level, with the
 * reference's output layout: MeanITEs S x n and CovITEs S x n x n, sample index fastest;
 * CovITEs carries the + pred_noise*I of src/estimation.jl:82.  Either output may be NULL. */
int gpslc_ite_distributions(gpslc_ctx* ctx, int64_t S, const double* U, const double* uyLS,
                            const double* xyLS, const double* tyLS, const double* yScale,
                            const double* yNoise, double doT, double pred_noise,
                            double* MeanITEs, double* CovITEs);

/* likelihoodDistribution(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y, doT) (src/likelihood.jl:8-52 and
 * its three reduced methods :55-94, :97-136, :139-174) for ONE parameter set, as the reference exports it:
 * the dense n x n blocks CovWW, CovWWs, CovWWp and the four posterior blocks CovC11..CovC22 (:46-49),
 * column-major; any output may be NULL.  (Y, the first element of the reference's tuple, is the caller's.) */
int gpslc_likelihood_distribution(gpslc_ctx* ctx, const double* U, const double* uyLS, const double* xyLS,
                                  double tyLS, double yScale, double yNoise, double doT, double* CovWW,
                                  double* CovWWs, double* CovWWp, double* CovC11, double* CovC12,
                                  double* CovC21, double* CovC22);

/* SATEsamples (src/estimation.jl:148-163): out[j*spp + d] = mean[j] + var[j] * z — the variance
 * is used as the standard deviation, as the reference does (src/estimation.jl:159).  Host-only
 * arithmetic; z_or_null (S*spp) or Philox stream `seed`, stream id 2^40 + j. */
int gpslc_sate_samples(const double* meanSATE, const double* varSATE, int64_t S, int32_t spp,
                       uint64_t seed, const double* z_or_null, double* out /* S*spp */);

/* summarizeEstimates(samples; credible_interval) (src/driver.jl:129-149): per-individual Mean and the
 * (1-ci)/2 and 1-(1-ci)/2 quantiles (Julia's Statistics.quantile, type 7) of an n x m sample matrix
 * (column-major, samples[i + n*j]); rows of up to 16384 samples are sorted in LDS, longer ones (S * spp of a large
 * posterior) go through an exact radix select.  The _dev variant reads device memory with explicit strides
 * (sample (i, j) at samples[i*row_stride + j*col_stride]) so that level l of gpslc_predict_dev's ite_draws
 * (L x n x M, level fastest) is summarised in place with samples = draws + l, row_stride = L,
 * col_stride = L*n, and writes device outputs: the draw tensor never leaves HBM. */
int gpslc_summarize(gpslc_ctx* ctx, const double* samples, int64_t n, int64_t m, double credible_interval,
                    double* mean, double* lower, double* upper);
int gpslc_summarize_dev(gpslc_ctx* ctx, const double* samples, int64_t n, int64_t m, int64_t row_stride,
                        int64_t col_stride, double credible_interval, double* mean, double* lower, double* upper);

/* 1-based failing pivot (0 = ok) of every posterior sample of the last predict / y_logpdf /
 * ite_distributions call; codes > n refer to the CovITE factorisation (pivot - n). */
int gpslc_last_info(const gpslc_ctx* ctx, int32_t* info, int64_t S);

/* ---- posterior pack (replaces Serialization of a GPSLCObject, src/io.jl:14-34) ----
 * Flat little-endian file: magic "GPSLCPK1"; 6 x int64 {n, nX, nU, S, binaryT, 0}; 7 x f64 hyper-parameters
 * {nU or -1, nOuter, nMHInner, nESInner, nBurnIn, stepSize, predictionCovarianceNoise} (src/types.jl:22-30);
 * then the f64 arrays X[n,nX] T[n] Y[n] U[n,nU,S] uyLS[nU,S] xyLS[nX,S] tyLS[S] yNoise[S] yScale[S], column-
 * major — what extractParameters (src/utils.jl:92-124) yields for the retained samples, stacked.
 * Host-only functions (no ctx, no GPU). */
typedef struct gpslc_pack_header {
    int64_t n, nX, nU, S, binary_t, reserved;
    double hyper[7];
} gpslc_pack_header;

int gpslc_pack_save(const char* path, const gpslc_pack_header* h, const double* X, const double* T,
                    const double* Y, const double* U, const double* uyLS, const double* xyLS,
                    const double* tyLS, const double* yNoise, const double* yScale);
int gpslc_pack_read_header(const char* path, gpslc_pack_header* h);
/* Reads the data and the posterior samples [s0, s1) (0 <= s0 <= s1 <= S) into caller buffers sized for
 * s1 - s0 samples — a rank of a sharded prediction loads only its own block.  Any output may be NULL (skipped). */
int gpslc_pack_load(const char* path, int64_t s0, int64_t s1, double* X, double* T, double* Y, double* U,
                    double* uyLS, double* xyLS, double* tyLS, double* yNoise, double* yScale);

/* ---- measurement hooks (bench.py, profiles/) ----------------------------------------- */

/* Accumulated HIP-event statistics since the last reset, recorded (on the launching stream) when the ctx was
 * created with GPSLC_FLAG_PROFILE: launches, total device milliseconds, total algorithmic work.  Kernel classes:
 *   0  tile_gemm_nt_kernel<1, 0> in the factorisation of A (trailing updates: the dominant kernel)   work = flop
 *   1  tile_fused_strip_kernel in the factorisation of A (in-panel column update + panel solve)        work = flop
 *   2  the predictive-draw kernels of a launch_draws call (normal generation + triangular product)      work = draws
 *   3  every f64-MFMA tile-update launch of the full-ITE-covariance path (W solve, SYRK, factor)        work = flop
 *   4  potrf_tasks_kernel: the whole factorisation of A as one persistent launch (gpslc_set_task_schedule)
 *      work = flop, textbook count n^3 / 3 + (right-hand sides) n^2 per matrix
 * gpslc_profile_get is class 0. */
int gpslc_profile_reset(gpslc_ctx* ctx);
int gpslc_profile_get(gpslc_ctx* ctx, int64_t* launches, double* total_ms, double* total_flop);
int gpslc_profile_get_class(gpslc_ctx* ctx, int32_t kernel_class, int64_t* launches, double* total_ms,
                            double* total_flop);

/* library / build identification, e.g. "gpslc_hip 0.1 gfx950" */
const char* gpslc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GPSLC_HIP_H */
