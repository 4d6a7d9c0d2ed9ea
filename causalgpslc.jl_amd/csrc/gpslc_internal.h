// Internal declarations shared by the HIP translation units of libgpslc_hip.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------------------------------
// Tiled storage.  Every dense matrix the factorisation touches lives in HBM as 128 x 128
// fp64 tiles (128 KiB, column-major inside the tile: element (r, c) at c*128 + r), so a
// K-slab of a tile is one contiguous run and every global access is a full-line stream.
// A "tile matrix" is addressed through a TRef: lower-packed (tile (i, j), j <= i, at
// i(i+1)/2 + j) or rectangular (i*ld + j).
// ---------------------------------------------------------------------------------------
#define GP_TS 128
#define GP_TSQ (GP_TS * GP_TS)

// ---------------------------------------------------------------------------------------
// Per-device launch state.  hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only
// and one process may hold a ctx per GPU, each driven by its own thread: every kernel
// instantiation remembers, per device, whether the opt-in was applied.  Lock-free; two threads racing on the
// same device both apply the (idempotent) attribute.
// ---------------------------------------------------------------------------------------
#include <atomic>
#include <cstdlib>
struct DeviceOnce {
    std::atomic<unsigned long long> mask{0};
};
inline void lds_opt_in(DeviceOnce& o, const void* fn, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (o.mask.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    o.mask.fetch_or(bit, std::memory_order_release);
}
// compute units of the current device (cached per device)
inline int device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int v = cus[dev & 63].load(std::memory_order_relaxed);
    if (v == 0) {
        v = 256;
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        cus[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v;
}
// Measurement switches (tile visiting order, grid sizes, timing-only kernels, in-kernel stamps) exist only in the
// -DGPSLC_DIAG build that tools/ and the profiling scripts use (libgpslc_hip_diag.so).  In the production library
// the environment cannot change a result or a schedule: every switch reads as its default.
inline int diag_env(const char* name, int dflt) {
#ifdef GPSLC_DIAG
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

struct TRef {
    double* base;        // first tile of batch element 0
    long long bstride;   // doubles between consecutive batch elements
    int kind;            // 0 = lower packed, 1 = rectangular
    int ro, co;          // tile offsets added to (i, j)
    int ld;              // tiles per tile-row (rectangular)
    int bdiv;            // > 1: batch element b uses matrix b / bdiv (several (sample, level) units share one factor of A)
};

__host__ __device__ inline long long tref_index(const TRef& t, int i, int j) {
    long long ii = (long long)i + t.ro, jj = (long long)j + t.co;
    return t.kind == 0 ? ii * (ii + 1) / 2 + jj : ii * (long long)t.ld + jj;
}
__host__ __device__ inline double* tref_tile(const TRef& t, long long b, int i, int j) {
    if (t.bdiv > 1) b /= t.bdiv;
    return t.base + b * t.bstride + tref_index(t, i, j) * (long long)GP_TSQ;
}

// C(i, j) (-)= sum_{kk in [k0, k1)} A(i, kk) * B(j, kk)^T over a set of output tiles.
struct GemmArgs {
    TRef A, B, C;
    int shape;        // 0: lower triangle (incl. diagonal) of an mi x mi tile square, 1: mi x mj rectangle
    int i0, j0;       // output tile (i, j) = (i0 + ii, j0 + jj)
    int mi, mj;
    int k0, k1;
    int accumulate;   // 1: C -= A B^T, 0: C = A B^T
    int nbatch;
    int ntiles;       // output tiles per batch element
    int short_row0;   // output tile rows >= short_row0 (augmented right-hand-side rows) carry only
    int short_rows;   // `short_rows` live rows: dead 16-row sub-tiles are skipped (0 = feature off)
    int sym;          // A and B are the same tile matrix and C(t, t) needs only its lower triangle.  1: noted for
                      // the flop accounting only; 2: launch_tile_gemm skips the full-size diagonal tiles, which
                      // launch_syrk_diag (36 of 64 sub-tile products, 9 per wave) computes instead; 3: as 2, and
                      // the augmented-row tiles (short_row0, j) of those columns ride with the diagonal items too
    int fuse;         // 1 (accumulate launches of one tile column, mj == 1): each item also applies the panel
    TRef F;           //    product with tile (0, fk) of F = the inverted diagonal blocks (see k_tilegemm.hip)
    int fk;
    int skip_gdiag;   // the (short) augmented diagonal tile (short_row0, short_row0) is not an item of this launch: nobody
                      // reads -R R^T (EpiArgs::from_rows)
    int* info;        // launch_diag_update_potrf: per-batch-element info words and the code base of tile row 0 (launch_diag's)
    int info_base;
    int* queue;       // optional: 16 zero-initialised ints (per-XCD ticket counters [0..8), exit counters [8..16))
                      // owned by the launching stream; the kernel leaves them zeroed again.  null = static stride
    const unsigned short* order;  // optional (ii, jj) pairs: output-tile visiting order (L2-blocked), or null
    int nt_c;                     // C tiles are loaded / stored with the non-temporal hint (streamed once per launch:
                                  // they should not displace the operand slabs the co-resident workgroups share in L2)
    int diag_skip;                // DIAGNOSTIC ONLY (GPSLC_GEMM_DIAG): 1 = skip in-loop global loads, 2 = also LDS writes
    unsigned long long* dbg;      // diagnostic builds only: per-workgroup s_memtime stamps, or null
};

// per-posterior-sample inputs of the Gram build (device pointers, already offset to sample 0 of the call)
struct SampleParams {
    const double* U;       // n x nU x S
    const double* uyLS;    // nU x S
    const double* xyLS;    // nX x S
    const double* tyLS;    // S
    const double* yScale;  // S
    const double* yNoise;  // S
    long long u_sstride;   // doubles between the U blocks of consecutive samples (n*nU; 0 = shared)
};

struct GramArgs {
    const double* X;   // n x nX (ctx or override)
    const double* T;   // n
    SampleParams p;
    long long s0;      // first sample of this chunk
    int n, nX, nU, nt;
    TRef M;            // lower-packed tile matrix of the chunk
    double* part;      // [b][2][nt][Np] partial column sums of B (0) and K (1)
    int with_sums;
    int f32;           // evaluate the RBF kernel in fp32 (GPSLC_FLAG_FP32_KERNEL)
    int binary_t;      // every T is 0 or 1: e_ij is 1 or exp(-1/tyLS^2), no per-pair exp (bit-identical)
    unsigned long long* dbg;   // measurement build only: per-workgroup s_memtime stamps [entry, staged, columns done, end], or null
};

// ---------------------------------------------------------------------------------------
// The whole left-looking factorisation of a batch as ONE persistent launch (round 6, k_tilegemm.hip: potrf_tasks_kernel).
// Tasks: diag(k) = update + Cholesky + inverse of diagonal tile k (the body of diag_update_potrf_kernel; k = 0: of
// diag_potrf_inv_la_kernel), strip(i, k) = column update + panel product of tile (i, k) (a work item of tile_fused_strip_kernel).
// The host lays the tasks of a launch out as eight ticket queues (one per XCD: a matrix never leaves its queue, so the
// 7 - k strips of a column share the B panel in one L2) in an order in which every task follows its producers; progress
// words per matrix in global memory carry the dependencies between workgroups (agent-scope release / acquire).
// ---------------------------------------------------------------------------------------
#include "task_list.h"     // descriptor layout + the host-built task order (plain C++: tests/c/task_list_test.cpp exercises it on the CPU)
struct PotrfTaskArgs {
    GemmArgs g;             // A = B = C = the tile matrix, F = the inverted diagonal blocks, k0 = 0, short_row0 = nt, short_rows,
                            // sym = 3 with an augmented row riding along, info / info_base, nbatch
    const unsigned* list;   // TASK_LIST_HDR header words, then the descriptors of the eight queues
    int* sync;              // TASK_SYNC_HDR + TASK_SYNC_STRIDE * nbatch ints, zeroed by the launcher's caller before every launch
    int nt;
    double* alpha;          // non-null: the list ends every matrix with its back-substitution, alpha = L^-T z -> alpha[b][nt 128]
    int fence_mode;         // measurement build only (0 = release / acquire as documented)
    unsigned long long* dbg;   // measurement build only: per-task stamps, or null
};
void launch_potrf_tasks(const PotrfTaskArgs& a, long long ntasks, int mt, hipStream_t st);

// launchers (implemented in the k_*.hip files); all asynchronous on `st`
void launch_tile_gemm(const GemmArgs& g, hipStream_t st);
void launch_syrk_diag(const GemmArgs& g, int carry_aug, hipStream_t st);   // g.mi full-size diagonal tiles from (i0, i0)
void launch_diag_update_potrf(const GemmArgs& g, int carry_aug, hipStream_t st);  // in-panel diagonal tile: update + factor + inverse
void launch_diag(const TRef& M, int k, double* inv, long long inv_bstride, int* info,
                 int info_base, int nbatch, hipStream_t st);
void launch_gram(const GramArgs& g, int nbatch, hipStream_t st);
// substitution-based (backward-stable) diagonal-tile factorisation and panel solve for near-singular matrices (k_robust.hip)
void launch_diag_robust(const TRef& M, int k, int* info, int info_base, int nbatch, hipStream_t st, int info_div = 1);
void launch_trsm_robust(const TRef& X, const TRef& L, int k, int i0, int count, int nbatch, hipStream_t st);

struct RhsArgs {
    const double* T; const double* Y; const double* tyLS; const double* doT;
    long long y_sstride;   // right-hand side 0 of sample s is Y + s*y_sstride (0 = shared)
    long long s0; int n, nt, naug, L; int with_sums;
    const double* part; double* bsum; double* ksum; double* sumdelta;  // bsum/ksum [b][Np], sumdelta [b][L]
    TRef M;
    int live_rows;   // > 0 (single augmented tile row of <= 32 right-hand sides, epilogue sums from the rows of R): only the
                     // first 32 rows of the augmented tiles are written (right-hand sides, then zeros) and the augmented
                     // diagonal tile not at all — every reader of those tiles touches the live 16- / 32-row blocks only
};
void launch_rhs(const RhsArgs& r, int nbatch, hipStream_t st);

struct EpiArgs {
    TRef M; int n, nt, naug, L; long long s0; long long S;
    const double* sumdelta; double pred_noise;
    double* meanSATE; double* varSATE;   // S x L or null
    double* logdet; double* quad;        // S or null
    int from_rows;   // 1 (single augmented tile row): z.z, z.w_l, w_l.w_l are summed from the rows of R themselves — the
                     // augmented diagonal tile is then never updated (potrf_tiles(..., skip_aug_diag))
};
void launch_epilogue(const EpiArgs& e, int nbatch, hipStream_t st);

struct BackArgs {
    TRef M; const double* inv; long long inv_bstride; int nt; int naug;
    double* zwork;   // [b][Np] in: z (copied from the augmented row), out: alpha
};
void launch_backsolve(const BackArgs& a, int nbatch, hipStream_t st);

struct IteMeanArgs {
    const double* X; const double* T; SampleParams p; long long s0; long long S;
    int n, nX, nU, nt, L; const double* doT;
    const double* alpha;   // [b][Np]
    const double* Y;       // right-hand side alpha solves for: A alpha = Y (sample s at Y + s*y_sstride)
    long long y_sstride;
    const double* yNoise;  // S (= p.yNoise; kept separate so that the generic node paths can pass their own)
    double* meanITE;       // element (i, s, l) at i*si + s*ss + l*sl
    long long si, ss, sl;
    int f32;
};
void launch_ite_mean(const IteMeanArgs& a, int nbatch, hipStream_t st);

void launch_rbf_log(const double* X1, const double* X2, long long n, int d, const double* ls,
                    int ls_len, double* out, hipStream_t st);
void launch_process_cov(const double* in, long long n, double scale, double noise, double* out,
                        hipStream_t st);

// unit B (full ITE covariance) helpers
struct DtArgs {
    const double* X; const double* T; SampleParams p; long long s0;
    int n, nX, nU, nt;
    const double* doT;     // device: intervention levels; batch element b = (sample s0 + b / lc, level l0 + b % lc)
    int l0, lc;
    double pred_noise;
    TRef W;    // nt x nt rectangular: receives D (rows = i, cols = j), D_ij = B_ij (r_j - e_ij)
    TRef Cm;   // lower packed nt: receives Delta + pred_noise*I (identity on the padding)
};
void launch_dt_build(const DtArgs& a, int nbatch, hipStream_t st);

struct GatherCovArgs {
    TRef Cm; int n, nt; long long s0, S; double* out; // out: S x n x n, sample fastest
};
void launch_gather_cov(const GatherCovArgs& a, int nbatch, hipStream_t st);

struct DrawArgs {
    TRef Lc; int n, nt; long long s0, S; int l, L, spp;   // batch element b = (sample s0 + b / lc, level l + b % lc)
    int lc;
    const double* mean;   // meanITE n x S x L
    const double* z;      // caller's normals, n x spp x S x L, or null
    double* zgen;         // z == null, spp > 128: workspace [nbatch][spp][n] the library's Philox normals are generated into
    double* zt;           // spp <= 128: workspace [nbatch][1 | 2 | 4 | 8 blocks of 16 draws][16 nt 128] — the unit's normals (caller's or Philox) in
                          // the MFMA operand image the streaming draw kernel reads (draws_zt_index, k_solve.hip), zero-padded
    unsigned long long seed;
    long long rs0, rS;    // Philox stream of batch element b: (rs0 + b / lc) + rS * (l + b % lc) (gpslc_set_ensemble)
    // element (instance i, sample offset sb = b / lc, level offset lb = b % lc, draw d) of this launch goes to
    // out[obase + sb*osb + lb*osl + i*osi + d*osd]: the reference tensor L x n x (S*spp) directly (L == 1), or the
    // level-sweep staging buffer [sample][level][d][i]
    double* out;
    long long obase, osb, osl, osi, osd;
};
void launch_draws(const DrawArgs& a, int nbatch, hipStream_t st);
void launch_draws_scatter(const double* tmp, double* out, long long n, int L, int spp, long long s0, int nbatch,
                          hipStream_t st);

// generic multivariate-normal pieces (U-prior node and friends)
struct DenseLoadArgs { const double* cov; int n, nt; TRef M; };   // column-major n x n -> lower tiles
void launch_dense_load(const DenseLoadArgs& a, hipStream_t st);
struct RowsRhsArgs { const double* x; long long S; int n, nt, naug; TRef M; int row0; int rect; };   // rows q = x[:, q]
void launch_rows_rhs(const RowsRhsArgs& a, hipStream_t st);
struct QuadRowsArgs { TRef M; int n, nt, naug; long long S; double* logdet; double* quad; };
struct RowNormArgs { TRef W; int nt, naug; long long S; double* quad; };   // quad[q] = ||row q of W||^2
void launch_row_norms(const RowNormArgs& a, hipStream_t st);
void launch_quad_rows(const QuadRowsArgs& a, hipStream_t st);

// likelihoodDistribution (src/likelihood.jl:8-174): dense blocks as rectangular tile matrices
struct LdBuildArgs {
    const double* X; const double* T; SampleParams p; long long s0;
    int n, nX, nU, nt; double doT;
    TRef K, Ks, KsT, Kss;   // nt x nt rectangular each: CovWW, CovWWs, CovWWs', CovWsWs
};
void launch_ld_build(const LdBuildArgs& a, hipStream_t st);
struct RectGatherArgs { TRef R; int n, nt; double* out; double diag_add; };   // -> column-major n x n
void launch_rect_gather(const RectGatherArgs& a, hipStream_t st);

// single-launch node score for small n (k_small.hip): one workgroup per node, matrix resident in LDS
struct SmallNode {
    const double* Fs;       // n x nF, column-major, ALREADY divided by the lengthscales: Fs[i, f] = F[i, f] * (1 / ls[f])
    const double* target;   // n
    double scale, noise;
    int nF;
    int pad_;
    const double* cov;      // non-null: dense n x n covariance in DEVICE memory (lower triangle read); the matrix is
    double covscale;        //   covscale * cov instead of the RBF Gram matrix (the :U => u => :U prior nodes)
};
#define SMALL_INLINE_NODES 4
struct SmallArgs {
    const SmallNode* nodes; // descriptors of the nodes >= SMALL_INLINE_NODES (device-visible memory)
    SmallNode inl[SMALL_INLINE_NODES];   // the first nodes travel in the kernel arguments: no dependent host-memory read
    int n, NB;              // NB = ceil(n / 16) block rows
    double* out;            // [count][4]: logdet, quad, info (1-based failing pivot, 0 = ok), reserved
    double* stamps;         // measurement build only: [count][8] phase timings, else null
    double* scratch;        // mid-size kernel: per-node device scratch (finished block columns + scaled features)
    long long scratch_stride;
    double* draw;           // non-null: [count][n], node i also returns chol(K_i) * target_i (target = standard normals):
                            //   a draw from N(0, K_i) — the auxiliary vector of an elliptical slice, a prior draw
};
size_t small_gp_lds_bytes(int n, int nF);
void launch_small_gp(const SmallArgs& a, int count, int nF_max, hipStream_t st);
// left-looking variant for 176 < n <= 640: only the current block column in LDS, finished columns in an L2-resident scratch
size_t mid_gp_scratch_doubles(int n, int nF);
size_t mid_gp_lds_bytes(int n, int nF);
bool mid_gp_fits(int n);
void launch_mid_gp(const SmallArgs& a, int count, int nF_max, hipStream_t st);

// summarizeEstimates (src/driver.jl:129-149): per-row mean and two type-7 quantiles of an n x m sample matrix
struct SummArgs {
    const double* x; long long rs, cs;   // sample (i, j) at x[i*rs + j*cs]
    int n, m, mpad; double lowerQ, upperQ;
    double* mean; double* lower; double* upper;
};
void launch_summarize(const SummArgs& a, hipStream_t st);
