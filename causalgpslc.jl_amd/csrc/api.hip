// Host side of libgpslc_hip.so: context, workspace, chunk scheduling over HIP streams, and the
// C ABI of include/gpslc_hip.h.  No exception crosses the ABI (everything is caught and mapped to
// a status code); every HIP error is reported through gpslc_last_error.
#include "../../include/gpslc_hip.h"
#include "gpslc_internal.h"
#include "batch_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

struct HipFail {
    hipError_t e;
    const char* what;
    int line;
};
#define HC(x)                                                         \
    do {                                                              \
        hipError_t _e = (x);                                          \
        if (_e != hipSuccess) throw HipFail{_e, #x, __LINE__};        \
    } while (0)

struct Arena {
    char* base = nullptr;
    size_t bytes = 0;
    size_t off = 0;
    void reset() { off = 0; }
    template <class T>
    T* take(size_t count) {
        size_t a = (off + 255) & ~size_t(255);
        size_t need = count * sizeof(T);
        if (a + need > bytes) throw std::bad_alloc();
        off = a + need;
        return reinterpret_cast<T*>(base + a);
    }
};

// Growable device staging pool of a ctx (host-pointer entry points: uploads, outputs, info words).  take() never
// moves earlier allocations: when the current block is full a new one is chained; the next reset() merges the
// chain into one block of the total size, so a steady-state call sequence allocates nothing.
struct PoolArena {
    struct Block { char* base; size_t bytes; };
    std::vector<Block> blocks;
    size_t off = 0;
    void release() {
        for (auto& b : blocks) (void)hipFree(b.base);
        blocks.clear();
        off = 0;
    }
    void reset() {
        if (blocks.size() > 1) {
            size_t total = 0;
            for (auto& b : blocks) total += b.bytes;
            (void)hipDeviceSynchronize();
            release();
            add_block(total);
        }
        off = 0;
    }
    void add_block(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); throw std::bad_alloc(); }
        blocks.push_back(Block{static_cast<char*>(p), bytes});
        off = 0;
    }
    template <class T>
    T* take(size_t count) {
        const size_t need = std::max<size_t>(count * sizeof(T), 8);
        size_t a = (off + 255) & ~size_t(255);
        if (blocks.empty() || a + need > blocks.back().bytes) {
            add_block(std::max<size_t>((need + 255) & ~size_t(255), size_t(1) << 20));
            a = 0;
        }
        off = a + need;
        return reinterpret_cast<T*>(blocks.back().base + a);
    }
};

struct DevBuf {   // RAII device allocation for the host-pointer entry points
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void alloc(size_t bytes) {
        if (bytes == 0) bytes = 8;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; throw std::bad_alloc(); }
    }
    template <class T> T* as() { return static_cast<T*>(p); }
};

struct ProfRec {
    hipEvent_t a, b;
    double flop;
    int cls;     // kernel class, see gpslc_profile_get_class (include/gpslc_hip.h)
};
constexpr int kProfClasses = 5;

// device-resident task list of one (tile count, augmented row, batch size, group size) shape of the persistent
// factorisation launch (potrf_tasks_kernel); built once per shape and kept (see task_list_for)
struct TaskList {
    int nt = 0, back = 0, nb = 0, G = 0, rows = 0;
    bool aug_full = false;
    unsigned* dev = nullptr;
    long long ntasks = 0;
    unsigned long long used = 0;
};

}  // namespace

struct gpslc_ctx {
    int device = 0;
    int64_t n = 0;
    int nX = 0, nU = 0;
    uint32_t flags = 0;
    int nt = 0;
    double *dX = nullptr, *dT = nullptr, *dY = nullptr;
    bool has_data = false;
    bool binary_t = false;   // every treatment is exactly 0 or 1 (detected in gpslc_set_data)
    int max_batch = 0;   // 0 = auto
    int panel = 8;
    bool panel_set = false;        // gpslc_set_tuning gave a panel width: the persistent launch then serves nt <= panel only
    int64_t ens_off = 0, ens_S = 0;   // gpslc_set_ensemble: placement of a call's samples for the Philox stream ids
    int nstreams = 1;   // chunks of one call alternate over this many HIP streams (2 buys ~1-2 %, see profiles/)
    std::vector<hipStream_t> streams;
    std::vector<Arena> arenas;     // one per stream slot
    std::vector<int*> queues;      // one ticket-counter block (16 ints) per stream slot, see GemmArgs::queue
    // persistent factorisation launch (potrf_tasks_kernel): progress words per stream slot, cached task lists, and whether
    // a call has used it (its time-out word is then checked when the call's streams have drained)
    std::vector<int*> task_sync;
    std::vector<size_t> task_sync_ints;
    std::vector<TaskList> task_lists;
    unsigned long long task_clock = 0;
    bool task_used = false;
    int task_min_nt = 2, task_max_nt = TASK_MAX_NT;   // tile counts in this range take the persistent launch (128 < N <= 4096: with
                                            // groups of 32 it wins at every tile count the descriptor can hold — N = 256 +1..3 %, 384 +6 %, 512 +10 %,
                                            // 1024 +6 %, 1536 +4 %, 2048 +2.4 %, 3072 +1.8 %, 4096 +1.0..1.3 % against panels of 8 + trailing
                                            // updates, profiles/r06_ab_experiments.md §1d)
    int task_min_batch = 256;      // ... when the chunk holds at least this many matrices: a persistent launch over few
                                   // matrices is a chain of hand-offs (N = 1024: 2.1 ms for 8 matrices against 1.5 ms with one
                                   // launch per column; even at 256, profiles/r06_ab_experiments.md §1)
    int task_group = 32;           // matrices per group of the task order (see build_task_list)
    int task_rows = 2;             // consecutive tile rows of a column per strip task up to 8 tiles per side (beyond: one — long K
                                   // loops amortise the task's fetch / acquire / publish by themselves, and finer tasks balance
                                   // better: N = 1536 .. 4096 +0.6 .. 1.2 %, profiles/r06_ab_experiments.md §1d)
    Arena scratch;                 // call-level buffers (internal MeanITE of a draws-only call, ...)
    PoolArena io;                  // staging of the host-pointer entry points and per-call info words
    // single-launch small-n node scores (k_small.hip): pinned, device-visible host staging (descriptors, inputs,
    // results) and host copies of the ctx's data for assembling the :Y node's feature block
    char* pin = nullptr;
    size_t pin_bytes = 0;
    // two pinned bounce chunks for large device -> pageable-host hand-overs (copy_out_large), allocated on first use
    char* bounce[2] = {nullptr, nullptr};
    std::vector<double> hX, hT, hY;
    std::vector<double> stage_f, stage_rest;   // host staging of gpslc_nodes_logpdf's batched pass (grown, never shrunk)
    std::string err;
    std::vector<int32_t> last_info;
    // cached factor of the last dense covariance given to gpslc_mvn_logpdf (SigmaU is constant per data set)
    double* mvn_tiles = nullptr;     // tiled factor (substitution-based: no inverted diagonal blocks are kept)
    double* mvn_dense = nullptr;     // small n: the dense covariance itself (every evaluation refactorises it in LDS)
    double mvn_logdet = 0.0;
    int mvn_info = 0;
    bool mvn_valid = false;
    // L2-blocked visiting orders of the lower-triangular tile sets, keyed by the triangle size m
    std::vector<unsigned short*> tri_order;
    int order_block = 8;
    // profiling of the dominant kernel
    std::vector<ProfRec> prof;
    size_t prof_used = 0;
    int64_t prof_launches[kProfClasses] = {};     // per kernel class, see ProfRec::cls
    double prof_ms[kProfClasses] = {}, prof_flop[kProfClasses] = {};
};

namespace {

void set_err(gpslc_ctx* c, const std::string& s) {
    if (c) c->err = s;
}

// HIP events around a launch region on `st` (GPSLC_FLAG_PROFILE only): accumulates into kernel class `cls`
struct ProfScope {
    gpslc_ctx* c;
    ProfRec* r = nullptr;
    hipStream_t st;
    ProfScope(gpslc_ctx* c_, int cls, double work, hipStream_t st_) : c(c_), st(st_) {
        if (!(c->flags & GPSLC_FLAG_PROFILE)) return;
        if (c->prof_used == c->prof.size()) {
            ProfRec n{};
            if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return;
            c->prof.push_back(n);
        }
        r = &c->prof[c->prof_used++];
        r->flop = work;
        r->cls = cls;
        (void)hipEventRecord(r->a, st);
    }
    ~ProfScope() { if (r) (void)hipEventRecord(r->b, st); }
};

int fail_hip(gpslc_ctx* c, const HipFail& f) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) at api.hip:%d in `%s`", (int)f.e, hipGetErrorString(f.e),
             f.line, f.what);
    set_err(c, buf);
    return GPSLC_ERR_HIP;
}

void arena_reserve(gpslc_ctx* c, Arena& a, size_t bytes) {
    if (a.bytes >= bytes) return;
    if (a.base) {
        HC(hipDeviceSynchronize());
        HC(hipFree(a.base));
        a.base = nullptr;
        a.bytes = 0;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        throw std::bad_alloc();
    }
    a.base = static_cast<char*>(p);
    a.bytes = bytes;
    (void)c;
}

void ensure_streams(gpslc_ctx* c) {
    while ((int)c->streams.size() < c->nstreams) {
        hipStream_t s;
        HC(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        c->streams.push_back(s);
    }
    if ((int)c->arenas.size() < c->nstreams) c->arenas.resize(c->nstreams);
    while ((int)c->queues.size() < c->nstreams) {
        int* q = nullptr;
        HC(hipMalloc((void**)&q, 16 * sizeof(int)));
        HC(hipMemset(q, 0, 16 * sizeof(int)));
        c->queues.push_back(q);
    }
}

// ---- profiled launch of the accumulate-mode tile kernel ---------------------------------
// GPSLC_SYRK_DIAG=0 keeps the diagonal tiles in the general kernel (A/B switch for measurements)
static int sym_mode() {
    static const int m = diag_env("GPSLC_SYRK_DIAG", 1) == 0 ? 1 : 2;
    return m;
}

static int col_sym_mode() {
    // the one diagonal tile of an in-panel column update also goes to the lower-triangle kernel (+0.2 %, measured)
    static const int m = diag_env("GPSLC_SYRK_DIAG_COL", 1) == 0 ? 1 : sym_mode();
    return m;
}

// with a single short augmented tile row, its off-diagonal tiles ride with the diagonal items (GemmArgs::sym == 3)
static int aug_sym(int sym, int short_rows) {
    static const int on = diag_env("GPSLC_SYRK_AUG", 1) == 0 ? 0 : 1;
    // up to 32 live rows (31 levels): beyond that the diagonal kernel runs out of registers and the augmented
    // tiles carry enough real work to stay ordinary items
    return (sym == 2 && short_rows > 0 && short_rows <= 32 && on) ? 3 : sym;
}

// GPSLC_FUSE_PANEL=0: separate panel-product launches for every column (measurement switch)
static int fuse_mode() {
    static const int m = diag_env("GPSLC_FUSE_PANEL", 1) == 0 ? 0 : 1;
    return m;
}

// the diagonal tile (and, with sym == 3, the augmented-row tile) of a symmetric column update, on its own
static void launch_sym_diag_tiles(const GemmArgs& g, hipStream_t st) {
    GemmArgs d = g;
    const int cand = g.shape == 0 ? g.mi : 1;
    d.mi = g.short_rows > 0 ? std::max(0, std::min(cand, g.short_row0 - g.i0)) : cand;
    launch_syrk_diag(d, g.sym == 3 && g.short_rows > 0, st);
}

void gemm(gpslc_ctx* c, const GemmArgs& g0, hipStream_t st, int prof_base = 0) {
    GemmArgs g = g0;
    g.diag_skip = 0;
    g.dbg = nullptr;
    static const int nt_c = diag_env("GPSLC_NT_C", 0);
    g.nt_c = (nt_c && g.shape == 0) ? 1 : 0;      // trailing updates only
    static const int use_queue = diag_env("GPSLC_GEMM_QUEUE", 1);
    g.queue = nullptr;
    if (use_queue)
        for (size_t i = 0; i < c->streams.size() && i < c->queues.size(); ++i)
            if (c->streams[i] == st) g.queue = c->queues[i];
    if (g.ntiles <= 0 || g.nbatch <= 0 || (g.k1 <= g.k0 && g.accumulate && !g.fuse)) return;     // fuse with an empty K range: panel product only
#ifdef GPSLC_DIAG
    // measurement build only: timing-only kernel variants (results are garbage by construction) and in-kernel
    // stamps of one trailing update, written to gpurun_out/gemm_dbg.bin
    static const int diag_skip = diag_env("GPSLC_GEMM_DIAG", 0);
    g.diag_skip = diag_skip;
    static const int dbg_m = diag_env("GPSLC_GEMM_DBG", 0);
    static bool dbg_done = false;
    DevBuf dbg_buf;
    size_t dbg_words = 0;
    // GPSLC_GEMM_DBG_FUSEK=<K>: the first FUSED in-panel launch whose K loop is K tiles deep instead
    static const int dbg_fk = diag_env("GPSLC_GEMM_DBG_FUSEK", -1);
    const bool dbg_fused = dbg_fk >= 0 && g.fuse && g.accumulate && (g.k1 - g.k0) == dbg_fk;
    if (!dbg_done && ((dbg_fk < 0 && dbg_m > 0 && g.shape == 0 && g.mi == dbg_m) || dbg_fused)) {
        dbg_words = (size_t)g.ntiles * g.nbatch * 8;
        dbg_buf.alloc(dbg_words * 8);
        HC(hipMemset(dbg_buf.p, 0, dbg_words * 8));
        g.dbg = dbg_buf.as<unsigned long long>();
        dbg_done = true;
    }
#endif
    // sym == 2: the full-size diagonal tiles (those above the augmented rows) go to the lower-triangle kernel
    auto launch_diag_tiles = [&]() {
        if (g.fuse) return;     // potrf_tiles launched them before the diagonal-block kernel (launch_sym_diag_tiles)
        if (g.sym >= 2 && g.i0 == g.j0 && (g.diag_skip == 0 || g.diag_skip == 3)) {
            GemmArgs d = g;
            const int cand = g.shape == 0 ? g.mi : 1;      // a column update holds one diagonal tile
            d.mi = g.short_rows > 0 ? std::max(0, std::min(cand, g.short_row0 - g.i0)) : cand;
            launch_syrk_diag(d, g.sym == 3 && g.short_rows > 0 && g.diag_skip == 0, st);
        }
    };
    const bool prof = (c->flags & GPSLC_FLAG_PROFILE) && g.accumulate;
    if (prof) {
        // algorithmic flop: full tiles count 2*128^3 per K tile, items in the (single) augmented row only
        // their live right-hand-side rows
        double short_items = 0;
        if (g.short_rows > 0) {
            const int last = g.i0 + g.mi - 1;                       // last output tile row of the launch
            if (last >= g.short_row0) short_items = (g.shape == 0) ? (double)g.mi : (double)g.mj;
        }
        // diagonal tiles of a symmetric launch need only their lower triangle: half a tile product, as in the
        // textbook N^3/3 count (the kernel runs 36 of the 64 sub-tile products, GemmArgs::sym); the short
        // augmented diagonal tile is already counted by its live rows
        double diag_items = 0.0;
        if (g.sym && g.i0 == g.j0) diag_items = (g.shape == 0) ? (double)g.mi : 1.0;
        if (short_items > 0 && g.shape == 0) diag_items -= 1.0;
        // sym == 2: this kernel does not run the full-size diagonal tiles at all (launch_syrk_diag does, outside the
        // timed bracket, so that the HIP-event average equals rocprofv3's average for the dominant kernel)
        const double diag_out = (g.sym >= 2 && g.i0 == g.j0 && (g.diag_skip == 0 || g.diag_skip == 3)) ? 1.0 : 0.5;
        // sym == 3: the off-diagonal augmented-row tiles are not this kernel's either (they ride with the diagonal items)
        double short_exec = short_items;
        if (g.sym == 3 && short_items > 0) short_exec = (g.shape == 0) ? 1.0 : 0.0;
        if (g.skip_gdiag && g.shape == 0 && short_exec >= 1.0) short_exec -= 1.0;     // the augmented diagonal tile is not run
        const double rows = GP_TS * ((double)g.ntiles - short_items - diag_out * diag_items)
                          + (double)g.short_rows * short_exec;
        double flop = 2.0 * GP_TS * GP_TS * rows * (double)(g.k1 - g.k0) * (double)g.nbatch;
        // fused panel product: one triangular solve per tile row = 128^2 * 128 multiply-adds... counted as the
        // textbook n^2 b flop of a TRSM (the kernel runs 56 % of the dense 2*128^3 product)
        if (g.fuse) flop += (double)GP_TS * GP_TS * rows * (double)g.nbatch;
        {
            ProfScope ps(c, prof_base ? prof_base : (g.fuse ? 1 : 0), flop, st);
            launch_tile_gemm(g, st);
        }
        launch_diag_tiles();
    } else {
        launch_tile_gemm(g, st);
        launch_diag_tiles();
    }
    HC(hipGetLastError());   // launch-configuration errors (LDS opt-in, grid) surface here, not at the end of the call
#ifdef GPSLC_DIAG
    if (dbg_buf.p) {
        HC(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(dbg_words);
        HC(hipMemcpy(h.data(), dbg_buf.p, dbg_words * 8, hipMemcpyDeviceToHost));
        FILE* f = fopen("gpurun_out/gemm_dbg.bin", "wb");
        if (f) { fwrite(h.data(), 8, dbg_words, f); fclose(f); }
    }
#endif
}

void prof_collect(gpslc_ctx* c) {
    for (size_t i = 0; i < c->prof_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->prof[i].a, c->prof[i].b) == hipSuccess) {
            const int k = c->prof[i].cls;
            c->prof_ms[k] += ms;
            c->prof_flop[k] += c->prof[i].flop;
            c->prof_launches[k] += 1;
        }
    }
    c->prof_used = 0;
}

// Visiting order of the m(m+1)/2 lower-triangular output tiles: G x G super-blocks in row-major order,
// row-major inside.  The kernel hands XCD x the x-th contiguous run of work items, so the ~64
// workgroups an XCD runs concurrently cover about one super-block: they walk the K slabs of the same
// G + G operand tile rows together and share them in that XCD's L2 instead of each streaming its own
// copy from HBM (operand traffic / ~G).
const unsigned short* tri_order(gpslc_ctx* c, int m) {
    const int G = c->order_block;
    if (G <= 1 || m <= 2) return nullptr;
    if ((int)c->tri_order.size() <= m) c->tri_order.resize(m + 1, nullptr);
    if (c->tri_order[m]) return c->tri_order[m];
    std::vector<unsigned short> h;
    h.reserve((size_t)m * (m + 1));
    for (int I = 0; I < m; I += G)
        for (int J = 0; J <= I; J += G)
            for (int i = I; i < std::min(I + G, m); ++i)
                for (int j = J; j < std::min(J + G, m) && j <= i; ++j) {
                    h.push_back((unsigned short)i);
                    h.push_back((unsigned short)j);
                }
    unsigned short* d = nullptr;
    HC(hipMalloc((void**)&d, h.size() * sizeof(unsigned short)));
    HC(hipMemcpy(d, h.data(), h.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    c->tri_order[m] = d;
    return d;
}

TRef lower_ref(double* base, long long bstride) { return TRef{base, bstride, 0, 0, 0, 0}; }
TRef rect_ref(double* base, long long bstride, int ld) { return TRef{base, bstride, 1, 0, 0, ld}; }

// ---- the persistent factorisation launch (potrf_tasks_kernel, k_tilegemm.hip): task order = build_task_list, task_list.h ----
const TaskList& task_list_for(gpslc_ctx* c, int nt, int back, int nb, int G, int rows, bool aug_full) {
    for (auto& t : c->task_lists)
        if (t.nt == nt && t.back == back && t.nb == nb && t.G == G && t.rows == rows && t.aug_full == aug_full) { t.used = ++c->task_clock; return t; }
    if (c->task_lists.size() >= 8) {       // evict the least recently used shape (nothing of it may still be in flight)
        size_t v = 0;
        for (size_t i = 1; i < c->task_lists.size(); ++i)
            if (c->task_lists[i].used < c->task_lists[v].used) v = i;
        HC(hipDeviceSynchronize());
        HC(hipFree(c->task_lists[v].dev));
        c->task_lists.erase(c->task_lists.begin() + (long)v);
    }
    TaskList t;
    t.nt = nt; t.back = back; t.nb = nb; t.G = G; t.rows = rows; t.aug_full = aug_full;
    static const int merge_diag = diag_env("GPSLC_TASK_MERGE", 1);      // measurement switch: 0 = strip(k + 1, k) as a task of its own
    std::vector<unsigned> h = build_task_list(nt, back, nb, G, rows, aug_full, &t.ntasks, merge_diag);
    HC(hipMalloc((void**)&t.dev, h.size() * sizeof(unsigned)));
    HC(hipMemcpy(t.dev, h.data(), h.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    t.used = ++c->task_clock;
    c->task_lists.push_back(t);
    return c->task_lists.back();
}

// the persistent launch serves: one left-looking panel over the whole width (nt <= task_max_nt), the inverse-based
// factorisation, ONE short augmented row whose tiles ride with the diagonal tasks and whose diagonal tile nobody reads
// (EpiArgs::from_rows) — the shape of run_predict's factorisation of A at N <= 128 task_max_nt
bool potrf_tasks_ok(const gpslc_ctx* c, int nt, int ntot, int short_rows, bool skip_aug_diag, int nb, const double* inv) {
    static const int on = diag_env("GPSLC_TASKS", 1);
    // one left-looking panel only: a panel width given through gpslc_set_tuning keeps its meaning (nt beyond it takes the panel
    // schedule); with the default width the persistent launch factorises the whole matrix as ONE panel up to task_max_nt
    const int pmax = c->panel_set ? std::max(1, c->panel) : TASK_MAX_NT;
    if (!on || !inv || nt < std::max(2, c->task_min_nt) || nt > std::min(std::min(c->task_max_nt, TASK_MAX_NT), pmax) ||
        nb >= TASK_MAX_BATCH || nb < c->task_min_batch)
        return false;
    return ntot == nt + 1 && short_rows > 0 && short_rows <= GP_TS && skip_aug_diag;
}

// back_alpha (optional, [nb][nt 128]): every matrix's task chain ends with its back-substitution alpha = L^-T z (z = right-hand
// side 0 after the forward solve the factorisation carries): launch_backsolve's result, bit for bit
void potrf_tasks(gpslc_ctx* c, const TRef& M, int nt, int ntot, double* inv, long long inv_bstride, int* info, int info_base,
                 int nb, hipStream_t st, int short_rows, double* back_alpha) {
    size_t slot = 0;
    for (size_t i = 0; i < c->streams.size(); ++i)
        if (c->streams[i] == st) slot = i;
    if (c->task_sync.size() <= slot) { c->task_sync.resize(slot + 1, nullptr); c->task_sync_ints.resize(slot + 1, 0); }
    const size_t ints = TASK_SYNC_HDR + (size_t)TASK_SYNC_STRIDE * nb;
    if (c->task_sync_ints[slot] < ints) {
        if (c->task_sync[slot]) { HC(hipDeviceSynchronize()); HC(hipFree(c->task_sync[slot])); c->task_sync[slot] = nullptr; }
        HC(hipMalloc((void**)&c->task_sync[slot], ints * sizeof(int)));
        c->task_sync_ints[slot] = ints;
    }
    static const int g_env = diag_env("GPSLC_TASK_G", 0);
    static const int r_env = diag_env("GPSLC_TASK_ROWS", 0);
    const TaskList& tl = task_list_for(c, nt, back_alpha ? 1 : 0, nb, g_env > 0 ? g_env : c->task_group,
                                       std::max(1, std::min(4, r_env > 0 ? r_env : (nt > 8 ? 1 : c->task_rows))), short_rows > 32);
    HC(hipMemsetAsync(c->task_sync[slot], 0, ints * sizeof(int), st));
    PotrfTaskArgs a{};
    a.g.A = M; a.g.B = M; a.g.C = M;
    a.g.F = TRef{inv, inv_bstride, 1, 0, 0, 0};
    a.g.shape = 1; a.g.k0 = 0; a.g.accumulate = 1; a.g.fuse = 1; a.g.nbatch = nb;
    a.g.short_row0 = nt; a.g.short_rows = short_rows; a.g.sym = 3;
    a.g.info = info; a.g.info_base = info_base;
    a.list = tl.dev; a.sync = c->task_sync[slot]; a.nt = nt; a.alpha = back_alpha;
    a.fence_mode = diag_env("GPSLC_TASK_FENCE", 0x30);      // measurement build: bit 0 = no release fence (WRONG results: the
                                                            // price of the fence), bits 4..5 = priority of the diagonal tasks
    c->task_used = true;
    const double Np = (double)nt * GP_TS;
#ifdef GPSLC_DIAG
    // measurement build, GPSLC_TASK_DBG=<n>: per-task stamps of the n-th launch -> gpurun_out/task_dbg.bin (tools/task_stamps.py)
    static const int dbg_at = diag_env("GPSLC_TASK_DBG", 0);
    static int dbg_seen = 0;
    DevBuf dbg_buf;
    if (dbg_at > 0 && ++dbg_seen == dbg_at) {
        dbg_buf.alloc((size_t)tl.ntasks * 64);
        HC(hipMemset(dbg_buf.p, 0, (size_t)tl.ntasks * 64));
        a.dbg = dbg_buf.as<unsigned long long>();
    }
#endif
    {
        ProfScope ps(c, 4, (Np * Np * Np / 3.0 + (double)a.g.short_rows * Np * Np) * (double)nb, st);
        launch_potrf_tasks(a, tl.ntasks, short_rows > 32 ? 0 : (short_rows + 15) / 16, st);
    }
    HC(hipGetLastError());
#ifdef GPSLC_DIAG
    if (dbg_buf.p) {
        HC(hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)tl.ntasks * 8);
        HC(hipMemcpy(h.data(), dbg_buf.p, h.size() * 8, hipMemcpyDeviceToHost));
        FILE* f = fopen("gpurun_out/task_dbg.bin", "wb");
        if (f) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
#endif
}

// after the call's streams have drained: a time-out inside a persistent factorisation launch is an internal error
void check_task_timeout(gpslc_ctx* c) {
    if (!c->task_used) return;
    c->task_used = false;
    for (size_t i = 0; i < c->task_sync.size(); ++i) {
        if (!c->task_sync[i]) continue;
        int w = 0;
        HC(hipMemcpy(&w, c->task_sync[i] + 8, sizeof(int), hipMemcpyDeviceToHost));
        if (w != 0) throw std::runtime_error("persistent factorisation launch timed out waiting for a producer task");
    }
}

// Blocked Cholesky of the leading nt x nt tiles of the lower-packed ntot x ntot tile matrix M; the
// rows nt..ntot-1 are carried along (augmented rows): after the call they hold R = rows * L^-T and the
// trailing (ntot-nt)^2 block its Schur complement.  Panels of `pw` tile columns: left-looking inside a
// panel, one right-looking trailing update (K = pw*128) per panel.
// tasks: the caller checks the persistent launch's time-out word when its streams have drained (check_task_timeout);
// back_alpha: where the back-substitution alpha = L^-T z may be delivered by that launch.  Returns true when it was (the
// caller then skips launch_backsolve).
bool potrf_tiles(gpslc_ctx* c, const TRef& M, int nt, int ntot, double* inv, long long inv_bstride,
                 int* info, int info_base, int nb, hipStream_t st, int aug_rows = 0, int prof_base = 0,
                 bool robust = false, int info_div = 1, bool skip_aug_diag = false, bool tasks = false,
                 double* back_alpha = nullptr) {
    // aug_rows > 0: the tile rows nt.. hold only that many live rows in total (right-hand sides);
    // a single augmented tile row is the common case and the only one the kernel shortens
    const int short_rows = (aug_rows > 0 && ntot == nt + 1) ? aug_rows : 0;
    const int pw = std::max(1, c->panel);
    // the tile kernels leave their ticket counters zeroed; re-arm them anyway so that an aborted launch
    // (device error in an earlier call) can never make a later factorisation skip work items
    for (size_t i = 0; i < c->streams.size() && i < c->queues.size(); ++i)
        if (c->streams[i] == st) HC(hipMemsetAsync(c->queues[i], 0, 16 * sizeof(int), st));
    TRef invref = TRef{inv, inv_bstride, 1, 0, 0, 0};   // tile (j, kk) -> inv[kk]
    if (robust && ntot == nt) {
        // Near-singular matrices (CovITE + 1e-10 I, SigmaU with its 1e-13 jitter): no multiplication by an inverted block
        // anywhere — the diagonal tile and the panel below it are solved by substitution (k_robust.hip); the column and
        // trailing updates stay on the MFMA tile kernel (a product is backward stable whatever the conditioning).
        for (int k = 0; k < nt; ++k) {
            const int ka = (k / pw) * pw;
            const int kend = std::min(ka + pw, nt);
            if (k > ka) {
                GemmArgs g{};
                g.A = M; g.B = M; g.C = M;
                g.shape = 1; g.i0 = k; g.j0 = k; g.mi = nt - k; g.mj = 1; g.sym = col_sym_mode();
                g.k0 = ka; g.k1 = k; g.accumulate = 1; g.nbatch = nb; g.ntiles = g.mi;
                gemm(c, g, st, prof_base);
            }
            launch_diag_robust(M, k, info, info_base, nb, st, info_div);
            launch_trsm_robust(M, M, k, k + 1, nt - k - 1, nb, st);
            if (k == kend - 1 && nt - kend > 0) {
                GemmArgs g{};
                g.A = M; g.B = M; g.C = M;
                const int m = nt - kend;
                g.shape = 0; g.i0 = kend; g.j0 = kend; g.mi = m; g.mj = m; g.sym = sym_mode();
                g.k0 = ka; g.k1 = kend; g.accumulate = 1; g.nbatch = nb; g.ntiles = m * (m + 1) / 2;
                g.order = tri_order(c, m);
                gemm(c, g, st, prof_base);
            }
        }
        HC(hipGetLastError());
        return false;
    }
    // small tile counts: the whole factorisation (and the back-substitution) as ONE persistent launch of tile tasks
    if (tasks && info_div == 1 && prof_base == 0 && potrf_tasks_ok(c, nt, ntot, short_rows, skip_aug_diag, nb, inv)) {
        potrf_tasks(c, M, nt, ntot, inv, inv_bstride, info, info_base, nb, st, short_rows, back_alpha);
        return back_alpha != nullptr;
    }
    for (int k = 0; k < nt; ++k) {
        const int ka = (k / pw) * pw;
        const int kend = std::min(ka + pw, nt);
        bool fused = false;
        if (k > ka) {   // column update inside the panel: tile(i,k) -= sum_{kk in [ka,k)} tile(i,kk) tile(k,kk)^T
            GemmArgs g{};
            g.A = M; g.B = M; g.C = M;
            g.shape = 1; g.i0 = k; g.j0 = k; g.mi = ntot - k; g.mj = 1; g.sym = aug_sym(col_sym_mode(), short_rows);
            g.k0 = ka; g.k1 = k; g.accumulate = 1; g.nbatch = nb; g.ntiles = g.mi;
            g.short_row0 = nt; g.short_rows = short_rows;
            if (g.sym >= 2 && fuse_mode() && ntot - k - 1 > 0) {
                // diagonal tile first, then its factor + inverse, then ONE pass over the column: update and panel
                // product of every tile below the diagonal (the column makes one HBM round trip instead of two)
                // GPSLC_DIAG_FOLD (measurement switch, default on): update + factorisation of the diagonal tile in ONE
                // launch (diag_update_potrf_kernel) instead of tile_syrk_diag_kernel followed by diag_potrf_inv_la_kernel
                static const int fold = diag_env("GPSLC_DIAG_FOLD", 1);
                const bool fold_ok = fold && info_div == 1;
                if (fold_ok) {
                    GemmArgs d = g;
                    d.F = invref; d.info = info; d.info_base = info_base;
                    launch_diag_update_potrf(d, g.sym == 3 && short_rows > 0, st);
                } else {
                    launch_sym_diag_tiles(g, st);
                    launch_diag(M, k, inv, inv_bstride, info, info_base, nb, st);
                }
                g.fuse = 1; g.F = invref; g.fk = k;
                gemm(c, g, st, prof_base);
                fused = true;
            } else {
                gemm(c, g, st, prof_base);
            }
        }
        if (!fused) launch_diag(M, k, inv, inv_bstride, info, info_base, nb, st);
        if (!fused && ntot - k - 1 > 0) {   // panel: tile(i,k) = tile(i,k) * inv(L_kk)^T
            // GPSLC_PANEL0_STRIP (measurement switch, default on, round 5): the first column of a panel has no column update — its
            // panel product runs as the strip kernel's second phase alone (empty K range: the tile goes from HBM into the
            // accumulators, the fragments of inv(L_kk) from L2 into registers, no LDS staging), same summation order as the
            // general product it replaces (tile_gemm_nt_kernel<0, 0>)
            static const int p0strip = diag_env("GPSLC_PANEL0_STRIP", 1);
            GemmArgs g{};
            g.A = M; g.C = M;
            g.shape = 1; g.i0 = k + 1; g.j0 = k; g.mi = ntot - k - 1; g.mj = 1;
            g.nbatch = nb; g.ntiles = g.mi;
            g.short_row0 = nt; g.short_rows = short_rows;
            if (p0strip && fuse_mode()) {
                g.B = M; g.k0 = k; g.k1 = k; g.accumulate = 1; g.fuse = 1; g.F = invref; g.fk = k;
            } else {
                g.B = invref; g.k0 = k; g.k1 = k + 1; g.accumulate = 0;
            }
            gemm(c, g, st, prof_base);
        }
        // skip_aug_diag (single short augmented row, epilogue sums from the rows of R): the augmented diagonal tile is
        // never updated — after the last panel it would be the launch's only item
        const bool skip_gd = skip_aug_diag && short_rows > 0;
        if (k == kend - 1 && ntot - kend > (skip_gd ? 1 : 0)) {   // trailing update with the whole panel
            GemmArgs g{};
            g.A = M; g.B = M; g.C = M;
            const int m = ntot - kend;
            g.skip_gdiag = skip_gd ? 1 : 0;
            g.shape = 0; g.i0 = kend; g.j0 = kend; g.mi = m; g.mj = m; g.sym = aug_sym(sym_mode(), short_rows);
            g.k0 = ka; g.k1 = kend; g.accumulate = 1; g.nbatch = nb; g.ntiles = m * (m + 1) / 2;
            g.order = tri_order(c, m);
            g.short_row0 = nt; g.short_rows = short_rows;
            gemm(c, g, st, prof_base);
        }
    }
    return false;
}

struct PredictIO {
    int64_t S = 0;
    SampleParams p{};
    const double* X = nullptr;   // device X to use (ctx or override)
    int L = 0;
    const double* doT = nullptr;
    double pred_noise = 0;
    int spp = 0;
    uint64_t seed = 0;
    const double* z = nullptr;
    double *meanSATE = nullptr, *varSATE = nullptr, *meanITE = nullptr, *ite_draws = nullptr;
    double *logdet = nullptr, *quad = nullptr;
    // ITEDistributions-style outputs (single level): MeanITEs S x n, CovITEs S x n x n
    double *MeanITEs = nullptr, *CovITEs = nullptr;
    int* info = nullptr;   // device, S
    int nU = -1, nX = -1;            // feature counts of this call (default: the ctx's)
    const double* Y = nullptr;       // right-hand side 0 (default: the ctx's Y) and its per-sample stride
    long long y_sstride = 0;
    bool p_shared_u = false;         // the feature block is shared by all samples (u_sstride stays 0)
    int64_t ens_off = 0, ens_S = 0;  // ens_S > 0: this call's placement in a larger ensemble (else the ctx's, gpslc_set_ensemble)
    // node draws (gpslc_nodes_draw beyond the single-workgroup kernels): ndraw[:, s] = chol(A_s) nz[:, s] from the batched
    // tiled factor of A_s (n x S each, device; nzero = n x S zeros, the draw kernel's mean)
    const double* nz = nullptr;
    const double* nzero = nullptr;
    double* ndraw = nullptr;
};

// chunk and unit-B sub-batch sizes of a call (batch_plan.h holds the arithmetic; here: what the schedule would like and what
// the device has)
BatchPlan auto_batch(gpslc_ctx* c, int64_t S, int L, size_t per_sample_bytes, long long want_bb, size_t unit_bytes, size_t lvl_extra) {
    // enough matrices in flight that the per-step diagonal-block kernel (one workgroup per matrix) and the
    // launch quantisation of the late, small trailing updates are amortised: 1024 at N = 4096 (82 GB of the
    // 288 GB; measured 2013 / 2040 / 2055 / 2064 samples/s at batch 256 / 512 / 1024 / 2048)
    // small matrices: up to 16,384 per chunk (N = 1024: 4,096 / 8,192 / 16,384 per chunk -> 83.8 k / 84.6 k / 86.5 k samples/s,
    // profiles/r04_ab_experiments.md §16 — every launch's ramp and tail are amortised over more items); memory permitting
    long long b = 1048576LL / ((long long)c->nt * c->nt);
    b = std::max<long long>(32, std::min<long long>(b, 16384));
    if (c->max_batch > 0) b = c->max_batch;
    size_t free_b = 0, tot_b = 0;
    HC(hipMemGetInfo(&free_b, &tot_b));
    size_t have = 0;
    for (auto& a : c->arenas) have += a.bytes;
    // the AUTOMATIC chunk also stays under 30 % of the device's memory (arenas never shrink, and the host process usually
    // shares the device: torch's caching allocator under sharded.py / bench.py, other contexts of gpslc_predict_multi on one
    // GPU); 1,024 matrices at N = 4096 are 28 %.  gpslc_set_tuning(max_batch) may ask for more: then only what is free counts.
    const BatchPlan p = plan_batch(b, want_bb, S, L, per_sample_bytes, unit_bytes, lvl_extra, free_b, have, tot_b, c->nstreams,
                                   c->max_batch <= 0);
    if (!p.ok) throw std::bad_alloc();
    return p;
}

// the chunked ensemble driver (device pointers everywhere)
void run_predict(gpslc_ctx* c, const PredictIO& io_in) {
    PredictIO io = io_in;
    if (io.nU < 0) io.nU = c->nU;
    if (io.nX < 0) io.nX = c->nX;
    if (!io.Y) { io.Y = c->dY; io.y_sstride = 0; }
    if (io.p.u_sstride == 0 && io.nU > 0 && io.p.U != nullptr && !io.p_shared_u) io.p.u_sstride = (long long)c->n * io.nU;
    ensure_streams(c);
    const int n = (int)c->n, nt = c->nt;
    const int L = io.L;
    const int naug = (L + 1 + GP_TS - 1) / GP_TS;
    const int ntot = nt + naug;
    const long long Np = (long long)nt * GP_TS;
    const long long tiles_per = (long long)ntot * (ntot + 1) / 2;
    const bool want_sate = io.meanSATE || io.varSATE;
    const bool want_cov = io.CovITEs != nullptr;
    const bool want_draws = io.ite_draws != nullptr;
    const bool want_mean = io.meanITE || io.MeanITEs || want_draws;
    const bool with_sums = want_sate;
    const long long nlow = (long long)nt * (nt + 1) / 2;

    // unit-B sub-batch (full ITE covariance): W (nt^2 tiles) + Cm (nlow tiles) + inv (nt tiles)
    const bool unitB = want_cov || want_draws;
    const size_t unitB_per = unitB ? (size_t)((long long)nt * nt + nlow) * GP_TSQ * 8 : 0;

    size_t per = (size_t)tiles_per * GP_TSQ * 8      // tiles
               + (size_t)nt * GP_TSQ * 8             // inv
               + (size_t)(with_sums ? 2 * nt * Np * 8 : 0)
               + (size_t)(2 * Np * 8)                // bsum, ksum
               + (size_t)(std::max(L, 1) * 8)        // sumdelta
               + (size_t)(2 * Np * 8)                // zwork + alpha
               + (size_t)(io.ndraw ? 16 * Np * 8 : 0)   // operand image of the node draw's normals
               + 1024;
    int Bb_target = 0;
    if (unitB) {
        // unit-B sub-batch: 128 units at N = 4096 (203 MB each, 26 GB).  Round 5, 256 units of 8 samples x 32 levels on one box
        // (profiles/r05_ab_experiments.md §3): 16 / 32 / 64 / 128 / 256 per sub-batch -> 322 / 349 / 364 / 372 / 373 units/s —
        // the late columns of a factorisation have (nt - k) x sub-batch items for 512 workgroup slots; flat from 128 on
        Bb_target = (int)std::max<long long>(1, std::min<long long>(128, 131072LL / ((long long)nt * nt)));
        static const int bb_env = diag_env("GPSLC_UNITB_BATCH", 0);      // measurement switch: the sub-batch curve (profiles/r05)
        if (bb_env > 0) Bb_target = bb_env;
    }
    // draws: normals workspace of one level of the sub-batch + the level-sweep staging buffer (see the unit-B loop)
    // (per unit of the sub-batch: the staging buffer holds max(1, Bb / L) samples x L levels <= Bb pairs, or one sample's
    // L > Bb levels — sized below through the extra term)
    // spp <= 128: the streaming draw kernel reads the unit's normals (caller's or Philox) from an operand image of
    // 16 Np doubles per block of 16 draws (DrawArgs::zt) instead
    const bool zimage = want_draws && io.spp <= 128;
    // per unit: 1 / 2 / 4 / 8 blocks of 16 draws (draws_nq, k_solve.hip: the stream kernel's template parameter)
    const size_t zimage_doubles = zimage ? (size_t)16 * (io.spp <= 16 ? 1 : io.spp <= 32 ? 2 : io.spp <= 64 ? 4 : 8) * Np : 0;
    const size_t draws_per = want_draws ? ((zimage ? zimage_doubles * 8 + 256 : (io.z ? 0 : (size_t)io.spp * n * 8 + 256)) +
                                           (L > 1 ? (size_t)io.spp * n * 8 + 256 : 0)) : 0;
    const size_t unitB_all = unitB_per + draws_per;
    // chunk size and sub-batch from what the device has (the sub-batch's workspace is sized from the pairs the call
    // really has, and the 30 % rule of the automatic chunk covers its per-sample part only)
    const BatchPlan plan = auto_batch(c, io.S, L, per, Bb_target, unitB_all, want_draws ? (size_t)io.spp * n * 8 : 0);
    const int Bt = (int)plan.Bt;
    const int Bb = (int)plan.Bb;      // (sample, level) pairs per unit-B sub-batch: no more than the call has
    const size_t need = (size_t)Bt * per + plan.fixed;
    for (int i = 0; i < c->nstreams; ++i) arena_reserve(c, c->arenas[i], need);

    // internal MeanITE buffer when the caller did not ask for it but the draws need it
    double* meanITE = io.meanITE;
    if (want_draws && !meanITE) {
        arena_reserve(c, c->scratch, std::max(c->scratch.bytes, (size_t)n * io.S * L * 8 + (1 << 16)));
        c->scratch.reset();
        meanITE = c->scratch.take<double>((size_t)n * io.S * L);
    }

    HC(hipMemsetAsync(io.info, 0, sizeof(int) * io.S, c->streams[0]));
    HC(hipStreamSynchronize(c->streams[0]));

    int chunk = 0;
    for (int64_t s0 = 0; s0 < io.S; s0 += Bt, ++chunk) {
        const int nb = (int)std::min<int64_t>(Bt, io.S - s0);
        const int slot = chunk % c->nstreams;
        hipStream_t st = c->streams[slot];
        Arena& ar = c->arenas[slot];
        ar.reset();
        double* tiles = ar.take<double>((size_t)nb * tiles_per * GP_TSQ);
        double* inv = ar.take<double>((size_t)nb * nt * GP_TSQ);
        double* part = with_sums ? ar.take<double>((size_t)nb * 2 * nt * Np) : nullptr;
        double* bsum = ar.take<double>((size_t)nb * Np);
        double* ksum = ar.take<double>((size_t)nb * Np);
        double* sumdelta = ar.take<double>((size_t)nb * std::max(L, 1));
        double* zwork = ar.take<double>((size_t)2 * nb * Np);
        const long long bstride = tiles_per * GP_TSQ;
        const long long inv_bs = (long long)nt * GP_TSQ;
        TRef M = lower_ref(tiles, bstride);

        GramArgs ga{};
        ga.X = io.X; ga.T = c->dT; ga.p = io.p; ga.s0 = s0;
        ga.n = n; ga.nX = io.nX; ga.nU = io.nU; ga.nt = nt; ga.M = M; ga.part = part;
        ga.with_sums = with_sums ? 1 : 0;
#ifdef GPSLC_DIAG
        static const int gram_dbg = diag_env("GPSLC_GRAM_DBG", 0);
        static bool gram_dbg_done = false;
        DevBuf gdbg;
        if (gram_dbg && !gram_dbg_done) {
            const size_t words = (size_t)8 * nb * (nt * (nt + 1) / 2);
            gdbg.alloc(words * 8);
            HC(hipMemset(gdbg.p, 0, words * 8));
            ga.dbg = gdbg.as<unsigned long long>();
        }
#endif
        ga.f32 = (c->flags & GPSLC_FLAG_FP32_KERNEL) ? 1 : 0;
        ga.binary_t = (c->binary_t && io.nU == c->nU && io.nX == c->nX && io.Y == c->dY) ? 1 : 0;   // ctx data only
        launch_gram(ga, nb, st);
#ifdef GPSLC_DIAG
        if (ga.dbg) {      // GPSLC_GRAM_DBG: dump the first launch's per-workgroup stamps
            HC(hipStreamSynchronize(st));
            const size_t words = (size_t)8 * nb * (nt * (nt + 1) / 2);
            std::vector<unsigned long long> h(words);
            HC(hipMemcpy(h.data(), ga.dbg, words * 8, hipMemcpyDeviceToHost));
            FILE* f = fopen("gpurun_out/gram_dbg.bin", "wb");
            if (f) { fwrite(h.data(), 8, words, f); fclose(f); }
            gram_dbg_done = true;
        }
#endif

        RhsArgs ra{};
        ra.T = c->dT; ra.Y = io.Y; ra.y_sstride = io.y_sstride; ra.tyLS = io.p.tyLS; ra.doT = io.doT; ra.s0 = s0;
        ra.n = n; ra.nt = nt; ra.naug = naug; ra.L = with_sums ? L : 0; ra.with_sums = with_sums ? 1 : 0;
        ra.part = part; ra.bsum = bsum; ra.ksum = ksum; ra.sumdelta = sumdelta; ra.M = M;
        static const int epi_rows_on = diag_env("GPSLC_EPI_ROWS", 1);      // measurement switch (A/B of the extra tile update)
        const bool epi_rows = (naug == 1) && epi_rows_on;
        {
            const int live = (with_sums ? L : 0) + 1;                         // right-hand sides: Y and one c_l per level
            ra.live_rows = (epi_rows && live <= 32) ? 16 * ((live + 15) / 16) : 0;
        }
        launch_rhs(ra, nb, st);

        // NB: the MeanITE pass takes K alpha as Y - yNoise alpha (k_solve.hip, ite_mean_kernel): it relies on alpha solving
        // EXACTLY (K_gram + yNoise I) alpha = Y.  Any future jitter, robust fallback or different right-hand side in this
        // factorisation must be reflected there (tests: test_mean_ite_tiny_noise_and_near_coincident_levels).
        const bool back_done = potrf_tiles(c, M, nt, ntot, inv, inv_bs, io.info + s0, 0, nb, st, (with_sums ? L : 0) + 1, 0, false, 1,
                                           epi_rows, /*tasks=*/true, want_mean ? zwork + (long long)nb * Np : nullptr);

        EpiArgs ea{};
        ea.M = M; ea.n = n; ea.nt = nt; ea.naug = naug; ea.L = with_sums ? L : 0; ea.s0 = s0; ea.S = io.S;
        ea.sumdelta = sumdelta; ea.pred_noise = io.pred_noise;
        ea.meanSATE = io.meanSATE; ea.varSATE = io.varSATE; ea.logdet = io.logdet; ea.quad = io.quad;
        ea.from_rows = epi_rows ? 1 : 0;
        launch_epilogue(ea, nb, st);

        if (io.ndraw) {
            // one draw from N(0, A_s) per parameter set with the caller's normals: L_s z_s on the factor just computed —
            // Gen's mvnormal(zeros(n), cov) of an elliptical slice's auxiliary vector (src/inference.jl:225-232) and of the
            // :logitT prior draw (src/model_likelihood.jl:25-33) — the predictive-draw kernel streaming L once
            double* zt = ar.take<double>((size_t)nb * 16 * Np);
            DrawArgs dr{};
            dr.Lc = M; dr.n = n; dr.nt = nt; dr.s0 = s0; dr.S = io.S; dr.l = 0; dr.lc = 1; dr.L = 1; dr.spp = 1;
            dr.mean = io.nzero; dr.z = io.nz; dr.zt = zt; dr.out = io.ndraw;
            dr.obase = (long long)n * s0; dr.osb = n; dr.osl = 0; dr.osi = 1; dr.osd = n;
            launch_draws(dr, nb, st);
        }

        if (want_mean) {
            BackArgs ba{};
            ba.M = M; ba.inv = inv; ba.inv_bstride = inv_bs; ba.nt = nt; ba.naug = naug; ba.zwork = zwork;
            if (!back_done) launch_backsolve(ba, nb, st);
            const double* alpha = zwork + (long long)nb * Np;
            IteMeanArgs ia{};
            ia.X = io.X; ia.T = c->dT; ia.p = io.p; ia.s0 = s0; ia.S = io.S;
            ia.n = n; ia.nX = io.nX; ia.nU = io.nU; ia.nt = nt; ia.L = L; ia.doT = io.doT; ia.alpha = alpha;
            ia.Y = io.Y; ia.y_sstride = io.y_sstride; ia.yNoise = io.p.yNoise;
            ia.f32 = (c->flags & GPSLC_FLAG_FP32_KERNEL) ? 1 : 0;
            if (meanITE) {
                ia.meanITE = meanITE; ia.si = 1; ia.ss = n; ia.sl = (long long)n * io.S;
                launch_ite_mean(ia, nb, st);
            }
            if (io.MeanITEs) {   // reference layout S x n (single level)
                ia.meanITE = io.MeanITEs; ia.si = io.S; ia.ss = 1; ia.sl = 0;
                launch_ite_mean(ia, nb, st);
            }
        }

        if (unitB) {
            // Unit B batches over (posterior sample, level) PAIRS: a sub-batch is gs samples x lc levels (gs * lc <= Bb),
            // batch element b = (sample b / lc, level l0 + b % lc).  Every pair has its own W / CovITE workspace; the
            // factor of A and its inverted diagonal blocks are shared by a sample's levels (TRef::bdiv = lc) — the shape of
            // predictCounterfactualEffects (src/prediction.jl:30-33): few samples, ~100 levels.
            double* Wt = ar.take<double>((size_t)Bb * nt * nt * GP_TSQ);
            double* Ct = ar.take<double>((size_t)Bb * nlow * GP_TSQ);
            const int lc_max = std::min(L, Bb);                   // levels per sub-batch
            const int gs_max = std::max(1, Bb / lc_max);          // samples per sub-batch
            // draws: the library's own normals of one sub-batch (generated once per unit), and for a level sweep the
            // staging buffer [sample][level][d][i] that is rearranged into the level-fastest tensor sample group by group
            double* zgen = (want_draws && !io.z && !zimage) ? ar.take<double>((size_t)Bb * io.spp * n) : nullptr;
            double* zt = zimage ? ar.take<double>((size_t)Bb * zimage_doubles) : nullptr;
            double* dtmp = (want_draws && L > 1) ? ar.take<double>((size_t)gs_max * L * io.spp * n) : nullptr;
            const long long wbs = (long long)nt * nt * GP_TSQ, cbs = nlow * GP_TSQ;
            for (int g0 = 0; g0 < nb; g0 += gs_max) {
                const int gs = std::min(gs_max, nb - g0);
                for (int l0 = 0; l0 < L; l0 += lc_max) {
                    const int lc = std::min(lc_max, L - l0);
                    const int ub = gs * lc;                           // (sample, level) pairs of this sub-batch
                    TRef W = rect_ref(Wt, wbs, nt);
                    TRef Cm = lower_ref(Ct, cbs);
                    TRef Ls = lower_ref(tiles + (long long)g0 * bstride, bstride);
                    Ls.bdiv = lc;
                    DtArgs da{};
                    da.X = io.X; da.T = c->dT; da.p = io.p; da.s0 = s0 + g0;
                    da.n = n; da.nX = io.nX; da.nU = io.nU; da.nt = nt; da.doT = io.doT; da.l0 = l0; da.lc = lc;
                    da.pred_noise = io.pred_noise; da.W = W; da.Cm = Cm;
                    launch_dt_build(da, ub, st);
                    // W <- D L^-T, left-looking over tile columns; the panel product with inv(L_kk)^T is applied by the
                    // same work item that finishes the column update (the tile makes one HBM round trip, as in the
                    // factorisation), column 0 has no update and takes the panel product alone
                    TRef invref = TRef{inv + (long long)g0 * inv_bs, inv_bs, 1, 0, 0, 0};
                    invref.bdiv = lc;
                    // Left-looking over the whole width (every column update in the strip kernel).  GPSLC_W_PANEL = w > 0
                    // (measurement build) blocks it like the factorisation instead — left-looking inside panels of w tile
                    // columns, one right-looking update of everything to the right of a panel on the quadrant kernel.
                    // Measured in round 4 (profiles/r04_ab_experiments.md §7): w = 4 / 8 / 16 -> 354 / 360 / 363.5 units/s
                    // against 363 unblocked: with a deep K loop the strip kernel is as fast as the quadrant kernel's
                    // rectangle update plus the extra pass over W.  Not the default.
                    static const int wpw_env = diag_env("GPSLC_W_PANEL", 0);
                    const int wpw = wpw_env <= 0 ? nt : wpw_env;
                    for (int k = 0; k < nt; ++k) {
                        const int ka = (k / wpw) * wpw;
                        const int kend = std::min(ka + wpw, nt);
                        GemmArgs g{};
                        g.A = W; g.C = W;
                        g.shape = 1; g.i0 = 0; g.j0 = k; g.mi = nt; g.mj = 1; g.nbatch = ub; g.ntiles = nt;
                        // fused up to a K depth of w_fuse_maxk tiles (measurement switch; +0.7 % fused at every depth)
                        static const int w_fuse_maxk = diag_env("GPSLC_FUSE_W_MAXK", 32);
                        if (k - ka <= w_fuse_maxk && fuse_mode()) {      // k == ka: empty K range, the panel product alone (round 5)
                            g.B = Ls; g.k0 = ka; g.k1 = k; g.accumulate = 1;
                            g.fuse = 1; g.F = invref; g.fk = k;
                            gemm(c, g, st, 3);
                        } else {
                            if (k > ka) {
                                g.B = Ls; g.k0 = ka; g.k1 = k; g.accumulate = 1;
                                gemm(c, g, st, 3);
                            }
                            g.B = invref; g.k0 = k; g.k1 = k + 1; g.accumulate = 0;
                            gemm(c, g, st, 3);
                        }
                        if (k == kend - 1 && kend < nt) {      // the panel is solved: update everything to its right
                            GemmArgs t{};
                            t.A = W; t.B = Ls; t.C = W;
                            t.shape = 1; t.i0 = 0; t.j0 = kend; t.mi = nt; t.mj = nt - kend;
                            t.k0 = ka; t.k1 = kend; t.accumulate = 1; t.nbatch = ub; t.ntiles = t.mi * t.mj;
                            gemm(c, t, st, 3);
                        }
                    }
                    {   // CovITE (+ jitter) = Delta - W W^T, lower tiles
                        GemmArgs g{};
                        g.A = W; g.B = W; g.C = Cm;
                        g.shape = 0; g.i0 = 0; g.j0 = 0; g.mi = nt; g.mj = nt; g.sym = sym_mode();
                        g.k0 = 0; g.k1 = nt; g.accumulate = 1; g.nbatch = ub; g.ntiles = (int)nlow;
                        g.order = tri_order(c, nt);
                        gemm(c, g, st, 3);
                    }
                    if (want_cov) {       // ITEDistributions: single level (lc == 1)
                        GatherCovArgs gc{};
                        gc.Cm = Cm; gc.n = n; gc.nt = nt; gc.s0 = s0 + g0; gc.S = io.S; gc.out = io.CovITEs;
                        launch_gather_cov(gc, ub, st);
                    }
                    if (want_draws) {
                        potrf_tiles(c, Cm, nt, nt, nullptr, 0, io.info + s0 + g0, n, ub, st, 0, 3, /*robust=*/true, /*info_div=*/lc);
                        DrawArgs dr{};
                        dr.Lc = Cm; dr.n = n; dr.nt = nt; dr.s0 = s0 + g0; dr.S = io.S; dr.l = l0; dr.lc = lc; dr.L = L;
                        dr.spp = io.spp; dr.mean = meanITE; dr.z = io.z; dr.zgen = zgen; dr.zt = zt; dr.seed = io.seed;
                        {
                            const int64_t eS = io.ens_S > 0 ? io.ens_S : c->ens_S, eo = io.ens_S > 0 ? io.ens_off : c->ens_off;
                            dr.rs0 = dr.s0 + (eS > 0 ? eo : 0); dr.rS = eS > 0 ? eS : io.S;
                        }
                        if (L == 1) {     // the reference tensor directly: n x (S*spp), instance fastest
                            dr.out = io.ite_draws;
                            dr.obase = (long long)n * io.spp * (s0 + g0); dr.osb = (long long)n * io.spp; dr.osl = 0;
                            dr.osi = 1; dr.osd = n;
                        } else {          // staging [sample][level][d][i]
                            dr.out = dtmp;
                            dr.obase = (long long)l0 * io.spp * n; dr.osb = (long long)L * io.spp * n;
                            dr.osl = (long long)io.spp * n; dr.osi = 1; dr.osd = n;
                        }
                        ProfScope ps(c, 2, (double)ub * io.spp, st);     // unit C: work = draws
                        launch_draws(dr, ub, st);
                    }
                }
                if (want_draws && L > 1)
                    launch_draws_scatter(dtmp, io.ite_draws, n, L, io.spp, s0 + g0, gs, st);
            }
        }
    }
    for (auto st : c->streams) HC(hipStreamSynchronize(st));
    HC(hipGetLastError());
    check_task_timeout(c);
    if (c->flags & GPSLC_FLAG_PROFILE) prof_collect(c);
    c->last_info.resize(io.S);
    HC(hipMemcpy(c->last_info.data(), io.info, sizeof(int) * io.S, hipMemcpyDeviceToHost));
}

int first_info(const gpslc_ctx* c) {
    for (int32_t v : c->last_info)
        if (v != 0) return v;
    return 0;
}

template <class F>
int guarded(gpslc_ctx* c, F&& f) {
    try {
        if (c) HC(hipSetDevice(c->device));
        return f();
    } catch (const HipFail& h) {
        return fail_hip(c, h);
    } catch (const std::bad_alloc&) {
        set_err(c, "device workspace allocation failed");
        return GPSLC_ERR_NOMEM;
    } catch (const std::runtime_error& e) {
        set_err(c, std::string("internal error: ") + e.what());
        return GPSLC_ERR_INTERNAL;
    } catch (...) {
        set_err(c, "internal error");
        return GPSLC_ERR_INTERNAL;
    }
}

int bad_arg(gpslc_ctx* c, int k, const char* what) {
    char buf[160];
    snprintf(buf, sizeof buf, "argument #%d is invalid: %s", k, what);
    set_err(c, buf);
    return -k;
}

// host-side Philox (same stream definition as the device code / the oracle)
void philox_host(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t o[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
double philox_normal_host(uint64_t seed, uint64_t stream, uint64_t e) {
    uint32_t w[4];
    const uint64_t pair = e >> 1;
    philox_host((uint32_t)pair, (uint32_t)(pair >> 32), (uint32_t)stream, (uint32_t)(stream >> 32),
                (uint32_t)seed, (uint32_t)(seed >> 32), w);
    const uint64_t A = ((uint64_t)w[0] << 21) ^ ((uint64_t)w[1] >> 11);
    const uint64_t B = ((uint64_t)w[2] << 21) ^ ((uint64_t)w[3] >> 11);
    const double u1 = ((double)A + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)B + 0.5) * (1.0 / 9007199254740992.0);
    const double rad = std::sqrt(-2.0 * std::log(u1));
    const double ang = 6.283185307179586476925286766559 * u2;
    return (e & 1) ? rad * std::sin(ang) : rad * std::cos(ang);
}

double* up(DevBuf& b, const double* host, size_t count) {
    b.alloc(count * sizeof(double));
    if (count) HC(hipMemcpy(b.p, host, count * sizeof(double), hipMemcpyHostToDevice));
    return b.as<double>();
}
// upload into the ctx's staging pool (no allocation in steady state)
double* up(gpslc_ctx* c, const double* host, size_t count) {
    double* d = c->io.take<double>(count);
    if (count) HC(hipMemcpy(d, host, count * sizeof(double), hipMemcpyHostToDevice));
    return d;
}

int check_common(gpslc_ctx* c, int64_t S, const double* U, const double* uyLS, const double* xyLS,
                 const double* tyLS, const double* yScale, const double* yNoise) {
    if (!c) return -1;
    if (!c->has_data) { set_err(c, "gpslc_set_data has not been called"); return GPSLC_ERR_NODATA; }
    if (S < 0) return bad_arg(c, 2, "S < 0");
    if (c->nU > 0 && S > 0 && (!U || !uyLS)) return bad_arg(c, 3, "U / uyLS must not be NULL when nU > 0");
    if (c->nX > 0 && S > 0 && !xyLS) return bad_arg(c, 5, "xyLS must not be NULL when nX > 0");
    if (S > 0 && (!tyLS || !yScale || !yNoise)) return bad_arg(c, 6, "tyLS / yScale / yNoise must not be NULL");
    return 0;
}

// ---- single-launch small-n node scores ----------------------------------------------------------------------------
constexpr size_t kLdsBytes = 160 * 1024;

struct HostNode {            // host view of one node (gpslc_node with plain pointers)
    int nF;
    const double* F[2];      // up to two column groups + one extra column are concatenated into the feature block:
    int nFpart[2];           //   F[0] (nFpart[0] columns), F[1] (nFpart[1] columns), then `col` (one column) if non-null
    const double* col;
    const double* ls[2];     // lengthscales of the two groups; ls_col for the extra column
    double ls_col;
    double scale, noise;
    const double* target;
    const double* dev_cov;   // dense-covariance node: n x n in device memory (else null), and its scale factor
    double covscale;
};

void pin_reserve(gpslc_ctx* c, size_t bytes) {
    if (c->pin_bytes >= bytes) return;
    if (c->pin) { HC(hipDeviceSynchronize()); HC(hipHostFree(c->pin)); c->pin = nullptr; c->pin_bytes = 0; }
    void* p = nullptr;
    bytes = (bytes + 4095) & ~size_t(4095);
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); throw std::bad_alloc(); }
    c->pin = static_cast<char*>(p);
    c->pin_bytes = bytes;
}

// A node score can run as ONE workgroup of one launch: matrix resident in LDS (n <= 160..176), or the left-looking
// kernel with the finished block columns in an L2-resident scratch (n <= 640).  The latter keeps one CU busy per node
// for 100+ us: worth it up to a few hundred nodes per call, beyond that the batched tiled path wins.
bool fast_path_ok(const gpslc_ctx* c, int nF_max, int64_t count) {
    if (c->flags & GPSLC_FLAG_FP32_KERNEL) return false;
    if (small_gp_lds_bytes((int)c->n, nF_max) <= kLdsBytes) return count <= 4096;
    return mid_gp_fits((int)c->n) && count <= 512;
}

// Device -> host hand-over of a LARGE result (draw tensors, MeanITE of a level sweep: GB).  A plain hipMemcpy into pageable
// memory runs at ~10 GB/s (the runtime's own bounce copies + first-touch page faults of a fresh destination, on one
// thread) against 57 GB/s of DMA into pinned memory (tools/bench_small_n_predict.py).  Here the DMA of chunk k + 1 into
// one of two pinned chunks overlaps the copy of chunk k into the caller's buffer, and that copy is split over a few
// threads: 1.1 GB in 77 ms instead of 110 (a huge-page hint on the destination changed nothing: NumPy's large arrays
// already carry it).  Small results keep the plain call; a caller who hands over a PINNED buffer gets the DMA rate.
constexpr size_t kBounceBytes = size_t(64) << 20;
void copy_out_large(gpslc_ctx* c, void* dst, const void* src_dev, size_t bytes) {
    bool plain = bytes < 2 * kBounceBytes;
    if (!plain) {                                     // a pinned (registered) destination takes the DMA directly
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) == hipSuccess) plain = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();                 // ordinary pageable memory: not known to the runtime
    }
    if (plain) { HC(hipMemcpy(dst, src_dev, bytes, hipMemcpyDeviceToHost)); return; }
    ensure_streams(c);
    for (int i = 0; i < 2; ++i)
        if (!c->bounce[i]) {
            void* p = nullptr;
            if (hipHostMalloc(&p, kBounceBytes, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                HC(hipMemcpy(dst, src_dev, bytes, hipMemcpyDeviceToHost));      // no pinned memory to be had: the plain way
                return;
            }
            c->bounce[i] = static_cast<char*>(p);
        }
    hipStream_t st = c->streams[0];
    const size_t nchunk = (bytes + kBounceBytes - 1) / kBounceBytes;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nthr = std::min(8u, hw);
    auto chunk_bytes = [&](size_t k) { return std::min(kBounceBytes, bytes - k * kBounceBytes); };
    HC(hipMemcpyAsync(c->bounce[0], src_dev, chunk_bytes(0), hipMemcpyDeviceToHost, st));
    for (size_t k = 0; k < nchunk; ++k) {
        HC(hipStreamSynchronize(st));                                           // chunk k has landed in bounce[k & 1]
        if (k + 1 < nchunk)
            HC(hipMemcpyAsync(c->bounce[(k + 1) & 1], static_cast<const char*>(src_dev) + (k + 1) * kBounceBytes,
                              chunk_bytes(k + 1), hipMemcpyDeviceToHost, st));
        const size_t cb = chunk_bytes(k);
        char* d = static_cast<char*>(dst) + k * kBounceBytes;
        const char* b = c->bounce[k & 1];
        const size_t per = ((cb + nthr - 1) / nthr + 4095) & ~size_t(4095);
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthr; ++t) {
            const size_t o = (size_t)t * per;
            if (o >= cb) break;
            const size_t len = std::min(per, cb - o);
            try { pool.emplace_back([=]() { memcpy(d + o, b + o, len); }); }
            catch (...) { memcpy(d + o, b + o, len); }        // no thread to be had: copy this slice here
        }
        memcpy(d, b, std::min(per, cb));
        for (auto& th : pool) th.join();
    }
}

// The same hand-over for a result whose rows are CONTIGUOUS ON THE DEVICE (rows x row_bytes) and `dpitch` bytes apart in the
// caller's array: a shard's block of an (n x S x L) array of a level sweep — level l of the shard is one run of n S_r doubles
// at n (s0 + S l) of the caller's array (gpslc_predict_multi).  The device side is DMA'd in 64 MiB chunks as above; the host
// threads scatter a chunk to its rows.  Small results: one hipMemcpy2D.
void copy_out_rows(gpslc_ctx* c, void* dst, size_t dpitch, const void* src_dev, size_t row_bytes, size_t rows) {
    if (rows == 0 || row_bytes == 0) return;
    if (rows == 1 || dpitch == row_bytes) { copy_out_large(c, dst, src_dev, row_bytes * rows); return; }
    const size_t bytes = row_bytes * rows;
    bool plain = bytes < 2 * kBounceBytes;
    if (!plain) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) == hipSuccess) plain = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    if (!plain) {
        ensure_streams(c);
        for (int i = 0; i < 2 && !plain; ++i)
            if (!c->bounce[i]) {
                void* p = nullptr;
                if (hipHostMalloc(&p, kBounceBytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); plain = true; }
                else c->bounce[i] = static_cast<char*>(p);
            }
    }
    if (plain) { HC(hipMemcpy2D(dst, dpitch, src_dev, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost)); return; }
    hipStream_t st = c->streams[0];
    const size_t nchunk = (bytes + kBounceBytes - 1) / kBounceBytes;
    const unsigned nthr = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    auto chunk_bytes = [&](size_t k) { return std::min(kBounceBytes, bytes - k * kBounceBytes); };
    // bytes [o, o + len) of the device block -> their rows of the caller's array
    auto scatter = [=](const char* b, size_t o, size_t len) {
        while (len > 0) {
            const size_t row = o / row_bytes, within = o - row * row_bytes;
            const size_t m = std::min(len, row_bytes - within);
            memcpy(static_cast<char*>(dst) + row * dpitch + within, b, m);
            b += m; o += m; len -= m;
        }
    };
    HC(hipMemcpyAsync(c->bounce[0], src_dev, chunk_bytes(0), hipMemcpyDeviceToHost, st));
    for (size_t k = 0; k < nchunk; ++k) {
        HC(hipStreamSynchronize(st));
        if (k + 1 < nchunk)
            HC(hipMemcpyAsync(c->bounce[(k + 1) & 1], static_cast<const char*>(src_dev) + (k + 1) * kBounceBytes,
                              chunk_bytes(k + 1), hipMemcpyDeviceToHost, st));
        const size_t cb = chunk_bytes(k), o0 = k * kBounceBytes;
        const char* b = c->bounce[k & 1];
        const size_t per = ((cb + nthr - 1) / nthr + 4095) & ~size_t(4095);
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthr; ++t) {
            const size_t o = (size_t)t * per;
            if (o >= cb) break;
            const size_t len = std::min(per, cb - o);
            try { pool.emplace_back([=]() { scatter(b + o, o0 + o, len); }); }
            catch (...) { scatter(b + o, o0 + o, len); }
        }
        scatter(b, o0, std::min(per, cb));
        for (auto& th : pool) th.join();
    }
}

// scores `count` nodes in ONE launch; logdet/quad/info per node come back through the pinned buffer.
// Returns the first failing pivot (0 = all fine); logpdf[i] = -(n log 2pi + logdet_i + quad_i) / 2.
// draw_out (optional, host, n x count): node i also returns chol(K_i) * target_i.
int small_nodes_logpdf(gpslc_ctx* c, int count, const HostNode* nodes, double* logpdf, double* draw_out = nullptr) {
    ensure_streams(c);
    const size_t n = (size_t)c->n;
    size_t doubles = 0;
    int nF_max = 0;
    for (int i = 0; i < count; ++i) {
        doubles += n * nodes[i].nF + n;
        nF_max = std::max(nF_max, nodes[i].nF);
    }
    const size_t off_out = ((size_t)count * sizeof(SmallNode) + 63) & ~size_t(63);
    // 4 results + 8 stamp words per node (+ 8 x 8 per-wave words of node 0 for the mid-size kernel: measurement build)
    const size_t off_data = off_out + ((size_t)count * 12 + 64) * sizeof(double);
    const size_t off_draw = off_data + doubles * sizeof(double);          // [count][n] draws (pinned, written by the kernel)
    pin_reserve(c, off_draw + (draw_out ? (size_t)count * n * sizeof(double) : 0));
    void* dbase = nullptr;
    HC(hipHostGetDevicePointer(&dbase, c->pin, 0));
    char* dev = static_cast<char*>(dbase);
    SmallNode* hn = reinterpret_cast<SmallNode*>(c->pin);
    double* hout = reinterpret_cast<double*>(c->pin + off_out);
    double* hd = reinterpret_cast<double*>(c->pin + off_data);
    double* dd = reinterpret_cast<double*>(dev + off_data);
    size_t o = 0;
    // features are staged already divided by their lengthscale: x * (1 / l), the arithmetic of the general path
    auto put_scaled = [&](const double* col, double l) {
        const double il = 1.0 / l;
        for (size_t r = 0; r < n; ++r) hd[o + r] = col[r] * il;
        o += n;
    };
    for (int i = 0; i < count; ++i) {
        const HostNode& h = nodes[i];
        SmallNode& sn = hn[i];
        sn.nF = h.nF; sn.pad_ = 0; sn.scale = h.scale; sn.noise = h.noise;
        sn.cov = h.dev_cov; sn.covscale = h.covscale;
        sn.Fs = dd + o;
        for (int g = 0; g < 2; ++g)
            for (int f = 0; f < h.nFpart[g]; ++f) put_scaled(h.F[g] + (size_t)f * n, h.ls[g][f]);
        if (h.col) put_scaled(h.col, h.ls_col);
        sn.target = dd + o;
        memcpy(hd + o, h.target, n * sizeof(double));
        o += n;
    }
    SmallArgs a{};
    a.nodes = reinterpret_cast<const SmallNode*>(dev);
    for (int i = 0; i < std::min(count, SMALL_INLINE_NODES); ++i) a.inl[i] = hn[i];
    a.n = (int)n; a.NB = (int)((n + 15) / 16);
    a.out = reinterpret_cast<double*>(dev + off_out);
    a.stamps = nullptr;
    a.draw = draw_out ? reinterpret_cast<double*>(dev + off_draw) : nullptr;
    const bool in_lds = small_gp_lds_bytes((int)n, nF_max) <= kLdsBytes;
    if (!in_lds) {       // mid-size kernel: per-node scratch for the finished block columns and the scaled features
        const size_t per = (mid_gp_scratch_doubles((int)n, nF_max) + 31) & ~size_t(31);
        arena_reserve(c, c->scratch, per * (size_t)count * sizeof(double) + 256);
        c->scratch.reset();
        a.scratch = c->scratch.take<double>(per * (size_t)count);
        a.scratch_stride = (long long)per;
    }
#ifdef GPSLC_DIAG
    static const int want_stamps = diag_env("GPSLC_SMALL_STAMPS", 0);
    if (want_stamps) a.stamps = a.out + 4 * (size_t)count;
#endif
    hipStream_t st = c->streams[0];
    if (in_lds) launch_small_gp(a, count, nF_max, st);
    else launch_mid_gp(a, count, nF_max, st);
    HC(hipGetLastError());
    HC(hipStreamSynchronize(st));
#ifdef GPSLC_DIAG
    if (want_stamps && in_lds) {
        static int printed = 0;
        const double* sp = hout + 4 * (size_t)count;
        if (printed++ < want_stamps)
            fprintf(stderr, "small_gp stamps (shader clocks): inputs %.0f gram %.0f factor+panel %.0f (wave 0 inside the pivot chains: %.0f) update %.0f total %.0f\n",
                    sp[0], sp[1], sp[2], sp[4], sp[3], sp[5]);
    }
    if (want_stamps && !in_lds) {      // node 0, per wave: cumulative shader clocks of the phases over all block columns
        static int printed = 0;
        const double* sp = hout + 12 * (size_t)count;
        if (printed++ < want_stamps)
            for (int w = 0; w < 8; ++w)
                fprintf(stderr, "mid_gp wave %d: gram %.0f k-loop %.0f to-LDS %.0f wait %.0f factor+store %.0f wait %.0f\n", w,
                        sp[8 * w], sp[8 * w + 1], sp[8 * w + 2], sp[8 * w + 3], sp[8 * w + 4], sp[8 * w + 5]);
    }
#endif
    if (draw_out) memcpy(draw_out, c->pin + off_draw, (size_t)count * n * sizeof(double));
    const double l2pi = 1.8378770664093454835606594728112;
    int first = 0;
    c->last_info.resize((size_t)count);
    for (int i = 0; i < count; ++i) {
        const int info = (int)hout[4 * i + 2];
        c->last_info[i] = info;
        if (info != 0 && first == 0) first = info;
        logpdf[i] = -0.5 * ((double)n * l2pi + hout[4 * i] + hout[4 * i + 1]);
    }
    return first;
}

}  // namespace

extern "C" {

const char* gpslc_version(void) { return "gpslc_hip 0.1 gfx950"; }

int gpslc_create(gpslc_ctx** out, int device, int64_t n, int32_t nX, int32_t nU, uint32_t flags) {
    if (!out) return -1;
    *out = nullptr;
    if (n < 1) return -3;
    if (nX < 0) return -4;
    if (nU < 0) return -5;
    if (nX + nU > 32) return -4;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return GPSLC_ERR_NODEVICE; }
    if (device < 0 || device >= ndev) return -2;
    gpslc_ctx* c = new (std::nothrow) gpslc_ctx();
    if (!c) return GPSLC_ERR_NOMEM;
    c->device = device; c->n = n; c->nX = nX; c->nU = nU; c->flags = flags;
    c->nt = (int)((n + GP_TS - 1) / GP_TS);
    c->order_block = diag_env("GPSLC_ORDER_BLOCK", c->order_block);
    int rc = guarded(c, [&]() {
        hipDeviceProp_t prop;
        HC(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            set_err(c, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
            return GPSLC_ERR_NODEVICE;
        }
        HC(hipMalloc((void**)&c->dX, sizeof(double) * std::max<int64_t>(1, n * nX)));
        HC(hipMalloc((void**)&c->dT, sizeof(double) * n));
        HC(hipMalloc((void**)&c->dY, sizeof(double) * n));
        // defined contents before gpslc_set_data: the node-score entry points that bring their own features and
        // target (gpslc_gp_logpdf, gpslc_mvn_logpdf) work on a ctx without data
        HC(hipMemset(c->dX, 0, sizeof(double) * std::max<int64_t>(1, n * nX)));
        HC(hipMemset(c->dT, 0, sizeof(double) * n));
        HC(hipMemset(c->dY, 0, sizeof(double) * n));
        ensure_streams(c);
        return GPSLC_OK;
    });
    if (rc != GPSLC_OK) { gpslc_destroy(c); return rc; }
    *out = c;
    return GPSLC_OK;
}

int gpslc_destroy(gpslc_ctx* c) {
    if (!c) return GPSLC_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto s : c->streams) (void)hipStreamDestroy(s);
    for (auto& a : c->arenas) if (a.base) (void)hipFree(a.base);
    for (int* q : c->queues) if (q) (void)hipFree(q);
    for (int* q : c->task_sync) if (q) (void)hipFree(q);
    for (auto& t : c->task_lists) if (t.dev) (void)hipFree(t.dev);
    if (c->scratch.base) (void)hipFree(c->scratch.base);
    c->io.release();
    if (c->pin) (void)hipHostFree(c->pin);
    for (char* b : c->bounce) if (b) (void)hipHostFree(b);
    for (auto& r : c->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto p : c->tri_order) if (p) (void)hipFree(p);
    if (c->mvn_tiles) (void)hipFree(c->mvn_tiles);
    if (c->mvn_dense) (void)hipFree(c->mvn_dense);
    if (c->dX) (void)hipFree(c->dX);
    if (c->dT) (void)hipFree(c->dT);
    if (c->dY) (void)hipFree(c->dY);
    delete c;
    return GPSLC_OK;
}

static int set_data_impl(gpslc_ctx* c, const double* X, const double* T, const double* Y, hipMemcpyKind kind) {
    if (!c) return -1;
    if (c->nX > 0 && !X) return bad_arg(c, 2, "X is NULL but nX > 0");
    if (!T) return bad_arg(c, 3, "T is NULL");
    if (!Y) return bad_arg(c, 4, "Y is NULL");
    return guarded(c, [&]() {
        if (c->nX > 0) HC(hipMemcpy(c->dX, X, sizeof(double) * c->n * c->nX, kind));
        HC(hipMemcpy(c->dT, T, sizeof(double) * c->n, kind));
        HC(hipMemcpy(c->dY, Y, sizeof(double) * c->n, kind));
        c->hX.resize((size_t)c->n * c->nX);
        c->hT.resize(c->n);
        c->hY.resize(c->n);
        if (c->nX > 0) HC(hipMemcpy(c->hX.data(), c->dX, sizeof(double) * c->n * c->nX, hipMemcpyDeviceToHost));
        HC(hipMemcpy(c->hT.data(), c->dT, sizeof(double) * c->n, hipMemcpyDeviceToHost));
        HC(hipMemcpy(c->hY.data(), c->dY, sizeof(double) * c->n, hipMemcpyDeviceToHost));
        c->binary_t = true;
        for (double t : c->hT) if (t != 0.0 && t != 1.0) { c->binary_t = false; break; }
        c->has_data = true;
        c->ens_off = 0; c->ens_S = 0;      // a new data set starts outside any ensemble placement
        return GPSLC_OK;
    });
}
int gpslc_set_data(gpslc_ctx* c, const double* X, const double* T, const double* Y) {
    return set_data_impl(c, X, T, Y, hipMemcpyHostToDevice);
}
int gpslc_set_data_dev(gpslc_ctx* c, const double* X, const double* T, const double* Y) {
    return set_data_impl(c, X, T, Y, hipMemcpyDeviceToDevice);
}

int gpslc_set_tuning(gpslc_ctx* c, int32_t max_batch, int32_t panel_tiles, int32_t n_streams) {
    if (!c) return -1;
    if (max_batch < 0) return -2;
    if (panel_tiles < 0) return -3;
    if (n_streams < 0 || n_streams > 8) return -4;
    if (max_batch > 0) c->max_batch = max_batch;
    if (panel_tiles > 0) { c->panel = panel_tiles; c->panel_set = true; }
    if (n_streams > 0) c->nstreams = n_streams;
    return GPSLC_OK;
}

int gpslc_set_task_schedule(gpslc_ctx* c, int32_t min_tiles, int32_t max_tiles, int32_t min_matrices, int32_t group) {
    if (!c) return -1;
    if (min_tiles > TASK_MAX_NT) return -2;
    if (max_tiles > TASK_MAX_NT) return -3;
    if (group > 4096) return -5;
    if (min_tiles > 0) c->task_min_nt = min_tiles;
    if (max_tiles >= 0) c->task_max_nt = max_tiles;
    if (min_matrices > 0) c->task_min_batch = min_matrices;
    if (group > 0) c->task_group = group;
    return GPSLC_OK;
}

const char* gpslc_last_error(const gpslc_ctx* c) { return c ? c->err.c_str() : "null context"; }

int gpslc_set_ensemble(gpslc_ctx* c, int64_t sample_offset, int64_t S_total) {
    if (!c) return -1;
    if (sample_offset < 0) return -2;
    if (S_total < 0 || (S_total > 0 && sample_offset >= S_total)) return -3;
    c->ens_off = S_total > 0 ? sample_offset : 0;
    c->ens_S = S_total;
    return 0;
}

static int rbf_log_check(gpslc_ctx* c, const double* X1, const double* X2, int64_t n, int32_t d, const double* ls,
                         int32_t ls_len, double* out) {
    if (!c) return -1;
    if (!X1) return bad_arg(c, 2, "X1 is NULL");
    if (!X2) return bad_arg(c, 3, "X2 is NULL");
    if (n < 1) return bad_arg(c, 4, "n < 1");
    if (d < 1) return bad_arg(c, 5, "d < 1");
    if (!ls) return bad_arg(c, 6, "ls is NULL");
    if (ls_len != 1 && ls_len != d) return bad_arg(c, 7, "vector lengthscale doesn't match individual");
    if (!out) return bad_arg(c, 8, "out is NULL");
    return 0;
}

int gpslc_rbf_log_dev(gpslc_ctx* c, const double* X1, const double* X2, int64_t n, int32_t d, const double* ls,
                      int32_t ls_len, double* out) {
    int rc = rbf_log_check(c, X1, X2, n, d, ls, ls_len, out);
    if (rc) return rc;
    return guarded(c, [&]() {
        ensure_streams(c);
        launch_rbf_log(X1, X2, n, d, ls, ls_len, out, c->streams[0]);
        HC(hipGetLastError());
        HC(hipStreamSynchronize(c->streams[0]));
        return GPSLC_OK;
    });
}

int gpslc_rbf_log(gpslc_ctx* c, const double* X1, const double* X2, int64_t n, int32_t d, const double* ls,
                  int32_t ls_len, double* out) {
    int rc = rbf_log_check(c, X1, X2, n, d, ls, ls_len, out);
    if (rc) return rc;
    return guarded(c, [&]() {
        ensure_streams(c);
        c->io.reset();
        double* dA = up(c, X1, (size_t)n * d);
        double* dB = up(c, X2, (size_t)n * d);
        double* dl = up(c, ls, (size_t)ls_len);
        double* o = c->io.take<double>((size_t)n * n);
        launch_rbf_log(dA, dB, n, d, dl, ls_len, o, c->streams[0]);
        HC(hipGetLastError());
        HC(hipStreamSynchronize(c->streams[0]));
        copy_out_large(c, out, o, sizeof(double) * (size_t)n * n);
        return GPSLC_OK;
    });
}

int gpslc_process_cov_dev(gpslc_ctx* c, const double* logcov, int64_t n, double scale, double noise, double* out) {
    if (!c) return -1;
    if (!logcov) return bad_arg(c, 2, "logcov is NULL");
    if (n < 1) return bad_arg(c, 3, "n < 1");
    if (!out) return bad_arg(c, 6, "out is NULL");
    return guarded(c, [&]() {
        ensure_streams(c);
        launch_process_cov(logcov, n, scale, noise, out, c->streams[0]);
        HC(hipGetLastError());
        HC(hipStreamSynchronize(c->streams[0]));
        return GPSLC_OK;
    });
}

int gpslc_process_cov(gpslc_ctx* c, const double* logcov, int64_t n, double scale, double noise, double* out) {
    if (!c) return -1;
    if (!logcov) return bad_arg(c, 2, "logcov is NULL");
    if (n < 1) return bad_arg(c, 3, "n < 1");
    if (!out) return bad_arg(c, 6, "out is NULL");
    return guarded(c, [&]() {
        ensure_streams(c);
        c->io.reset();
        double* dA = up(c, logcov, (size_t)n * n);
        double* o = c->io.take<double>((size_t)n * n);
        launch_process_cov(dA, n, scale, noise, o, c->streams[0]);
        HC(hipGetLastError());
        HC(hipStreamSynchronize(c->streams[0]));
        copy_out_large(c, out, o, sizeof(double) * (size_t)n * n);
        return GPSLC_OK;
    });
}

// argument checks of gpslc_predict / gpslc_predict_multi; argoff shifts the reported argument number (code AND message) to the
// caller's own signature (gpslc_predict_multi has two more leading arguments than gpslc_predict: nctx, ctxs)
static int predict_check(gpslc_ctx* c, int64_t S, const double* U, const double* uyLS, const double* xyLS,
                         const double* tyLS, const double* yScale, const double* yNoise, int32_t L,
                         const double* doT, int32_t spp, const double* ite_draws, int argoff = 0) {
    if (!c) return -1;
    if (!c->has_data) { set_err(c, "gpslc_set_data has not been called"); return GPSLC_ERR_NODATA; }
    if (S < 0) return bad_arg(c, 2 + argoff, "S < 0");
    if (c->nU > 0 && S > 0 && (!U || !uyLS)) return bad_arg(c, 3 + argoff, "U / uyLS must not be NULL when nU > 0");
    if (c->nX > 0 && S > 0 && !xyLS) return bad_arg(c, 5 + argoff, "xyLS must not be NULL when nX > 0");
    if (S > 0 && (!tyLS || !yScale || !yNoise)) return bad_arg(c, 6 + argoff, "tyLS / yScale / yNoise must not be NULL");
    if (L < 1) return bad_arg(c, 9 + argoff, "L < 1");
    if (!doT) return bad_arg(c, 10 + argoff, "doT is NULL");
    if (ite_draws && spp < 1) return bad_arg(c, 12 + argoff, "spp < 1 with ite_draws requested");
    // a placement left over from an earlier, smaller call would make the stream ids (off + s) + S_total * l of this call's last
    // samples collide with the next level's streams: refuse instead of drawing correlated normals
    if (c->ens_S > 0 && c->ens_off + S > c->ens_S)
        return bad_arg(c, 2 + argoff, "S exceeds the room gpslc_set_ensemble left: sample_offset + S > S_total");
    return 0;
}

// device pointers everywhere; the info words come from the staging pool (the caller has reset it)
static int predict_dev_inner(gpslc_ctx* c, int64_t S, const double* U, const double* uyLS, const double* xyLS,
                             const double* tyLS, const double* yScale, const double* yNoise, int32_t L,
                             const double* doT, double pred_noise, int32_t spp, uint64_t seed, const double* z,
                             double* meanSATE, double* varSATE, double* meanITE, double* ite_draws,
                             int64_t ens_off = 0, int64_t ens_S = 0) {
    PredictIO io;
    io.S = S; io.p = SampleParams{U, uyLS, xyLS, tyLS, yScale, yNoise}; io.X = c->dX;
    io.L = L; io.doT = doT; io.pred_noise = pred_noise; io.spp = spp; io.seed = seed; io.z = z;
    io.meanSATE = meanSATE; io.varSATE = varSATE; io.meanITE = meanITE; io.ite_draws = ite_draws;
    io.ens_off = ens_off; io.ens_S = ens_S;
    io.info = c->io.take<int>((size_t)S);
    run_predict(c, io);
    return first_info(c);
}

int gpslc_predict_dev(gpslc_ctx* c, int64_t S, const double* U, const double* uyLS, const double* xyLS,
                      const double* tyLS, const double* yScale, const double* yNoise, int32_t L,
                      const double* doT, double pred_noise, int32_t spp, uint64_t seed, const double* z,
                      double* meanSATE, double* varSATE, double* meanITE, double* ite_draws) {
    int rc = predict_check(c, S, U, uyLS, xyLS, tyLS, yScale, yNoise, L, doT, spp, ite_draws);
    if (rc) return rc;
    if (S == 0) { c->last_info.clear(); return GPSLC_OK; }
    return guarded(c, [&]() {
        c->io.reset();
        return predict_dev_inner(c, S, U, uyLS, xyLS, tyLS, yScale, yNoise, L, doT, pred_noise, spp, seed, z,
                                 meanSATE, varSATE, meanITE, ite_draws);
    });
}

// Where a host-pointer prediction over the samples [s0, s0 + S) of S_total sits in the caller's arrays (gpslc_predict_multi:
// one shard; gpslc_predict: s0 = 0, S_total = S).  Per-sample inputs and the outputs are addressed in the caller's FULL
// arrays: element (s, l) of an S_total x L array at (s0 + s) + S_total l, block (s, l) of an n x S_total x L array at
// n ((s0 + s) + S_total l); the level-fastest draw tensor keeps a shard's columns contiguous.
struct HostPlacement {
    int64_t s0 = 0, S_total = 0;
    int64_t ens_off = 0, ens_S = 0;     // Philox stream placement of the shard (ens_S > 0), else the ctx's
};

// gpslc_predict's body: uploads, run_predict, delivery — every level's run of a shard goes from the device straight to its place
// in the caller's array (no host staging)
static int predict_host(gpslc_ctx* c, int64_t S, const double* U, const double* uyLS, const double* xyLS, const double* tyLS,
                        const double* yScale, const double* yNoise, int32_t L, const double* doT, double pred_noise, int32_t spp,
                        uint64_t seed, const double* z, double* meanSATE, double* varSATE, double* meanITE, double* ite_draws,
                        const HostPlacement& pl) {
    return guarded(c, [&]() {
        const size_t n = (size_t)c->n;
        const size_t s0 = (size_t)pl.s0, St = (size_t)pl.S_total;
        c->io.reset();
        const double* dU = c->nU ? up(c, U + n * c->nU * s0, n * c->nU * S) : nullptr;
        const double* duy = c->nU ? up(c, uyLS + (size_t)c->nU * s0, (size_t)c->nU * S) : nullptr;
        const double* dxy = c->nX ? up(c, xyLS + (size_t)c->nX * s0, (size_t)c->nX * S) : nullptr;
        const double* dty = up(c, tyLS + s0, S);
        const double* dys = up(c, yScale + s0, S);
        const double* dyn = up(c, yNoise + s0, S);
        const double* ddo = up(c, doT, L);
        const double* dz = nullptr;
        if (z && ite_draws) {      // caller's normals n x spp x S_total x L: level l of the shard is one run of n spp S doubles
            double* d = c->io.take<double>(n * spp * S * L);
            const size_t run = n * spp * (size_t)S * sizeof(double);
            HC(hipMemcpy2D(d, run, z + n * spp * s0, n * spp * St * sizeof(double), run, (size_t)L, hipMemcpyHostToDevice));
            dz = d;
        }
        double* oms = meanSATE ? c->io.take<double>((size_t)S * L) : nullptr;
        double* ovs = varSATE ? c->io.take<double>((size_t)S * L) : nullptr;
        double* omi = meanITE ? c->io.take<double>(n * S * L) : nullptr;
        double* odr = ite_draws ? c->io.take<double>((size_t)L * n * S * spp) : nullptr;
        int st = predict_dev_inner(c, S, dU, duy, dxy, dty, dys, dyn, L, ddo, pred_noise, spp, seed, dz,
                                   oms, ovs, omi, odr, pl.ens_off, pl.ens_S);
        if (st < 0) return st;
        const size_t sb = (size_t)S * sizeof(double);
        if (meanSATE) HC(hipMemcpy2D(meanSATE + s0, St * sizeof(double), oms, sb, sb, (size_t)L, hipMemcpyDeviceToHost));
        if (varSATE) HC(hipMemcpy2D(varSATE + s0, St * sizeof(double), ovs, sb, sb, (size_t)L, hipMemcpyDeviceToHost));
        if (meanITE) copy_out_rows(c, meanITE + n * s0, n * St * sizeof(double), omi, n * sb, (size_t)L);
        // level-fastest tensor L x n x (S_total spp): a sample's columns are one contiguous run
        if (ite_draws) copy_out_large(c, ite_draws + (size_t)L * n * spp * s0, odr, sizeof(double) * (size_t)L * n * S * spp);
        return st;
    });
}

int gpslc_predict(gpslc_ctx* c, int64_t S, const double* U, const double* uyLS, const double* xyLS,
                  const double* tyLS, const double* yScale, const double* yNoise, int32_t L, const double* doT,
                  double pred_noise, int32_t spp, uint64_t seed, const double* z, double* meanSATE,
                  double* varSATE, double* meanITE, double* ite_draws) {
    int rc = predict_check(c, S, U, uyLS, xyLS, tyLS, yScale, yNoise, L, doT, spp, ite_draws);
    if (rc) return rc;
    if (S == 0) { c->last_info.clear(); return GPSLC_OK; }
    HostPlacement pl;
    pl.S_total = S;
    return predict_host(c, S, U, uyLS, xyLS, tyLS, yScale, yNoise, L, doT, pred_noise, spp, seed, z, meanSATE, varSATE, meanITE,
                        ite_draws, pl);
}

int gpslc_shard_range(int64_t S, int32_t nblocks, int32_t k, int64_t* s0, int64_t* s1) {
    if (S < 0) return -1;
    if (nblocks < 1) return -2;
    if (k < 0 || k >= nblocks) return -3;
    if (!s0) return -4;
    if (!s1) return -5;
    const int64_t q = S / nblocks, r = S % nblocks;
    *s0 = k * q + std::min<int64_t>(k, r);
    *s1 = *s0 + q + (k < r ? 1 : 0);
    return GPSLC_OK;
}

// The sharded ensemble behind the ABI (the loop src/prediction.jl:30-33 over src/estimation.jl:78-84): contiguous
// blocks of the posterior-sample index over the contexts, one host thread per context, data replicated (every ctx holds its own
// copy: gpslc_set_data), no traffic between the devices while they compute.  The "gather" is each device's own device-to-host
// copy into ITS block of the caller's arrays — nctx PCIe links in parallel, nothing funnels through one GPU — and since round 6
// every level's run of a shard goes from the device (through the ctx's pinned bounce chunks) straight to its place in the
// caller's array: no per-shard host staging, no second pass over the results.  A shard's ensemble placement travels as a
// parameter: the contexts' own gpslc_set_ensemble state is read (ctxs[0]'s places the whole call), never written.
int gpslc_predict_multi(int32_t nctx, gpslc_ctx* const* ctxs, int64_t S, const double* U, const double* uyLS,
                        const double* xyLS, const double* tyLS, const double* yScale, const double* yNoise, int32_t L,
                        const double* doT, double pred_noise, int32_t spp, uint64_t seed, const double* z,
                        double* meanSATE, double* varSATE, double* meanITE, double* ite_draws, int32_t* info_or_null) {
    if (nctx < 1) return -1;
    if (!ctxs) return -2;
    for (int k = 0; k < nctx; ++k) {
        if (!ctxs[k]) return -2;
        for (int j = 0; j < k; ++j)
            if (ctxs[j] == ctxs[k]) return bad_arg(ctxs[0], 2, "the same ctx is listed twice (one ctx per shard: calls on a ctx are serialised)");
    }
    gpslc_ctx* c0 = ctxs[0];
    for (int k = 1; k < nctx; ++k) {
        if (ctxs[k]->n != c0->n || ctxs[k]->nX != c0->nX || ctxs[k]->nU != c0->nU)
            return bad_arg(c0, 2, "the contexts differ in n / nX / nU");
        // one precision mode per call: a ctx created with GPSLC_FLAG_FP32_KERNEL would silently mix arithmetics across the shards
        if ((ctxs[k]->flags & GPSLC_FLAG_FP32_KERNEL) != (c0->flags & GPSLC_FLAG_FP32_KERNEL))
            return bad_arg(c0, 2, "the contexts differ in GPSLC_FLAG_FP32_KERNEL");
        if (!ctxs[k]->has_data) { set_err(c0, "gpslc_set_data has not been called on every ctx"); return GPSLC_ERR_NODATA; }
    }
    // argument numbers of THIS signature, in the code and in the message (S is #3, ...)
    int rc0 = predict_check(c0, S, U, uyLS, xyLS, tyLS, yScale, yNoise, L, doT, spp, ite_draws, /*argoff=*/1);
    if (rc0) return rc0;
    if (S == 0) { for (int k = 0; k < nctx; ++k) ctxs[k]->last_info.clear(); return GPSLC_OK; }
    try {
        // placement of the whole call (ctxs[0]'s, if the caller set one: this call may itself be one node's share)
        const int64_t base_off = c0->ens_S > 0 ? c0->ens_off : 0, total = c0->ens_S > 0 ? c0->ens_S : S;
        struct Shard { int64_t s0 = 0, Sr = 0; int rc = 0; };
        std::vector<Shard> sh((size_t)nctx);
        for (int k = 0; k < nctx; ++k) {
            int64_t s1 = 0;
            (void)gpslc_shard_range(S, nctx, k, &sh[(size_t)k].s0, &s1);
            sh[(size_t)k].Sr = s1 - sh[(size_t)k].s0;
        }
        auto work = [&](int k) {
            Shard& h = sh[(size_t)k];
            if (h.Sr == 0) { ctxs[k]->last_info.clear(); return; }
            HostPlacement pl;
            pl.s0 = h.s0; pl.S_total = S; pl.ens_off = base_off + h.s0; pl.ens_S = total;
            h.rc = predict_host(ctxs[k], h.Sr, U, uyLS, xyLS, tyLS, yScale, yNoise, L, doT, pred_noise, spp, seed, z, meanSATE,
                                varSATE, meanITE, ite_draws, pl);
        };
        std::vector<std::thread> pool;
        for (int k = 1; k < nctx; ++k) {
            try { pool.emplace_back(work, k); }
            catch (...) { work(k); }                 // no thread to be had: this shard runs here, in turn
        }
        work(0);
        for (auto& t : pool) t.join();
        int rc = 0;
        for (int k = 0; k < nctx && rc == 0; ++k)
            if (sh[(size_t)k].rc < 0) {
                rc = sh[(size_t)k].rc;
                if (k > 0) set_err(c0, "shard " + std::to_string(k) + " (device " + std::to_string(ctxs[k]->device) + "): " + ctxs[k]->err);
            }
        for (int k = 0; k < nctx && rc == 0; ++k) rc = sh[(size_t)k].rc;        // blocks are in sample order: the first failing pivot
        if (info_or_null)
            for (int k = 0; k < nctx; ++k) {
                const Shard& h = sh[(size_t)k];
                if (h.Sr == 0) continue;
                if ((int64_t)ctxs[k]->last_info.size() == h.Sr) memcpy(info_or_null + h.s0, ctxs[k]->last_info.data(), sizeof(int32_t) * h.Sr);
                else for (int64_t j = 0; j < h.Sr; ++j) info_or_null[h.s0 + j] = 0;
            }
        return rc;
    } catch (const std::bad_alloc&) {
        set_err(c0, "host allocation failed");
        return GPSLC_ERR_NOMEM;
    } catch (...) {
        set_err(c0, "internal error");
        return GPSLC_ERR_INTERNAL;
    }
}

int gpslc_ite_distributions(gpslc_ctx* c, int64_t S, const double* U, const double* uyLS, const double* xyLS,
                            const double* tyLS, const double* yScale, const double* yNoise, double doT,
                            double pred_noise, double* MeanITEs, double* CovITEs) {
    int rc = check_common(c, S, U, uyLS, xyLS, tyLS, yScale, yNoise);
    if (rc) return rc;
    if (S == 0) { c->last_info.clear(); return GPSLC_OK; }
    return guarded(c, [&]() {
        const size_t n = (size_t)c->n;
        c->io.reset();
        const double* dU = c->nU ? up(c, U, n * c->nU * S) : nullptr;
        const double* duy = c->nU ? up(c, uyLS, (size_t)c->nU * S) : nullptr;
        const double* dxy = c->nX ? up(c, xyLS, (size_t)c->nX * S) : nullptr;
        const double* dty = up(c, tyLS, S);
        const double* dys = up(c, yScale, S);
        const double* dyn = up(c, yNoise, S);
        const double* ddo = up(c, &doT, 1);
        double* om = MeanITEs ? c->io.take<double>((size_t)S * n) : nullptr;
        double* oc = CovITEs ? c->io.take<double>((size_t)S * n * n) : nullptr;
        PredictIO io;
        io.S = S; io.p = SampleParams{dU, duy, dxy, dty, dys, dyn}; io.X = c->dX;
        io.L = 1; io.doT = ddo; io.pred_noise = pred_noise;
        io.MeanITEs = om; io.CovITEs = oc; io.info = c->io.take<int>((size_t)S);
        run_predict(c, io);
        if (MeanITEs) HC(hipMemcpy(MeanITEs, om, sizeof(double) * S * n, hipMemcpyDeviceToHost));
        if (CovITEs) copy_out_large(c, CovITEs, oc, sizeof(double) * (size_t)S * n * n);
        return first_info(c);
    });
}

// logpdf[s] = -(n log 2pi + logdet_s + quad_s) / 2 from the device-side epilogue values
static void finish_logpdf(gpslc_ctx* c, int64_t S, const double* d_logdet, const double* d_quad, double* logpdf) {
    std::vector<double> ld(S), q(S);
    HC(hipMemcpy(ld.data(), d_logdet, sizeof(double) * S, hipMemcpyDeviceToHost));
    HC(hipMemcpy(q.data(), d_quad, sizeof(double) * S, hipMemcpyDeviceToHost));
    const double l2pi = 1.8378770664093454835606594728112;
    for (int64_t s = 0; s < S; ++s) logpdf[s] = -0.5 * ((double)c->n * l2pi + ld[s] + q[s]);
}

int gpslc_y_logpdf(gpslc_ctx* c, int64_t S, const double* U, const double* X_or_null, const double* Y_or_null,
                   const double* uyLS, const double* xyLS, const double* tyLS, const double* yScale,
                   const double* yNoise, double* logpdf) {
    int rc = check_common(c, S, U, uyLS, xyLS, tyLS, yScale, yNoise);
    if (rc) return rc;
    if (!logpdf) return bad_arg(c, 11, "logpdf is NULL");
    if (S == 0) { c->last_info.clear(); return GPSLC_OK; }
    if (fast_path_ok(c, c->nU + c->nX + 1, S)) {
        // small n: every parameter set is one workgroup of ONE launch (k_small.hip); T enters as a feature column
        return guarded(c, [&]() {
            const size_t n = (size_t)c->n;
            std::vector<HostNode> hn((size_t)S);
            for (int64_t s = 0; s < S; ++s) {
                HostNode& h = hn[s];
                h = HostNode{};
                h.nF = c->nU + c->nX + 1;
                h.F[0] = c->nU ? U + s * n * c->nU : nullptr; h.nFpart[0] = c->nU; h.ls[0] = c->nU ? uyLS + s * c->nU : nullptr;
                h.F[1] = c->nX ? (X_or_null ? X_or_null : c->hX.data()) : nullptr; h.nFpart[1] = c->nX;
                h.ls[1] = c->nX ? xyLS + s * c->nX : nullptr;
                h.col = c->hT.data(); h.ls_col = tyLS[s];
                h.scale = yScale[s]; h.noise = yNoise[s];
                h.target = Y_or_null ? Y_or_null : c->hY.data();
            }
            return small_nodes_logpdf(c, (int)S, hn.data(), logpdf);
        });
    }
    return guarded(c, [&]() {
        const size_t n = (size_t)c->n;
        c->io.reset();
        const double* dU = c->nU ? up(c, U, n * c->nU * S) : nullptr;
        const double* dX = (X_or_null && c->nX) ? up(c, X_or_null, n * c->nX) : c->dX;
        const double* dYo = Y_or_null ? up(c, Y_or_null, n) : nullptr;
        const double* duy = c->nU ? up(c, uyLS, (size_t)c->nU * S) : nullptr;
        const double* dxy = c->nX ? up(c, xyLS, (size_t)c->nX * S) : nullptr;
        const double* dty = up(c, tyLS, S);
        const double* dys = up(c, yScale, S);
        const double* dyn = up(c, yNoise, S);
        const double zero = 0.0;
        const double* ddo = up(c, &zero, 1);
        double* old = c->io.take<double>((size_t)S);
        double* oq = c->io.take<double>((size_t)S);
        PredictIO io;
        io.S = S; io.p = SampleParams{dU, duy, dxy, dty, dys, dyn}; io.X = dX;
        if (dYo) { io.Y = dYo; io.y_sstride = 0; }    // the value being scored (Gen passes it to logpdf), else the ctx's Y
        io.L = 0; io.doT = ddo; io.logdet = old; io.quad = oq; io.info = c->io.take<int>((size_t)S);
        run_predict(c, io);
        finish_logpdf(c, S, old, oq, logpdf);
        return first_info(c);
    });
}

// general (tiled, multi-launch) path of gpslc_gp_logpdf; arguments already validated
// draws_or_null (host, n x S; needs t_shared == 0): also chol(K_s) target_s — the targets are then standard normals
static int gp_logpdf_general(gpslc_ctx* c, int64_t S, int32_t nF, const double* F, int32_t f_shared, const double* ls,
                             const double* scale, const double* noise, const double* target, int32_t t_shared,
                             double* logpdf, double* draws_or_null = nullptr) {
    const size_t n = (size_t)c->n;
    c->io.reset();
    const double* dF = nF ? up(c, F, n * nF * (f_shared ? 1 : S)) : nullptr;
    const double* dls = nF ? up(c, ls, (size_t)nF * S) : nullptr;
    const double* dsc = up(c, scale, S);
    const double* dno = up(c, noise, S);
    const double* dtg = up(c, target, n * (t_shared ? 1 : S));
    std::vector<double> inf(S, INFINITY);        // tyLS = inf switches the treatment term off: e_ij = exp(-0) = 1
    const double* dty = up(c, inf.data(), S);
    const double zero = 0.0;
    const double* ddo = up(c, &zero, 1);
    double* old = c->io.take<double>((size_t)S);
    double* oq = c->io.take<double>((size_t)S);
    PredictIO io;
    io.S = S;
    io.p = SampleParams{dF, dls, nullptr, dty, dsc, dno, f_shared ? 0 : (long long)n * nF};
    io.p_shared_u = f_shared != 0;
    io.X = nullptr; io.nU = nF; io.nX = 0;
    io.Y = dtg; io.y_sstride = t_shared ? 0 : (long long)n;
    io.L = 0; io.doT = ddo; io.logdet = old; io.quad = oq; io.info = c->io.take<int>((size_t)S);
    double* ddraw = nullptr;
    if (draws_or_null) {
        ensure_streams(c);
        ddraw = c->io.take<double>(n * (size_t)S);
        double* zero_mean = c->io.take<double>(n * (size_t)S);
        HC(hipMemsetAsync(zero_mean, 0, n * (size_t)S * sizeof(double), c->streams[0]));
        HC(hipStreamSynchronize(c->streams[0]));
        io.nz = dtg; io.nzero = zero_mean; io.ndraw = ddraw;
    }
    run_predict(c, io);
    if (logpdf) finish_logpdf(c, S, old, oq, logpdf);
    if (draws_or_null) HC(hipMemcpy(draws_or_null, ddraw, n * (size_t)S * sizeof(double), hipMemcpyDeviceToHost));
    return first_info(c);
}

int gpslc_gp_logpdf(gpslc_ctx* c, int64_t S, int32_t nF, const double* F, int32_t f_shared, const double* ls,
                    const double* scale, const double* noise, const double* target, int32_t t_shared,
                    double* logpdf) {
    if (!c) return -1;
    if (S < 0) return bad_arg(c, 2, "S < 0");
    if (nF < 0 || nF > 32) return bad_arg(c, 3, "nF must be in 0..32");
    if (nF > 0 && (!F || !ls)) return bad_arg(c, 4, "F / ls must not be NULL when nF > 0");
    if (S > 0 && (!scale || !noise)) return bad_arg(c, 7, "scale / noise must not be NULL");
    if (S > 0 && !target) return bad_arg(c, 9, "target is NULL");
    if (!logpdf) return bad_arg(c, 11, "logpdf is NULL");
    if (S == 0) { c->last_info.clear(); return GPSLC_OK; }
    return guarded(c, [&]() {
        const size_t n = (size_t)c->n;
        if (fast_path_ok(c, nF, S)) {
            std::vector<HostNode> hn((size_t)S);
            for (int64_t s = 0; s < S; ++s) {
                HostNode& h = hn[s];
                h = HostNode{};
                h.nF = nF;
                h.F[0] = nF ? F + (f_shared ? 0 : s * n * nF) : nullptr; h.nFpart[0] = nF; h.ls[0] = nF ? ls + s * nF : nullptr;
                h.scale = scale[s]; h.noise = noise[s];
                h.target = target + (t_shared ? 0 : s * n);
            }
            return small_nodes_logpdf(c, (int)S, hn.data(), logpdf);
        }
        return gp_logpdf_general(c, S, nF, F, f_shared, ls, scale, noise, target, t_shared, logpdf);
    });
}

// ONE batched pass of the general tiled path over all nodes of a call (S = count parameter sets with per-set feature blocks
// and per-set targets); draws_or_null: also chol(K_i) target_i per node (gpslc_nodes_draw)
static int nodes_general(gpslc_ctx* c, int32_t count, const gpslc_node* nodes, int nF_max, double* logpdf, double* draws_or_null) {
    // Nodes with fewer than nF_max feature columns are padded with zero
    // columns of lengthscale 1: a padding column adds (0 * 1 - 0 * 1)^2 = +0.0 to every squared distance, so a
    // node's Gram matrix — and its score — is bit-identical to the one its own feature count would give.
    // Staging lives in the ctx (this call sits in the MCMC inner loop: no allocation once the buffers have grown),
    // and nodes that all share ONE feature block — the nX `:X => k => :X` nodes, F = U for every k — hand it over once
    // (f_shared) instead of count padded copies.
    const size_t n = (size_t)c->n;
    bool same_f = true;
    for (int i = 1; i < count; ++i) same_f = same_f && nodes[i].F == nodes[0].F && nodes[i].nF == nodes[0].nF;
    std::vector<double>& Fp = c->stage_f;
    std::vector<double>& rest = c->stage_rest;       // [ls | scale | noise | targets]
    const size_t ls_n = (size_t)std::max(nF_max, 1) * count;
    rest.resize(ls_n + 2 * (size_t)count + n * (size_t)count);
    double* lsp = rest.data();
    double* sc = lsp + ls_n;
    double* no = sc + count;
    double* tg = no + count;
    std::fill(lsp, lsp + ls_n, 1.0);
    if (!same_f) {
        Fp.resize(n * (size_t)nF_max * count);
        std::fill(Fp.begin(), Fp.end(), 0.0);
    }
    for (int i = 0; i < count; ++i) {
        const gpslc_node& q = nodes[i];
        if (q.nF > 0) {
            if (!same_f) memcpy(Fp.data() + (size_t)i * n * nF_max, q.F, n * (size_t)q.nF * sizeof(double));
            memcpy(lsp + (size_t)i * nF_max, q.ls, (size_t)q.nF * sizeof(double));
        }
        sc[i] = q.scale; no[i] = q.noise;
        memcpy(tg + (size_t)i * n, q.target, n * sizeof(double));
    }
    const double* Fsrc = nF_max == 0 ? nullptr : (same_f ? nodes[0].F : Fp.data());
    return gp_logpdf_general(c, count, nF_max, Fsrc, same_f ? 1 : 0, nF_max ? lsp : nullptr, sc, no, tg, 0, logpdf, draws_or_null);
}

int gpslc_nodes_logpdf(gpslc_ctx* c, int32_t count, const gpslc_node* nodes, double* logpdf) {
    if (!c) return -1;
    if (count < 0) return bad_arg(c, 2, "count < 0");
    if (count > 0 && !nodes) return bad_arg(c, 3, "nodes is NULL");
    if (count > 0 && !logpdf) return bad_arg(c, 4, "logpdf is NULL");
    int nF_max = 0;
    for (int i = 0; i < count; ++i) {
        const gpslc_node& q = nodes[i];
        if (q.nF < 0 || q.nF > 32 || (q.nF > 0 && (!q.F || !q.ls)) || !q.target)
            return bad_arg(c, 3, "node with nF outside 0..32 or a NULL feature / lengthscale / target pointer");
        nF_max = std::max(nF_max, (int)q.nF);
    }
    if (count == 0) { c->last_info.clear(); return GPSLC_OK; }
    return guarded(c, [&]() {
        if (fast_path_ok(c, nF_max, count)) {
            std::vector<HostNode> hn((size_t)count);
            for (int i = 0; i < count; ++i) {
                HostNode& h = hn[i];
                h = HostNode{};
                h.nF = nodes[i].nF;
                h.F[0] = nodes[i].F; h.nFpart[0] = nodes[i].nF; h.ls[0] = nodes[i].ls;
                h.scale = nodes[i].scale; h.noise = nodes[i].noise; h.target = nodes[i].target;
            }
            return small_nodes_logpdf(c, count, hn.data(), logpdf);
        }
        return nodes_general(c, count, nodes, nF_max, logpdf, nullptr);
    });
}

int gpslc_nodes_draw(gpslc_ctx* c, int32_t count, const gpslc_node* nodes, double* draws, double* logpdf_or_null) {
    if (!c) return -1;
    if (count < 0) return bad_arg(c, 2, "count < 0");
    if (count > 0 && !nodes) return bad_arg(c, 3, "nodes is NULL");
    if (count > 0 && !draws) return bad_arg(c, 4, "draws is NULL");
    int nF_max = 0;
    for (int i = 0; i < count; ++i) {
        const gpslc_node& q = nodes[i];
        if (q.nF < 0 || q.nF > 32 || (q.nF > 0 && (!q.F || !q.ls)) || !q.target)
            return bad_arg(c, 3, "node with nF outside 0..32 or a NULL feature / lengthscale / target pointer");
        nF_max = std::max(nF_max, (int)q.nF);
    }
    if (count == 0) { c->last_info.clear(); return GPSLC_OK; }
    return guarded(c, [&]() {
        std::vector<double> lp((size_t)count);
        int st;
        if (fast_path_ok(c, nF_max, count)) {
            std::vector<HostNode> hn((size_t)count);
            for (int i = 0; i < count; ++i) {
                HostNode& h = hn[i];
                h = HostNode{};
                h.nF = nodes[i].nF;
                h.F[0] = nodes[i].F; h.nFpart[0] = nodes[i].nF; h.ls[0] = nodes[i].ls;
                h.scale = nodes[i].scale; h.noise = nodes[i].noise; h.target = nodes[i].target;
            }
            st = small_nodes_logpdf(c, count, hn.data(), lp.data(), draws);
        } else {
            // beyond the single-workgroup kernels (n > 640, many nodes, fp32 kernel mode): the batched tiled factorisation of
            // every node's covariance, then L z on the predictive-draw kernel (round 6)
            st = nodes_general(c, count, nodes, nF_max, lp.data(), draws);
        }
        if (logpdf_or_null) memcpy(logpdf_or_null, lp.data(), (size_t)count * sizeof(double));
        return st;
    });
}

// caches the dense covariance (small n: every evaluation refactorises it in LDS) or its tiled factor (substitution-based:
// SigmaU * uNoise is near-singular by construction, 1e-13 jitter, src/utils.jl:17-33) in the context; mvn_info = its LAPACK-style info
static void mvn_cache(gpslc_ctx* c, const double* cov, bool small) {
    const size_t n = (size_t)c->n;
    c->mvn_valid = false;
    if (small) {
        if (!c->mvn_dense) HC(hipMalloc((void**)&c->mvn_dense, n * n * sizeof(double)));
        HC(hipMemcpy(c->mvn_dense, cov, n * n * sizeof(double), hipMemcpyHostToDevice));
        // validate once (as the general path does when it caches the factor): info of cov itself
        std::vector<double> zeros(n, 0.0);
        HostNode probe{};
        probe.dev_cov = c->mvn_dense; probe.covscale = 1.0; probe.target = zeros.data();
        double dummy = 0.0;
        c->mvn_info = small_nodes_logpdf(c, 1, &probe, &dummy);
        c->mvn_valid = true;
        return;
    }
    ensure_streams(c);
    const int nt = c->nt;
    const long long nlow = (long long)nt * (nt + 1) / 2;
    hipStream_t st = c->streams[0];
    if (!c->mvn_tiles) HC(hipMalloc((void**)&c->mvn_tiles, (size_t)nlow * GP_TSQ * 8));
    DevBuf bcov, old, oq, info;
    const double* dcov = up(bcov, cov, n * n);
    old.alloc(8); oq.alloc(8); info.alloc(sizeof(int));
    HC(hipMemsetAsync(info.p, 0, sizeof(int), st));
    TRef M = lower_ref(c->mvn_tiles, nlow * GP_TSQ);
    launch_dense_load(DenseLoadArgs{dcov, (int)n, nt, M}, st);
    potrf_tiles(c, M, nt, nt, nullptr, 0, info.as<int>(), 0, 1, st, 0, 0, /*robust=*/true);
    launch_quad_rows(QuadRowsArgs{M, (int)n, nt, 0, 0, old.as<double>(), oq.as<double>()}, st);
    HC(hipStreamSynchronize(st));
    HC(hipGetLastError());
    HC(hipMemcpy(&c->mvn_logdet, old.p, 8, hipMemcpyDeviceToHost));
    HC(hipMemcpy(&c->mvn_info, info.p, sizeof(int), hipMemcpyDeviceToHost));
    c->mvn_valid = true;
}

int gpslc_mvn_logpdf(gpslc_ctx* c, int64_t S, const double* cov, const double* covscale, const double* x,
                     double* logpdf) {
    if (!c) return -1;
    if (S < 0) return bad_arg(c, 2, "S < 0");
    if (!cov && !c->mvn_valid) return bad_arg(c, 3, "cov is NULL and no factor is cached");
    if (S > 0 && !x) return bad_arg(c, 5, "x is NULL");
    if (S > 0 && !logpdf) return bad_arg(c, 6, "logpdf is NULL");
    if (fast_path_ok(c, 0, std::max<int64_t>(S, 1))) {
        // small n: keep the dense matrix on the device; every evaluation is one workgroup that scales, factorises
        // and solves in LDS (k_small.hip) — cheaper than the tiled forward solve against a cached factor
        return guarded(c, [&]() {
            const size_t n = (size_t)c->n;
            if (cov) mvn_cache(c, cov, true);
            else if (!c->mvn_dense) return bad_arg(c, 3, "cov is NULL and the cached factor belongs to the tiled path (S > 512 earlier)");
            c->last_info.assign((size_t)S, c->mvn_info);
            if (S == 0 || c->mvn_info != 0) {
                for (int64_t s = 0; s < S; ++s) logpdf[s] = NAN;
                return c->mvn_info;
            }
            std::vector<HostNode> hn((size_t)S);
            for (int64_t s = 0; s < S; ++s) {
                hn[s] = HostNode{};
                hn[s].dev_cov = c->mvn_dense;
                hn[s].covscale = covscale ? covscale[s] : 1.0;
                hn[s].target = x + s * n;
            }
            const int st = small_nodes_logpdf(c, (int)S, hn.data(), logpdf);
            if (st > 0) for (int64_t s = 0; s < S; ++s) if (c->last_info[s] != 0) logpdf[s] = NAN;
            return st;
        });
    }
    return guarded(c, [&]() {
        ensure_streams(c);
        const int n = (int)c->n, nt = c->nt;
        const long long nlow = (long long)nt * (nt + 1) / 2;
        hipStream_t st = c->streams[0];
        if (cov) mvn_cache(c, cov, false);   // factor once, keep L in the context
        else if (!c->mvn_tiles) return bad_arg(c, 3, "cov is NULL and no tiled factor is cached");
        c->last_info.assign((size_t)S, c->mvn_info);
        if (S == 0 || c->mvn_info != 0) {
            for (int64_t s = 0; s < S; ++s) logpdf[s] = NAN;
            return c->mvn_info;
        }
        // z_s = L^-1 x_s for all S vectors: rows of W = X^T L^-T, tile-level left-looking solve
        const int naug = (int)((S + GP_TS - 1) / GP_TS);
        c->io.reset();
        const double* dx = up(c, x, (size_t)n * S);
        double* wt = c->io.take<double>((size_t)naug * nt * GP_TSQ);
        double* oq = c->io.take<double>((size_t)S);
        TRef W = rect_ref(wt, (long long)naug * nt * GP_TSQ, nt);
        TRef Ls = lower_ref(c->mvn_tiles, nlow * GP_TSQ);
        launch_rows_rhs(RowsRhsArgs{dx, S, n, nt, naug, W, 0, 1}, st);
        const int short_rows = naug == 1 ? (int)S : 0;
        // right-looking: z_k = w_k L_kk^-T by substitution (no inverted block: the factor is near-singular), then every
        // remaining column block in parallel on the MFMA tile kernel
        for (int k = 0; k < nt; ++k) {
            launch_trsm_robust(W, Ls, k, 0, naug, 1, st);
            if (k + 1 < nt) {   // W(:, j) -= W(:, k) L(j, k)^T for j > k
                GemmArgs u{};
                u.A = W; u.B = Ls; u.C = W;
                u.shape = 1; u.i0 = 0; u.j0 = k + 1; u.mi = naug; u.mj = nt - k - 1;
                u.k0 = k; u.k1 = k + 1; u.accumulate = 1; u.nbatch = 1; u.ntiles = naug * (nt - k - 1);
                u.short_row0 = 0; u.short_rows = short_rows;
                gemm(c, u, st);
            }
        }
        launch_row_norms(RowNormArgs{W, nt, naug, S, oq}, st);
        HC(hipStreamSynchronize(st));
        HC(hipGetLastError());
        if (c->flags & GPSLC_FLAG_PROFILE) prof_collect(c);
        std::vector<double> q(S);
        HC(hipMemcpy(q.data(), oq, sizeof(double) * S, hipMemcpyDeviceToHost));
        const double l2pi = 1.8378770664093454835606594728112;
        for (int64_t s = 0; s < S; ++s) {
            const double sc = covscale ? covscale[s] : 1.0;
            logpdf[s] = -0.5 * ((double)n * l2pi + (double)n * std::log(sc) + c->mvn_logdet + q[s] / sc);
        }
        return GPSLC_OK;
    });
}

// draws[:, s] = chol(covscale_s cov) z[:, s]: Gen's mvnormal(zeros(n), uCov) with the host's normals — the auxiliary vector of
// elliptical_slice(trace, :U => k => :U, zeros(n), uCov) (src/inference.jl:48-54) and the prior draw generateUfromSigmaU
// (src/model_likelihood.jl:4-10) — from the covariance gpslc_mvn_logpdf caches (chol(s C) = sqrt(s) chol(C))
int gpslc_mvn_draw(gpslc_ctx* c, int64_t S, const double* cov, const double* covscale, const double* z, double* draws) {
    if (!c) return -1;
    if (S < 0) return bad_arg(c, 2, "S < 0");
    if (!cov && !c->mvn_valid) return bad_arg(c, 3, "cov is NULL and no factor is cached");
    if (S > 0 && !z) return bad_arg(c, 5, "z is NULL");
    if (S > 0 && !draws) return bad_arg(c, 6, "draws is NULL");
    const bool small = fast_path_ok(c, 0, std::max<int64_t>(S, 1));
    return guarded(c, [&]() {
        const size_t n = (size_t)c->n;
        if (cov) mvn_cache(c, cov, small);
        else if (small ? !c->mvn_dense : !c->mvn_tiles) return bad_arg(c, 3, "cov is NULL and the cached covariance belongs to the other path");
        c->last_info.assign((size_t)S, c->mvn_info);
        if (S == 0) return GPSLC_OK;
        if (c->mvn_info != 0) {
            for (size_t e = 0; e < n * (size_t)S; ++e) draws[e] = NAN;
            return c->mvn_info;
        }
        if (small) {
            std::vector<HostNode> hn((size_t)S);
            for (int64_t s = 0; s < S; ++s) {
                hn[s] = HostNode{};
                hn[s].dev_cov = c->mvn_dense;
                hn[s].covscale = covscale ? covscale[s] : 1.0;
                hn[s].target = z + s * n;
            }
            std::vector<double> lp((size_t)S);
            return small_nodes_logpdf(c, (int)S, hn.data(), lp.data(), draws);
        }
        // tiled: L z on the predictive-draw kernel (every unit streams the ONE cached factor: batch stride 0), scaled on the host
        ensure_streams(c);
        hipStream_t st = c->streams[0];
        const int nt = c->nt;
        const long long Np = (long long)nt * GP_TS;
        c->io.reset();
        const double* dz = up(c, z, n * (size_t)S);
        double* zero_mean = c->io.take<double>(n * (size_t)S);
        double* out = c->io.take<double>(n * (size_t)S);
        double* zt = c->io.take<double>((size_t)S * 16 * Np);
        HC(hipMemsetAsync(zero_mean, 0, n * (size_t)S * sizeof(double), st));
        DrawArgs dr{};
        dr.Lc = lower_ref(c->mvn_tiles, 0); dr.n = (int)n; dr.nt = nt; dr.s0 = 0; dr.S = S; dr.l = 0; dr.lc = 1; dr.L = 1; dr.spp = 1;
        dr.mean = zero_mean; dr.z = dz; dr.zt = zt; dr.out = out;
        dr.obase = 0; dr.osb = (long long)n; dr.osl = 0; dr.osi = 1; dr.osd = (long long)n;
        launch_draws(dr, (int)S, st);
        HC(hipStreamSynchronize(st));
        HC(hipGetLastError());
        HC(hipMemcpy(draws, out, n * (size_t)S * sizeof(double), hipMemcpyDeviceToHost));
        if (covscale)
            for (int64_t s = 0; s < S; ++s) {
                const double r = std::sqrt(covscale[s]);
                for (size_t i = 0; i < n; ++i) draws[s * n + i] *= r;
            }
        return GPSLC_OK;
    });
}

int gpslc_likelihood_distribution(gpslc_ctx* c, const double* U, const double* uyLS, const double* xyLS,
                                  double tyLS, double yScale, double yNoise, double doT, double* CovWW,
                                  double* CovWWs, double* CovWWp, double* CovC11, double* CovC12,
                                  double* CovC21, double* CovC22) {
    const double ty = tyLS, ysc = yScale, yno = yNoise;
    int rc = check_common(c, 1, U, uyLS, xyLS, &ty, &ysc, &yno);
    if (rc) return rc;
    return guarded(c, [&]() {
        ensure_streams(c);
        hipStream_t st = c->streams[0];
        const int n = (int)c->n, nt = c->nt;
        const long long nlow = (long long)nt * (nt + 1) / 2, nsq = (long long)nt * nt;
        DevBuf bU, buy, bxy, bty, bys, byn, info, tiles, inv, part, rect, outb;
        const double* dU = c->nU ? up(bU, U, (size_t)n * c->nU) : nullptr;
        const double* duy = c->nU ? up(buy, uyLS, c->nU) : nullptr;
        const double* dxy = c->nX ? up(bxy, xyLS, c->nX) : nullptr;
        const double* dty = up(bty, &ty, 1);
        const double* dys = up(bys, &ysc, 1);
        const double* dyn = up(byn, &yno, 1);
        SampleParams sp{dU, duy, dxy, dty, dys, dyn, (long long)n * c->nU};
        info.alloc(sizeof(int));
        HC(hipMemsetAsync(info.p, 0, sizeof(int), st));
        tiles.alloc((size_t)nlow * GP_TSQ * 8);
        inv.alloc((size_t)nt * GP_TSQ * 8);
        rect.alloc((size_t)6 * nsq * GP_TSQ * 8);        // K, Ks, Ks', Kss, W1, W2
        outb.alloc((size_t)n * n * 8);
        double* rb = rect.as<double>();
        const long long rs = nsq * GP_TSQ;
        TRef Kt = rect_ref(rb, rs, nt), Kst = rect_ref(rb + rs, rs, nt), KsTt = rect_ref(rb + 2 * rs, rs, nt),
             Ksst = rect_ref(rb + 3 * rs, rs, nt), W1 = rect_ref(rb + 4 * rs, rs, nt), W2 = rect_ref(rb + 5 * rs, rs, nt);
        TRef M = lower_ref(tiles.as<double>(), nlow * GP_TSQ);
        // A = K + yNoise I (lower tiles) and its factor
        GramArgs ga{};
        ga.X = c->dX; ga.T = c->dT; ga.p = sp; ga.s0 = 0; ga.n = n; ga.nX = c->nX; ga.nU = c->nU; ga.nt = nt;
        ga.M = M; ga.part = nullptr; ga.with_sums = 0; ga.f32 = 0;
        launch_gram(ga, 1, st);
        potrf_tiles(c, M, nt, nt, inv.as<double>(), (long long)nt * GP_TSQ, info.as<int>(), 0, 1, st);
        LdBuildArgs la{c->dX, c->dT, sp, 0, n, c->nX, c->nU, nt, doT, Kt, Kst, KsTt, Ksst};
        launch_ld_build(la, st);
        auto emit = [&](const TRef& R, double* host, double diag_add) {
            if (!host) return;
            launch_rect_gather(RectGatherArgs{R, n, nt, outb.as<double>(), diag_add}, st);
            HC(hipStreamSynchronize(st));
            copy_out_large(c, host, outb.p, (size_t)n * n * 8);
        };
        emit(Kt, CovWW, 0.0);
        emit(Kst, CovWWs, 0.0);
        emit(Kt, CovWWp, yno);
        if (CovC11 || CovC12 || CovC21 || CovC22) {
            // W1 = K L^-T, W2 = Ks' L^-T  (copies, then the tile-level left-looking solve)
            HC(hipMemcpyAsync(W1.base, Kt.base, (size_t)rs * 8, hipMemcpyDeviceToDevice, st));
            HC(hipMemcpyAsync(W2.base, KsTt.base, (size_t)rs * 8, hipMemcpyDeviceToDevice, st));
            TRef invref = TRef{inv.as<double>(), (long long)nt * GP_TSQ, 1, 0, 0, 0};
            for (TRef* Wp : {&W1, &W2}) {
                for (int k = 0; k < nt; ++k) {
                    if (k > 0) {
                        GemmArgs g{};
                        g.A = *Wp; g.B = M; g.C = *Wp;
                        g.shape = 1; g.i0 = 0; g.j0 = k; g.mi = nt; g.mj = 1;
                        g.k0 = 0; g.k1 = k; g.accumulate = 1; g.nbatch = 1; g.ntiles = nt;
                        gemm(c, g, st);
                    }
                    GemmArgs g{};
                    g.A = *Wp; g.B = invref; g.C = *Wp;
                    g.shape = 1; g.i0 = 0; g.j0 = k; g.mi = nt; g.mj = 1;
                    g.k0 = k; g.k1 = k + 1; g.accumulate = 0; g.nbatch = 1; g.ntiles = nt;
                    gemm(c, g, st);
                }
            }
            // C11 = K - W1 W1', C12 = Ks - W1 W2', C21 = Ks' - W2 W1', C22 = Kss - W2 W2'   (src/likelihood.jl:46-49)
            auto block = [&](const TRef& Cm, const TRef& Wa, const TRef& Wb, double* host) {
                if (!host) return;
                GemmArgs g{};
                g.A = Wa; g.B = Wb; g.C = Cm;
                g.shape = 1; g.i0 = 0; g.j0 = 0; g.mi = nt; g.mj = nt;
                g.k0 = 0; g.k1 = nt; g.accumulate = 1; g.nbatch = 1; g.ntiles = nt * nt;
                gemm(c, g, st);
                emit(Cm, host, 0.0);
            };
            block(Kt, W1, W1, CovC11);      // K, Ks, Ks', Kss are consumed in place: emitted above already
            block(Kst, W1, W2, CovC12);
            block(KsTt, W2, W1, CovC21);
            block(Ksst, W2, W2, CovC22);
        }
        HC(hipStreamSynchronize(st));
        HC(hipGetLastError());
        int hinfo = 0;
        HC(hipMemcpy(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost));
        c->last_info.assign(1, hinfo);
        return hinfo;
    });
}

static int summarize_impl(gpslc_ctx* c, const double* dx, int64_t n, int64_t m, int64_t rs, int64_t cs,
                          double ci, double* dmean, double* dlo, double* dhi) {
    ensure_streams(c);
    int mpad = 1;                                     // LDS image of a row (sorted in place): rows up to 16384 samples
    while (mpad < m && mpad < 16384) mpad <<= 1;      // (longer rows: radix select, no image)
    const double lowerQ = (1.0 - ci) / 2.0;          // src/driver.jl:130-131
    const double upperQ = 1.0 - lowerQ;
    launch_summarize(SummArgs{dx, rs, cs, (int)n, (int)m, mpad, lowerQ, upperQ, dmean, dlo, dhi}, c->streams[0]);
    HC(hipStreamSynchronize(c->streams[0]));
    HC(hipGetLastError());
    return GPSLC_OK;
}

int gpslc_summarize_dev(gpslc_ctx* c, const double* samples, int64_t n, int64_t m, int64_t row_stride,
                        int64_t col_stride, double credible_interval, double* mean, double* lower, double* upper) {
    if (!c) return -1;
    if (!samples) return bad_arg(c, 2, "samples is NULL");
    if (n < 1) return bad_arg(c, 3, "n < 1");
    if (m < 1 || m > 2147483647LL) return bad_arg(c, 4, "m must be in 1..2^31-1");
    if (!(credible_interval > 0.0 && credible_interval < 1.0)) return bad_arg(c, 7, "credible_interval not in (0,1)");
    if (!mean || !lower || !upper) return bad_arg(c, 8, "output is NULL");
    return guarded(c, [&]() { return summarize_impl(c, samples, n, m, row_stride, col_stride, credible_interval,
                                                    mean, lower, upper); });
}

int gpslc_summarize(gpslc_ctx* c, const double* samples, int64_t n, int64_t m, double credible_interval,
                    double* mean, double* lower, double* upper) {
    if (!c) return -1;
    if (!samples) return bad_arg(c, 2, "samples is NULL");
    if (n < 1) return bad_arg(c, 3, "n < 1");
    if (m < 1 || m > 2147483647LL) return bad_arg(c, 4, "m must be in 1..2^31-1");
    if (!(credible_interval > 0.0 && credible_interval < 1.0)) return bad_arg(c, 5, "credible_interval not in (0,1)");
    if (!mean || !lower || !upper) return bad_arg(c, 6, "output is NULL");
    return guarded(c, [&]() {
        DevBuf bx, o;
        const double* dx = up(bx, samples, (size_t)n * m);
        o.alloc(sizeof(double) * 3 * n);
        double* d = o.as<double>();
        summarize_impl(c, dx, n, m, 1, n, credible_interval, d, d + n, d + 2 * n);
        HC(hipMemcpy(mean, d, sizeof(double) * n, hipMemcpyDeviceToHost));
        HC(hipMemcpy(lower, d + n, sizeof(double) * n, hipMemcpyDeviceToHost));
        HC(hipMemcpy(upper, d + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
        return GPSLC_OK;
    });
}

int gpslc_sate_samples(const double* meanSATE, const double* varSATE, int64_t S, int32_t spp, uint64_t seed,
                       const double* z, double* out) {
    if (!meanSATE) return -1;
    if (!varSATE) return -2;
    if (S < 0) return -3;
    if (spp < 0) return -4;
    if (!out) return -7;
    for (int64_t j = 0; j < S; ++j)
        for (int32_t d = 0; d < spp; ++d) {
            const int64_t i = j * spp + d;
            const double zz = z ? z[i] : philox_normal_host(seed, (1ull << 40) + (uint64_t)j, (uint64_t)d);
            out[i] = meanSATE[j] + varSATE[j] * zz;   // variance used as sigma: src/estimation.jl:159
        }
    return GPSLC_OK;
}

int gpslc_last_info(const gpslc_ctx* c, int32_t* info, int64_t S) {
    if (!c) return -1;
    if (!info) return -2;
    if (S != (int64_t)c->last_info.size()) return -3;
    if (S) memcpy(info, c->last_info.data(), sizeof(int32_t) * S);
    return GPSLC_OK;
}

int gpslc_profile_reset(gpslc_ctx* c) {
    if (!c) return -1;
    for (int k = 0; k < kProfClasses; ++k) { c->prof_launches[k] = 0; c->prof_ms[k] = 0; c->prof_flop[k] = 0; }
    c->prof_used = 0;
    return GPSLC_OK;
}
int gpslc_profile_get_class(gpslc_ctx* c, int32_t cls, int64_t* launches, double* total_ms, double* total_flop) {
    if (!c) return -1;
    if (cls < 0 || cls >= kProfClasses) return bad_arg(c, 2, "kernel class must be 0..4");
    if (launches) *launches = c->prof_launches[cls];
    if (total_ms) *total_ms = c->prof_ms[cls];
    if (total_flop) *total_flop = c->prof_flop[cls];
    return GPSLC_OK;
}
int gpslc_profile_get(gpslc_ctx* c, int64_t* launches, double* total_ms, double* total_flop) {
    return gpslc_profile_get_class(c, 0, launches, total_ms, total_flop);
}

}  // extern "C"
