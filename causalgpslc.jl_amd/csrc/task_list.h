// Task descriptors and the host-built task order of the persistent factorisation launch (potrf_tasks_kernel, k_tilegemm.hip) —
// plain C++ (no HIP in here), so that the CPU test suite can compile it and check the order against the kernel's wait rules
// (tests/c/task_list_test.cpp: every task behind its producers in its own queue = the launch cannot deadlock; every tile exactly once).
#pragma once
#include <algorithm>
#include <vector>
#ifdef __HIPCC__
#define TASK_HD __host__ __device__
#else
#define TASK_HD
#endif

#define TASK_NONE 0xFFFFFFFFu
#define TASK_LIST_HDR 32          // words: [0, 8) first descriptor of queue x, [8, 16) descriptors of queue x
#define TASK_SYNC_HDR 32          // ints: [0, 8) ticket heads, [8] time-out word, [9] tasks finished
#define TASK_SYNC_STRIDE 40       // ints per matrix: [0] diagonal tiles finished, [1 + i] finished tiles of tile row i (i <= nt)
#define TASK_MAX_NT 32
// descriptor: bits 0..16 batch element, 17..18 (strips) number of consecutive tile rows of the column the task covers minus 1,
// 19..23 column k, 24..29 (first) tile row i (up to nt = 32: the augmented tile row), 30..31 kind: 0 strip(i.., k), 1 diag(k), 2 strip(k + 1, k) that also applies the
// panel product to the augmented tile (nt, k), 3 the back-substitution of the matrix
enum { TASK_STRIP = 0, TASK_DIAG = 1, TASK_STRIP_AUG = 2, TASK_BACK = 3 };
#define TASK_MAX_BATCH (1 << 17)
#define TASK_B(d) ((int)((d) & 0x1FFFF))
#define TASK_ROWS(d) ((int)(((d) >> 17) & 3) + 1)
#define TASK_K(d) ((int)(((d) >> 19) & 31))
#define TASK_I(d) ((int)(((d) >> 24) & 63))
TASK_HD inline unsigned task_pack(int b, int k, int i, int kind, int rows = 1) {
    return (unsigned)b | ((unsigned)(rows - 1) << 17) | ((unsigned)k << 19) | ((unsigned)i << 24) | ((unsigned)kind << 30);
}

// ---- the persistent factorisation launch (potrf_tasks_kernel, k_tilegemm.hip) -------------------------------------------
// Task order of one queue (= one XCD's contiguous run of matrices).  Stages of a matrix: s = 2k: diag(k), s = 2k + 1: the
// strips of column k.  Matrices are taken in groups of G; step t of the order holds stage s of group t - s for every s — a
// skewed wavefront, so that (1) every task follows its producers (stage s - 1 of the same group sits one whole step
// earlier: ~ G * (nt + nt (nt + 1) / 2) tickets, several rounds of the XCD's 64 workgroup slots — a consumer practically
// never finds its producer unfinished), and (2) every stretch of the order mixes the latency-bound diagonal tasks of some
// groups with the MFMA-bound strips of others.  Inside a stage the strips of one matrix are consecutive tickets: they run
// at the same time on one XCD and share the B panel L(k, 0..k-1) in its L2.
inline std::vector<unsigned> build_task_list(int nt, int back, int nb, int G, int rows_per_task, bool aug_full, long long* ntasks_out,
                                             int merge_diag = 1) {
    std::vector<unsigned> out(TASK_LIST_HDR, 0u);
    const int NS = 2 * nt + (back ? 1 : 0);       // back: one more stage, the back-substitution of the finished factor
    long long total = 0;
    const int wq = nb >> 3, wrm = nb & 7;
    for (int x = 0; x < 8; ++x) {
        const int x0 = x * wq + std::min(x, wrm), xc = wq + (x < wrm ? 1 : 0);
        const size_t first = out.size() - TASK_LIST_HDR;
        const int NG = (xc + G - 1) / G;
        for (int t = 0; t < NG + NS - 1; ++t)
            for (int s = 0; s < NS; ++s) {
                const int g = t - s;
                if (g < 0 || g >= NG) continue;
                const int k = s >> 1;
                for (int j = g * G; j < std::min(xc, (g + 1) * G); ++j) {
                    const int b = x0 + j;
                    if (s == 2 * nt) { out.push_back(task_pack(b, 0, 0, TASK_BACK)); continue; }
                    // a diagonal task goes on with the strip of tile row k + 1 and the augmented tile of its column (rows = 2 in
                    // its descriptor): the next diagonal task waits for exactly those two.  The last column has no strip: its
                    // augmented tile is a task of its own
                    if ((s & 1) == 0) { out.push_back(task_pack(b, k, k, TASK_DIAG, (merge_diag && k + 1 < nt) ? 2 : 1)); continue; }
                    // aug_full (more than 32 right-hand sides): the augmented row is a tile row like the others, strip(nt, k) in
                    // every column; otherwise its tiles ride with the diagonal tasks and only the last column's is a task of its own
                    if (k + 1 < nt) {
                        if (!merge_diag) out.push_back(task_pack(b, k, k + 1, aug_full ? TASK_STRIP : TASK_STRIP_AUG));
                        if (aug_full) out.push_back(task_pack(b, k, nt, TASK_STRIP));
                    } else out.push_back(task_pack(b, k, nt, TASK_STRIP));
                    for (int i = k + 2; i < nt; i += rows_per_task)
                        out.push_back(task_pack(b, k, i, TASK_STRIP, std::min(rows_per_task, nt - i)));
                }
            }
        out[x] = (unsigned)first;
        out[8 + x] = (unsigned)(out.size() - TASK_LIST_HDR - first);
        total += out[8 + x];
    }
    *ntasks_out = total;
    return out;
}

