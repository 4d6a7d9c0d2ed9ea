// Host-side check of the task order of the persistent factorisation launch (causalgpslc.jl_amd/csrc/task_list.h) against the wait /
// publish rules of potrf_tasks_kernel (k_tilegemm.hip), restated here.  Inside a queue tickets are handed out in list order to
// RUNNING workgroups, so "every producer of a task sits EARLIER in the task's own queue" is the whole no-deadlock argument:
// replaying each queue sequentially, every wait must already be satisfied by the tasks before it.  Also: a matrix lives in exactly
// one queue, every tile is produced exactly once and column by column, the right-hand-side row ends complete.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../causalgpslc.jl_amd/csrc/task_list.h"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails++ < 20) { printf("FAIL %s:%d ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

static void run(int nt, int nb, int G, int rows, int back, bool aug_full, int merge) {
    long long ntasks = 0;
    std::vector<unsigned> L = build_task_list(nt, back, nb, G, rows, aug_full, &ntasks, merge);
    CHECK((long long)L.size() == TASK_LIST_HDR + ntasks, "size");
    std::vector<int> prog((size_t)nb * TASK_SYNC_STRIDE, 0), queue_of(nb, -1), backs(nb, 0);
    long long seen = 0;
    for (int x = 0; x < 8; ++x) {
        const unsigned first = L[x], len = L[8 + x];
        for (unsigned t = 0; t < len; ++t, ++seen) {
            const unsigned d = L[TASK_LIST_HDR + first + t];
            const int b = TASK_B(d), k = TASK_K(d), i = TASK_I(d), r = TASK_ROWS(d), kind = (int)(d >> 30);
            CHECK(b >= 0 && b < nb, "matrix index %d", b);
            if (b < 0 || b >= nb) continue;
            CHECK(queue_of[b] == -1 || queue_of[b] == x, "matrix %d in queues %d and %d", b, queue_of[b], x);
            queue_of[b] = x;
            int* p = &prog[(size_t)b * TASK_SYNC_STRIDE];
            if (kind == TASK_BACK) {               // kernel: task_wait(prog, nt); task_wait(prog + 1 + nt, nt)
                CHECK(back, "back task without back");
                CHECK(p[0] >= nt && p[1 + nt] >= nt, "back(%d) before its factor: diag %d aug %d", b, p[0], p[1 + nt]);
                ++backs[b];
            } else if (kind == TASK_DIAG) {        // kernel: k > 0: row k up to column k - 1, the augmented row (MT > 0), row k + 1 when merged
                if (k > 0) {
                    CHECK(p[1 + k] >= k, "diag(%d,%d): row %d at %d", b, k, k, p[1 + k]);
                    if (!aug_full) CHECK(p[1 + nt] >= k, "diag(%d,%d): aug row at %d", b, k, p[1 + nt]);
                    if (r > 1) CHECK(p[2 + k] >= k, "diag(%d,%d): row %d at %d", b, k, k + 1, p[2 + k]);
                }
                CHECK(p[0] == k, "diag(%d,%d) out of order: %d", b, k, p[0]);
                p[0] = k + 1;
                if (r > 1) {                       // goes on with strip(k + 1, k) (+ the augmented tile when it rides along)
                    CHECK(k + 1 < nt, "merged strip beyond the last column");
                    CHECK(p[2 + k] == k, "merged strip(%d,%d,%d) out of order: %d", b, k + 1, k, p[2 + k]);
                    p[2 + k] = k + 1;
                    if (!aug_full) { CHECK(p[1 + nt] == k, "aug tile (%d,%d) out of order: %d", b, k, p[1 + nt]); p[1 + nt] = k + 1; }
                }
            } else {                               // strips: diag(k) done; their own rows up to column k - 1
                CHECK(p[0] >= k + 1, "strip(%d,%d,%d) before diag: %d", b, i, k, p[0]);
                CHECK(i + r - 1 <= nt && i > k, "strip rows %d..%d of column %d", i, i + r - 1, k);
                for (int q = 0; q < r; ++q) {
                    if ((i < nt || aug_full) && k > 0) CHECK(p[1 + i + q] >= k, "strip(%d,%d,%d): row at %d", b, i + q, k, p[1 + i + q]);
                    CHECK(p[1 + i + q] == k, "strip(%d,%d,%d) out of order: %d", b, i + q, k, p[1 + i + q]);
                    p[1 + i + q] = k + 1;
                }
                if (kind == TASK_STRIP_AUG) { CHECK(p[1 + nt] == k, "aug tile (%d,%d) out of order: %d", b, k, p[1 + nt]); p[1 + nt] = k + 1; }
            }
        }
    }
    CHECK(seen == ntasks, "task count");
    for (int b = 0; b < nb; ++b) {
        const int* p = &prog[(size_t)b * TASK_SYNC_STRIDE];
        CHECK(queue_of[b] >= 0, "matrix %d has no tasks", b);
        CHECK(p[0] == nt, "matrix %d: %d diagonal tiles", b, p[0]);
        for (int i = 1; i < nt; ++i) CHECK(p[1 + i] == i, "matrix %d row %d: %d tiles", b, i, p[1 + i]);
        CHECK(p[1 + nt] == nt, "matrix %d: augmented row at %d", b, p[1 + nt]);
        CHECK(backs[b] == (back ? 1 : 0), "matrix %d: %d back tasks", b, backs[b]);
    }
}

int main() {
    int n = 0;
    const int nts[] = {2, 3, 4, 5, 8, 9, 16, 31, 32}, nbs[] = {1, 3, 7, 8, 9, 64, 125, 1000}, Gs[] = {1, 3, 8, 32, 64, 4096};
    for (int nt : nts) for (int nb : nbs) for (int G : Gs) for (int rows = 1; rows <= 4; ++rows)
        for (int back = 0; back < 2; ++back) for (int aug = 0; aug < 2; ++aug) for (int merge = 0; merge < 2; ++merge) {
            if (nb == 1000 && (nt > 9 || rows > 2)) continue;      // keep the run short
            run(nt, nb, G, rows, back, aug != 0, merge);
            ++n;
        }
    // descriptor fields at their limits
    const unsigned d = task_pack(TASK_MAX_BATCH - 1, 31, 32, TASK_STRIP, 4);
    CHECK(TASK_B(d) == TASK_MAX_BATCH - 1 && TASK_K(d) == 31 && TASK_I(d) == 32 && TASK_ROWS(d) == 4 && (d >> 30) == TASK_STRIP, "pack");
    CHECK(TASK_SYNC_STRIDE >= TASK_MAX_NT + 2, "progress words");
    printf("%s %d\n", fails ? "FAILED" : "OK", n);
    return fails ? 1 : 0;
}
