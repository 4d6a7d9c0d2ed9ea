/* A consumer of libgpslc_hip.so written in plain C (what any FFI does): no Python, no torch.
 * Scores Gaussian-process nodes whose answer has a closed form and checks the library against it:
 *   nF = 0  ->  covariance = scale * 11' + noise * I:
 *       logdet = (n - 1) log(noise) + log(noise + n scale)
 *       x' C^-1 x = x'x / noise - scale (sum x)^2 / (noise (noise + n scale))          (Sherman-Morrison)
 * through gpslc_gp_logpdf (S parameter sets), gpslc_nodes_logpdf (heterogeneous nodes) and, for n large enough to
 * leave the single-workgroup kernels, the tiled path; then the ensemble driver (gpslc_set_data, gpslc_predict,
 * gpslc_set_ensemble, gpslc_predict_multi): the reference's exact-zero identities at n = 300, the placement of a sample in an
 * ensemble, and the ensemble sharded over two contexts.
 * Exit code 0 = all within 1e-10 relative (exact where stated). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "gpslc_hip.h"

static double closed_form(int n, double scale, double noise, const double* x) {
    double sx = 0.0, sxx = 0.0;
    for (int i = 0; i < n; ++i) { sx += x[i]; sxx += x[i] * x[i]; }
    const double logdet = (n - 1) * log(noise) + log(noise + n * scale);
    const double quad = sxx / noise - scale * sx * sx / (noise * (noise + n * scale));
    return -0.5 * (n * log(2.0 * 3.14159265358979323846) + logdet + quad);
}

int main(void) {
    const int sizes[3] = {150, 272, 700};
    int bad = 0;
    printf("%s\n", gpslc_version());
    for (int t = 0; t < 3; ++t) {
        const int n = sizes[t];
        gpslc_ctx* ctx = NULL;
        int st = gpslc_create(&ctx, 0, n, 0, 0, GPSLC_FLAG_DEFAULT);
        if (st != GPSLC_OK) { printf("gpslc_create failed: %d\n", st); return 2; }
        double* x = (double*)malloc(sizeof(double) * n);
        for (int i = 0; i < n; ++i) x[i] = sin(0.37 * i) + 0.01 * i;
        /* (a) S = 3 parameter sets in one call */
        const double scale[3] = {0.7, 1.9, 0.05}, noise[3] = {0.4, 1.3, 2.5};
        double out[3];
        st = gpslc_gp_logpdf(ctx, 3, 0, NULL, 1, NULL, scale, noise, x, 1, out);
        if (st != GPSLC_OK) { printf("gpslc_gp_logpdf: status %d (%s)\n", st, gpslc_last_error(ctx)); return 3; }
        for (int s = 0; s < 3; ++s) {
            const double ref = closed_form(n, scale[s], noise[s], x);
            const double rel = fabs(out[s] - ref) / fabs(ref);
            printf("n=%d set %d: %.12g vs closed form %.12g (rel %.1e)\n", n, s, out[s], ref, rel);
            if (!(rel <= 1e-10)) ++bad;
        }
        /* (b) the fused node call */
        gpslc_node nodes[2] = {{0, 0, NULL, NULL, 1.1, 0.9, x}, {0, 0, NULL, NULL, 0.3, 0.6, x}};
        double out2[2];
        st = gpslc_nodes_logpdf(ctx, 2, nodes, out2);
        if (st != GPSLC_OK) { printf("gpslc_nodes_logpdf: status %d (%s)\n", st, gpslc_last_error(ctx)); return 4; }
        for (int s = 0; s < 2; ++s) {
            const double ref = closed_form(n, nodes[s].scale, nodes[s].noise, x);
            if (!(fabs(out2[s] - ref) <= 1e-10 * fabs(ref))) { printf("node %d: %.12g vs %.12g\n", s, out2[s], ref); ++bad; }
        }
        /* (b') a prior draw: chol(K) e_0 is the first column of the factor, K[:,0] / sqrt(K[0,0]) */
        if (n <= 640) {
            double* z = (double*)calloc((size_t)n, sizeof(double));
            double* w = (double*)malloc(sizeof(double) * n);
            z[0] = 1.0;
            gpslc_node dn = {0, 0, NULL, NULL, 1.1, 0.9, z};
            st = gpslc_nodes_draw(ctx, 1, &dn, w, NULL);
            if (st != GPSLC_OK) { printf("gpslc_nodes_draw: status %d (%s)\n", st, gpslc_last_error(ctx)); return 5; }
            const double d0 = sqrt(1.1 + 0.9);
            for (int i = 0; i < n; ++i) {
                const double ref = (i == 0 ? 2.0 : 1.1) / d0;
                if (!(fabs(w[i] - ref) <= 1e-12)) { printf("draw row %d: %.15g vs %.15g\n", i, w[i], ref); ++bad; break; }
            }
            free(z); free(w);
        }
        /* (c) argument errors come back as negative status codes, never as a crash */
        if (gpslc_gp_logpdf(ctx, 1, 0, NULL, 1, NULL, scale, noise, NULL, 1, out) != -9) ++bad;
        /* (d) a matrix that is not positive definite -> LAPACK-style info (scale * 11' with negative noise) */
        const double nneg = -0.5, sc1 = 1.0;
        st = gpslc_gp_logpdf(ctx, 1, 0, NULL, 1, NULL, &sc1, &nneg, x, 1, out);
        if (st <= 0) { printf("expected a positive info for a non-PD matrix, got %d\n", st); ++bad; }
        free(x);
        gpslc_destroy(ctx);
    }
    /* (e) the ensemble driver from plain C: gpslc_set_data + gpslc_predict on a 300-instance data set (three tiles per side).
     *   - every treatment equal to the intervention level: MeanSATE, MeanITE exactly 0, VarSATE exactly pred_noise / n
     *     (test/estimation.jl:6-136 of the reference, at n > 1);
     *   - gpslc_set_ensemble: sample 1 of a 3-sample ensemble predicted on its own draws the normals the 3-sample call
     *     draws for it (bit for bit). */
    {
        const int n = 300, nX = 2, nU = 1, S = 3, L = 2, spp = 2;
        gpslc_ctx* ctx = NULL;
        if (gpslc_create(&ctx, 0, n, nX, nU, GPSLC_FLAG_DEFAULT) != GPSLC_OK) { printf("gpslc_create (predict) failed\n"); return 6; }
        double* X = (double*)malloc(sizeof(double) * n * nX);
        double* T = (double*)malloc(sizeof(double) * n);
        double* Y = (double*)malloc(sizeof(double) * n);
        double* U = (double*)malloc(sizeof(double) * n * nU * S);
        for (int i = 0; i < n; ++i) {
            X[i] = sin(0.11 * i); X[n + i] = cos(0.07 * i);
            T[i] = 0.75; Y[i] = sin(0.3 * i) + 0.2 * X[i];
            for (int s = 0; s < S; ++s) U[i + n * s] = cos(0.05 * (i / 10) + s);
        }
        const double uyLS[3] = {1.1, 0.9, 1.4}, xyLS[6] = {1.0, 1.3, 0.8, 1.2, 1.5, 0.7}, tyLS[3] = {1.0, 1.2, 0.9};
        const double yScale[3] = {1.0, 0.8, 1.3}, yNoise[3] = {0.5, 0.7, 0.4};
        const double doT[2] = {0.75, 0.1}, pn = 1e-10;
        int st = gpslc_set_data(ctx, X, T, Y);
        if (st != GPSLC_OK) { printf("gpslc_set_data: %d\n", st); return 6; }
        double mS[6], vS[6];
        double* mI = (double*)malloc(sizeof(double) * n * S * L);
        double* dr = (double*)malloc(sizeof(double) * L * n * S * spp);
        st = gpslc_predict(ctx, S, U, uyLS, xyLS, tyLS, yScale, yNoise, L, doT, pn, spp, 5, NULL, mS, vS, mI, dr);
        if (st != GPSLC_OK) { printf("gpslc_predict: status %d (%s)\n", st, gpslc_last_error(ctx)); return 6; }
        for (int s = 0; s < S; ++s) {     /* level 0 == every T: exact zeros; element (s, l) at s + S*l */
            if (mS[s] != 0.0 || vS[s] != (n * pn) / ((double)n * (double)n)) { printf("exact-zero identity broken for sample %d: %.17g %.17g\n", s, mS[s], vS[s]); ++bad; }
            for (int i = 0; i < n; ++i) if (mI[i + n * s] != 0.0) { printf("MeanITE not exactly 0 at (%d, %d)\n", i, s); ++bad; break; }
            if (!(mS[s + S] != 0.0) || !(vS[s + S] > 0.0)) { printf("level 1 degenerate for sample %d\n", s); ++bad; }
        }
        /* sample 1 alone, placed at offset 1 of an ensemble of 3 */
        double m1[2], v1[2];
        double* d1 = (double*)malloc(sizeof(double) * L * n * spp);
        if (gpslc_set_ensemble(ctx, 1, 3) != GPSLC_OK) { printf("gpslc_set_ensemble failed\n"); ++bad; }
        st = gpslc_predict(ctx, 1, U + (size_t)n * nU, uyLS + 1, xyLS + 2, tyLS + 1, yScale + 1, yNoise + 1, L, doT, pn, spp, 5, NULL,
                           m1, v1, NULL, d1);
        if (st != GPSLC_OK) { printf("gpslc_predict (one sample): status %d (%s)\n", st, gpslc_last_error(ctx)); return 6; }
        if (m1[0] != mS[1] || m1[1] != mS[1 + S] || v1[1] != vS[1 + S]) { printf("one-sample SATE differs from the ensemble's\n"); ++bad; }
        /* draws: L x n x (S*spp), level fastest; sample 1's columns are 1*spp .. 2*spp - 1 */
        for (size_t e = 0; e < (size_t)L * n * spp; ++e)
            if (d1[e] != dr[(size_t)L * n * spp * 1 + e]) { printf("placed draws differ at element %zu\n", e); ++bad; break; }
        if (gpslc_set_ensemble(ctx, 5, 3) != -3 || gpslc_set_ensemble(ctx, 0, 0) != GPSLC_OK) { printf("gpslc_set_ensemble argument check\n"); ++bad; }
        printf("predict from C: MeanSATE(level 1) = %.12g %.12g %.12g\n", mS[S], mS[S + 1], mS[S + 2]);
        /* (f) gpslc_predict_multi: the same ensemble sharded over two contexts (both on device 0 here; one per GPU on a node):
         *     every output, the seeded draws included, equals the single-context call bit for bit. */
        {
            gpslc_ctx* c2 = NULL;
            if (gpslc_create(&c2, 0, n, nX, nU, GPSLC_FLAG_DEFAULT) != GPSLC_OK || gpslc_set_data(c2, X, T, Y) != GPSLC_OK) { printf("second ctx failed\n"); return 7; }
            gpslc_ctx* both[2] = {ctx, c2};
            double mS2[6], vS2[6];
            int32_t info[3] = {-1, -1, -1};
            double* mI2 = (double*)malloc(sizeof(double) * n * S * L);
            double* dr2 = (double*)malloc(sizeof(double) * L * n * S * spp);
            st = gpslc_predict_multi(2, both, S, U, uyLS, xyLS, tyLS, yScale, yNoise, L, doT, pn, spp, 5, NULL, mS2, vS2, mI2, dr2, info);
            if (st != GPSLC_OK) { printf("gpslc_predict_multi: status %d (%s)\n", st, gpslc_last_error(ctx)); return 7; }
            for (int e = 0; e < S * L; ++e) if (mS2[e] != mS[e] || vS2[e] != vS[e]) { printf("multi: SATE differs at %d\n", e); ++bad; break; }
            for (size_t e = 0; e < (size_t)n * S * L; ++e) if (mI2[e] != mI[e]) { printf("multi: MeanITE differs at %zu\n", e); ++bad; break; }
            for (size_t e = 0; e < (size_t)L * n * S * spp; ++e) if (dr2[e] != dr[e]) { printf("multi: draws differ at %zu\n", e); ++bad; break; }
            if (info[0] | info[1] | info[2]) { printf("multi: info not zero\n"); ++bad; }
            gpslc_ctx* twice[2] = {ctx, ctx};
            if (gpslc_predict_multi(2, twice, S, U, uyLS, xyLS, tyLS, yScale, yNoise, L, doT, pn, spp, 5, NULL, mS2, vS2, NULL, NULL, NULL) != -2) { printf("multi: a ctx listed twice must be refused\n"); ++bad; }
            free(mI2); free(dr2);
            gpslc_destroy(c2);
        }
        free(X); free(T); free(Y); free(U); free(mI); free(dr); free(d1);
        gpslc_destroy(ctx);
    }
    if (bad) printf("FAILED: %d mismatches\n", bad); else printf("ok\n");
    return bad ? 1 : 0;
}
