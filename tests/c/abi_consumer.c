/* A consumer of libgpslc_hip.so written in plain C (what any FFI does): no Python, no torch.
 * Scores Gaussian-process nodes whose answer has a closed form and checks the library against it:
 *   nF = 0  ->  covariance = scale * 11' + noise * I:
 *       logdet = (n - 1) log(noise) + log(noise + n scale)
 *       x' C^-1 x = x'x / noise - scale (sum x)^2 / (noise (noise + n scale))          (Sherman-Morrison)
 * through gpslc_gp_logpdf (S parameter sets), gpslc_nodes_logpdf (heterogeneous nodes) and, for n large enough to
 * leave the single-workgroup kernels, the tiled path.  Exit code 0 = all within 1e-10 relative. */
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
    if (bad) printf("FAILED: %d mismatches\n", bad); else printf("ok\n");
    return bad ? 1 : 0;
}
