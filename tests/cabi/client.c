/* A C client of include/plp.h -- no Python, no torch: what a foreign-language binding (ctypes, cffi, cgo, JNI)
 * sees.  Solves the reference's known-answer LPs (tests/polytope_test.py:510-548), reduces the polytope of
 * test_reduce (:601-622) and runs a containment query; prints one line per result for the pytest wrapper.
 * Build: gcc -std=c99 -I include tests/cabi/client.c -L polytope_amd -lplp_hip -Wl,-rpath,$PWD/polytope_amd */
#include <stdio.h>
#include <stdint.h>
#include "plp.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != PLP_OK) { \
    printf("FAIL %s -> %d: %s\n", #call, rc_, plp_last_error()); return 1; } } while (0)

int main(void)
{
    printf("version %d devices %d\n", plp_version(), plp_device_count());
    if (plp_device_count() < 1) { printf("NODEVICE\n"); return 2; }
    plp_ctx *ctx = NULL;
    CHECK(plp_ctx_create(0, &ctx));

    /* min x s.t. -x <= 1  ->  x = -1 */
    { double c[1] = {1.0}, G[1] = {-1.0}, h[1] = {1.0}, x[1], fun[1]; int32_t st[1], it[1];
      CHECK(plp_lp_solve_batch(ctx, 1, 1, 1, c, G, h, NULL, x, fun, st, it));
      printf("lp1 status %d x %.17g fun %.17g\n", st[0], x[0], fun[0]); }
    /* min x+y s.t. -x <= 1, -y <= 1 ; and an unbounded one in the same batch (min x s.t. x <= 1) */
    { double c[4] = {1.0, 1.0, 1.0, 0.0}, G[8] = {-1.0, 0.0, 0.0, -1.0, 1.0, 0.0, 0.0, 0.0},
             h[4] = {1.0, 1.0, 1.0, 0.0}, x[4], fun[2]; int32_t st[2], m[2] = {2, 1};
      CHECK(plp_lp_solve_batch(ctx, 2, 2, 2, c, G, h, m, x, fun, st, NULL));
      printf("lp2 status %d x %.17g %.17g | status %d\n", st[0], x[0], x[1], st[1]); }
    /* reduce of test_reduce's polytope: row 1 ([1, .1] x <= 50.5) is redundant */
    { double A[10] = {1.0, 0.1, 1.0, 0.1, -1.0, 0.0, 0.0, 1.0, 0.0, -1.0}, b[5] = {50.0, 50.5, -40.0, 1.0, 0.0};
      /* the reference normalises rows in the Polytope constructor; rows 0,1 have norm sqrt(1.01) */
      const double s = 0.99503719020998915;  /* 1/sqrt(1.01) */
      A[0] *= s; A[1] *= s; A[2] *= s; A[3] *= s; b[0] *= s; b[1] *= s;
      uint64_t keep[1]; int32_t flags[1], nlp[1]; double r[1], xc[2];
      CHECK(plp_reduce_batch(ctx, 1, 5, 2, A, b, NULL, 1e-7, keep, flags, r, xc, nlp));
      printf("reduce keep 0x%llx flags %d nlp %d r %.12f\n", (unsigned long long)keep[0], flags[0], nlp[0], r[0]); }
    /* containment: unit square, 4 points as columns */
    { double A[8] = {1, 0, 0, 1, -1, 0, 0, -1}, b[4] = {1, 1, 0, 0}, X[8] = {0.5, 1.5, 1.0, -0.1, 0.5, 0.5, 1.0, 0.5};
      uint8_t out[4];
      CHECK(plp_contains(ctx, 1, 4, 2, A, b, NULL, 4, X, 1e-7, 0, out));
      printf("contains %d %d %d %d\n", out[0], out[1], out[2], out[3]); }
    /* bounding boxes: the rectangle [2,5]x[-1,3] (origin outside) and an infeasible strip (handed back: status 1) */
    { double A[16] = {1, 0, 0, 1, -1, 0, 0, -1,   1, 0, -1, 0, 0, 1, 0, -1}, b[8] = {5, 3, -2, 1,   1, -2, 1, 1};
      double lb[4], ub[4]; int32_t st[2];
      CHECK(plp_bbox_batch(ctx, 2, 4, 2, A, b, NULL, lb, ub, st));
      printf("bbox status %d lb %.9f %.9f ub %.9f %.9f | status %d\n", st[0], lb[0], lb[1], ub[0], ub[1], st[1]); }
    /* quickhull of the unit cube's corners + 3 interior points (built-in LU, no LAPACK): 12 triangles, 8 vertices */
    { double X[33]; int n = 0;
      for (int i = 0; i < 8; ++i) { X[n++] = (i & 1) - 0.5; X[n++] = ((i >> 1) & 1) - 0.5; X[n++] = ((i >> 2) & 1) - 0.5; }
      const double in[9] = {0.1, 0.0, -0.2, -0.3, 0.2, 0.1, 0.0, 0.0, 0.05};
      for (int i = 0; i < 9; ++i) X[n++] = in[i];
      int64_t simplex[4] = {0, 1, 2, 4};
      /* translate to the simplex centroid, as the caller of plp_quickhull_run has to (quickhull.py:188-192) */
      double xc[3] = {0, 0, 0};
      for (int k = 0; k < 4; ++k) for (int c = 0; c < 3; ++c) xc[c] += X[simplex[k] * 3 + c] / 4;
      for (int i = 0; i < 11; ++i) for (int c = 0; c < 3; ++c) X[i * 3 + c] -= xc[c];
      plp_qh_result *res = NULL;
      CHECK(plp_quickhull_run(ctx, 11, 3, X, simplex, 1e-7, NULL, &res));
      int64_t nf = 0, it = 0, made = 0;
      CHECK(plp_qh_result_sizes(res, &nf, &it, &made));
      double nrm[36 * 3], off[36]; int64_t verts[36 * 3]; int used[11] = {0}, nv = 0;
      if (nf <= 36) { CHECK(plp_qh_result_copy(res, nrm, off, verts));
        for (int i = 0; i < nf * 3; ++i) if (!used[verts[i]]) { used[verts[i]] = 1; ++nv; } }
      printf("quickhull facets %lld vertices %d\n", (long long)nf, nv);
      CHECK(plp_qh_result_free(res)); }
    /* region_diff search: unit square minus the strip 0.5 <= x <= 1.5: one piece, the square's rows + "x <= 0.5" */
    { const double A[24] = {1, 0, 0, 1, -1, 0, 0, -1,   -1, 0, 1, 0, 0, -1, 0, 1,   1, 0, -1, 0, 0, 1, 0, -1},
                   b[12] = {1, 1, 0, 0,   -0.5, 1.5, 1, 2,   0.5, -1.5, -1, -2};
      int32_t mi[1] = {4};
      plp_rdiff_result *res = NULL;
      CHECK(plp_region_diff_search(ctx, 2, 4, 1, mi, A, b, 1e-7, &res));
      int64_t nl = 0, nr = 0, nlp = 0, nb = 0;
      CHECK(plp_rdiff_result_sizes(res, &nl, &nr, &nlp, &nb));
      int32_t kind[8], off[9], rows[64];
      if (nl <= 8 && nr <= 64) { CHECK(plp_rdiff_result_copy(res, kind, off, rows));
        printf("region_diff pieces %lld:", (long long)nl);
        for (int k = 0; k < nl; ++k) { printf(" kind %d rows", kind[k]); for (int t = off[k]; t < off[k + 1]; ++t) printf(" %d", rows[t]); }
        printf("\n"); }
      CHECK(plp_rdiff_result_free(res)); }
    /* misuse is reported, not crashed on */
    { double c[1] = {1.0}; int rc = plp_lp_solve_batch(ctx, 1, 100000, 1, c, c, c, NULL, c, c, (int32_t *)c, NULL);  /* no LDS for 100000 rows */
      printf("envelope rc %d (%s)\n", rc, rc == PLP_EUNSUPPORTED ? "PLP_EUNSUPPORTED" : "?"); }
    CHECK(plp_ctx_destroy(ctx));
    printf("done\n");
    return 0;
}
