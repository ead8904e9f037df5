/*
 * plp.h -- C ABI of libplp_hip.so: the MI355X (gfx950) engine for the batched small-LP hot
 * path of tulip-control/polytope.  Plain pointers and sizes only; no torch / C++ types.
 *
 * The reference is pure Python and has no FFI of its own; the boundary this library sits
 * behind is the reference's solver plug-in surface and the numpy kernels listed below
 * (paths relative to the reference checkout).  Each entry point cites what it replaces.
 *
 * Conventions
 *   - all floating point data is IEEE binary64 ("double"), C-contiguous, caller-owned;
 *     nothing is retained after a call returns (reduce() passes aliases of arrays it
 *     mutates around the call: polytope/polytope.py:1146-1151).
 *   - a batch of polytopes is packed as A[B][m_max][d], b[B][m_max] with an optional
 *     int32 m[B] (rows actually used, <= m_max; NULL = all m_max).  d <= 16.  m_max <= 64 for the fused
 *     kernels (reduce, bounding boxes, pair LPs: register-resident dictionaries); plp_lp_solve_batch and
 *     plp_cheby_batch also take m_max > 64 -- region_diff stacks m_poly + sum(active rows) without a limit
 *     (polytope/polytope.py:2212-2224) -- as long as one dictionary fits the CU's 160 KB of LDS
 *     (about 890 rows at n = 17, 3000 at n = 4); beyond that PLP_EUNSUPPORTED.
 *   - per-LP status codes are scipy.optimize.linprog's, as returned by
 *     polytope.solvers.lpsolve (polytope/solvers.py:76-106, 155-158):
 *        0 optimal, 1 iteration limit, 2 infeasible, 3 unbounded, 4 numerical trouble.
 *     An LP that fails is NOT an error of the call: it is status != 0 with x/fun = NaN.
 *   - function return value: PLP_OK, or PLP_E* for API misuse / HIP errors
 *     (plp_last_error() has the text).  Nothing throws across this boundary.
 *   - "_dev" variants take DEVICE pointers and a HIP stream (void* = hipStream_t, NULL =
 *     default stream), enqueue the work and return without synchronising.  The plain
 *     variants take HOST pointers, copy in/out through the context's scratch buffers and
 *     block until results are host-visible.
 *   - a plp_ctx is not thread safe; use one per thread (the reference is single-threaded).
 */
#ifndef PLP_H
#define PLP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLP_OK 0
#define PLP_EINVAL 1       /* bad argument (NULL pointer, negative size, ...)            */
#define PLP_EUNSUPPORTED 2 /* size outside the engine's envelope (d > 16, too many rows)  */
#define PLP_EHIP 3         /* HIP runtime error                                          */
#define PLP_ENODEVICE 4    /* no gfx950 device visible                                   */
#define PLP_ENONFINITE 5   /* an input held inf or nan (only with plp_ctx_set_check_finite) */

/* flags[] bits written by plp_reduce_batch (see polytope/polytope.py:1053-1163) */
#define PLP_RF_EMPTY 1  /* not full-dimensional: reduce() returns Polytope()       (:1081-1082) */
#define PLP_RF_EARLY 2  /* returned at neq <= nx+1, minrep stays False             (:1114-1116, :1136-1138) */
#define PLP_RF_MINREP 4 /* all redundancy LPs done, minrep = True                  (:1161-1163) */
#define PLP_RF_LPFAIL 8 /* a bounding-box LP ended 1/4: reference raises RuntimeError (:1378-1384) */
/* set beside PLP_RF_EMPTY when the verdict is the ENGINE's, not the polytope's: the fused kernel's Chebyshev LP ended neither
 * optimal nor infeasible, or "optimal" at a centre that violates a row (rows a hair apart).  The fused reduce is not verified:
 * ask plp_cheby_batch (verified) about such a polytope, as polytope_amd.polytope.reduce does. */
#define PLP_RF_F1OPEN 32

typedef struct plp_ctx plp_ctx;

int plp_version(void);
/* number of visible HIP devices whose architecture is gfx950 (0 if none / no runtime) */
int plp_device_count(void);
const char *plp_last_error(void);

int plp_ctx_create(int device, plp_ctx **out);
/* on != 0: the host-pointer entry points plp_lp_solve_batch / plp_cheby_batch / plp_bbox_batch / plp_reduce_batch return
 * PLP_ENONFINITE, writing no output, when c / G / h resp. A / b hold an inf or a nan -- the ValueError
 * scipy.optimize.linprog raises for such input behind solvers.lpsolve (solvers.py:152-154).  Large batches are checked
 * by the threads that stage them for upload, at no extra pass over the data.  Off by default: the kernels then report
 * such LPs with status 4. */
int plp_ctx_set_check_finite(plp_ctx *ctx, int on);
int plp_ctx_destroy(plp_ctx *ctx);
/* block until everything enqueued on `stream` (NULL = default stream) has finished */
int plp_ctx_synchronize(plp_ctx *ctx, void *stream);

/*
 * Batched lpsolve():  B independent LPs   min c'x  s.t.  G x <= h,  x free.
 * Replaces: polytope.solvers.lpsolve / _solve_lp_using_scipy
 *           (polytope/solvers.py:76-106, :149-158) called in Python loops at
 *           polytope/polytope.py:1150, :1288, :1371, :1393.
 * c[B][n], G[B][m_max][n], h[B][m_max], m[B] or NULL;  n <= 17.  m_max > 64: LDS-resident engine.
 * Out: x[B][n], fun[B] (NaN unless status 0), status[B], iters[B] (may be NULL).
 */
int plp_lp_solve_batch(plp_ctx *ctx, int64_t B, int m_max, int n, const double *c, const double *G,
                       const double *h, const int32_t *m, double *x, double *fun, int32_t *status,
                       int32_t *iters);
int plp_lp_solve_batch_dev(plp_ctx *ctx, void *stream, int64_t B, int m_max, int n, const double *c,
                           const double *G, const double *h, const int32_t *m, double *x, double *fun,
                           int32_t *status, int32_t *iters);

/*
 * Batched Chebyshev ball, LP form F1:  max r  s.t.  a_i.x + ||a_i|| r <= b_i.
 * Replaces: cheby_ball (polytope/polytope.py:1241-1300; G,h,c built at :1283-1287) and
 *           therefore is_fulldim (:962-985).
 * Out: r[B] = x[-1] of the LP (may be negative: the caller applies ":1291 r < 0 -> empty"),
 *      xc[B][d], status[B] raw LP status.
 */
int plp_cheby_batch(plp_ctx *ctx, int64_t B, int m_max, int d, const double *A, const double *b,
                    const int32_t *m, double *r, double *xc, int32_t *status);
int plp_cheby_batch_dev(plp_ctx *ctx, void *stream, int64_t B, int m_max, int d, const double *A,
                        const double *b, const int32_t *m, double *r, double *xc, int32_t *status);

/*
 * Bounding boxes of a batch of polytopes (1 <= d <= 16, m_max <= 64): per polytope the Chebyshev LP and then the 2d LPs
 * min +-e_i.x (form F3) started from its centre -- one launch instead of 2d generic LPs per polytope (d > 8, and
 * d = 5..8 with more than 32 rows in batches beyond 1024 polytopes: one polytope per wavefront, every LP with a
 * wave-uniform pivot; d >= 14: the 2d LPs without a stored dictionary).
 * Replaces: the LP loops of bounding_box (polytope/polytope.py:1367-1409).
 * Out: lb[B][d], ub[B][d] (-inf / +inf where the LP is unbounded, :1376 / :1398) and status[B]:
 *      0 = lb/ub hold the box,
 *      1 = not handled here (no Chebyshev centre with r >= 1e-6: empty, flat or unbounded-ball polytopes; LPs that
 *          need Bland's rule, for d > 8 also LPs of more than 32 pivots): lb/ub are NaN and the caller solves the 2d generic LPs (plp_lp_solve_batch), whose
 *          statuses 2 / 3 then take the reference's branches (:1378-1380, :1400-1402).
 */
int plp_bbox_batch(plp_ctx *ctx, int64_t B, int m_max, int d, const double *A, const double *b, const int32_t *m,
                   double *lb, double *ub, int32_t *status);
int plp_bbox_batch_dev(plp_ctx *ctx, void *stream, int64_t B, int m_max, int d, const double *A, const double *b,
                       const int32_t *m, double *lb, double *ub, int32_t *status);

/*
 * Fused reduce() of a batch of polytopes (none of them minrep):  F1, parallel-row dedupe,
 * bounding-box prefilter (2d LPs F3, when rows > 3d) and one redundancy LP F2 per row.
 * Replaces: reduce (polytope/polytope.py:1053-1163), including the is_fulldim/cheby_ball
 *           call at :1081 and the bounding_box call at :1119 (:1314-1411).
 * Out: keep[B]  bit i set <=> input row i is kept;
 *      flags[B] PLP_RF_* ;  r[B], xc[B][d] the Chebyshev ball of the INPUT polytope
 *      (r = 0, xc = NaN when cheby_ball would return (0, None));
 *      nlp[B]   number of LPs the reference would have issued (= LPs solved here).
 */
int plp_reduce_batch(plp_ctx *ctx, int64_t B, int m_max, int d, const double *A, const double *b,
                     const int32_t *m, double abs_tol, uint64_t *keep, int32_t *flags, double *r,
                     double *xc, int32_t *nlp);
int plp_reduce_batch_dev(plp_ctx *ctx, void *stream, int64_t B, int m_max, int d, const double *A,
                         const double *b, const int32_t *m, double abs_tol, uint64_t *keep,
                         int32_t *flags, double *r, double *xc, int32_t *nlp);

/*
 * In-run accounting of plp_reduce_batch[_dev]: nlp[] counts the LPs the REFERENCE issues for a polytope
 * (polytope/polytope.py:1081, :1119, :1142-1151: 1 + 2d + one per row that reaches the redundancy loop).  Not all of them
 * run the simplex here: a redundancy LP whose verdict "keep" a feasible witness point proves (the presolve)
 * is answered without one.  This call returns the number of LPs that DID run the simplex in the fused reduce launches
 * issued through `ctx` since the counter was last reset (all streams; the first call of a context switches the counting
 * on and returns 0).  It enqueues a copy on `stream` and blocks until it has landed, so it sees every launch enqueued on
 * that stream before it.  reset != 0: the counter restarts at zero.  The second pass that redoes polytopes flagged for
 * Bland's rule and the kernels for more than 64 rows do not count.
 */
int plp_reduce_counters(plp_ctx *ctx, void *stream, uint64_t *simplex_runs, int reset);

/*
 * The verifier behind plp_lp_solve_batch / plp_cheby_batch / plp_bbox_batch (round 6; csrc/plp_verify.hpp says why): every
 * answer of the LP engines is certified against the original rows from its final basis -- its vertex recomputed, its
 * multipliers' signs checked, every row tested -- and what does not certify (or is reported unbounded / at a limit) is
 * solved again by a careful double-double engine.  This call returns, for the LAST verified batch issued on `stream`
 * through `ctx`, how many of its LPs went to the careful engine (0 on random, ragged, rescaled, flat or lattice data;
 * ~5 % of the LPs of polytopes with rows 1e-16 .. 1e-5 rad apart; every unbounded LP).  It blocks until that batch is done.
 * PLP_VERIFY=0 in the environment switches the verifier off (A/B measurements: the answers then are the engines' own).
 */
int plp_verify_counters(plp_ctx *ctx, void *stream, int64_t *careful_lps);

/*
 * The same for polytopes of ANY row count whose rows and dictionary fit the LDS of a CU (about 500 rows at d = 16,
 * 2000 at d = 3): `reduce` has no row limit in the reference (polytope/polytope.py:1053-1163), Polytope.intersect
 * stacks m1 + m2 rows (:268-275) and region_diff's leaves as many as the search collected (:2276).
 * keep[B][W], W = (m_max + 63) / 64 words per polytope, bit i of word i / 64 <=> input row i is kept; everything else
 * as plp_reduce_batch (to which m_max <= 64 is passed on, W = 1).  PLP_EUNSUPPORTED when a polytope does not fit.
 */
int plp_reduce_wide_batch(plp_ctx *ctx, int64_t B, int m_max, int d, const double *A, const double *b,
                          const int32_t *m, double abs_tol, uint64_t *keep, int32_t *flags, double *r,
                          double *xc, int32_t *nlp);
int plp_reduce_wide_batch_dev(plp_ctx *ctx, void *stream, int64_t B, int m_max, int d, const double *A,
                              const double *b, const int32_t *m, double abs_tol, uint64_t *keep,
                              int32_t *flags, double *r, double *xc, int32_t *nlp);

/*
 * Containment of N points in P polytopes:  all_i( A_p[i,:].x - b_p[i] < abs_tol ).
 * Replaces: Polytope.contains (polytope/polytope.py:206-218), Region.contains (:732-746),
 *           is_inside (:1017-1029), __contains__ (:191-204, :723-730).
 * X[d][N]: column vectors exactly as the reference takes them.
 * mode 0 (Region.contains): out[N]    = OR over the P polytopes (all P*N tests evaluated)
 * mode 1 (per polytope)   : out[P][N] = Polytope.contains of each polytope
 */
int plp_contains(plp_ctx *ctx, int P, int m_max, int d, const double *A, const double *b, const int32_t *m,
                 int64_t N, const double *X, double abs_tol, int mode, uint8_t *out);
int plp_contains_dev(plp_ctx *ctx, void *stream, int P, int m_max, int d, const double *A, const double *b,
                     const int32_t *m, int64_t N, const double *X, double abs_tol, int mode, uint8_t *out);

/*
 * quickhull outside-set assignment and furthest point.
 * Replaces: distance() (polytope/quickhull.py:117-121), the assignment loops (:224-245,
 *           :311-336: a point goes to the FIRST facet with distance > abs_tol) and
 *           Facet.get_furthest (:87-102: first maximum wins).
 * X[N][d] (rows = points, as quickhull takes them), normals[F][d], offsets[F].
 * Out: facet_of_point[N] (-1 = inside every facet), dist[N] (0 when unassigned),
 *      argmax[F] (index of the furthest point assigned to facet f, -1 if none), maxd[F] (its
 *      distance; 0 if none).
 */
int plp_assign(plp_ctx *ctx, int64_t N, int d, const double *X, int F, const double *normals,
               const double *offsets, double abs_tol, int32_t *facet_of_point, double *dist,
               int64_t *argmax, double *maxd);
int plp_assign_dev(plp_ctx *ctx, void *stream, int64_t N, int d, const double *X, int F,
                   const double *normals, const double *offsets, double abs_tol, int32_t *facet_of_point,
                   double *dist, int64_t *argmax, double *maxd);

/*
 * quickhull main loop: outside sets kept on the device for the whole run.
 * Replaces: the per-iteration pooling of the visible facets' outside points and their
 *           re-assignment to the new facets (polytope/quickhull.py:273-283, :311-336), the initial
 *           assignment (:224-245) and Facet.get_furthest (:87-102).
 * A plp_hull holds the N points ([N][d] rows, uploaded once), one int32 owner per point (facet id;
 * -1 = not outside any facet) and the point's distance to its owner.  Facet ids are handed out by
 * the session in creation order; id 0 is the virtual facet that owns every point at creation.
 *
 * plp_hull_reassign: flag the n_dead facets dead_ids[] (the visible set) dead; every point they own
 *   goes to the FIRST of the n_new facets normals[n_new][d], offsets[n_new] (creation order) with
 *   n.p - offset > abs_tol, else to -1.  The new facets get ids *new_id0 .. *new_id0 + n_new - 1.
 *   Out (per new facet): count[] points received, argmax[] the point furthest from it (-1 if none;
 *   lowest point index among exactly equal distances), maxd[] that distance (0 if none).
 * plp_hull_drop: owner[idx[i]] = -1 (start-simplex members, the apex just taken from its facet).  The
 *   indices are copied before the call returns; the update itself is enqueued and ordered before the
 *   session's next call (reassign and read synchronise).
 * A call moves its small arrays through one pinned block: one H2D copy, the kernels, one D2H copy.
 * plp_hull_read: copy owner[N] / dist[N] to the host (either may be NULL).
 * The "_dev" form is stateless: device pointers X, owner, dist, dead[new_id0] (uint8 per facet id),
 * normals, offsets, argmax, maxd, count; it enqueues on `stream` and returns.
 */
typedef struct plp_hull plp_hull;
int plp_hull_create(plp_ctx *ctx, int64_t N, int d, const double *X, plp_hull **out);
int plp_hull_destroy(plp_hull *h);
int plp_hull_drop(plp_hull *h, int64_t n, const int64_t *idx);
int plp_hull_reassign(plp_hull *h, int n_dead, const int32_t *dead_ids, int n_new, const double *normals,
                      const double *offsets, double abs_tol, int32_t *new_id0, int64_t *argmax,
                      double *maxd, int64_t *count);
int plp_hull_read(plp_hull *h, int32_t *owner, double *dist);
int plp_hull_reassign_dev(plp_ctx *ctx, void *stream, int64_t N, int d, const double *X, int32_t *owner,
                          double *dist, const uint8_t *dead, int new_id0, int n_new, const double *normals,
                          const double *offsets, double abs_tol, int64_t *argmax, double *maxd,
                          int64_t *count);

/*
 * Adjacency of all pairs of n cells (one polytope each): adj[i][j] = 1 iff the two cells, both
 * inflated by abs_tol, have an intersection with Chebyshev radius > abs_tol/10.
 * Replaces: the O(n^2) loop of find_adjacent_regions (polytope/prop2partition.py:46-63) over
 *           is_adjacent(a, b, overlap=True) (polytope/polytope.py:1843-1866): one Chebyshev LP per
 *           pair on the stacked rows [A_i; A_j], [b_i + abs_tol; b_j + abs_tol].
 * A[n][m_max][d], b[n][m_max], m[n] or NULL; 2*m_max <= 64, d <= 16 (d >= 9: one pair per wavefront).  Out: adj[n][n] (symmetric,
 * ones on the diagonal).  The n(n-1)/2 pair LPs are formed on the device from the resident cells.
 */
int plp_adjacent_pairs(plp_ctx *ctx, int n, int m_max, int d, const double *A, const double *b,
                       const int32_t *m, double abs_tol, uint8_t *adj);
int plp_adjacent_pairs_dev(plp_ctx *ctx, void *stream, int n, int m_max, int d, const double *A,
                           const double *b, const int32_t *m, double abs_tol, uint8_t *adj);
/*
 * Overlap of all pairs of n cells: out[i][j] = 1 iff the stack of cells i and j (nothing inflated) has
 * Chebyshev radius > abs_tol, i.e. is_fulldim(cell_i.intersect(cell_j)); ones on the diagonal.
 * Replaces: the O(n^2) pair loop of Partition.are_disjoint (polytope/prop2partition.py:123-192).
 */
int plp_overlap_pairs(plp_ctx *ctx, int n, int m_max, int d, const double *A, const double *b,
                      const int32_t *m, double abs_tol, uint8_t *out);
int plp_overlap_pairs_dev(plp_ctx *ctx, void *stream, int n, int m_max, int d, const double *A,
                          const double *b, const int32_t *m, double abs_tol, uint8_t *out);
/*
 * Cross pairs of TWO lists of cells held in one table (the first n1 cells, then n2 cells; A[n1+n2][m_max][d] ...):
 * out[a * n2 + c] = 1 iff the stack [cell a of the first list; cell c of the second] has a Chebyshev radius > thresh
 * (status 0), else 0.  Replaces: the scan region_diff opens with, one Chebyshev LP per cell of the subtrahend
 * (polytope/polytope.py:2148-2158), repeated by mldivide for every member of the minuend (:1484-1496) and by
 * Partition.refines for every pair of elements (prop2partition.py:194-207) -- here all n1 * n2 LPs in one launch, the
 * stacked rows formed on the device from the resident table.  2 * m_max <= 64, d <= 16.
 */
int plp_overlap_cross(plp_ctx *ctx, int n1, int n2, int m_max, int d, const double *A, const double *b,
                      const int32_t *m, double thresh, uint8_t *out);
int plp_overlap_cross_dev(plp_ctx *ctx, void *stream, int n1, int n2, int m_max, int d, const double *A,
                          const double *b, const int32_t *m, double thresh, uint8_t *out);

/*
 * plp_adjacent_pairs for a slice of the pair space (one rank's shard when the O(n^2) loop is split across
 * GPUs): pairs pair_lo <= p < pair_hi in the order p = i (i - 1) / 2 + j, j < i; out[p - pair_lo].
 */
int plp_adjacent_pairs_range(plp_ctx *ctx, int n, int m_max, int d, const double *A, const double *b,
                             const int32_t *m, double abs_tol, int64_t pair_lo, int64_t pair_hi,
                             uint8_t *out);
int plp_adjacent_pairs_range_dev(plp_ctx *ctx, void *stream, int n, int m_max, int d, const double *A,
                                 const double *b, const int32_t *m, double abs_tol, int64_t pair_lo,
                                 int64_t pair_hi, uint8_t *out);

/*
 * Quickhull's main loop (polytope/quickhull.py:224-345) as native host code over a plp_hull session: facet graph,
 * visibility search, horizon, new facets and their links on the host, every pass over the points one plp_hull_reassign.
 * X0[N][d]: the points translated so that the start simplex' centroid is the origin (:188-192); simplex[d + 1]: the
 * indices of the start simplex (chosen by the caller, who owns the random number stream, :165-185).
 * lapack_dgesv: optional pointer to LAPACK's dgesv (Fortran calling convention) for the facet hyperplanes (:66-85);
 * with the one numpy.linalg.solve runs, the rows come out bit-identical to the reference's; NULL = built-in LU.
 * Result: the hull's facets in the reference's order: normals[n][d] (unit, outward), offsets[n] (translated
 * coordinates: n.x = offset), verts[n][d] point indices.  PLP_EINVAL + plp_quickhull_last_error() for a singular
 * hyperplane system / degenerate neighbouring facets (where the reference raises).
 */
typedef struct plp_qh_result plp_qh_result;
int plp_quickhull_run(plp_ctx *ctx, int64_t N, int d, const double *X0, const int64_t *simplex, double abs_tol,
                      void *lapack_dgesv, plp_qh_result **out);
int plp_qh_result_sizes(const plp_qh_result *r, int64_t *n_facets, int64_t *iterations, int64_t *facets_made);
int plp_qh_result_copy(const plp_qh_result *r, double *normals, double *offsets, int64_t *verts);
int plp_qh_result_free(plp_qh_result *r);
const char *plp_quickhull_last_error(void);

/*
 * The search of region_diff: poly minus the union of N cells (polytope/polytope.py:2201-2281), run by the library.
 * The caller has done the reference's preparation (:2144-2199): cells sorted by the Chebyshev radius of their stack
 * with poly, mi[j] >= 1 new constraints per cell, and the table  A[m + 2M][d], b[m + 2M]  (M = sum mi) = poly's m rows,
 * the cells' new rows in that order, and the negations of those M rows -- each row already scaled to unit length as
 * Polytope.__init__ does (:130-138), because every LP of the search is the Chebyshev ball (:1283-1288) of
 * Polytope(A[rows], b[rows]) for some row list.  The table is uploaded once; the library walks the reference's search
 * (same visiting order, same tests "R > abs_tol", including its index arithmetic on INDICES / counter) and solves the
 * LPs in batches that are described by row-index lists only and gathered on the device; each batch holds what the
 * search needs now plus what it will need next if the current node is not empty (its scan and the first child of
 * every cell), so there is one launch and one synchronisation per visited node instead of one per LP.
 * Result: the pieces in the reference's order, as row lists: kind[k] = 0 -> Polytope(A[rows], b[rows]) as is (:2229),
 * 1 -> reduce() of it (:2276).  PLP_EINVAL with "row index out of range" where the reference raises IndexError.
 */
typedef struct plp_rdiff_result plp_rdiff_result;
int plp_region_diff_search(plp_ctx *ctx, int d, int m, int N, const int32_t *mi, const double *A, const double *b,
                           double abs_tol, plp_rdiff_result **out);
int plp_rdiff_result_sizes(const plp_rdiff_result *r, int64_t *n_leaves, int64_t *n_rows, int64_t *n_lps,
                           int64_t *n_batches);
/* kind[n_leaves], off[n_leaves + 1], rows[n_rows]: piece k holds rows[off[k] .. off[k+1]) */
int plp_rdiff_result_copy(const plp_rdiff_result *r, int32_t *kind, int32_t *off, int32_t *rows);
int plp_rdiff_result_free(plp_rdiff_result *r);

/* cross-lane primitive self-test (group size 8/16/32/64); host out_d[128], out_u[128] */
int plp_selftest(plp_ctx *ctx, int group_size, double *out_d, uint32_t *out_u);

#ifdef __cplusplus
}
#endif
#endif /* PLP_H */
