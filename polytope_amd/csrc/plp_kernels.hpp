// plp_kernels.hpp -- host-side launchers of the HIP kernels (internal to libplp_hip.so).
// Every launcher returns 0, or 2 for an unsupported size; launches are asynchronous on `st`.
#pragma once
#include "plp_common.hpp"

namespace plp {

// lanes per LP group for polytopes with up to m_max rows (-1: unsupported)
static inline int group_size_for(int m_max) {
    if (m_max < 0 || m_max > MAX_M) return -1;
    if (m_max <= 8) return 8;
    if (m_max <= 16) return 16;
    if (m_max <= 32) return 32;
    return 64;
}

int launch_lp(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
              double* x, double* fun, int* status, int* iters, hipStream_t st);
int launch_lp_phase(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
                    double* x, double* fun, int* status, int* iters, hipStream_t st, int phase, int* more);

int launch_cheby(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* r,
                 double* xc, int* status, hipStream_t st);

// dictionary in LDS, one LP per wavefront, any row count that fits 160 KB (plp_lds.hip): the engine for m_max > 64
int launch_lp_lds(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
                  double* x, double* fun, int* status, int* iters, hipStream_t st);
int launch_cheby_lds(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* r,
                     double* xc, int* status, hipStream_t st);
size_t lds_lp_bytes(int m_max, int nc);  // 0: does not fit

// Chebyshev LPs on row subsets rows[off[p] .. off[p+1]) of one resident table (region_diff's search): LPs sel[0..nlp)
// of the batch; out[p] = radius as cheby_ball reads it (0 unless optimal with r >= 0).  plp_rdiff.hip / plp_lds.hip
int launch_cheby_gather_r(int d, long long n0, long long n1, long long n2, const int* off, const int* rows, const int* sel,
                          const double* A, const double* b, double* out, hipStream_t st);
// the LP server of a region_diff search (plp_rdiff.hip, d <= 4): one resident launch, batches through a host-mapped mailbox
int launch_rdiff_server(int d, int nwg, const unsigned long long* mail, const int* rec, void* out, unsigned long long* alive,
                        unsigned long long* dstate, const double* A, const double* b, unsigned long long last_word,
                        unsigned idle_polls, hipStream_t st);
void launch_rdiff_publish(long long n, const double* src, double* host_out, unsigned long long* host_flag,
                          unsigned long long seq, hipStream_t st);
// lists of at most 64 rows, d = 5..16: one LP per wavefront (plp_wide.hip); returns 1 when it does not apply
int launch_cheby_gather_w(int d, long long nlp, const int* off, const int* rows, const int* sel, const double* A,
                          const double* b, double* out, hipStream_t st);
int launch_cheby_gather_lds(int d, int m_cap, long long nlp, const int* off, const int* rows, const int* sel,
                            const double* A, const double* b, double* out, hipStream_t st);

// fused reduce() of polytopes with more than 64 rows: rows and dictionary in LDS, keep = ceil(m_max / 64) words per
// polytope (plp_lds.hip); returns 2 when a polytope does not fit the CU's LDS
int launch_reduce_lds(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double abs_tol,
                      unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st);

// one LP per wavefront, one row per lane, wave-uniform pivot column (plp_wide.hip): the engine for d >= 9
int launch_cheby_w(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* r,
                   double* xc, int* status, hipStream_t st);

// generic LPs, one per wavefront, both phases (n = 5..16, m_max <= 64, plp_lp_wide.hip); returns 1 when it does not apply
int launch_lp_w(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
                double* x, double* fun, int* status, int* iters, hipStream_t st);

// generic LPs, four rows per lane, origin-feasible ones only (n <= 8, plp_cheby_r.hip): the others get
// status ST_RETRY for the general kernel; returns 1 when it does not apply
int launch_lp_r(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
                double* x, double* fun, int* status, int* iters, hipStream_t st);

// four rows per lane (d <= 8, plp_cheby_r.hip); returns 1 when it does not apply
int launch_cheby_r(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* r,
                   double* xc, int* status, hipStream_t st);

// bounding boxes, Chebyshev LP + 2d LPs from its centre per polytope (d <= 8, plp_bbox_r.hip): status 0 = lb/ub
// valid, 1 = polytope left to the generic LPs; returns 1 when the kernel does not apply
// What the fused bounding-box kernels hand to the verifier (plp_verify.hip), per box LP (2d per polytope, lower_0, upper_0,
// lower_1, ...): `basis8` -- its final basis, d signed bytes (>= 0: an active row; -1 - j: the free variable x'_j left at the
// centre) -- with `centre` (d doubles per polytope), or `xfin` -- the point it ended on (d doubles; the one-LP-per-lane
// kernel, whose walk has no basis of d entries when it ends on a face).  `mode` (out): 0 nothing written, 1 bases, 2 points.
struct BoxHandover {
    signed char* basis8;
    double* centre;
    double* xfin;
    int mode;
};
int launch_bbox(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* lb,
                double* ub, int* status, hipStream_t st, BoxHandover* ho = nullptr);
// d = 9..16 (plp_bbox_lazy.hip); same contract
int launch_bbox_lazy(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* lb,
                     double* ub, int* status, hipStream_t st, BoxHandover* ho = nullptr);

// adjacency of all pairs of n cells (2*m_max <= 64, d <= 8) into the n x n matrix adj (compact == nullptr),
// or of the pairs p_lo <= p < p_hi, p = i (i - 1) / 2 + j, j < i, into compact[p - p_lo]
// a pair counts iff the Chebyshev radius of the two stacked cells, each b inflated by `inflate`, exceeds `thresh`
// cross_n1 > 0 (compact only): the table holds two lists, n1 cells then n - n1 cells, and pair p = a (n - n1) + c is cell a
// of the first list stacked on cell c of the second (p_lo <= p < p_hi within [0, n1 (n - n1)))
int launch_adjacent(int n, int m_max, int d, const double* A, const double* b, const int* mrows, double inflate,
                    double thresh, unsigned char* adj, long long p_lo, long long p_hi, unsigned char* compact,
                    hipStream_t st, int cross_n1 = 0);

// the same, one pair per wavefront (d = 5..16, plp_wide.hip); returns 1 when it does not apply
int launch_adjacent_w(int n, int m_max, int d, const double* A, const double* b, const int* mrows, double inflate,
                      double thresh, unsigned char* adj, long long p_lo, long long p_hi, unsigned char* compact,
                      hipStream_t st, int cross_n1 = 0);

// Device counter (or nullptr) the fused reduce kernels of the calling thread add their simplex-run count to: set by
// plp_reduce_batch_dev from the context around the launch (plp_reduce_counters reads it back)
extern thread_local unsigned long long* t_reduce_ctr;
// Device word (or nullptr) and the value the fast kernels raise it to when they hand a polytope to the general kernel
// (RF_RETRY): the second pass of launch_reduce leaves at once when the word does not hold the call's value
extern thread_local unsigned long long* t_reduce_retry;
extern thread_local unsigned long long t_reduce_epoch;

int launch_reduce(long long B, int m_max, int d, const double* A, const double* b, const int* mrows,
                  double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                  hipStream_t st);
int launch_reduce_phase(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double abs_tol,
                        unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st, int phase);

// four dictionary rows per lane (d <= 8); returns 1 when it does not apply
int launch_reduce_r(long long B, int m_max, int d, const double* A, const double* b, const int* mrows,
                    double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                    hipStream_t st);

// `scratch` (contains_scratch_bytes(P, m_max) bytes of device memory, or nullptr): the per-row thresholds of the
// comparison form (plp_points.hip); without it the subtraction stays in the kernel
int launch_contains(int P, int m_max, int d, const double* A, const double* b, const int* mrows, long long N,
                    const double* X, double abs_tol, int mode, unsigned char* out, void* scratch, hipStream_t st);
size_t contains_scratch_bytes(int P, int m_max);

int launch_assign(long long N, int d, const double* X, int F, const double* normals, const double* offsets,
                  double abs_tol, int* facet_of_point, double* dist, long long* argmax, double* maxd,
                  void* scratch, size_t scratch_bytes, hipStream_t st);
size_t assign_scratch_bytes(long long N, int F);

// quickhull outside sets on resident points (plp_hull.hip)
int launch_hull_reassign(long long N, int d, const double* X, int* owner, double* dist, const unsigned char* dead,
                         int new_id0, int n_new, const double* normals, const double* offsets, double abs_tol,
                         long long* argmax, double* maxd, long long* count, hipStream_t st);
void launch_hull_mark(int n, const int* ids, unsigned char* dead, hipStream_t st);
void launch_hull_drop(long long n, const long long* idx, int* owner, hipStream_t st);

int launch_selftest(int gs, double* out_d, unsigned* out_u, hipStream_t st);

}  // namespace plp
