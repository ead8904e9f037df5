// plp_reduce_tpl.hip -- fused reduce() for SMALL polytopes (rows <= 16, d <= 3): one polytope
// per LANE, the whole dictionary of the current LP in that lane's VGPRs (gfx950).
//
// Same reference behaviour as plp_reduce.hip (polytope/polytope.py:1053-1163) and the same
// pivot rules as plp_simplex.hpp; what changes is the mapping.  reduce_kernel gives a 16-row LP
// a 16-lane group, so one wavefront instruction advances 4 LPs and every pivot needs cross-lane
// reductions and broadcasts; measured on MI355X that kernel is VALU-issue bound (SQ_ACTIVE_INST_VALU
// ~96 %).  Here a 64-lane wavefront carries 64 polytopes, each lane walks through the LP sequence
// of its own polytope (F1, 2d x F3, one F2 per surviving row) at its own pace inside one flat
// loop -- "if my LP is finished: record it and set up the next one; do one pivot" -- so there is
// no cross-lane traffic at all and no lock-step between LPs of different length.  Row and column
// indices that differ per lane are handled by compile-time-unrolled scans with v_cndmask / EXEC
// masking (a 16 x NC tableau is 16*(NC+1) doubles = 128..160 VGPRs).
//
// LDS holds the tile's input rows transposed to [element][lane] (stride 65 doubles, conflict-free
// both for the coalesced staging writes and for the per-lane reads), 40 KB per 64 polytopes.
#include "plp_kernels.hpp"
#include "plp_common.hpp"

namespace plp {

constexpr int TPL_M = 16;     // rows per polytope handled by this kernel
constexpr int TPL_LD = 65;    // LDS leading dimension (doubles) per element row

// ------------------------------------------------------------------------------------------
// Dictionary of one LP in one lane:  basic_i = beta[i] - sum_j T[i][j] nb_j ;
//                                    (-zeta) = negz   - sum_j cost[j] nb_j
template <int NC>
struct TplLP {
    double T[TPL_M][NC];
    double beta[TPL_M];
    double cost[NC];
    double negz;
    unsigned act;     // bit i: row i takes part in ratio tests
    unsigned rid[4];  // basic variable id of row i, 8 bits each (id+1; 0 = artificial, unused here)
    unsigned cid;     // nonbasic variable id of column j, 8 bits each (NC <= 4)
    unsigned cfree;   // bit j: column j holds a free structural variable
    unsigned rneg;    // bit i: the basic free variable of row i is stored negated
    unsigned cneg;    // bit j: the free variable of column j is stored negated
    int n, ndeg, iters, maxit;

    __device__ __forceinline__ void reset(int n_, int mrows) {
        n = n_;
        cfree = (1u << n_) - 1u;
        rneg = 0u;
        cneg = 0u;
        ndeg = 0;
        iters = 0;
        maxit = 50 * (mrows + n_) + 100;
        negz = 0.0;
        cid = 0u;
#pragma unroll
        for (int j = 0; j < NC; ++j) cid |= (unsigned)(j + 1) << (8 * j);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            rid[w] = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) rid[w] |= (unsigned)(n_ + 4 * w + q + 1) << (8 * q);
        }
    }

    __device__ __forceinline__ unsigned row_id(int i) const { return (rid[i >> 2] >> (8 * (i & 3))) & 255u; }

    // One simplex iteration.  `go`: this lane has an LP in progress.  `forced` >= 0: INIT pivot --
    // that column enters and the row with the smallest SIGNED beta_i / T_i,forced leaves.
    // Returns -1 (pivoted or idle) or the final status of the LP.
    __device__ __forceinline__ int step(bool go, int forced) {
        // ------------------------------------------------ entering column (Dantzig / Bland)
        int e = -1;
        double best = 0.0;
        bool epos = false;
        const bool bland = ndeg >= BLAND_AFTER;
        unsigned bestid = 0xffffu;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const double c = cost[j];
            const double ac = fabs(c);
            const bool elig = (ac > TOL_D) & ((((cfree >> j) & 1u) != 0u) | (c < 0.0));
            const unsigned id = (cid >> (8 * j)) & 255u;
            const bool take = elig & (bland ? (id < bestid) : (ac > best));
            e = take ? j : e;
            best = take ? ac : best;
            bestid = take ? id : bestid;
            epos = take ? (c > 0.0) : epos;
        }
        const bool init = forced >= 0;
        if (init) { e = forced; epos = false; }
        if (!go) return -1;
        if (e < 0) return ST_OPT;
        if (iters >= maxit) return ST_ITER;
        const bool flip = epos;  // free variable entering downwards: x := -x
        // ------------------------------------------------ entering column values + ratio test
        // min beta_i / a_i over eligible rows, compared by cross-multiplication (a > 0):
        //   b_i / a_i < b_n / a_n  <=>  b_i a_n < b_n a_i ;  the first (lowest) row wins ties
        double ac[TPL_M];
        int r = -1;
        double bn = 0.0, an = 1.0;
#pragma unroll
        for (int i = 0; i < TPL_M; ++i) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < NC; ++j) a = (j == e) ? T[i][j] : a;
            a = flip ? -a : a;
            ac[i] = a;
            const bool elig = (((act >> i) & 1u) != 0u) & (a > (init ? 0.0 : TOL_PIV));
            const double bi = init ? beta[i] : (beta[i] > 0.0 ? beta[i] : 0.0);
            const bool better = elig & ((r < 0) | (bi * an < bn * a));
            r = better ? i : r;
            bn = better ? bi : bn;
            an = better ? a : an;
        }
        if (r < 0) return ST_UNBND;
        if (bland & !init) {  // Bland: among the rows that tie with the minimum, lowest basic id
            unsigned rbest = 0xffffu;
            int r2 = r;
#pragma unroll
            for (int i = 0; i < TPL_M; ++i) {
                const bool elig = (((act >> i) & 1u) != 0u) & (ac[i] > TOL_PIV);
                const double bi = beta[i] > 0.0 ? beta[i] : 0.0;
                const bool tie = elig & (bi * an == bn * ac[i]);
                const unsigned id = row_id(i);
                const bool take = tie & (id < rbest);
                rbest = take ? id : rbest;
                r2 = take ? i : r2;
            }
            r = r2;
        }
        if (!init) ndeg = (bn <= DEGEN_EPS * an) ? ndeg + 1 : 0;
        // ------------------------------------------------ pivot row
        double prow[NC], pb = 0.0, ar = 1.0;
#pragma unroll
        for (int j = 0; j < NC; ++j) prow[j] = 0.0;
#pragma unroll
        for (int i = 0; i < TPL_M; ++i) {
            if (i == r) {
#pragma unroll
                for (int j = 0; j < NC; ++j) prow[j] = T[i][j];
                pb = beta[i];
                ar = ac[i];
            }
        }
        const double x0 = __builtin_amdgcn_rcp(ar);
        const double x1 = fma(x0, fma(-ar, x0, 1.0), x0);
        const double p = fma(x1, fma(-ar, x1, 1.0), x1);
        double rho[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) rho[j] = prow[j] * p;
        const double rhob = pb * p;
        // ------------------------------------------------ update (the pivot row is fixed up below)
#pragma unroll
        for (int i = 0; i < TPL_M; ++i) {
            const double f = ac[i];
#pragma unroll
            for (int j = 0; j < NC; ++j) T[i][j] = fma(-f, rho[j], T[i][j]);
            beta[i] = fma(-f, rhob, beta[i]);
        }
        const double ce = init ? cost_at(e) : -best;  // sign-flipped reduced cost of the entering column
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const double c = fma(-ce, rho[j], cost[j]);
            cost[j] = (j == e) ? -(ce * p) : c;
        }
        negz = fma(-ce, rhob, negz);
        // entering column: T[i][e] = -a_i p   (EXEC-masked per column)
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            if (j == e) {
#pragma unroll
                for (int i = 0; i < TPL_M; ++i) T[i][j] = -(ac[i] * p);
            }
        }
        // pivot row: T[r][j] = rho_j, T[r][e] = p, beta[r] = rhob
#pragma unroll
        for (int j = 0; j < NC; ++j) rho[j] = (j == e) ? p : rho[j];
#pragma unroll
        for (int i = 0; i < TPL_M; ++i) {
            if (i == r) {
#pragma unroll
                for (int j = 0; j < NC; ++j) T[i][j] = rho[j];
                beta[i] = rhob;
            }
        }
        // ------------------------------------------------ bookkeeping: entering <-> leaving variable
        {
            const unsigned vin = (cid >> (8 * e)) & 255u;
            unsigned rword = 0u;  // rid[r >> 2] without a dynamically indexed register array
#pragma unroll
            for (int w = 0; w < 4; ++w) rword = ((r >> 2) == w) ? rid[w] : rword;
            const unsigned vout = (rword >> (8 * (r & 3))) & 255u;
            const unsigned inneg = ((cneg >> e) & 1u) ^ (flip ? 1u : 0u);
            const unsigned outneg = (rneg >> r) & 1u;
            const bool efree = ((cfree >> e) & 1u) != 0u;
            cid = (cid & ~(255u << (8 * e))) | (vout << (8 * e));
            const int sh = 8 * (r & 3);
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if ((r >> 2) == w) rid[w] = (rid[w] & ~(255u << sh)) | (vin << sh);
            cneg = (cneg & ~(1u << e)) | (outneg << e);
            rneg = (rneg & ~(1u << r)) | (inneg << r);
            cfree &= ~(1u << e);
            if (efree) act &= ~(1u << r);  // a free variable never leaves again
            iters += 1;
        }
        return -1;
    }

    __device__ __forceinline__ double cost_at(int e) const {
        double c = 0.0;
#pragma unroll
        for (int j = 0; j < NC; ++j) c = (j == e) ? cost[j] : c;
        return c;
    }
};

__device__ __forceinline__ double lds_at(const double* s, int elem, int lane) { return s[elem * TPL_LD + lane]; }

template <int D>
__global__ __launch_bounds__(64) void reduce_tpl_kernel(long long B, int m_max,
                                                        const double* __restrict__ Ag,
                                                        const double* __restrict__ bg,
                                                        const int* __restrict__ mrows, double abs_tol,
                                                        unsigned long long* __restrict__ keep_out,
                                                        int* __restrict__ flags_out,
                                                        double* __restrict__ r_out,
                                                        double* __restrict__ xc_out,
                                                        int* __restrict__ nlp_out) {
    constexpr int M = TPL_M;
    __shared__ double sA[M * D * TPL_LD];  // [row*D + k][lane]
    __shared__ double sb[M * TPL_LD];      // [row][lane]
    const int lane = threadIdx.x;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);

    for (long long tile = (long long)blockIdx.x * 64; tile < B; tile += (long long)gridDim.x * 64) {
        const int ntile = (B - tile) < 64 ? (int)(B - tile) : 64;
        __syncthreads();
        {
            const int rowsz = m_max * D;
            const double* src = Ag + tile * rowsz;
            for (int idx = lane; idx < ntile * rowsz; idx += 64) {
                const int p = idx / rowsz, rem = idx - p * rowsz;
                sA[rem * TPL_LD + p] = src[idx];
            }
            const double* srcb = bg + tile * m_max;
            for (int idx = lane; idx < ntile * m_max; idx += 64) {
                const int p = idx / m_max, row = idx - p * m_max;
                sb[row * TPL_LD + p] = srcb[idx];
            }
        }
        __syncthreads();
        const long long pg = tile + lane;
        const bool valid = lane < ntile;
        const int m = valid ? (mrows ? mrows[pg] : m_max) : 0;
        const unsigned rowmask = (m >= 32) ? 0xffffffffu : ((1u << m) - 1u);  // m <= 16

        // ---------------------------------------------------------------- F1 (polytope.py:1283-1288)
        double xc[D];
        double rr = 0.0;
        bool ball = false;
        unsigned live = 0u;
        {
            TplLP<D + 1> S;
            S.reset(D + 1, m);
            bool finite = true, infeasible0 = false;
            unsigned act = 0u;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const bool has = valid & (i < m);
                double nrm2 = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double a = has ? lds_at(sA, i * D + k, lane) : 0.0;
                    S.T[i][k] = a;
                    nrm2 = nrm2 + a * a;
                    finite = finite & isfinite(a);
                }
                const double bi = has ? lds_at(sb, i, lane) : 0.0;
                finite = finite & isfinite(bi);
                const double nrm = sqrt(nrm2);
                const bool zero = !(nrm > 0.0);
                S.T[i][D] = nrm;
                S.beta[i] = bi;
                infeasible0 = infeasible0 | (has & zero & (bi < -TOL_FEAS));
                act |= (has & !zero) ? (1u << i) : 0u;
            }
            S.act = act;
#pragma unroll
            for (int j = 0; j < D; ++j) S.cost[j] = 0.0;
            S.cost[D] = -1.0;
            int status = -1;
            if (!valid | !finite) status = ST_NUM;
            else if (infeasible0) status = ST_INFEAS;
            bool first = true;
            while (__any(status < 0)) {
                const int st = S.step(status < 0, first ? D : -1);
                if (first) {
#pragma unroll
                    for (int i = 0; i < M; ++i)
                        if (((S.act >> i) & 1u) && S.beta[i] < 0.0) S.beta[i] = 0.0;  // rounding of the forced pivot
                }
                first = false;
                if (status < 0 && st >= 0) status = st;
            }
            // x_j = +-beta of the row whose basic id is j (0 when nonbasic)
            double xs[D + 1];
#pragma unroll
            for (int j = 0; j <= D; ++j) xs[j] = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const unsigned id = S.row_id(i) - 1u;
                const double v = ((S.rneg >> i) & 1u) ? -S.beta[i] : S.beta[i];
#pragma unroll
                for (int j = 0; j <= D; ++j) xs[j] = (id == (unsigned)j) ? v : xs[j];
            }
#pragma unroll
            for (int j = 0; j < D; ++j) xc[j] = xs[j];
            rr = xs[D];
            ball = (status == ST_OPT) & (rr >= 0.0);  // cheby_ball: status 0 and r >= 0 (:1289-1293)
        }
        const bool fulldim = ball & (rr > abs_tol);
        // ---------------------------------------------------------------- dedupe (:1094-1110)
        {
            double nr[M][D], bnorm[M];
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double nrm2 = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) { nr[i][k] = lds_at(sA, i * D + k, lane); nrm2 = nrm2 + nr[i][k] * nr[i][k]; }
                const double an = 1.0 / sqrt(nrm2);
#pragma unroll
                for (int k = 0; k < D; ++k) nr[i][k] = nr[i][k] * an;
                bnorm[i] = lds_at(sb, i, lane) * an;
            }
            unsigned removed = 0u;
#pragma unroll
            for (int i = 0; i < M; ++i) {
#pragma unroll
                for (int j = i + 1; j < M; ++j) {
                    double dot = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) dot = dot + nr[i][k] * nr[j][k];
                    const bool par = (((rowmask >> i) & (rowmask >> j) & 1u) != 0u) & (dot > 1.0 - abs_tol);
                    const bool dropj = bnorm[i] < bnorm[j];
                    removed |= par ? (dropj ? (1u << j) : (1u << i)) : 0u;
                }
            }
            live = valid ? (rowmask & ~removed) : 0u;
        }
        int flags = fulldim ? 0 : RF_EMPTY;
        int nlp = 1;
        unsigned keep = 0u;
        int stage = 0;  // 0 done, 1 bounding box pending, 2 redundancy LPs pending
        if (fulldim) {
            const int neq = __popc(live);
            if (neq <= D + 1) { flags = RF_EARLY; keep = live; }
            else stage = (neq > 3 * D) ? 1 : 2;
        }
        // ---------------------------------------------------------------- F3 / F2: one flat loop
        {
            TplLP<D> S;
            S.reset(D, 0);
            S.act = 0u;
#pragma unroll
            for (int i = 0; i < M; ++i) { S.beta[i] = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) S.T[i][j] = 0.0; }
#pragma unroll
            for (int j = 0; j < D; ++j) S.cost[j] = 0.0;
            bool need_next = stage != 0;   // set up my next LP at the top of the next iteration
            bool running = false;          // an LP is in progress
            int lpstat = -1;
            int item = -1;                 // F3: 0..2D-1 ; F2: row index k
            unsigned todo = 0u;            // F2 rows still to do
            double lbv[D], ubv[D];
            double cxc = 0.0;
            bool lpfail = false;
#pragma unroll
            for (int k = 0; k < D; ++k) { lbv[k] = 0.0; ubv[k] = 0.0; }
            if (stage == 2) { todo = live; nlp += __popc(live); }
            while (__any(stage != 0)) {
                // ---------------- A: record the LP that just finished, pick and set up the next one
                if (need_next) {
                    if (running) {
                        running = false;
                        if (stage == 1) {
                            // zeta = c.x' = -negz ; x_k = xc_k + x'_k ; lower: c = +e_k, upper: c = -e_k
                            const int k = item >> 1;
                            const bool up = item & 1;
                            double xck = 0.0;
#pragma unroll
                            for (int kk = 0; kk < D; ++kk) xck = (kk == k) ? xc[kk] : xck;
                            double val;
                            if (lpstat == ST_OPT) val = up ? (xck + S.negz) : (xck - S.negz);
                            else if (lpstat == ST_UNBND) val = up ? pinf : -pinf;
                            else { val = qnan; lpfail = true; }
#pragma unroll
                            for (int kk = 0; kk < D; ++kk) {
                                if (kk == k) { if (up) ubv[kk] = val; else lbv[kk] = val; }
                            }
                            if (item == 2 * D - 1) {
                                // prefilter (:1131-1134), sums in k order
                                unsigned out = 0u;
#pragma unroll
                                for (int i = 0; i < M; ++i) {
                                    double s1 = 0.0, s2 = 0.0;
#pragma unroll
                                    for (int kk = 0; kk < D; ++kk) {
                                        const double a = lds_at(sA, i * D + kk, lane);
                                        const double pa = (a > 0.0 ? 1.0 : 0.0) * a;
                                        s1 = s1 + pa * (ubv[kk] - lbv[kk]);
                                        s2 = s2 + a * lbv[kk];
                                    }
                                    out |= ((s1 - (lds_at(sb, i, lane) - s2)) < -1e-4) ? (1u << i) : 0u;
                                }
                                live &= ~out;
                                nlp += 2 * D;
                                if (lpfail) flags |= RF_LPFAIL;
                                if (__popc(live) <= D + 1) { flags |= RF_EARLY; keep = live; stage = 0; }
                                else { stage = 2; todo = live; nlp += __popc(live); item = -1; }
                            }
                        } else if (stage == 2) {
                            const int k = item;
                            const double fun = cxc - S.negz;  // c.xc + zeta
                            const double bk = sb[k * TPL_LD + lane];
                            const double hk = (bk + 0.1) - 0.1;
                            const double obj = -fun - hk;     // (:1156)
                            if ((lpstat == ST_OPT && obj > abs_tol) || lpstat == ST_UNBND) keep |= 1u << k;
                            if (todo == 0u) { flags |= RF_MINREP; stage = 0; }
                        }
                    }
                    if (stage != 0) {
                        // ---- next LP of my polytope
                        if (stage == 1) item = item + 1;
                        else { item = __ffs((int)todo) - 1; todo &= todo - 1u; }
                        S.reset(D, __popc(live));
                        S.act = live;
                        int kcost = 0;
                        double csign = 0.0;
                        if (stage == 1) { kcost = item >> 1; csign = (item & 1) ? -1.0 : 1.0; }
                        cxc = 0.0;
#pragma unroll
                        for (int kk = 0; kk < D; ++kk) {
                            double c;
                            if (stage == 1) c = (kk == kcost) ? csign : 0.0;
                            else c = -sA[(item * D + kk) * TPL_LD + lane];  // f = -A[k,:]  (:1145)
                            S.cost[kk] = c;
                            cxc = fma(c, xc[kk], cxc);
                        }
#pragma unroll
                        for (int i = 0; i < M; ++i) {
                            double s = 0.0;
#pragma unroll
                            for (int kk = 0; kk < D; ++kk) {
                                const double a = lds_at(sA, i * D + kk, lane);
                                S.T[i][kk] = a;
                                s = fma(a, xc[kk], s);
                            }
                            // dictionary translated to the Chebyshev centre; F2: h[k] += 0.1 for this LP,
                            // rows k' < k carry the (+0.1, -0.1) round trip (:1149-1151)
                            const double b0 = lds_at(sb, i, lane);
                            const double bup = b0 + 0.1;
                            const double brt = bup - 0.1;
                            double be = b0;
                            if (stage == 2) be = (i < item) ? brt : ((i == item) ? bup : b0);
                            const double bsh = be - s;
                            S.beta[i] = bsh > 0.0 ? bsh : 0.0;
                        }
                        running = true;
                    }
                    need_next = false;
                }
                // ---------------- B: one simplex iteration of my current LP
                const int st = S.step(running, -1);
                if (running && st >= 0) { lpstat = st; need_next = true; }
            }
        }
        // ---------------------------------------------------------------- results
        if (valid) {
            keep_out[pg] = (unsigned long long)keep;
            flags_out[pg] = flags;
            nlp_out[pg] = nlp;
            r_out[pg] = ball ? rr : 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) xc_out[pg * D + k] = ball ? xc[k] : qnan;
        }
    }
}

template <int D>
static int launch_reduce_tpl_d(long long B, int m_max, const double* A, const double* b, const int* mrows,
                               double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc,
                               int* nlp, hipStream_t st) {
    long long blocks = (B + 63) / 64;
    if (blocks > (1ll << 20)) blocks = 1ll << 20;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(reduce_tpl_kernel<D>, dim3((unsigned)blocks), dim3(64), 0, st, B, m_max, A, b, mrows, abs_tol,
                       keep, flags, r, xc, nlp);
    return 0;
}

// returns 0 when the small-polytope kernel took the batch, 1 when the caller should use reduce_kernel
int launch_reduce_tpl(long long B, int m_max, int d, const double* A, const double* b, const int* mrows,
                      double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                      hipStream_t st) {
    if (m_max > TPL_M || m_max < 1 || d > 3 || d < 1) return 1;
    switch (d) {
        case 1: return launch_reduce_tpl_d<1>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
        case 2: return launch_reduce_tpl_d<2>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
        case 3: return launch_reduce_tpl_d<3>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
        default: return 1;
    }
}

}  // namespace plp
