// plp_lp_wide.hip -- generic LPs  min c'x  s.t.  G x <= h  (solvers.lpsolve, polytope/solvers.py:76-106, :149-158), one LP
// per WAVEFRONT, one dictionary row per lane, wave-uniform pivot column and row: the engine of plp_wide.hpp with what a
// generic LP needs on top of a Chebyshev LP --
//   * phase 1 (origin infeasible): the artificial variable t (id -1) in the last column, -1 on every active row, forced
//     first pivot "t enters on the smallest right-hand side", Dantzig on the cost row e_t while the real objective is
//     CARRIED through the pivots as a second cost row (cost2 / negz2 in the wavefront's LDS block);
//   * the hand-over: t basic above TOL_FEAS -> infeasible; t basic at ~0 -> one forced pivot on the largest element of
//     its row (or, if there is none, the row takes no further part); the column that holds t is dead from then on, the
//     carried row becomes the cost row, right-hand sides rounded below zero are clamped;
//   * phase 2.
// The steps, their order and the arithmetic of every dictionary entry are those of SimplexR::run_two_phase /
// Simplex::step (plp_simplex_r.hpp, plp_simplex.hpp) and of oracle/plp_oracle.c: plpo_lp_solve; Bland's rule after
// BLAND_AFTER degenerate pivots runs inside the loop.  ~140 executed VALU instructions per pivot against ~500 of the
// one-row-per-lane lane-group kernel (lp_kernel<N>, plp_lp.hip), which at 33..64 rows holds one LP per wavefront as well.
#include <stdlib.h>

#include "plp_kernels.hpp"
#include "plp_wave.hpp"
#include "plp_wide.hpp"

namespace plp {

using namespace wide;

namespace {

template <int NC>
struct WideSharedC {
    double cost[NC + 1];
    double cost2[NC + 1];  // the carried objective (phase 1)
    double rho[NC + 1];    // rho[NC] = scaled right-hand side of the pivot row
    int cv[NC + 1];        // (id + 1) << 1 | negated ; t has id -1
};

struct GenScalars {  // wave-uniform state of one LP
    unsigned cfree, dead;
    int iters, maxit;
    double negz, negz2;
};

// One pivot on (column e, row = lane r): the pivot row scales itself in place and goes to LDS, every other row and both
// cost rows are updated from it.  ce / c2e: the reduced costs of column e as stored; flip: the (free) variable enters
// downwards, x := -x; a: my entry of the (sign-normalised) entering column, pinv its reciprocal.
template <int NC, class TV>
__device__ __forceinline__ void gen_apply(const int lane, TV& Tv, double& T16, double& beta, int& rowvar, int& rowneg,
                                          bool& rowact, WideSharedC<NC>& sh, GenScalars& S, double& cc, double& cc2,
                                          const bool carry, const int e,
                                          const bool flip, const double ce, const double c2e, const double a, const double pinv,
                                          const int r, const bool clamp) {
    const double p = uniform_lane(pinv, r);
    const int vin = sh.cv[e];
    const bool efree = (S.cfree >> e) & 1u;
    const bool is_r = lane == r;
    if (is_r) {
#pragma unroll
        for (int j = 0; j < NC; ++j) { const double v = ROW_GET(j) * pinv; ROW_SET(j, v); sh.rho[j] = v; }
        beta = beta * pinv;
        sh.rho[NC] = beta;
        sh.cv[e] = ((rowvar + 1) << 1) | rowneg;
        rowvar = (vin >> 1) - 1;
        rowneg = (vin & 1) ^ (flip ? 1 : 0);
        rowact = !efree;  // a free variable never leaves again
    }
    wave_sync();
    const double f = is_r ? 0.0 : a;
    const double rb = sh.rho[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) ROW_SET(j, fma(-f, sh.rho[j], ROW_GET(j)));
    row_put<NC>(Tv, T16, e, is_r ? pinv : -(f * p));
    beta = fma(-f, rb, beta);
    const double fc = flip ? -ce : ce;
    const double fc2 = flip ? -c2e : c2e;
    S.negz = fma(-fc, rb, S.negz);
    if (carry) S.negz2 = fma(-fc2, rb, S.negz2);
#if PLP_WIDE_CREG
    {   // both cost rows stay in registers (lane j: column j; 0 beyond the columns), see plp_wide.hpp
        const double rj = sh.rho[lane < NC ? lane : NC];
        const double n1 = (lane == e) ? -(fc * p) : fma(-fc, rj, cc);
        cc = lane < NC ? n1 : 0.0;
        if (carry) {
            const double n2 = (lane == e) ? -(fc2 * p) : fma(-fc2, rj, cc2);
            cc2 = lane < NC ? n2 : 0.0;
        }
    }
#else
    if (lane < NC) {
        const double rj = sh.rho[lane];
        sh.cost[lane] = (lane == e) ? -(fc * p) : fma(-fc, rj, sh.cost[lane]);
        if (carry) sh.cost2[lane] = (lane == e) ? -(fc2 * p) : fma(-fc2, rj, sh.cost2[lane]);
    }
#endif
    S.cfree &= ~(1u << e);
    S.iters += 1;
    if (clamp & rowact & (beta < 0.0)) beta = 0.0;  // rounding of a forced pivot
    wave_sync();
}

// Dantzig / Bland loop from a primal-feasible dictionary; `forced`: the first pivot is "column NC-1 enters, the active row
// with the smallest q0 leaves" (phase 1's start).  Returns the status.
template <int NC, class TV>
__device__ __forceinline__ int gen_run(const int lane, TV& Tv, double& T16, double& beta, int& rowvar, int& rowneg,
                                       bool& rowact, WideSharedC<NC>& sh, GenScalars& S, double& cc, double& cc2,
                                       const bool carry, bool forced_first, const double q0) {
    int ndeg = 0;
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    int status = -1;
#if PLP_WIDE_PEEL
    auto pivot = [&](auto forced_c) __attribute__((always_inline)) -> bool {
        constexpr bool forced = decltype(forced_c)::value;   // (the forced first pivot is an instance of its own)
#else
    bool forced = forced_first;
    auto pivot = [&]() __attribute__((always_inline)) -> bool {
#endif
        const bool bland = ndeg >= BLAND_AFTER;
        int e;
        double ce, c2e = 0.0;
        bool flip = false;
        if (forced) {
            e = NC - 1;
#if PLP_WIDE_CREG
            ce = uniform_lane(cc, NC - 1);
            if (carry) c2e = uniform_lane(cc2, NC - 1);
#else
            ce = sh.cost[NC - 1];
            if (carry) c2e = sh.cost2[NC - 1];
#endif
        } else {
#if PLP_WIDE_CREG
            const double c = cc;
#else
            const double c = lane < NC ? sh.cost[lane] : 0.0;
#endif
            const bool alive = (lane < NC) & (((S.dead >> (lane & 31)) & 1u) == 0u);
            const bool elig = alive & (fabs(c) > TOL_D) & ((((S.cfree >> (lane & 31)) & 1u) != 0u) | (c < 0.0));
            const uint64_t eb = __ballot(elig);
            if (eb == 0) { status = ST_OPT; return false; }
            if (S.iters >= S.maxit) { status = ST_ITER; return false; }
            if (!bland) {
                const unsigned kh = elig ? ((unsigned)__double2hiint(c) & 0x7fffffffu) : 0u;
                const unsigned mh = low_max_u32<NC>(kh);
                uint64_t top = __ballot(elig & (kh == mh));
                if (top & (top - 1ull)) {
                    const unsigned kl = (elig & (kh == mh)) ? (unsigned)__double2loint(c) : 0u;
                    const unsigned ml = low_max_u32<NC>(kl);
                    top = __ballot(elig & (kh == mh) & (kl == ml));
                }
                e = __ffsll((long long)top) - 1;
            } else {
                const int id = elig ? sh.cv[lane] : 0x7fffffff;
                const int idmin = wave_min_i32(id);
                e = __ffsll((long long)__ballot(elig & (id == idmin))) - 1;
            }
            e = __builtin_amdgcn_readfirstlane(e);
            ce = uniform_lane(c, e);
#if PLP_WIDE_CREG
            if (carry) c2e = uniform_lane(cc2, e);
#else
            if (carry) c2e = uniform_lane(lane < NC ? sh.cost2[lane] : 0.0, e);
#endif
            flip = ce > 0.0;
        }
        e = __builtin_amdgcn_readfirstlane(e);
        double a = row_at<NC>(Tv, T16, e);
        a = flip ? -a : a;
        const double pinv = rcpn(a);
        bool erow;
        double q;
        if (forced) { erow = rowact; q = q0; }
        else { erow = rowact & (a > TOL_PIV); q = (beta > 0.0 ? beta : 0.0) * pinv; }
        q = erow ? q : pinf;
        const int qh = __double2hiint(q), ql = __double2loint(q);
        const int sm = qh >> 31;
        const unsigned kh = (unsigned)(qh ^ (sm | (int)0x80000000));
        const unsigned kl = (unsigned)(ql ^ sm);
        const unsigned mh = wave_min_u32(kh);
        const uint64_t hib = __ballot(kh == mh);
        unsigned ml;
        if (hib & (hib - 1ull)) ml = wave_min_u32((kh == mh) ? kl : 0xffffffffu);
        else ml = (unsigned)__builtin_amdgcn_readlane((int)kl, __ffsll((long long)hib) - 1);
        if (mh >= 0xfff00000u) { status = ((mh == 0xfff00000u) & (ml == 0u)) ? ST_UNBND : ST_NUM; return false; }
        const bool tie = erow & (kh == mh) & (kl == ml);
        const int mhs = (int)(mh ^ 0x80000000u);
        const double qmin = __hiloint2double(mhs >= 0 ? mhs : (int)~mh, mhs >= 0 ? (int)ml : (int)~ml);
        int r;
        if (bland & !forced) {
            const int id = tie ? rowvar + 1 : 0x7fffffff;
            const int idmin = wave_min_i32(id);
            r = __ffsll((long long)__ballot(tie & (id == idmin))) - 1;
        } else {
            r = __ffsll((long long)__ballot(tie)) - 1;
        }
        r = __builtin_amdgcn_readfirstlane(r);
        if (!forced) ndeg = (qmin <= DEGEN_EPS) ? ndeg + 1 : 0;
        gen_apply<NC>(lane, Tv, T16, beta, rowvar, rowneg, rowact, sh, S, cc, cc2, carry, e, flip, ce, c2e, a, pinv, r, forced);
#if !PLP_WIDE_PEEL
        forced = false;
#endif
        return true;
    };
#if PLP_WIDE_PEEL
    {
        bool go = true;
        if (forced_first) go = pivot(std::integral_constant<bool, true>{});
        while (go) go = pivot(std::integral_constant<bool, false>{});
    }
#else
    while (pivot() && pivot()) {}
#endif
    return status;
}

}  // namespace

template <int N>
__global__ __launch_bounds__(64, 4) void lp_w_kernel(long long B, int m_max, const double* __restrict__ c,
                                                  const double* __restrict__ G, const double* __restrict__ h,
                                                  const int* __restrict__ mrows, double* __restrict__ x,
                                                  double* __restrict__ fun, int* __restrict__ status,
                                                  int* __restrict__ iters) {
    constexpr int NC = N + 1;  // + the phase-1 artificial
    __shared__ WideSharedC<NC> sh;
    const int lane = threadIdx.x;
    const long long lp = blockIdx.x;
    if (lp >= B) return;
    const int m = mrows ? mrows[lp] : m_max;
    const bool has = lane < m;
    typename RowVec<NC>::type Tv = (typename RowVec<NC>::type)(0.0);
    double T16 = 0.0;
    bool finite = true, zero = true;
    const double cj = lane < N ? c[lp * N + (lane < N ? lane : 0)] : 0.0;  // lane j looks after column j
    finite = finite & isfinite(cj);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double v = has ? G[(lp * m_max + lane) * N + j] : 0.0;
        ROW_SET(j, v);
        zero = zero & (v == 0.0);
        finite = finite & isfinite(v);
    }
    const double hi = has ? h[lp * m_max + lane] : 0.0;
    finite = finite & isfinite(hi);
    bool rowact = has & !zero;
    double beta = (has & zero) ? 0.0 : hi;
    int rowvar = N + lane, rowneg = 0;
    const bool infeasible0 = __ballot(has & zero & (hi < -TOL_FEAS)) != 0;  // 0 <= h_i < 0
    const bool bad = (__ballot(!finite) != 0) | (m > 64);
    const bool need_p1 = __ballot(rowact & (hi < 0.0)) != 0;
    ROW_SET(N, (need_p1 & rowact) ? -1.0 : 0.0);
    GenScalars S;
    S.cfree = (1u << N) - 1u;
    S.dead = need_p1 ? 0u : (1u << N);
    S.iters = 0;
    S.maxit = 50 * (m + N) + 100;
    S.negz = 0.0;
    S.negz2 = 0.0;
    double cc = 0.0, cc2 = 0.0;  // (PLP_WIDE_CREG) my column's entries of the two cost rows
    if (lane <= NC) {
#if PLP_WIDE_CREG
        if (lane < NC) {
            cc = need_p1 ? (lane == N ? 1.0 : 0.0) : (lane < N ? cj : 0.0);
            cc2 = (need_p1 & (lane < N)) ? cj : 0.0;
        }
#else
        sh.cost[lane] = need_p1 ? (lane == N ? 1.0 : 0.0) : (lane < N ? cj : 0.0);
        sh.cost2[lane] = (need_p1 & (lane < N)) ? cj : 0.0;
#endif
        sh.cv[lane] = lane == N ? 0 : ((lane + 1) << 1);  // column N holds t (id -1)
    }
    __syncthreads();
    int st;
    if (bad) {
        st = ST_NUM;
    } else if (infeasible0) {
        st = ST_INFEAS;
    } else if (!need_p1) {
        st = gen_run<NC>(lane, Tv, T16, beta, rowvar, rowneg, rowact, sh, S, cc, cc2, false, false, 0.0);
    } else {
        const int s1 = gen_run<NC>(lane, Tv, T16, beta, rowvar, rowneg, rowact, sh, S, cc, cc2, true, true, beta);
        if (s1 != ST_OPT) {
            st = (s1 == ST_ITER) ? ST_ITER : ST_NUM;  // (the auxiliary problem is never unbounded)
        } else {
            // ---- where phase 1 ended
            const uint64_t tb = __ballot(rowvar == ID_T);
            const int rt = tb ? __ffsll((long long)tb) - 1 : 0;
            const double tval = uniform_lane(beta, rt);
            if (tb && tval > TOL_FEAS) {
                st = ST_INFEAS;
            } else {
                if (tb) {  // t basic at ~0: out of the basis on the largest element of its row (columns still alive)
                    int eo = -1;
                    double big = TOL_PIV;
#pragma unroll
                    for (int j = 0; j < NC; ++j) {
                        const double aj = fabs(ROW_GET(j));
                        const bool tk = (aj > big) & (((S.dead >> j) & 1u) == 0u);
                        big = tk ? aj : big;
                        eo = tk ? j : eo;
                    }
                    const int ed = __builtin_amdgcn_readlane(eo, rt);
                    if (ed < 0) {  // row "0 = t": redundant, it takes no further part
                        if (lane == rt) rowact = false;
                    } else {
#if PLP_WIDE_CREG
                        const double cc1_ = cc, cc2_ = cc2;
#else
                        const double cc1_ = lane < NC ? sh.cost[lane] : 0.0;
                        const double cc2_ = lane < NC ? sh.cost2[lane] : 0.0;
#endif
                        const double a = row_at<NC>(Tv, T16, ed);
                        gen_apply<NC>(lane, Tv, T16, beta, rowvar, rowneg, rowact, sh, S, cc, cc2, true, ed, false,
                                      uniform_lane(cc1_, ed), uniform_lane(cc2_, ed), a, rcpn(a), rt, true);
                    }
                }
                // the column that now holds t is dropped; the carried cost row becomes active
                const uint64_t tcol = __ballot((lane < NC) && (sh.cv[lane < NC ? lane : 0] >> 1) == 0);
                S.dead |= (unsigned)tcol;
#if PLP_WIDE_CREG
                cc = cc2;
#else
                if (lane < NC) sh.cost[lane] = sh.cost2[lane];
#endif
                S.negz = S.negz2;
                if (rowact & (beta < 0.0)) beta = 0.0;
                __syncthreads();
                st = gen_run<NC>(lane, Tv, T16, beta, rowvar, rowneg, rowact, sh, S, cc, cc2, false, false, 0.0);
            }
        }
    }
    // ---- x: variable j sits in the row whose basic id is j (0 when nonbasic); fun = c.x by the oracle's FMA chain
    const bool ok = st == ST_OPT;
    const double mine = rowneg ? -beta : beta;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    double f = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const uint64_t ob = __ballot(rowvar == j);
        const double xj = ob ? uniform_lane(mine, __ffsll((long long)ob) - 1) : 0.0;
        f = fma(uniform_lane(cj, j), xj, f);
        if (lane == 0) x[lp * N + j] = ok ? xj : qnan;
    }
    if (lane == 0) {
        fun[lp] = ok ? f : qnan;
        status[lp] = st;
        if (iters) iters[lp] = S.iters;
    }
}

template <int N>
static int launch_lp_w_n(long long B, int m_max, const double* c, const double* G, const double* h, const int* mrows,
                         double* x, double* fun, int* status, int* iters, hipStream_t st) {
    if (B > 2147483647ll) return 1;
    hipLaunchKernelGGL((lp_w_kernel<N>), dim3((unsigned)(B < 1 ? 1 : B)), dim3(64), 0, st, B, m_max, c, G, h, mrows, x, fun,
                       status, iters);
    return 0;
}

#define PLP_CASE_LW(K) case K: return launch_lp_w_n<K>(B, m_max, c, G, h, mrows, x, fun, status, iters, st);

// one LP per wavefront, n = 5..16, m_max <= 64; returns 1 when it does not apply
int launch_lp_w(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
                double* x, double* fun, int* status, int* iters, hipStream_t st) {
    if (m_max < 1 || m_max > 64) return 1;
    switch (n) {
        PLP_CASE_LW(5) PLP_CASE_LW(6) PLP_CASE_LW(7) PLP_CASE_LW(8) PLP_CASE_LW(9) PLP_CASE_LW(10)
        PLP_CASE_LW(11) PLP_CASE_LW(12) PLP_CASE_LW(13) PLP_CASE_LW(14) PLP_CASE_LW(15) PLP_CASE_LW(16)
        default: return 1;
    }
}

}  // namespace plp
