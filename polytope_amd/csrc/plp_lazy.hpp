// plp_lazy.hpp -- the F2 / F3 LPs of a large polytope (d = 9..16, up to 64 rows) WITHOUT carrying the dictionary.
//
// From the Chebyshev centre these LPs end after two to four pivots (measured: 2.2 at (64,16), 3.0 at (64,12), 99.9 % of
// them within 16), yet every pivot of the dense engines rewrites all m x d entries of the dictionary and every LP
// starts by reloading them.  Of the dictionary after t pivots a pivot only needs ONE column (the entering one, m
// entries) and ONE row (the leaving one, d entries), and both follow from the polytope's own rows -- which stay in
// LDS, untouched -- and the t pivots so far:
//     T_s[i][j] = T_{s-1}[i][j] - u_s[i] * rho_s[j]        (i != r_s, j != e_s)
//     T_s[i][e_s] = -(u_s[i] * p_s),   T_s[r_s][j] = rho_s[j],   rho_s[e_s] = p_s
// with u_s = the entering column of pivot s (0 in the pivot row), rho_s = its scaled pivot row.  One polytope per
// wavefront: lane i keeps u_s[i] of ITS row in registers (two per pivot), lane j < d looks after column j (reduced
// cost and its entry of every rho_s, in registers too: what the other lanes need of rho_s is the one entry rho_s[e],
// a read-lane), e_s / r_s are wave-uniform.  No LDS traffic beyond the polytope's own rows.  A pivot costs O(m t + d t) instead of
// O(m d) and an LP starts with nothing to load but its cost vector.
// The recurrences are evaluated with the same operations in the same order as SimplexR::pivot_core applies them to the
// stored dictionary, so every number -- and therefore every pivot choice and every result -- is bit-identical to the
// one-row-per-lane instance of that engine (reduce_r_tile<D, 64, 1>), which the tests compare against.
// An LP that needs more than K pivots, or Bland's rule, ends with ST_RETRY: the polytope goes to the general kernel.
#pragma once
#include "plp_simplex_r.hpp"
#include "plp_wide.hpp"

namespace plp {
namespace lazy {

using wide::wave_max_u32;
using wide::wave_min_u32;
__device__ __forceinline__ double lane_value(double v, int lane) { return wide::uniform_lane(v, lane); }
// maximum over the lanes 0..15 of a u32 that is 0 in every other lane (the d <= 16 columns sit in the first DPP row)
__device__ __forceinline__ unsigned row0_max_u32(unsigned v) {
    PLP_W_DPP("v_max_u32_dpp", v, "quad_perm:[1,0,3,2]", "0xf");
    PLP_W_DPP("v_max_u32_dpp", v, "quad_perm:[2,3,0,1]", "0xf");
    PLP_W_DPP("v_max_u32_dpp", v, "row_half_mirror", "0xf");
    PLP_W_DPP("v_max_u32_dpp", v, "row_mirror", "0xf");
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0);
}

constexpr int K_STEPS = 32;  // pivots an LP may take here
#ifndef PLP_LAZY_KREG
#define PLP_LAZY_KREG 4
#endif
constexpr int K_REG = PLP_LAZY_KREG;     // ... of which the first K_REG keep u_s / rho_s in named registers: nine LPs in ten end
                             // within them; the later ones index private arrays (scratch memory: slow, rare)

// LDS behind the tile's arrays: only F1's block of the one-LP-per-wavefront engine (reduced costs, pivot row, column ids)
template <int D>
__host__ __device__ constexpr size_t lds_bytes() { return (sizeof(wide::WideShared<D + 1>) + 255) & ~(size_t)255; }

// Dantzig choice over the d reduced costs (lane j holds c_j): SimplexR::scan_enter's key -- |c| for a free column,
// -c otherwise -- largest key above TOL_D, lowest column on ties.  false: the dictionary is optimal.
template <int D>
__device__ __forceinline__ bool price(const int lane, const double c, const unsigned cfree, int& e, double& best, int& chi) {
    const int hi = __double2hiint(c);
    const bool fr = ((cfree >> (lane & 31)) & 1u) != 0u;
    const int khi = (hi ^ (int)0x80000000) & ~(fr ? (int)0x80000000 : 0);
    const double key = __hiloint2double(khi, __double2loint(c));
    const bool valid = (lane < D) & (key > TOL_D);
    if (__ballot(valid) == 0ull) return false;
    const unsigned kh = valid ? (unsigned)khi : 0u;  // key > 0: its bit pattern orders like an unsigned integer
    static_assert(D <= 16, "columns in the first DPP row");
    const unsigned mh = row0_max_u32(kh);
    uint64_t top = __ballot(valid & (kh == mh));
    if (top & (top - 1ull)) {  // several columns share the high word (rare): the low words decide
        const unsigned kl = (valid & (kh == mh)) ? (unsigned)__double2loint(c) : 0u;
        const unsigned ml = row0_max_u32(kl);
        top = __ballot(valid & (kh == mh) & (kl == ml));
    }
    e = __builtin_amdgcn_readfirstlane(__ffsll((long long)top) - 1);
    best = lane_value(key, e);
    chi = __builtin_amdgcn_readlane(hi, e);
    return true;
}

// min c.x' over { A x' <= beta } from x' = 0 (the rows of A in LDS at sA[i * D + j], row i = lane i; beta >= 0 in
// lane i; rowact: row i exists).  c: lane j < D holds c_j.
// Returns the status (ST_OPT / ST_UNBND / ST_NUM / ST_ITER / ST_RETRY); negz = -(optimal value) as in SimplexR.
// basis_out (optional): the final basis for the verifier (plp_verify.hip), D signed bytes -- >= 0 an active row, -1 - j the free
// variable x'_j still at zero; costs two read-lanes per pivot when asked for.
template <int D>
__device__ __forceinline__ int solve(const int lane, const int m_rows, const double* sA, double c, double beta, bool rowact,
                                     double& negz_out, signed char* __restrict__ basis_out = nullptr) {
    constexpr int K = K_STEPS;
    // (ur / qr / str are only ever indexed by unrolled loop counters: registers; ux / qx by t: private memory)
    double ur_[K_REG], qr_[K_REG];  // u_s of my row; my entry of rho_s (lanes j < D)
    int str_[K_REG];                // e_s | r_s << 8
#pragma unroll
    for (int s = 0; s < K_REG; ++s) { ur_[s] = 0.0; qr_[s] = 0.0; str_[s] = 0; }
    double ux[K - K_REG];
    double qx[K - K_REG];
    int steps = 0;  // lane s: e_s | r_s << 8 (steps beyond the first K_REG)
    unsigned cfree = (1u << D) - 1u;
    int t = 0, ndeg = 0;
    const int maxit = 50 * (__builtin_amdgcn_readfirstlane(m_rows) + D) + 100;  // (wave-uniform: keeps the loop scalar)
    double negz = 0.0;
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    int status;
    int e, chi;
    double best;
    int colvar = lane, rowvar = D + lane;  // (basis_out) lane j < D: the variable of column j; lane i: the basic variable of row i
    if (!price<D>(lane, c, cfree, e, best, chi)) {
        negz_out = negz;
        if (basis_out && lane < D) basis_out[lane] = (signed char)(-1 - lane);
        return ST_OPT;
    }
    for (;;) {
        if (ndeg >= BLAND_AFTER) { status = ST_RETRY; break; }  // (as SimplexR::run_fast)
        if (t >= maxit) { status = ST_ITER; break; }
        if (t >= K) { status = ST_RETRY; break; }
        const bool flip = chi >= 0;  // c > 0: the free variable enters downwards, x := -x
        const bool efree = ((cfree >> e) & 1u) != 0u;
        // ---- entering column of my row as of now
        double a = sA[lane * D + e];
#define PLP_LZ_COL(US_, QS_, ST_)                                                    \
        {                                                                            \
            const int st_ = (ST_);                                                   \
            const int e_s = st_ & 0xff, r_s = st_ >> 8;                              \
            const double pe = lane_value((QS_), e);                                  \
            const double us = (US_);                                                 \
            const double an = (e == e_s) ? -(us * pe) : fma(-us, pe, a);             \
            a = (lane == r_s) ? pe : an;                                             \
        }
#pragma unroll
        for (int s = 0; s < K_REG; ++s)
            if (t > s) PLP_LZ_COL(ur_[s], qr_[s], str_[s])
        for (int s = K_REG; s < t; ++s) PLP_LZ_COL(ux[s - K_REG], qx[s - K_REG], __builtin_amdgcn_readlane(steps, s))
        a = __hiloint2double(__double2hiint(a) ^ (flip ? (int)0x80000000 : 0), __double2loint(a));
        // ---- ratio test (one row per lane), exact f64 minimum on the order-preserving key, lowest lane on ties
        const double bi = max0_raw(beta);
        const bool elig = rowact & (a > TOL_PIV) & (bi < pinf);  // (pivot_core's `bi * an < bn * a` with bn = inf, an = 1)
        const double bn = elig ? bi : pinf;
        const double an_ = elig ? a : 1.0;
        const double pb = elig ? beta : 0.0;
        const double x0 = __builtin_amdgcn_rcp(an_);
        const double x1 = fma(x0, fma(-an_, x0, 1.0), x0);
        const double pinv = fma(x1, fma(-an_, x1, 1.0), x1);
        const double q = bn * pinv;
        const int qh = __double2hiint(q), ql = __double2loint(q);
        const int sm = qh >> 31;
        const unsigned kh = (unsigned)(qh ^ (sm | (int)0x80000000));
        const unsigned kl = (unsigned)(ql ^ sm);
        const unsigned mh = wave_min_u32(kh);
        const uint64_t hib = __ballot(kh == mh);
        unsigned ml;
        if (hib & (hib - 1ull)) ml = wave_min_u32((kh == mh) ? kl : 0xffffffffu);
        else ml = (unsigned)__builtin_amdgcn_readlane((int)kl, __ffsll((long long)hib) - 1);
        if (mh >= 0xfff00000u) { status = ((mh == 0xfff00000u) & (ml == 0u)) ? ST_UNBND : ST_NUM; break; }
        const int r = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot((kh == mh) & (kl == ml))) - 1);
        const double qmin = __hiloint2double((int)(mh ^ 0x80000000u), (int)ml);
        ndeg = __builtin_amdgcn_readfirstlane((qmin <= DEGEN_EPS) ? ndeg + 1 : 0);
        const double p = lane_value(pinv, r);
        const double rhob = lane_value(pb, r) * p;
        // ---- the pivot row as of now, lane j computing its entry j, then scaled: rho_t (its entry e is p)
        double v = lane < D ? sA[r * D + lane] : 0.0;
#define PLP_LZ_ROW(US_, QS_, ST_)                                                    \
        {                                                                            \
            const int st_ = (ST_);                                                   \
            const int e_s = st_ & 0xff, r_s = st_ >> 8;                              \
            const double ur = lane_value((US_), r);                                  \
            const double rj_ = (QS_);                                                \
            const double vn = (lane == e_s) ? -(ur * rj_) : fma(-ur, rj_, v);        \
            v = (r == r_s) ? rj_ : vn;                                               \
        }
#pragma unroll
        for (int s = 0; s < K_REG; ++s)
            if (t > s) PLP_LZ_ROW(ur_[s], qr_[s], str_[s])
        for (int s = K_REG; s < t; ++s) PLP_LZ_ROW(ux[s - K_REG], qx[s - K_REG], __builtin_amdgcn_readlane(steps, s))
        const double rj = lane < D ? ((lane == e) ? p : v * p) : 0.0;
        // ---- reduced costs, objective, my row
        const double fc = -best;
        c = (lane == e) ? -(fc * p) : fma(-fc, rj, c);
        negz = fma(-fc, rhob, negz);
        const bool is_r = lane == r;
        const double f = is_r ? 0.0 : a;
        const int stt = e | (r << 8);
#pragma unroll
        for (int s = 0; s < K_REG; ++s)
            if (t == s) { ur_[s] = f; qr_[s] = rj; str_[s] = stt; }
        if (t >= K_REG) { ux[t - K_REG] = f; qx[t - K_REG] = rj; }
        beta = is_r ? rhob : fma(-f, rhob, beta);
        if (is_r & efree) rowact = false;  // a free variable never leaves again
        cfree &= ~(1u << e);
        steps = (lane == t) ? stt : steps;
        if (basis_out) {
            const int vin = __builtin_amdgcn_readlane(colvar, e), vout = __builtin_amdgcn_readlane(rowvar, r);
            rowvar = is_r ? vin : rowvar;
            colvar = (lane == e) ? vout : colvar;
        }
        ++t;
        if (!price<D>(lane, c, cfree, e, best, chi)) { status = ST_OPT; break; }
    }
    negz_out = negz;
    if (basis_out && lane < D) basis_out[lane] = (signed char)(colvar < D ? -1 - colvar : colvar - D);
    return status;
}

}  // namespace lazy
}  // namespace plp
