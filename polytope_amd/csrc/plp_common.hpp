// plp_common.hpp -- constants shared by the HIP kernels and the C-ABI (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plp {

// LP status codes = scipy.optimize.linprog's (reference: polytope/solvers.py:76-106,155-158)
enum : int { ST_OPT = 0, ST_ITER = 1, ST_INFEAS = 2, ST_UNBND = 3, ST_NUM = 4 };
// internal: the fast pivot path met a dictionary that needs Bland's rule; the LP is redone by the
// general engine (never leaves the library)
constexpr int ST_RETRY = 5;
constexpr int ST_RETRY_P1 = 6;  // internal: generic LP that needs phase 1 (second kernel of launch_lp_r)

// flags written by the fused reduce kernel (reference: polytope/polytope.py:1053-1163)
enum : int {
    RF_EMPTY = 1,   // not full-dimensional -> reference returns Polytope()      (:1081-1082)
    RF_EARLY = 2,   // returned at neq <= nx+1, minrep stays False               (:1114-1116,:1136-1138)
    RF_MINREP = 4,  // went through the redundancy LPs, minrep = True            (:1161-1163)
    RF_LPFAIL = 8,  // a bounding-box LP ended with status 1/4 (RuntimeError)    (:1378-1384)
    RF_RETRY = 16,  // internal: redo this polytope with the general engine (second pass of launch_reduce)
    // RF_EMPTY because the Chebyshev LP did NOT end optimal (unbounded, or a limit) rather than with a radius below the
    // tolerance.  For a half-space or a cone that is the reference's verdict too (cheby_ball returns 0 for every status but 0,
    // :1289-1297) -- but on rows a hair apart the engine's pivot tolerance (TOL_PIV) can call a bounded ball unbounded
    // (tests/golden/g23, case 33: the reference reduces that polytope, the engine called it empty).  The fused kernels run
    // unverified; with this flag the caller re-examines the few polytopes concerned through the verified stand-alone LPs
    // (polytope_amd/polytope.py: _reduce_many).
    // Also set when the engine called its Chebyshev LP optimal at a centre that VIOLATES a row of the polytope (centre_off
    // below; tests/golden/found/lane93_t21_k20730.npz: radius right, centre 5 outside -- a pivot next to twin rows): the
    // presolve and the walks of F2 / F3 start from that centre, so nothing of the fused answer is usable.
    RF_F1OPEN = 32
};

// tolerances of the simplex core (identical in oracle/plp_oracle.c)
constexpr double TOL_D = 1e-9;      // reduced-cost tolerance
// Smallest admissible pivot.  1e-9 until round 5: two rows a hair apart (normals 1e-9 .. 1e-7 rad apart, or equal with right-hand
// sides 1e-10 apart -- what stacking polytopes that share a facet produces) leave, once one of them is in the basis, a
// dictionary row whose entries ARE that hair; a pivot on it multiplies the dictionary's rounding by 1e9, two in a row by
// 1e18: Chebyshev balls that stuck 0.05 .. 6 out of their polytope, an F1 called unbounded on a box (scripts/soak_lane.py,
// soak_wide.py, family `dup`; the oracle's simplex, which shares the constant, failed the same way on other members of the
// family).  At 1e-7 such a row is not a blocking row: the step may pass it by 1e-7 times its length -- HiGHS's own primal
// feasibility tolerance, i.e. what the reference's answers are good to on such rows -- and every one of those cases
// agrees with HiGHS (600 `dup` polytopes of (37,8): 4 wrong balls -> 0, largest difference 2e-8).
constexpr double TOL_PIV = 1e-7;
constexpr double TOL_FEAS = 1e-7;   // phase-1 infeasibility accepted (HiGHS primal tolerance)
// The fused reduce's F1 answer is checked where the centre is first used: row i with raw = b_i - a_i.xc fails when
// raw / |a_i| < -TOL_CENTRE max(|b_i| / |a_i|, max(1, |xc|_inf)) -- far above anything a healthy answer shows (1e-12) and far
// below what a broken pivot leaves (1e-2 .. 10); identical in oracle/plp_oracle.c (plpo_reduce).
constexpr double TOL_CENTRE = 1e-6;
__host__ __device__ __forceinline__ bool centre_off(double raw, double inv_nrm, double bi, double xs) {
#if defined(PLP_NO_CENTRE_CHECK)   // (A/B builds only: what the test costs)
    return false;
#endif
    return isfinite(inv_nrm) && (raw * inv_nrm < -TOL_CENTRE * fmax(fabs(bi) * inv_nrm, xs));
}
template <int D>
__host__ __device__ __forceinline__ double centre_scale(const double* xc) {
    double xs = 1.0;
#pragma unroll
    for (int k = 0; k < D; ++k) xs = fmax(xs, fabs(xc[k]));
    return xs;
}
constexpr double DEGEN_EPS = 1e-12; // step length regarded as degenerate
constexpr int BLAND_AFTER = 6;      // consecutive degenerate pivots before Bland's rule

constexpr int MAX_M = 64;  // rows per LP  (one lane per row, one LP per <=64-lane group)
constexpr int MAX_D = 16;  // space dimension
constexpr int WAVE = 64;   // CDNA wavefront
constexpr int BLOCK = 256; // 4 waves per workgroup

// plp_reduce_counters: the device counter is this many 8-byte words, 64 B apart (a power of two)
constexpr int PLP_CTR_SLOTS = 1024;

// np.sum(n * p) as numpy evaluates it on a contiguous vector of D doubles (the reference's quickhull distance,
// polytope/quickhull.py:121): the products first, then add.reduce's pairwise_sum -- below 8 elements one running sum in
// index order; from 8 on eight running sums r[j] over the blocks of eight, combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the D % 8 left-over elements one by one (oracle: np_sum_prod, pinned bit
// for bit by tests/golden g18).  No fma: the kernels are built with -ffp-contract=off.
template <int D>
__host__ __device__ __forceinline__ double np_dot(const double* __restrict__ n, const double* __restrict__ x) {
    if constexpr (D < 8) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) s = s + n[k] * x[k];
        return s;
    } else {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = n[j] * x[j];
        constexpr int BLK = D - D % 8;
#pragma unroll
        for (int i = 8; i < BLK; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = r[j] + n[i + j] * x[i + j];
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = BLK; i < D; ++i) res = res + n[i] * x[i];
        return res;
    }
}

// ids of variables: 0..n-1 structural free x_j ; n+i slack of row i ; -1 phase-1 artificial
constexpr int ID_T = -1;

}  // namespace plp
