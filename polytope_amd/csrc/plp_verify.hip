// plp_verify.hip -- the verifier kernels behind plp_lp_solve_batch / plp_cheby_batch / plp_bbox_batch (round 6; the
// arithmetic is plp_verify.hpp, which says why).  Three launches follow the engines' own, on the same stream:
//
//   verify_kernel<KIND, VN>   generic LPs and Chebyshev LPs (the engines hand over x): an optimal answer gets a basis read
//        off its x and the certificate; certified -> x / fun (r / xc) are REPLACED by the polished vertex (LU of the
//        original rows + refinement: the value no longer depends on the path the engine took), and an optimum out of
//        range becomes "unbounded"; anything else that is not a plain "infeasible" (a failed certificate, unbounded,
//        iteration limit, numerical trouble) is appended to the launch's list;  box LPs (2d per polytope): the fused
//        bounding-box kernels hand over each LP's final basis (d bytes) and the Chebyshev centre they started from, or
//        the point the LP ended on; same rule, the list gets (polytope, side);
//   careful_kernel<KIND>        the list, one LP per thread, solved from scratch by the double-double engine
//        (careful_solve) with its dictionary in global memory, element by element interleaved over the threads
//        (coalesced); empty list -> the launch ends at once (no host round trip decides whether it is needed).
//
// HBM traffic of the verifier: the rows twice more (they come from the L2 / Infinity Cache right behind the engine that
// read them), x in and out.  The certificate's work arrays live in LDS, sized by VN = 5 / 9 / 17 columns (64 / 32 / 8 LPs per
// workgroup): as private arrays they were scratch memory, and the verifier took ten times the engines' own time.
#include <stdlib.h>

#include "plp_kernels.hpp"
#include "plp_verify.hpp"

namespace plp {

using namespace verify;

namespace {

constexpr int VBLK = 64;
constexpr long long CAREFUL_SLOTS = 4096;  // LPs the careful engine solves side by side (its dictionaries: slots x ~21 KB at 64 rows)

struct Scratch {
    unsigned* count;  // entries of this launch's list; `other`: the counter of the stream's NEXT launch, zeroed by this one's careful
    unsigned* other;  // kernel (two counters in turn: no reset launch, no host round trip)
    int* list;
    double* hi;
    double* lo;
    int* rowinfo;
    long long cap;
};

__host__ __device__ inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

Scratch carve(void* base, long long cap, int m_max, int parity) {
    char* p = static_cast<char*>(base);
    Scratch s;
    s.count = reinterpret_cast<unsigned*>(p + ((parity & 1) ? 64 : 0));
    s.other = reinterpret_cast<unsigned*>(p + ((parity & 1) ? 0 : 64));
    p += 256;
    s.list = reinterpret_cast<int*>(p);
    p += pad256((size_t)cap * 4);
    const size_t nd = careful_doubles_per_lp(m_max) * CAREFUL_SLOTS;
    s.hi = reinterpret_cast<double*>(p);
    p += pad256(nd * 8);
    s.lo = reinterpret_cast<double*>(p);
    p += pad256(nd * 8);
    s.rowinfo = reinterpret_cast<int*>(p);
    s.cap = cap;
    return s;
}

__device__ __forceinline__ void push(const Scratch& s, int v) {
    const unsigned k = atomicAdd(s.count, 1u);
    if ((long long)k < s.cap) s.list[k] = v;
}

__device__ __forceinline__ double qnan() { return __longlong_as_double(0x7ff8000000000000ll); }

// ---- the verifier kernel.  LPB LPs per 64-lane workgroup, each with its work arrays in LDS (element e of LP l at
// ws[e * LPB + l]: consecutive LPs, consecutive banks) and GSL = 64 / LPB lanes: the leader lane runs the certificate's serial
// part (candidates -> basis -> LU -> polished vertex -> multipliers), all GSL lanes share the two passes over the rows.
//   KIND LP_GENERIC / LP_CHEBY: LP t of the batch, x handed over by the engine (x / fun; xc / r);
//   KIND LP_BOXSIDE: t = polytope * 2 d + side; the fused bounding-box kernels handed over the LP's basis (d bytes) and the
//   polytope's centre, or (the one-LP-per-lane kernel) the point the LP ended on.
struct VArgs {
    long long T;       // LPs
    int m_max, n;      // n: columns (d + 1 for LP_CHEBY, d for LP_BOXSIDE)
    const double* c;   // generic only
    const double* G;   // G, or A
    const double* h;   // h, or b
    const int* mrows;
    double* x;         // x [B][n] | xc [B][d] | lb [B][d]
    double* fun;       // fun [B]  | r [B]     | ub [B][d]
    int* status;       // [B]
    const signed char* basis8;
    const double* centre;
    const double* xfin;
};

template <int VN>
struct VShape {
#ifndef PLP_VF_LPB9
#define PLP_VF_LPB9 8
#endif
#ifndef PLP_VF_LPB17
#define PLP_VF_LPB17 4
#endif
    static constexpr int LPB = VN <= 5 ? 64 : (VN <= 9 ? PLP_VF_LPB9 : PLP_VF_LPB17);   // (LDS: 2 KB / 6 KB of work arrays per LP; 8 / 16 lanes per LP: measured against 4 / 8 and 16 / 32)
    static constexpr int GSL = VBLK / LPB;
};

// (-DPLP_VF_STOP=k: the kernel ends before stage k + 1 -- timing builds only, scripts/debug/verify_stages.sh)
#ifdef PLP_VF_STOP
#define PLP_VF_STAGE(k) if (PLP_VF_STOP == (k)) return;
#else
#define PLP_VF_STAGE(k)
#endif
template <int KIND, int VN>
__global__ __launch_bounds__(VBLK) void verify_kernel(VArgs a, Scratch sc) {
    constexpr int LPB = VShape<VN>::LPB, GSL = VShape<VN>::GSL;
    static_assert(GSL >= 2, "two lanes for the side-by-side solves");
    using CT = Cert<VN, LPB>;
    using Vec = typename CT::Vec;
    __shared__ double s_ws[CT::WS_DOUBLES * LPB];
    __shared__ int s_cn[LPB];
    __shared__ int s_mode[LPB];   // 0: nothing to do, 1: read the basis off x, 2: basis handed over, 3: to the list
    __shared__ int s_ok[LPB];
    __shared__ unsigned s_bad[LPB];
    __shared__ double s_xs[LPB], s_fun[LPB];
    __shared__ unsigned long long s_hs[LPB];   // bits of max(1, max_i |h_i| / |G_i|_inf), gathered by pass 2
    const int li = threadIdx.x / GSL, gl = threadIdx.x % GSL;
    const long long t = (long long)blockIdx.x * LPB + li;
    const bool valid = t < a.T;
    const int n = a.n;
    double* ws = s_ws + li;
    // ---- which LP
    long long p = valid ? t : 0;
    int side = 0;
    if constexpr (KIND == LP_BOXSIDE) {
        p = (valid ? t : 0) / (2 * n);
        side = (int)((valid ? t : 0) - p * 2 * n);
    }
    const int ng = KIND == LP_CHEBY ? n - 1 : n;
    LpView lp;
    lp.m = valid ? (a.mrows ? a.mrows[p] : a.m_max) : 0;
    lp.n = n;
    lp.kind = KIND;
    lp.side = side;
    lp.G = a.G + (size_t)p * a.m_max * ng;
    lp.h = a.h + (size_t)p * a.m_max;
    lp.c = KIND == LP_GENERIC ? a.c + (size_t)p * n : nullptr;
    // ---- leader: what is to be done, the point / basis into the workspace
    if (gl == 0) {
        int mode = 0;
        if (valid) {
            if constexpr (KIND == LP_BOXSIDE) {
                if (a.status[p] == 0) {   // (status 1: handed to the caller's generic LPs, which pass this kernel as LP_GENERIC)
                    const double val = ((side & 1) ? a.fun : a.x)[p * n + (side >> 1)];
                    if (!(fabs(val) < 1e300)) mode = 3;   // unbounded: the careful engine decides
                    else mode = a.xfin ? 1 : 2;
                }
            } else {
                const int st = a.status[p];
                // (infeasible: phase 1's verdict, read with a 1e-7 margin -- not a question of rounding; it stands)
                if (st == ST_OPT) mode = 1;
                else if (st != ST_INFEAS) {
                    mode = 3;
                }
            }
        }
        s_mode[li] = mode;
        s_cn[li] = 0;
        s_ok[li] = 0;
        s_bad[li] = 0u;
        s_hs[li] = (unsigned long long)__double_as_longlong(1.0);
    }
    __syncthreads();
    // ---- the engine's point / basis into the workspace, an entry per lane in turn (one lane alone pays a global-memory round
    // trip per entry: the loop does not unroll)
    if (s_mode[li] == 1 || s_mode[li] == 2) {
        const Vec xv = CT::at(ws, CT::O_X), bas = CT::at(ws, CT::O_BAS);
        for (int j = gl; j < n; j += GSL) {
            if constexpr (KIND == LP_BOXSIDE) {
                if (s_mode[li] == 1) xv[j] = a.xfin[((size_t)p * 2 * n + side) * n + j];
                else {
                    xv[j] = a.centre[(size_t)p * n + j];
                    bas[j] = (double)a.basis8[((size_t)p * 2 * n + side) * n + j];
                }
            } else if constexpr (KIND == LP_CHEBY) {
                xv[j] = j < n - 1 ? a.x[(size_t)p * (n - 1) + j] : a.fun[p];
            } else {
                xv[j] = a.x[(size_t)p * n + j];
            }
        }
    }
    __syncthreads();
    if ((gl == 0) & (s_mode[li] == 1)) {
        bool fin;
        s_xs[li] = CT::x_scale(lp, CT::at(ws, CT::O_X), &fin);
        if (!fin) s_mode[li] = 3;
    }
    __syncthreads();
    PLP_VF_STAGE(1)
    // ---- pass 1 (a basis is to be read off x): the candidate rows, unsorted, into the workspace's list
    if (s_mode[li] == 1) {
        const Vec xv = CT::at(ws, CT::O_X), cs = CT::at(ws, CT::O_CS), ci = CT::at(ws, CT::O_CI);
        const double xs = s_xs[li];
        for (int i = gl; i < lp.m; i += GSL) {
            double sl;
            if (CT::row_candidate(lp, i, xv, xs, &sl)) {
                const int k = atomicAdd(&s_cn[li], 1);
                if (k < CT::KC) { cs[k] = sl; ci[k] = (double)i; }
            }
        }
    }
    __syncthreads();
    PLP_VF_STAGE(2)
    // ---- leader: the basis
    if ((gl == 0) & (s_mode[li] == 1)) {
        const Vec cs = CT::at(ws, CT::O_CS), ci = CT::at(ws, CT::O_CI);
        int cn = s_cn[li];
        if (cn > CT::KC) cn = CT::KC + 1;
        else {   // in place, by (slack, row): the order the host version inserts in
            for (int k = 1; k < cn; ++k) {
                const double sk = cs[k], ik = ci[k];
                int q = k;
                while (q > 0 && (cs[q - 1] > sk || (cs[q - 1] == sk && ci[q - 1] > ik))) {
                    cs[q] = cs[q - 1];
                    ci[q] = ci[q - 1];
                    --q;
                }
                cs[q] = sk;
                ci[q] = ik;
            }
        }
        if (!CT::select_basis(lp, ws, cn)) s_mode[li] = 3;
    }
    __syncthreads();
    PLP_VF_STAGE(3)
    // ---- the basis matrix, a row per lane in turn; its factorisation with the row updates of an elimination step shared by
    // the LP's lanes (the leader finds the pivot and swaps, a barrier, every lane takes every GSL-th row below it, a barrier)
    const bool fac = (s_mode[li] == 1) | (s_mode[li] == 2);   // (the same for the lanes of an LP; the barriers are workgroup-wide)
    if (fac) {
        bool okr = true;
        for (int k = gl; k < n; k += GSL) {
            okr = okr & CT::basis_row(lp, true, ws, k);
            CT::cost_entry(lp, ws, k);
        }
        if (!okr) atomicOr(&s_bad[li], 2u);
    }
    __syncthreads();
    PLP_VF_STAGE(4)
    double big = 0.0, pmin = 1e300;
    bool live = fac && !(s_bad[li] & 2u);
    if (live & (gl == 0)) {
        big = CT::lu_begin(n, CT::at(ws, CT::O_GM), CT::at(ws, CT::O_PERM));
        if (!(big > 0.0)) atomicOr(&s_bad[li], 2u);
    }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        live = fac && !(s_bad[li] & 2u);
        if (live & (gl == 0)) {
            double pv;
            if (CT::lu_pivot(n, k, CT::at(ws, CT::O_LU), CT::at(ws, CT::O_PERM), big, &pv)) pmin = fmin(pmin, pv);
            else atomicOr(&s_bad[li], 2u);
        }
        __syncthreads();
        if (fac && !(s_bad[li] & 2u)) CT::lu_rows(n, k, CT::at(ws, CT::O_LU), gl, GSL);
        __syncthreads();
    }
    PLP_VF_STAGE(5)
    // ---- the two plain solves side by side (lane 0: the vertex, lane 1: the multipliers; one instruction stream), then the
    // leader: refinement where the basis is ill-conditioned, value, the multipliers' signs
    if (fac && !(s_bad[li] & 2u) && gl < 2) {
        const bool tr = gl == 1;
        CT::lu_solve_any(n, CT::at(ws, CT::O_LU), CT::at(ws, CT::O_PERM), CT::at(ws, tr ? CT::O_V : CT::O_RHS),
                         CT::at(ws, tr ? CT::O_Y : CT::O_Z), CT::at(ws, tr ? CT::O_DZ : CT::O_T), tr);
    }
    __syncthreads();
    if ((gl == 0) & fac) {
        double f = 0.0, zs = 1.0;
        if (!(s_bad[li] & 2u) && CT::vertex_and_dual_finish(lp, true, ws, pmin / big, &f, &zs, true)) {
            s_xs[li] = zs;
            s_fun[li] = f;
            s_ok[li] = 1;
        }
        s_bad[li] = 0u;
    }
    __syncthreads();
    PLP_VF_STAGE(6)
    // ---- pass 2: every row against the polished vertex
    if (s_ok[li]) {
        const Vec z = CT::at(ws, CT::O_Z);
        const double zs = s_xs[li];
        bool bad = false;
        double hs = 1.0;
        for (int i = gl; i < lp.m; i += GSL) bad = bad | !CT::row_feasible_hs(lp, i, z, zs, &hs);
        if (bad) atomicOr(&s_bad[li], 1u);
        if (hs > 1.0) atomicMax(&s_hs[li], (unsigned long long)__double_as_longlong(hs));   // (positive doubles order like their bits)
    }
    __syncthreads();
    if ((gl != 0) | (s_mode[li] == 0)) return;
    if (s_ok[li] & (s_bad[li] == 0u) & !LpView::far_vertex_hs(s_xs[li], __longlong_as_double((long long)s_hs[li]))) {
        const Vec z = CT::at(ws, CT::O_Z);
        const double f = s_fun[li];
        double cmax = 0.0;   // (-c is in the workspace)
        for (int j = 0; j < n; ++j) cmax = fmax(cmax, fabs(CT::at(ws, CT::O_V)[j]));
        const bool out = range_rule_c(lp, V_OPT, f, s_xs[li], cmax) != V_OPT;
        if constexpr (KIND == LP_BOXSIDE) {
            const double pinf = __longlong_as_double(0x7ff0000000000000ll);
            ((side & 1) ? a.fun : a.x)[p * n + (side >> 1)] = out ? ((side & 1) ? pinf : -pinf) : z[side >> 1];
        } else if constexpr (KIND == LP_CHEBY) {
            for (int j = 0; j < n - 1; ++j) a.x[(size_t)p * (n - 1) + j] = out ? qnan() : z[j];
            a.fun[p] = out ? qnan() : z[n - 1];
            if (out) a.status[p] = ST_UNBND;
        } else {
            for (int j = 0; j < n; ++j) a.x[(size_t)p * n + j] = out ? qnan() : z[j];
            a.fun[p] = out ? qnan() : f;
            if (out) a.status[p] = ST_UNBND;
        }
        return;
    }
    push(sc, (int)t);
}

// ---- the list: one LP per thread, from scratch, double-double
template <int KIND>
__global__ __launch_bounds__(VBLK) void careful_kernel(int m_max, int n, const double* __restrict__ c,
                                                       const double* __restrict__ G, const double* __restrict__ h,
                                                       const int* __restrict__ mrows, double* __restrict__ x,
                                                       double* __restrict__ fun, int* __restrict__ status,
                                                       double* __restrict__ ub, Scratch sc) {
    const long long slot = (long long)blockIdx.x * VBLK + threadIdx.x;
    if (slot == 0) *sc.other = 0u;
    unsigned cnt = *sc.count;
    if ((long long)cnt > sc.cap) cnt = (unsigned)sc.cap;
    CarefulMem M{sc.hi + slot, sc.lo + slot, sc.rowinfo + slot, CAREFUL_SLOTS};
    for (long long k = slot; k < (long long)cnt; k += CAREFUL_SLOTS) {
        const long long t = sc.list[k];
        LpView lp;
        lp.n = n;
        lp.kind = KIND;
        double xo[VNMAX], f = 0.0;
        if constexpr (KIND == LP_BOXSIDE) {
            // x = lb, ub = ub, fun unused, status = the polytope's (left alone: a side that cannot be solved makes it 1)
            const long long p = t / (2 * n);
            const int side = (int)(t - p * 2 * n);
            lp.m = mrows ? mrows[p] : m_max;
            lp.side = side;
            lp.G = G + (size_t)p * m_max * n;
            lp.h = h + (size_t)p * m_max;
            lp.c = nullptr;
            int st = careful_solve(lp, M, xo, &f, nullptr);
            double xm = 0.0;
            for (int j = 0; j < n; ++j) xm = fmax(xm, fabs(xo[j]));
            st = range_rule(lp, st, f, xm);
            double* out = (side & 1) ? ub : x;
            const double pinf = __longlong_as_double(0x7ff0000000000000ll);
            const int kx = side >> 1;
            if (st == V_OPT) out[p * n + kx] = xo[kx];
            else if (st == V_UNBND) out[p * n + kx] = (side & 1) ? pinf : -pinf;
            else { out[p * n + kx] = qnan(); status[p] = 1; }  // infeasible / limits: the caller's generic LPs decide (ref :1378-1402)
        } else {
            const long long p = t;
            const int ng = KIND == LP_CHEBY ? n - 1 : n;
            lp.m = mrows ? mrows[p] : m_max;
            lp.side = 0;
            lp.G = G + (size_t)p * m_max * ng;
            lp.h = h + (size_t)p * m_max;
            lp.c = KIND == LP_GENERIC ? c + (size_t)p * n : nullptr;
            int st = careful_solve(lp, M, xo, &f, nullptr);
            double xm = 0.0;
            for (int j = 0; j < n; ++j) xm = fmax(xm, fabs(xo[j]));
            st = range_rule(lp, st, f, xm);
            const bool opt = st == V_OPT;
            if constexpr (KIND == LP_CHEBY) {
                for (int j = 0; j < n - 1; ++j) x[(size_t)p * (n - 1) + j] = opt ? xo[j] : qnan();
                fun[p] = opt ? xo[n - 1] : qnan();
            } else {
                for (int j = 0; j < n; ++j) x[(size_t)p * n + j] = opt ? xo[j] : qnan();
                fun[p] = opt ? f : qnan();
            }
            status[p] = st;
        }
    }
}

bool verify_off() {
    static const int off = [] {
        const char* e = getenv("PLP_VERIFY");
        return (e && e[0] == '0') ? 1 : 0;
    }();
    return off != 0;
}

}  // namespace

// bytes of device scratch a verify launch over `nlp` LPs of up to m_max rows needs
size_t verify_scratch_bytes(long long nlp, int m_max) {
    const size_t nd = careful_doubles_per_lp(m_max) * CAREFUL_SLOTS;
    return 256 + pad256((size_t)nlp * 4) + 2 * pad256(nd * 8) + pad256((size_t)(m_max + 2) * CAREFUL_SLOTS * 4) + 256;
}

bool verify_enabled() { return !verify_off(); }

// ---- the same for LPs of N <= 5 columns (d <= 4: most of what the library is asked): ONE LP PER LANE, the column count a
// compile-time constant, the certificate's arrays in registers (Cert<N, 1, N>: static indices only) -- no LDS, no
// workgroup barriers, eight waves per SIMD instead of one.
template <int KIND, int N>
__global__ __launch_bounds__(256) void verify_small_kernel(VArgs a, Scratch sc) {
    using CT = Cert<N, 1, N>;
    using Vec = typename CT::Vec;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.T) return;
    constexpr int n = N;
    long long p = t;
    int side = 0;
    if constexpr (KIND == LP_BOXSIDE) {
        p = t / (2 * n);
        side = (int)(t - p * 2 * n);
    }
    constexpr int ng = KIND == LP_CHEBY ? n - 1 : n;
    LpView lp;
    lp.m = a.mrows ? a.mrows[p] : a.m_max;
    lp.n = n;
    lp.kind = KIND;
    lp.side = side;
    lp.G = a.G + (size_t)p * a.m_max * ng;
    lp.h = a.h + (size_t)p * a.m_max;
    lp.c = KIND == LP_GENERIC ? a.c + (size_t)p * n : nullptr;
    double ws[CT::WS_DOUBLES];
    const Vec xv = CT::at(ws, CT::O_X), bas = CT::at(ws, CT::O_BAS), z = CT::at(ws, CT::O_Z);
    int mode = 0;   // 0: nothing to do, 1: read the basis off x, 2: basis handed over, 3: to the list
    if constexpr (KIND == LP_BOXSIDE) {
        if (a.status[p] != 0) return;   // (handed to the caller's generic LPs, which pass this kernel as LP_GENERIC)
        const double val = ((side & 1) ? a.fun : a.x)[p * n + (side >> 1)];
        if (!(fabs(val) < 1e300)) mode = 3;   // unbounded: the careful engine decides
        else if (a.xfin) {
#pragma unroll
            for (int j = 0; j < n; ++j) xv[j] = a.xfin[((size_t)p * 2 * n + side) * n + j];
            mode = 1;
        } else {
#pragma unroll
            for (int j = 0; j < n; ++j) {
                xv[j] = a.centre[(size_t)p * n + j];
                bas[j] = (double)a.basis8[((size_t)p * 2 * n + side) * n + j];
            }
            mode = 2;
        }
    } else {
        const int st = a.status[p];
        if (st == ST_INFEAS) return;   // (phase 1's verdict, read with a 1e-7 margin -- not a question of rounding; it stands)
        if (st == ST_OPT) {
            if constexpr (KIND == LP_CHEBY) {
#pragma unroll
                for (int j = 0; j < n - 1; ++j) xv[j] = a.x[(size_t)p * (n - 1) + j];
                xv[n - 1] = a.fun[p];
            } else {
#pragma unroll
                for (int j = 0; j < n; ++j) xv[j] = a.x[(size_t)p * n + j];
            }
            mode = 1;
        } else {
            mode = 3;
        }
    }
    bool ok = false;
    double f = 0.0, zs = 1.0;
    if (mode == 1) ok = CT::basis_from_x(lp, ws);
    else if (mode == 2) ok = true;
    if (ok) ok = CT::vertex_and_dual(lp, true, true, ws, &f, &zs);
    if (ok) {
        double hs = 1.0;
        for (int i = 0; i < lp.m; ++i) ok = ok & CT::row_feasible_hs(lp, i, z, zs, &hs);
        if (ok) ok = !LpView::far_vertex_hs(zs, hs);
    }
    if (ok) {
        const bool out = range_rule(lp, V_OPT, f, zs) != V_OPT;
        if constexpr (KIND == LP_BOXSIDE) {
            const double pinf = __longlong_as_double(0x7ff0000000000000ll);
            double zk = 0.0;
#pragma unroll
            for (int j = 0; j < n; ++j) zk = (j == (side >> 1)) ? z[j] : zk;
            ((side & 1) ? a.fun : a.x)[p * n + (side >> 1)] = out ? ((side & 1) ? pinf : -pinf) : zk;
        } else if constexpr (KIND == LP_CHEBY) {
#pragma unroll
            for (int j = 0; j < n - 1; ++j) a.x[(size_t)p * (n - 1) + j] = out ? qnan() : z[j];
            a.fun[p] = out ? qnan() : z[n - 1];
            if (out) a.status[p] = ST_UNBND;
        } else {
#pragma unroll
            for (int j = 0; j < n; ++j) a.x[(size_t)p * n + j] = out ? qnan() : z[j];
            a.fun[p] = out ? qnan() : f;
            if (out) a.status[p] = ST_UNBND;
        }
        return;
    }
    push(sc, (int)t);
}

// `parity`: the launches of a stream take the scratch's two list counters in turn (carve)
template <int KIND>
static void launch_verify_kind(const VArgs& a, const Scratch& sc, hipStream_t st) {
    const dim3 gsmall((unsigned)((a.T + 255) / 256));
    if (a.n == 1 && KIND != LP_CHEBY) {
        hipLaunchKernelGGL((verify_small_kernel<KIND, 1>), gsmall, dim3(256), 0, st, a, sc);
    } else if (a.n == 2) {
        hipLaunchKernelGGL((verify_small_kernel<KIND, 2>), gsmall, dim3(256), 0, st, a, sc);
    } else if (a.n == 3) {
        hipLaunchKernelGGL((verify_small_kernel<KIND, 3>), gsmall, dim3(256), 0, st, a, sc);
    } else if (a.n == 4) {
        hipLaunchKernelGGL((verify_small_kernel<KIND, 4>), gsmall, dim3(256), 0, st, a, sc);
    } else if (a.n == 5) {
        hipLaunchKernelGGL((verify_small_kernel<KIND, 5>), gsmall, dim3(256), 0, st, a, sc);
    } else if (a.n <= 9) {
        constexpr int LPB = VShape<9>::LPB;
        hipLaunchKernelGGL((verify_kernel<KIND, 9>), dim3((unsigned)((a.T + LPB - 1) / LPB)), dim3(VBLK), 0, st, a, sc);
    } else {
        constexpr int LPB = VShape<17>::LPB;
        hipLaunchKernelGGL((verify_kernel<KIND, 17>), dim3((unsigned)((a.T + LPB - 1) / LPB)), dim3(VBLK), 0, st, a, sc);
    }
}

int launch_verify_lp(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
                     double* x, double* fun, int* status, void* scratch, int parity, hipStream_t st) {
    if (B < 1 || n < 1 || n > VNMAX || B > 2147483647ll) return 1;
    const Scratch sc = carve(scratch, B, m_max, parity);
    const VArgs a{B, m_max, n, c, G, h, mrows, x, fun, status, nullptr, nullptr, nullptr};
    launch_verify_kind<LP_GENERIC>(a, sc, st);
    hipLaunchKernelGGL((careful_kernel<LP_GENERIC>), dim3((unsigned)(CAREFUL_SLOTS / VBLK)), dim3(VBLK), 0, st, m_max, n, c, G, h,
                       mrows, x, fun, status, (double*)nullptr, sc);
    return 0;
}

int launch_verify_cheby(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* r,
                        double* xc, int* status, void* scratch, int parity, hipStream_t st) {
    const int n = d + 1;
    if (B < 1 || d < 1 || n > VNMAX || B > 2147483647ll) return 1;
    const Scratch sc = carve(scratch, B, m_max, parity);
    const VArgs a{B, m_max, n, nullptr, A, b, mrows, xc, r, status, nullptr, nullptr, nullptr};
    launch_verify_kind<LP_CHEBY>(a, sc, st);
    hipLaunchKernelGGL((careful_kernel<LP_CHEBY>), dim3((unsigned)(CAREFUL_SLOTS / VBLK)), dim3(VBLK), 0, st, m_max, n,
                       (const double*)nullptr, A, b, mrows, xc, r, status, (double*)nullptr, sc);
    return 0;
}

int launch_verify_box(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* lb, double* ub,
                      int* status, const signed char* basis8, const double* centre, const double* xfin, void* scratch,
                      int parity, hipStream_t st) {
    const long long T = B * 2 * d;
    if (B < 1 || d < 1 || d > MAX_D || T > 2147483647ll) return 1;
    const Scratch sc = carve(scratch, T, m_max, parity);
    const VArgs a{T, m_max, d, nullptr, A, b, mrows, lb, ub, status, basis8, centre, xfin};
    launch_verify_kind<LP_BOXSIDE>(a, sc, st);
    hipLaunchKernelGGL((careful_kernel<LP_BOXSIDE>), dim3((unsigned)(CAREFUL_SLOTS / VBLK)), dim3(VBLK), 0, st, m_max, d,
                       (const double*)nullptr, A, b, mrows, lb, (double*)nullptr, status, ub, sc);
    return 0;
}

}  // namespace plp
