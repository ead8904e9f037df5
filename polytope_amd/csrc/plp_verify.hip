// plp_verify.hip -- the verifier kernels behind plp_lp_solve_batch / plp_cheby_batch / plp_bbox_batch (round 6; the
// arithmetic is plp_verify.hpp, which says why).  Three launches follow the engines' own, on the same stream:
//
//   verify_x_kernel<KIND, VN>   one LP per thread (generic LPs and Chebyshev LPs: the engines hand over x): an optimal
//        answer gets a basis read off its x and the certificate; certified -> x / fun (r / xc) are REPLACED by the polished
//        vertex (LU of the original rows + refinement: the value no longer depends on the path the engine took), and an
//        optimum out of range becomes "unbounded"; anything else that is not a plain "infeasible" (a failed certificate,
//        unbounded, iteration limit, numerical trouble) is appended to the launch's list;
//   verify_box_kernel<VN>       one box LP per thread (2d per polytope): the fused bounding-box kernels hand over each
//        LP's final basis (d bytes) and the Chebyshev centre they started from; same rule, the list gets (polytope, side);
//   careful_kernel<KIND>        the list, one LP per thread, solved from scratch by the double-double engine
//        (careful_solve) with its dictionary in global memory, element by element interleaved over the threads
//        (coalesced); empty list -> the launch ends at once (no host round trip decides whether it is needed).
//
// HBM traffic of the verifier: the rows once more (they come from the L2 / Infinity Cache right behind the engine that
// read them), x in and out.  Per-thread arrays (LU, the Gram-Schmidt basis) are sized by VN = 5 / 9 / 17 columns.
#include <stdlib.h>

#include "plp_kernels.hpp"
#include "plp_verify.hpp"

namespace plp {

using namespace verify;

namespace {

constexpr int VBLK = 64;
constexpr long long CAREFUL_SLOTS = 4096;  // LPs the careful engine solves side by side (its dictionaries: slots x ~21 KB at 64 rows)

struct Scratch {
    unsigned* count;  // [0]: entries of the list
    int* list;
    double* hi;
    double* lo;
    int* rowinfo;
    long long cap;
};

__host__ __device__ inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

Scratch carve(void* base, long long cap, int m_max) {
    char* p = static_cast<char*>(base);
    Scratch s;
    s.count = reinterpret_cast<unsigned*>(p);
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

// ---- generic LPs / Chebyshev LPs: x is there
template <int KIND, int VN>
__global__ __launch_bounds__(VBLK) void verify_x_kernel(long long B, int m_max, int n, const double* __restrict__ c,
                                                        const double* __restrict__ G, const double* __restrict__ h,
                                                        const int* __restrict__ mrows, double* __restrict__ x,
                                                        double* __restrict__ fun, int* __restrict__ status, Scratch sc) {
    const long long p = (long long)blockIdx.x * VBLK + threadIdx.x;
    if (p >= B) return;
    const int st = status[p];
    if (st == ST_INFEAS) return;  // (phase 1's verdict, read with a 1e-7 margin: not a question of rounding)
    const int ng = KIND == LP_CHEBY ? n - 1 : n;
    LpView lp;
    lp.m = mrows ? mrows[p] : m_max;
    lp.n = n;
    lp.kind = KIND;
    lp.side = 0;
    lp.G = G + (size_t)p * m_max * ng;
    lp.h = h + (size_t)p * m_max;
    lp.c = KIND == LP_GENERIC ? c + (size_t)p * n : nullptr;
    if (st == ST_OPT) {
        // generic: x[p][n]; Chebyshev: x = xc[p][d] (passed as `x`), fun = r[p] (passed as `fun`)
        double xin[VN], xo[VN], f = 0.0;
        int basis[VN + 2];
        if constexpr (KIND == LP_CHEBY) {
            for (int j = 0; j < n - 1; ++j) xin[j] = x[(size_t)p * (n - 1) + j];
            xin[n - 1] = fun[p];
        } else {
            for (int j = 0; j < n; ++j) xin[j] = x[(size_t)p * n + j];
        }
        if (Cert<VN>::basis_from_x(lp, xin, basis) && Cert<VN>::certify(lp, V_OPT, basis, xin, xo, &f)) {
            const bool out = range_rule(lp, V_OPT, f) != V_OPT;
            if constexpr (KIND == LP_CHEBY) {
                for (int j = 0; j < n - 1; ++j) x[(size_t)p * (n - 1) + j] = out ? qnan() : xo[j];
                fun[p] = out ? qnan() : xo[n - 1];
            } else {
                for (int j = 0; j < n; ++j) x[(size_t)p * n + j] = out ? qnan() : xo[j];
                fun[p] = out ? qnan() : f;
            }
            if (out) status[p] = ST_UNBND;
            return;
        }
    }
    push(sc, (int)p);
}

// ---- the fused bounding boxes: value per side, basis per side, centre per polytope
template <int VN>
__global__ __launch_bounds__(VBLK) void verify_box_kernel(long long B, int m_max, int d, const double* __restrict__ A,
                                                          const double* __restrict__ b, const int* __restrict__ mrows,
                                                          double* __restrict__ lb, double* __restrict__ ub,
                                                          const int* __restrict__ status,
                                                          const signed char* __restrict__ basis8,
                                                          const double* __restrict__ centre,
                                                          const double* __restrict__ xfin, Scratch sc) {
    const long long t = (long long)blockIdx.x * VBLK + threadIdx.x;
    if (t >= B * 2 * d) return;
    const long long p = t / (2 * d);
    const int side = (int)(t - p * 2 * d);  // 2k: lower_k, 2k + 1: upper_k
    if (status[p] != 0) return;             // (handed to the caller's generic LPs, which pass verify_x_kernel)
    const int k = side >> 1;
    double* out = (side & 1) ? ub : lb;
    const double val = out[p * d + k];
    LpView lp;
    lp.m = mrows ? mrows[p] : m_max;
    lp.n = d;
    lp.kind = LP_BOXSIDE;
    lp.side = side;
    lp.G = A + (size_t)p * m_max * d;
    lp.h = b + (size_t)p * m_max;
    lp.c = nullptr;
    if (fabs(val) < 1e300) {
        int basis[VN + 2];
        double xo[VN], f = 0.0;
        bool have = true;
        const double* xref;
        if (xfin) {  // the one-LP-per-lane kernel: the point its walk ended on; the basis is read off it
            xref = xfin + ((size_t)p * 2 * d + side) * d;
            have = Cert<VN>::basis_from_x(lp, xref, basis);
        } else {
            const signed char* bs = basis8 + ((size_t)p * 2 * d + side) * d;
            for (int j = 0; j < d; ++j) basis[j] = bs[j];
            xref = centre + (size_t)p * d;
        }
        if (have && Cert<VN>::certify(lp, V_OPT, basis, xref, xo, &f)) {
            const bool oor = range_rule(lp, V_OPT, f) != V_OPT;
            const double pinf = __longlong_as_double(0x7ff0000000000000ll);
            out[p * d + k] = oor ? ((side & 1) ? pinf : -pinf) : xo[k];
            return;
        }
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
            st = range_rule(lp, st, f);
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
            st = range_rule(lp, st, f);
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

__global__ void verify_reset_kernel(unsigned* count) { *count = 0u; }

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

#define PLP_VN_DISPATCH(nn, CALL5, CALL9, CALL17) \
    do {                                          \
        if ((nn) <= 5) { CALL5; }                 \
        else if ((nn) <= 9) { CALL9; }            \
        else { CALL17; }                          \
    } while (0)

int launch_verify_lp(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
                     double* x, double* fun, int* status, void* scratch, hipStream_t st) {
    if (B < 1 || n < 1 || n > VNMAX || B > 2147483647ll) return 1;
    const Scratch sc = carve(scratch, B, m_max);
    hipLaunchKernelGGL(verify_reset_kernel, dim3(1), dim3(1), 0, st, sc.count);
    const dim3 grid((unsigned)((B + VBLK - 1) / VBLK));
    PLP_VN_DISPATCH(n,
                    hipLaunchKernelGGL((verify_x_kernel<LP_GENERIC, 5>), grid, dim3(VBLK), 0, st, B, m_max, n, c, G, h, mrows, x, fun, status, sc),
                    hipLaunchKernelGGL((verify_x_kernel<LP_GENERIC, 9>), grid, dim3(VBLK), 0, st, B, m_max, n, c, G, h, mrows, x, fun, status, sc),
                    hipLaunchKernelGGL((verify_x_kernel<LP_GENERIC, 17>), grid, dim3(VBLK), 0, st, B, m_max, n, c, G, h, mrows, x, fun, status, sc));
    hipLaunchKernelGGL((careful_kernel<LP_GENERIC>), dim3((unsigned)(CAREFUL_SLOTS / VBLK)), dim3(VBLK), 0, st, m_max, n, c, G, h,
                       mrows, x, fun, status, (double*)nullptr, sc);
    return 0;
}

int launch_verify_cheby(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* r,
                        double* xc, int* status, void* scratch, hipStream_t st) {
    const int n = d + 1;
    if (B < 1 || d < 1 || n > VNMAX || B > 2147483647ll) return 1;
    const Scratch sc = carve(scratch, B, m_max);
    hipLaunchKernelGGL(verify_reset_kernel, dim3(1), dim3(1), 0, st, sc.count);
    const dim3 grid((unsigned)((B + VBLK - 1) / VBLK));
    PLP_VN_DISPATCH(n,
                    hipLaunchKernelGGL((verify_x_kernel<LP_CHEBY, 5>), grid, dim3(VBLK), 0, st, B, m_max, n, (const double*)nullptr, A, b, mrows, xc, r, status, sc),
                    hipLaunchKernelGGL((verify_x_kernel<LP_CHEBY, 9>), grid, dim3(VBLK), 0, st, B, m_max, n, (const double*)nullptr, A, b, mrows, xc, r, status, sc),
                    hipLaunchKernelGGL((verify_x_kernel<LP_CHEBY, 17>), grid, dim3(VBLK), 0, st, B, m_max, n, (const double*)nullptr, A, b, mrows, xc, r, status, sc));
    hipLaunchKernelGGL((careful_kernel<LP_CHEBY>), dim3((unsigned)(CAREFUL_SLOTS / VBLK)), dim3(VBLK), 0, st, m_max, n,
                       (const double*)nullptr, A, b, mrows, xc, r, status, (double*)nullptr, sc);
    return 0;
}

int launch_verify_box(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* lb, double* ub,
                      int* status, const signed char* basis8, const double* centre, const double* xfin, void* scratch,
                      hipStream_t st) {
    const long long T = B * 2 * d;
    if (B < 1 || d < 1 || d > MAX_D || T > 2147483647ll) return 1;
    const Scratch sc = carve(scratch, T, m_max);
    hipLaunchKernelGGL(verify_reset_kernel, dim3(1), dim3(1), 0, st, sc.count);
    const dim3 grid((unsigned)((T + VBLK - 1) / VBLK));
    PLP_VN_DISPATCH(d,
                    hipLaunchKernelGGL((verify_box_kernel<5>), grid, dim3(VBLK), 0, st, B, m_max, d, A, b, mrows, lb, ub, status, basis8, centre, xfin, sc),
                    hipLaunchKernelGGL((verify_box_kernel<9>), grid, dim3(VBLK), 0, st, B, m_max, d, A, b, mrows, lb, ub, status, basis8, centre, xfin, sc),
                    hipLaunchKernelGGL((verify_box_kernel<17>), grid, dim3(VBLK), 0, st, B, m_max, d, A, b, mrows, lb, ub, status, basis8, centre, xfin, sc));
    hipLaunchKernelGGL((careful_kernel<LP_BOXSIDE>), dim3((unsigned)(CAREFUL_SLOTS / VBLK)), dim3(VBLK), 0, st, m_max, d,
                       (const double*)nullptr, A, b, mrows, lb, (double*)nullptr, status, ub, sc);
    return 0;
}

}  // namespace plp
