// plp_wide.hpp -- the one-LP-per-wavefront simplex engine (see plp_wide.hip for the design notes): wave reductions, the
// register-vector row, the per-wavefront LDS block and wide_run().  Shared by cheby_w_kernel (plp_wide.hip) and the F1
// stage of reduce_lazy_kernel (plp_reduce_r_impl.hpp).
#pragma once
#include "plp_kernels.hpp"
#include "plp_wave.hpp"
#include <type_traits>

namespace plp {
namespace wide {

#ifndef PLP_WIDE_CREG
#define PLP_WIDE_CREG 1
#endif
#ifndef PLP_WIDE_ROW9_V8
#define PLP_WIDE_ROW9_V8 1
#endif
#ifndef PLP_WIDE_PEEL
#define PLP_WIDE_PEEL 2
#endif
#ifndef PLP_WIDE_DPP1
#define PLP_WIDE_DPP1 1
#endif
#ifndef PLP_WIDE_SPLITLOAD
#define PLP_WIDE_SPLITLOAD 0
#endif
#ifndef PLP_WIDE_BPERM
#define PLP_WIDE_BPERM 0
#endif
#define PLP_DPP_BCAST15 0x142
#define PLP_DPP_BCAST31 0x143

// Ordering point for the wavefront's own LDS block (one wavefront writes and reads it: the LDS unit serves a wavefront's
// operations in issue order, what is needed is that the compiler keeps them in program order).  PLP_WIDE_WAVESYNC=0: the
// workgroup barrier -- the same thing in a 64-thread workgroup, and wrong in reduce_wsplit_kernel, whose wavefronts run
// different LPs.
#ifndef PLP_WIDE_WAVESYNC
#define PLP_WIDE_WAVESYNC 1
#endif
__device__ __forceinline__ void wave_sync() {
#if PLP_WIDE_WAVESYNC
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
    __syncthreads();
#endif
}
__device__ __forceinline__ int wave_min_i32(int v) {
    int t;
    t = __builtin_amdgcn_update_dpp(v, v, PLP_DPP_XOR1, 0xF, 0xF, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, PLP_DPP_XOR2, 0xF, 0xF, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, PLP_DPP_HMIRROR, 0xF, 0xF, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, PLP_DPP_MIRROR, 0xF, 0xF, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, PLP_DPP_BCAST15, 0xA, 0xF, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, PLP_DPP_BCAST31, 0xC, 0xF, false); v = t < v ? t : v;
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double uniform_lane(double v, int lane) {  // value of `lane` (wave-uniform index)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// wave-wide min / max of a u32 with the DPP operand folded into the VALU op (one instruction per level; hipcc does
// not fold v_mov_dpp into the consumer by itself); result wave-uniform (lane 63 holds it after the row_bcast levels)
#define PLP_W_DPP(OP, v, CTRL, RM) asm volatile("s_nop 1\n\t" OP " %0, %0, %0 " CTRL " row_mask:" RM " bank_mask:0xf" : "+v"(v))
#if PLP_WIDE_DPP1
// the six levels and the read-out as ONE asm statement: the hazard nops are written once (a VALU write needs two wait
// states before a DPP read of the same register); statement by statement the compiler adds its own s_nop in front of
// every one of them (12 scalar instructions per reduction, two reductions per pivot)
#define PLP_W_REDUCE(OP, v, res)                                                                     \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   \
                 "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"   \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"       \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"            \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"          \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"          \
                 "s_nop 0\n\tv_readlane_b32 %1, %0, 63"                                              \
                 : "+v"(v), "=s"(res))
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    unsigned res;
    PLP_W_REDUCE("v_min_u32_dpp", v, res);
    return res;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    unsigned res;
    PLP_W_REDUCE("v_max_u32_dpp", v, res);
    return res;
}
// the same over the first N lanes only (the pricing: lane j = column j, the lanes beyond hold 0): three DPP levels cover
// 8 lanes, four cover 16, every one of those lanes ends with the maximum and lane 0 is read
template <int N>
__device__ __forceinline__ unsigned low_max_u32(unsigned v) {
    if constexpr (N <= 8) {
        unsigned res;
        asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 0\n\tv_readlane_b32 %1, %0, 0"
                     : "+v"(v), "=s"(res));
        return res;
    } else if constexpr (N <= 16) {
        unsigned res;
        asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 0\n\tv_readlane_b32 %1, %0, 0"
                     : "+v"(v), "=s"(res));
        return res;
    } else {
        return wave_max_u32(v);
    }
}
#else
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    PLP_W_DPP("v_min_u32_dpp", v, "quad_perm:[1,0,3,2]", "0xf");
    PLP_W_DPP("v_min_u32_dpp", v, "quad_perm:[2,3,0,1]", "0xf");
    PLP_W_DPP("v_min_u32_dpp", v, "row_half_mirror", "0xf");
    PLP_W_DPP("v_min_u32_dpp", v, "row_mirror", "0xf");
    PLP_W_DPP("v_min_u32_dpp", v, "row_bcast:15", "0xa");
    PLP_W_DPP("v_min_u32_dpp", v, "row_bcast:31", "0xc");
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    PLP_W_DPP("v_max_u32_dpp", v, "quad_perm:[1,0,3,2]", "0xf");
    PLP_W_DPP("v_max_u32_dpp", v, "quad_perm:[2,3,0,1]", "0xf");
    PLP_W_DPP("v_max_u32_dpp", v, "row_half_mirror", "0xf");
    PLP_W_DPP("v_max_u32_dpp", v, "row_mirror", "0xf");
    PLP_W_DPP("v_max_u32_dpp", v, "row_bcast:15", "0xa");
    PLP_W_DPP("v_max_u32_dpp", v, "row_bcast:31", "0xc");
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
template <int N>
__device__ __forceinline__ unsigned low_max_u32(unsigned v) { return wave_max_u32(v); }
#endif
__device__ __forceinline__ double rcpn(double a) {
    const double x0 = __builtin_amdgcn_rcp(a);
    const double x1 = fma(x0, fma(-a, x0, 1.0), x0);
    return fma(x1, fma(-a, x1, 1.0), x1);
}

// The row of a lane: NC <= 17 doubles held as a 16-wide register vector (+ one scalar for the 17th column), so that
// T[e] for a wave-uniform e is ONE indexed register move (s_set_gpr_idx / v_movrel: the compiler lowers a dynamic
// element access with a uniform index that way) instead of a select chain or a branch tree.
typedef double v16d __attribute__((ext_vector_type(16)));
typedef double v8d __attribute__((ext_vector_type(8)));
// (eight doubles are enough up to 8 columns: 16 VGPRs less)
template <int NC> struct RowVec { typedef v16d type; };
template <> struct RowVec<1> { typedef v8d type; };
template <> struct RowVec<2> { typedef v8d type; };
template <> struct RowVec<3> { typedef v8d type; };
template <> struct RowVec<4> { typedef v8d type; };
template <> struct RowVec<5> { typedef v8d type; };
template <> struct RowVec<6> { typedef v8d type; };
template <> struct RowVec<7> { typedef v8d type; };
template <> struct RowVec<8> { typedef v8d type; };
#if PLP_WIDE_ROW9_V8
// nine columns (d = 8 and its F1 / phase-1 column): eight in the vector, the ninth in the scalar T16, as for 17 columns
template <> struct RowVec<9> { typedef v8d type; };
#endif
// (plain local variables, not a struct: the struct form was kept in scratch memory by the compiler)
#define ROW_W ((int)(sizeof(Tv) / sizeof(double)))
#define ROW_GET(j) ((j) < ROW_W ? Tv[(j) & (ROW_W - 1)] : T16)
#define ROW_SET(j, val) do { if ((j) < ROW_W) Tv[(j) & (ROW_W - 1)] = (val); else T16 = (val); } while (0)
template <int NC, class TV>
__device__ __forceinline__ double row_at(const TV& Tv, const double& T16, int e) {  // wave-uniform e
    if constexpr (NC <= ROW_W) return Tv[e & (ROW_W - 1)];
    else return e < ROW_W ? Tv[e & (ROW_W - 1)] : T16;
}
template <int NC, class TV>
__device__ __forceinline__ void row_put(TV& Tv, double& T16, int e, double val) {   // wave-uniform e
    if constexpr (NC <= ROW_W) {
        Tv[e & (ROW_W - 1)] = val;
    } else {  // (written as two selects on single elements: a branch here is if-converted into a select of the whole vector)
        const double old = Tv[e & (ROW_W - 1)];
        Tv[e & (ROW_W - 1)] = e < ROW_W ? val : old;
        T16 = e < ROW_W ? T16 : val;
    }
}

// Per-wavefront LDS: reduced costs, the scaled pivot row, the nonbasic variable of every column
template <int NC>
struct WideShared {
    double cost[NC + 1];
    double rho[NC + 1];   // rho[NC] = scaled right-hand side of the pivot row
    int cv[NC + 1];       // (id + 1) << 1 | negated
};

// Solve from a dictionary whose rows sit in T/beta (one per lane).  `forced`: first pivot = column NC-1 enters, the
// leaving row is the active one with the smallest signed ratio q0 (F1).  Returns the status; the optimal dictionary
// stays in T/beta/rowvar/rowneg.
template <int NC>
__device__ __forceinline__ int wide_run(const int lane, const int m, typename RowVec<NC>::type& Tv, double& T16, double& beta, int& rowvar,
                                        int& rowneg, bool& rowact, WideShared<NC>& sh, const int nfree, bool forced,
                                        const double q0, int& iters_out, double* negz = nullptr) {
    unsigned cfree = nfree >= 32 ? 0xffffffffu : ((1u << nfree) - 1u);
    int ndeg = 0, iters = 0;
    const int maxit = 50 * (m + nfree) + 100;
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    int status = -1;
#if PLP_WIDE_CREG
    // the reduced costs stay in a register (lane j: column j) from the set-up the caller left in sh.cost: one LDS
    // store -> fence -> load round trip less on the chain from one pivot to the next pricing
    double c = lane < NC ? sh.cost[lane < NC ? lane : 0] : 0.0;
#endif
#if PLP_WIDE_PEEL
    // three instances of the pivot -- the forced first one, Dantzig's rule, Bland's rule -- instead of one body that
    // tests `forced` and `ndeg >= BLAND_AFTER` as data at every step (scalar compares, selects and branches of every pivot)
    auto pivot = [&](auto forced_c, auto bland_c) __attribute__((always_inline)) -> bool {
        constexpr bool forced = decltype(forced_c)::value;
#if PLP_WIDE_PEEL == 2
        const bool bland = ndeg >= BLAND_AFTER;   // (PEEL == 2: only the forced pivot is an instance of its own)
#else
        constexpr bool bland = decltype(bland_c)::value;
#endif
#else
    auto pivot = [&]() __attribute__((always_inline)) -> bool {
        const bool bland = ndeg >= BLAND_AFTER;
#endif
        int e;
        double ce;       // reduced cost of the entering column as stored
        bool flip = false;
        if (forced) {
            e = NC - 1;
#if PLP_WIDE_CREG
            ce = uniform_lane(c, NC - 1);
#else
            ce = sh.cost[NC - 1];
#endif
        } else {
            // ---- pricing: lane j looks after column j
#if !PLP_WIDE_CREG
            const double c = lane < NC ? sh.cost[lane] : 0.0;
#endif
            const bool elig = (lane < NC) & (fabs(c) > TOL_D) & ((((cfree >> (lane & 31)) & 1u) != 0u) | (c < 0.0));
            const uint64_t eb = __ballot(elig);
            if (eb == 0) { status = ST_OPT; return false; }
            if (iters >= maxit) { status = ST_ITER; return false; }
            if (!bland) {  // largest |c|: its bit pattern orders like an unsigned integer
                const unsigned kh = elig ? ((unsigned)__double2hiint(c) & 0x7fffffffu) : 0u;
                const unsigned mh = low_max_u32<NC>(kh);
                uint64_t top = __ballot(elig & (kh == mh));
                if (top & (top - 1ull)) {  // several columns share the high word (rare): the low words decide
                    const unsigned kl = (elig & (kh == mh)) ? (unsigned)__double2loint(c) : 0u;
                    const unsigned ml = low_max_u32<NC>(kl);
                    top = __ballot(elig & (kh == mh) & (kl == ml));
                }
                e = __ffsll((long long)top) - 1;
            } else {  // Bland: lowest variable id among the eligible columns
                const int id = elig ? sh.cv[lane] : 0x7fffffff;
                const int idmin = wave_min_i32(id);
                e = __ffsll((long long)__ballot(elig & (id == idmin))) - 1;
            }
            e = __builtin_amdgcn_readfirstlane(e);
            ce = uniform_lane(c, e);
            flip = ce > 0.0;  // free variable entering downwards: x := -x
        }
        // ---- ratio test
        e = __builtin_amdgcn_readfirstlane(e);  // (wave-uniform on both paths; said again where the forced and the priced
                                                // column merge, or the indexed register access below becomes a select chain)
        double a = row_at<NC>(Tv, T16, e);
        a = flip ? -a : a;
        const double pinv = rcpn(a);
        bool erow;
        double q;
        if (forced) { erow = rowact; q = q0; }
        else { erow = rowact & (a > TOL_PIV); q = (beta > 0.0 ? beta : 0.0) * pinv; }
        q = erow ? q : pinf;
        // exact f64 minimum on the order-preserving u64 key (hi dword, then lo dword among the hi-minima)
        const int qh = __double2hiint(q), ql = __double2loint(q);
        const int sm = qh >> 31;
        const unsigned kh = (unsigned)(qh ^ (sm | (int)0x80000000));
        const unsigned kl = (unsigned)(ql ^ sm);
        const unsigned mh = wave_min_u32(kh);
        // one row alone at the minimal high word (the usual case): its low word is the minimum, no second reduction
        const uint64_t hib = __ballot(kh == mh);
        unsigned ml;
        if (hib & (hib - 1ull)) ml = wave_min_u32((kh == mh) ? kl : 0xffffffffu);
        else ml = (unsigned)__builtin_amdgcn_readlane((int)kl, __ffsll((long long)hib) - 1);
        if (mh >= 0xfff00000u) { status = ((mh == 0xfff00000u) & (ml == 0u)) ? ST_UNBND : ST_NUM; return false; }
        const bool tie = erow & (kh == mh) & (kl == ml);
        const int mhs = (int)(mh ^ 0x80000000u);   // (the minimum is >= 0 in a normal pivot; the forced one ignores ndeg)
        const double qmin = __hiloint2double(mhs >= 0 ? mhs : (int)~mh, mhs >= 0 ? (int)ml : (int)~ml);
        int r;
        if (bland & !forced) {  // lowest basic-variable id among the ties
            const int id = tie ? rowvar + 1 : 0x7fffffff;
            const int idmin = wave_min_i32(id);
            r = __ffsll((long long)__ballot(tie & (id == idmin))) - 1;
        } else {
            r = __ffsll((long long)__ballot(tie)) - 1;  // lowest row among ties
        }
        r = __builtin_amdgcn_readfirstlane(r);
        if (!forced) ndeg = (qmin <= DEGEN_EPS) ? ndeg + 1 : 0;
        // ---- the pivot row scales itself in place and goes to LDS
        const double p = uniform_lane(pinv, r);
        const int vin = sh.cv[e];
        const bool efree = (cfree >> e) & 1u;
        const bool is_r = lane == r;
        if (is_r) {   // (LDS stores and scalars only inside the branch: the row vector itself is updated branch-free below)
#pragma unroll
            for (int j = 0; j < NC; ++j) { const double v = ROW_GET(j) * pinv; ROW_SET(j, v); sh.rho[j] = v; }
            beta = beta * pinv;
            sh.rho[NC] = beta;
            sh.cv[e] = ((rowvar + 1) << 1) | rowneg;
            rowvar = (vin >> 1) - 1;
            rowneg = (vin & 1) ^ (flip ? 1 : 0);
            rowact = !efree;  // a free variable never leaves again
        }
#if PLP_WIDE_BPERM
        double rb;
        {
            // the scaled pivot row goes from lane r's registers to every lane through the LDS crossbar (ds_bpermute, no
            // memory): the update does not wait for the store -> load round trip, which only the cost row still needs
            const int raddr = r << 2;
            const double f = is_r ? 0.0 : a;
            rb = bcast_addr(beta, raddr);
#pragma unroll
            for (int j = 0; j < NC; ++j) ROW_SET(j, fma(-f, bcast_addr(ROW_GET(j), raddr), ROW_GET(j)));
            row_put<NC>(Tv, T16, e, is_r ? pinv : -(f * p));
            beta = fma(-f, rb, beta);
        }
        wave_sync();
#else
        wave_sync();
        const double rb = sh.rho[NC];
        {
            // row r: T * (1/a_r)  (f = 0);  every other row: T - a_i * rho  (scale 1)
            // (scaling the pivot row in place inside the branch above -- inline asm, so that it stays a branch -- costs
            // a copy of the row vector: 136 VGPRs instead of 102, three waves per SIMD; not kept)
            const double f = is_r ? 0.0 : a;
#if PLP_WIDE_SPLITLOAD
            // the pivot row is fetched in two halves (the scheduler would otherwise hoist all NC loads: 2 NC registers)
            constexpr int NH = NC > 8 ? (NC + 1) / 2 : NC;
#pragma unroll
            for (int j = 0; j < NH; ++j) ROW_SET(j, fma(-f, sh.rho[j], ROW_GET(j)));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = NH; j < NC; ++j) ROW_SET(j, fma(-f, sh.rho[j], ROW_GET(j)));
#else
#pragma unroll
            for (int j = 0; j < NC; ++j) ROW_SET(j, fma(-f, sh.rho[j], ROW_GET(j)));
#endif
            row_put<NC>(Tv, T16, e, is_r ? pinv : -(f * p));
            beta = fma(-f, rb, beta);
        }
#endif
        // ---- reduced costs (lane j = column j); the entering column is sign-normalised first
        if (negz) *negz = fma(-(flip ? -ce : ce), rb, *negz);  // objective row: -zeta, as SimplexR carries it
#if PLP_WIDE_CREG
        {
            const double fc = flip ? -ce : ce;
            const double cn = (lane == e) ? -(fc * p) : fma(-fc, sh.rho[lane < NC ? lane : NC], c);
            c = lane < NC ? cn : 0.0;
        }
#else
        if (lane < NC) {
            const double fc = flip ? -ce : ce;
            const double cj = sh.cost[lane];
            sh.cost[lane] = (lane == e) ? -(fc * p) : fma(-fc, sh.rho[lane], cj);
        }
#endif
        cfree &= ~(1u << e);
        iters += 1;
        if (forced) {
            if (rowact & (beta < 0.0)) beta = 0.0;  // rounding of the forced pivot
#if !PLP_WIDE_PEEL
            forced = false;
#endif
        }
        wave_sync();
        return true;
    };
#if PLP_WIDE_PEEL
    {
        using T_ = std::integral_constant<bool, true>;
        using F_ = std::integral_constant<bool, false>;
        bool go = true;
        if (forced) go = pivot(T_{}, F_{});
#if PLP_WIDE_PEEL == 2
        while (go) go = pivot(F_{}, F_{});
#else
        while (go) {
            if (ndeg < BLAND_AFTER) go = pivot(F_{}, F_{});
            else go = pivot(F_{}, T_{});
        }
#endif
    }
#else
    while (pivot() && pivot()) {}
#endif
    iters_out = iters;
    return status;
}

// One LP  min c.y  s.t.  A y <= beta  (y free, beta >= 0: the origin is feasible) of a polytope whose rows sit in LDS, one
// per lane (dead rows zeroed there): the dense twin of lazy::solve() -- same arguments, same numbers (the dictionary
// arithmetic is SimplexR's: bitwise the status and -zeta of reduce_r_kernel<D, 64, 1>), Bland's rule inside instead of a
// retry.  cj: lane j < D holds c_j.  `sh`: this wavefront's LDS block.
template <int D>
__device__ __forceinline__ int solve_dense(const int lane, const int nrows, const double* __restrict__ rowsA, const double cj,
                                           const double beta0, const bool act, double& negz, WideShared<D>& sh,
                                           signed char* __restrict__ basis_out = nullptr) {
    typename RowVec<D>::type Tv = (typename RowVec<D>::type)(0.0);
    double T16 = 0.0;
#pragma unroll
    for (int kk = 0; kk < D; ++kk) ROW_SET(kk, rowsA[lane * D + kk]);
    double beta = beta0;
    int rowvar = D + lane, rowneg = 0, iters = 0;
    bool rowact = act;
    wave_sync();  // (the last LP's reads of the block are done)
    if (lane <= D) {
        sh.cost[lane] = lane < D ? cj : 0.0;
        sh.cv[lane] = (lane + 1) << 1;
    }
    wave_sync();
    negz = 0.0;
    const int st = wide_run<D>(lane, nrows, Tv, T16, beta, rowvar, rowneg, rowact, sh, D, false, 0.0, iters, &negz);
    if (basis_out) {  // the final basis, for the verifier (plp_verify.hip): column j holds variable (cv >> 1) - 1
        if (lane < D) {
            const int id = (sh.cv[lane] >> 1) - 1;
            basis_out[lane] = (signed char)(id < D ? -1 - id : id - D);
        }
    }
    return st;
}

}  // namespace wide
}  // namespace plp
