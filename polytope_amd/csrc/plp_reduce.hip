// plp_reduce.hip -- fused redundancy removal for a packed batch of H-polytopes (gfx950).
//
// Reference behaviour restated (polytope/polytope.py:1053-1163, `reduce`), per polytope:
//   1. is_fulldim -> cheby_ball: LP F1, r > abs_tol                      (:1081, :962-985, :1241-1300)
//   2. pairwise parallel-row dedupe on unit rows                          (:1094-1112)
//   3. early return when neq <= nx+1                                      (:1114-1116)
//   4. if neq > 3 nx: bounding box = 2 nx LPs F3, prefilter rows          (:1118-1134, :1367-1409)
//   5. early return when neq <= nx+1                                      (:1136-1138)
//   6. one redundancy LP F2 per remaining row k (h[k] += 0.1 ... -= 0.1)  (:1142-1160)
// Output: 64-bit keep mask over the INPUT rows, flags, Chebyshev ball, number of LPs solved.
//
// Mapping: a 256-thread workgroup takes a tile of NG = 256/GS polytopes (GS lanes per LP
// group, GS >= rows).  The tile's rows are read from HBM once, coalesced, into LDS; every LP
// of the tile is then solved out of LDS by whichever group is free:
//   phase A  group p solves F1 of polytope p, then dedupes              (NG LPs)
//   phase B  the 2*nx F3 LPs of all polytopes that need a box           (work list, NG at a time)
//   phase C  the F2 LPs of all surviving rows                           (work list, NG at a time)
// F2/F3 start from the dictionary shifted to the Chebyshev centre (b - A xc > 0), so they
// need no phase 1.  HBM traffic = 8 m (d+1) bytes in + 12 + 8(d+1) + 4 bytes out per polytope.
#include "plp_kernels.hpp"
#include "plp_simplex.hpp"

namespace plp {

struct ReduceSmem {
    double* A;    // [NG][gs][D]
    double* b;    // [NG][gs]
    double* an;   // [NG][gs]   1/||a_i||
    double* xc;   // [NG][D]
    double* lb;   // [NG][D]
    double* ub;   // [NG][D]
    double* r;    // [NG]
    unsigned long long* live;  // [NG]
    unsigned long long* keep;  // [NG]
    int* flags;   // [NG]
    int* nlp;     // [NG]
    int* stage;   // [NG] 0 done, 1 needs box, 2 needs F2
    int* list;    // [NG*max(gs,2D)]
    int* count;   // [1]
};

static inline size_t reduce_smem_bytes(int gs, int D) {
    const int NG = BLOCK / gs;
    const int per = gs > 2 * D ? gs : 2 * D;
    size_t dbl = (size_t)NG * gs * D + 2 * (size_t)NG * gs + 3 * (size_t)NG * D + NG;
    size_t bytes = dbl * 8 + 2 * (size_t)NG * 8 + 3 * (size_t)NG * 4 + (size_t)NG * per * 4 + 16;
    return (bytes + 15) & ~(size_t)15;
}

template <int D>
__global__ __launch_bounds__(BLOCK) void reduce_kernel(long long B, int m_max, int gs,
                                                       const double* __restrict__ Ag,
                                                       const double* __restrict__ bg,
                                                       const int* __restrict__ mrows, double abs_tol,
                                                       unsigned long long* __restrict__ keep_out,
                                                       int* __restrict__ flags_out,
                                                       double* __restrict__ r_out,
                                                       double* __restrict__ xc_out,
                                                       int* __restrict__ nlp_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const Grp g(gs);
    const int NG = BLOCK / gs;
    const int gib = threadIdx.x / gs;
    const int i = g.gl;
    ReduceSmem sm;
    {
        double* p = reinterpret_cast<double*>(smem_raw);
        sm.A = p;  p += (size_t)NG * gs * D;
        sm.b = p;  p += (size_t)NG * gs;
        sm.an = p; p += (size_t)NG * gs;
        sm.xc = p; p += (size_t)NG * D;
        sm.lb = p; p += (size_t)NG * D;
        sm.ub = p; p += (size_t)NG * D;
        sm.r = p;  p += NG;
        sm.live = reinterpret_cast<unsigned long long*>(p);
        sm.keep = sm.live + NG;
        sm.flags = reinterpret_cast<int*>(sm.keep + NG);
        sm.nlp = sm.flags + NG;
        sm.stage = sm.nlp + NG;
        sm.list = sm.stage + NG;
        sm.count = sm.list + NG * (gs > 2 * D ? gs : 2 * D);
    }
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);

    for (long long tile = (long long)blockIdx.x * NG; tile < B; tile += (long long)gridDim.x * NG) {
        const int ntile = (B - tile) < NG ? (int)(B - tile) : NG;
        // ---------------------------------------------------------------- stage rows in LDS
        {
            const int rowsz = m_max * D;
            const int totA = ntile * rowsz;
            const double* src = Ag + tile * rowsz;
            for (int idx = threadIdx.x; idx < totA; idx += BLOCK) {
                const int p = idx / rowsz, rem = idx - p * rowsz;
                const int row = rem / D, k = rem - row * D;
                sm.A[((size_t)p * gs + row) * D + k] = src[idx];
            }
            const int totb = ntile * m_max;
            const double* srcb = bg + tile * m_max;
            for (int idx = threadIdx.x; idx < totb; idx += BLOCK) {
                const int p = idx / m_max, row = idx - p * m_max;
                sm.b[p * gs + row] = srcb[idx];
            }
        }
        __syncthreads();
        // ---------------------------------------------------------------- phase A: F1 + dedupe
        const long long pg = tile + gib;
        const bool valid = gib < ntile;
        const int m = valid ? (mrows ? mrows[pg] : m_max) : 0;
        const bool has_row = valid && i < m && m <= gs;
        double a[D];
        double bi = 0.0, an_i = 0.0;
        {
            Simplex<D + 1, false> S;
            S.reset(D + 1, m, i);
            bool finite = true;
            double nrm2 = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                a[k] = has_row ? sm.A[((size_t)gib * gs + i) * D + k] : 0.0;
                S.T[k] = a[k];
                nrm2 = nrm2 + a[k] * a[k];
                finite = finite && isfinite(a[k]);
            }
            bi = has_row ? sm.b[gib * gs + i] : 0.0;
            finite = finite && isfinite(bi);
            const double nrm = sqrt(nrm2);
            an_i = 1.0 / nrm;
            const bool zero = !(nrm > 0.0);
            S.T[D] = nrm;
            S.beta = bi;
            S.rowact = has_row && !zero;
            if (!S.rowact) { S.beta = 0.0; S.T[D] = 0.0; }
            const bool infeasible0 = grp_ballot(has_row && zero && bi < -TOL_FEAS, g) != 0;
            const bool bad = grp_ballot(!finite, g) != 0 || m > gs;
            S.cost[D] = -1.0;
            S.mode = M_INIT;
            S.init_col = D;
            S.init_q = bi / nrm;
            S.init_elig = S.rowact;
            S.mode_after_init = M_P2;
            if (!valid || bad) { S.mode = M_DONE; S.status = ST_NUM; }
            else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }
            S.run(g);

            const bool ok = S.status == ST_OPT;
            const double mine = S.x_value();
            const bool holds = S.holds_x();
            double rr = 0.0;
#pragma unroll
            for (int j = 0; j <= D; ++j) {
                const uint64_t ob = grp_ballot(holds && S.rowvar == j, g);
                const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
                const double xj = ob ? v : 0.0;
                if (j < D) { if (valid && i == 0) sm.xc[gib * D + j] = xj; }
                else rr = xj;
            }
            const bool ball = ok && rr >= 0.0;  // cheby_ball: status 0 and r >= 0 (:1289-1293)
            const bool fulldim = ball && rr > abs_tol;
            if (valid && i == 0) {
                sm.r[gib] = ball ? rr : 0.0;
                sm.flags[gib] = fulldim ? 0 : RF_EMPTY;
                sm.nlp[gib] = 1;
                sm.keep[gib] = 0ull;
                sm.stage[gib] = fulldim ? -1 : 0;  // -1: decided after dedupe
                if (!ball) {
#pragma unroll
                    for (int j = 0; j < D; ++j) sm.xc[gib * D + j] = qnan;
                }
            }
            if (has_row) sm.an[gib * gs + i] = an_i;
        }
        __syncthreads();
        {
            // dedupe (:1094-1110): unit rows with dot > 1 - abs_tol are the same hyperplane;
            // of a pair (p<q) the one with the larger normalised offset goes, ties drop p.
            bool removed = false;
            double ni[D];
#pragma unroll
            for (int k = 0; k < D; ++k) ni[k] = a[k] * an_i;
            const double bin_ = bi * an_i;
            for (int j = 0; j < m_max; ++j) {
                const bool jrow = valid && j < m;
                const double an_j = jrow ? sm.an[gib * gs + j] : 0.0;
                double dot = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double ajk = jrow ? sm.A[((size_t)gib * gs + j) * D + k] : 0.0;
                    dot = dot + ni[k] * (ajk * an_j);
                }
                const double bjn = (jrow ? sm.b[gib * gs + j] : 0.0) * an_j;
                const bool par = has_row && jrow && j != i && (dot > 1.0 - abs_tol);
                if (par) {
                    if (i < j) removed = removed || !(bin_ < bjn);
                    else removed = removed || (bjn < bin_);
                }
            }
            const uint64_t live = grp_ballot(has_row && !removed, g);
            const int neq = __popcll(live);
            if (valid && i == 0 && sm.stage[gib] == -1) {
                if (neq <= D + 1) { sm.flags[gib] = RF_EARLY; sm.keep[gib] = live; sm.stage[gib] = 0; }
                else sm.stage[gib] = (neq > 3 * D) ? 1 : 2;
            }
            if (valid && i == 0) sm.live[gib] = live;
            if (!valid && i == 0) { sm.stage[gib] = 0; sm.live[gib] = 0ull; }
        }
        __syncthreads();
        // ---------------------------------------------------------------- phase B: bounding boxes
        {
            unsigned need = 0u;
            for (int p = 0; p < NG; ++p) need |= (sm.stage[p] == 1 ? 1u : 0u) << p;
            const int total = 2 * D * __popc(need);
            if (sm.stage[gib] == 1) {
                const int off = 2 * D * __popc(need & ((1u << gib) - 1u));
                for (int k = i; k < 2 * D; k += gs) sm.list[off + k] = (gib << 8) | k;
            }
            __syncthreads();
            for (int it = 0; it * NG < total; ++it) {
                const int item = it * NG + gib;
                const bool iv = item < total;
                const int code = iv ? sm.list[item] : 0;
                const int p = code >> 8, k = code & 255;
                const uint64_t live = iv ? sm.live[p] : 0ull;
                const bool lrow = (live >> i) & 1ull;
                Simplex<D, false> S;
                S.reset(D, __popcll(live), i);
                double s = 0.0;
#pragma unroll
                for (int kk = 0; kk < D; ++kk) {
                    const double av = lrow ? sm.A[((size_t)p * gs + i) * D + kk] : 0.0;
                    S.T[kk] = av;
                    s = fma(av, iv ? sm.xc[p * D + kk] : 0.0, s);
                    S.cost[kk] = iv ? ((kk == (k < D ? k : k - D)) ? (k < D ? 1.0 : -1.0) : 0.0) : 0.0;
                }
                const double bsh = (lrow ? sm.b[p * gs + i] : 0.0) - s;
                S.beta = bsh > 0.0 ? bsh : 0.0;
                S.rowact = lrow;
                S.mode = iv ? M_P2 : M_DONE;
                S.run(g);
                const int kk = k < D ? k : k - D;
                const uint64_t ob = grp_ballot(S.holds_x() && S.rowvar == kk, g);
                const double v = bcast(S.x_value(), g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
                if (iv && i == 0) {
                    double val;
                    if (S.status == ST_OPT) val = sm.xc[p * D + kk] + (ob ? v : 0.0);
                    else if (S.status == ST_UNBND) val = (k < D) ? -pinf : pinf;
                    else { val = qnan; atomicOr(&sm.flags[p], RF_LPFAIL); }
                    if (k < D) sm.lb[p * D + kk] = val; else sm.ub[p * D + kk] = val;
                }
            }
        }
        __syncthreads();
        // ---------------------------------------------------------------- prefilter (:1131-1134)
        if (sm.stage[gib] == 1) {
            const uint64_t live = sm.live[gib];
            const bool lrow = (live >> i) & 1ull;
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double lbk = sm.lb[gib * D + k], ubk = sm.ub[gib * D + k];
                const double pa = (a[k] > 0.0 ? 1.0 : 0.0) * a[k];
                s1 = s1 + pa * (ubk - lbk);
                s2 = s2 + a[k] * lbk;
            }
            const bool out = (s1 - (bi - s2)) < -1e-4;
            const uint64_t live2 = live & ~grp_ballot(lrow && out, g);
            if (i == 0) {
                sm.nlp[gib] += 2 * D;
                sm.live[gib] = live2;
                if (__popcll(live2) <= D + 1) {
                    sm.flags[gib] |= RF_EARLY; sm.keep[gib] = live2; sm.stage[gib] = 0;
                } else sm.stage[gib] = 2;
            }
        }
        __syncthreads();
        // ---------------------------------------------------------------- phase C: redundancy LPs
        {
            int myoff = 0, total = 0;
            for (int p = 0; p < NG; ++p) {
                const int c = (sm.stage[p] == 2) ? __popcll(sm.live[p]) : 0;
                if (p < gib) myoff += c;
                total += c;
            }
            if (sm.stage[gib] == 2) {
                const uint64_t live = sm.live[gib];
                if ((live >> i) & 1ull) sm.list[myoff + __popcll(live & ((1ull << i) - 1ull))] = (gib << 8) | i;
            }
            __syncthreads();
            for (int it = 0; it * NG < total; ++it) {
                const int item = it * NG + gib;
                const bool iv = item < total;
                const int code = iv ? sm.list[item] : 0;
                const int p = code >> 8, k = code & 255;
                const uint64_t live = iv ? sm.live[p] : 0ull;
                const bool lrow = (live >> i) & 1ull;
                Simplex<D, false> S;
                S.reset(D, __popcll(live), i);
                double s = 0.0, cxc = 0.0;
                double cc[D];
#pragma unroll
                for (int kk = 0; kk < D; ++kk) {
                    const double av = lrow ? sm.A[((size_t)p * gs + i) * D + kk] : 0.0;
                    S.T[kk] = av;
                    s = fma(av, iv ? sm.xc[p * D + kk] : 0.0, s);
                    cc[kk] = iv ? -sm.A[((size_t)p * gs + k) * D + kk] : 0.0;  // f = -A[k,:]  (:1145)
                    S.cost[kk] = cc[kk];
                }
                // h[k] += 0.1 for this LP; rows k' < k carry the (+0.1, -0.1) round trip (:1149-1151)
                const double b0 = lrow ? sm.b[p * gs + i] : 0.0;
                const double bup = b0 + 0.1;
                const double brt = bup - 0.1;
                const double beff = (i < k) ? brt : ((i == k) ? bup : b0);
                const double bsh = beff - s;
                S.beta = bsh > 0.0 ? bsh : 0.0;
                S.rowact = lrow;
                S.mode = iv ? M_P2 : M_DONE;
                S.run(g);
                const double mine = S.x_value();
                const bool holds = S.holds_x();
                double fun = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const uint64_t ob = grp_ballot(holds && S.rowvar == j, g);
                    const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
                    const double xj = (iv ? sm.xc[p * D + j] : 0.0) + (ob ? v : 0.0);
                    fun = fma(cc[j], xj, fun);
                }
                (void)cxc;
                const double bk = iv ? sm.b[p * gs + k] : 0.0;
                const double hk = (bk + 0.1) - 0.1;
                const double obj = -fun - hk;  // (:1156)
                const bool keepk = (S.status == ST_OPT && obj > abs_tol) || S.status == ST_UNBND;
                if (iv && i == 0 && keepk) atomicOr(&sm.keep[p], 1ull << k);
            }
        }
        __syncthreads();
        // ---------------------------------------------------------------- results
        if (valid && i == 0) {
            int fl = sm.flags[gib];
            int nl = sm.nlp[gib];
            if (sm.stage[gib] == 2) { fl |= RF_MINREP; nl += __popcll(sm.live[gib]); }
            keep_out[pg] = sm.keep[gib];
            flags_out[pg] = fl;
            nlp_out[pg] = nl;
            r_out[pg] = sm.r[gib];
        }
        if (valid) {
            for (int k = i; k < D; k += gs) xc_out[pg * D + k] = sm.xc[gib * D + k];
        }
        __syncthreads();
    }
}

template <int D>
static int launch_reduce_d(long long B, int m_max, int gs, const double* A, const double* b, const int* mrows,
                           double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                           hipStream_t st) {
    const size_t smem = reduce_smem_bytes(gs, D);
    const long long NG = BLOCK / gs;
    long long blocks = (B + NG - 1) / NG;
    if (blocks > 256ll * 16) blocks = 256ll * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(reduce_kernel<D>, dim3((unsigned)blocks), dim3(BLOCK), smem, st, B, m_max, gs, A, b, mrows,
                       abs_tol, keep, flags, r, xc, nlp);
    return 0;
}

#define PLP_CASE_R(K) \
    case K: return launch_reduce_d<K>(B, m_max, gs, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);

int launch_reduce(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double abs_tol,
                  unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st) {
    const int gs = group_size_for(m_max);
    if (gs < 0 || d < 1 || d > MAX_D) return 2;
    switch (d) {
        PLP_CASE_R(1) PLP_CASE_R(2) PLP_CASE_R(3) PLP_CASE_R(4) PLP_CASE_R(5) PLP_CASE_R(6)
        PLP_CASE_R(7) PLP_CASE_R(8) PLP_CASE_R(9) PLP_CASE_R(10) PLP_CASE_R(11) PLP_CASE_R(12)
        PLP_CASE_R(13) PLP_CASE_R(14) PLP_CASE_R(15) PLP_CASE_R(16)
        default: return 2;
    }
}

}  // namespace plp
