// plp_reduce_r_impl.hpp -- reduce_r_kernel<D,GS,R> and its launcher (included by plp_reduce_r.hip, d <= 8, and
// plp_reduce_r2a/b.hip, d = 9..16): fused reduce() (polytope/polytope.py:1053-1163) with R = 4 rows per lane.
//
// Same pipeline and the same reference steps as plp_reduce.hip (F1 -> dedupe -> 2d F3 -> prefilter
// -> one F2 per surviving row, all LPs of a polytope solved by one lane group out of registers),
// but built on SimplexR (plp_simplex_r.hpp): a polytope with up to 16 / 32 / 64 rows occupies a
// group of 4 / 8 / 16 lanes, lane l holding rows 4l..4l+3.  A wavefront therefore advances 16 / 8 / 4
// polytopes per instruction instead of 4 / 2 / 1, and the per-pivot reductions are DPP quad steps.
// A workgroup (RBLOCK threads) takes NG = RBLOCK/GS polytopes per tile; their rows are read from HBM once,
// coalesced, into LDS (rows of F2's objective and the dedupe partners are read back from there).
#pragma once
#include <stdlib.h>

#include "plp_kernels.hpp"
#include "plp_simplex_r.hpp"
#include "plp_lazy.hpp"

namespace plp {

#ifdef PLP_STAGE_STATS
// debug build only (scripts/debug/stage_stats.sh): wave-level iterations against per-group pivots, per stage
static __device__ unsigned long long plp_stage_stats[16];
__device__ __forceinline__ int stat_wave_max(int v) {
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}
#define PLP_STAT_ADD(i, v) atomicAdd(&plp_stage_stats[i], (unsigned long long)(v))
#endif

constexpr int RR = 4;  // rows per lane (default; R8 variant: 8 rows per lane, groups of 2 lanes for m <= 16)

#ifndef PLP_REDUCE_R_BLOCK
// One wavefront per workgroup: at C2 (100000 polytopes = 6250 wavefronts over 4096 resident slots) the last
// round is spread over the CUs wave by wave instead of in blocks of four (measured 256: 0.306 ms, 128: 0.305,
// 64: 0.299), and the workgroup barriers cost nothing.
#define PLP_REDUCE_R_BLOCK 64
#endif
constexpr int RBLOCK = PLP_REDUCE_R_BLOCK;  // threads per workgroup

static inline int group_size_r(int m_max) {
    if (m_max <= 16) return 4;
    if (m_max <= 32) return 8;
    return 16;
}

// (bench shape: 16 polytopes x 16 rows x 5 doubles = 10 240 B per one-wavefront workgroup, and 16 of them -- four waves per
// SIMD -- are EXACTLY the CU's 160 KB: 384 B more per workgroup (the centres kept in LDS, tried in round 4 against the
// spills) and a CU holds 15, 0.194 -> 0.213 ms)
static inline size_t reduce_r_smem_bytes(int gs, int D, int R) {
    const int NG = RBLOCK / gs;
    return ((size_t)NG * gs * R * (D + 2) * 8 + 15) & ~(size_t)15;  // A rows, b, 1/||a||
}

// bit l of x (l < 16)  ->  bit 4l
__device__ __forceinline__ uint64_t spread4(uint64_t x) {
    x = (x | (x << 24)) & 0x000000ff000000ffull;
    x = (x | (x << 12)) & 0x000f000f000f000full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    x = (x | (x << 3)) & 0x1111111111111111ull;
    return x;
}

// bit l of x (l < GS)  ->  bit R*l
template <int R, int GS>
__device__ __forceinline__ uint64_t spread_rows(uint64_t x) {
    if constexpr (R == 4) return spread4(x);
    if constexpr (R == 1) return x;
    uint64_t out = 0ull;
#pragma unroll
    for (int l = 0; l < GS; ++l) out |= ((x >> l) & 1ull) << (R * l);
    return out;
}


// In-run count of the LPs that ran the simplex (plp_reduce_counters): the tile's sum of `v` over its lane-group leaders
// goes to the device counter with one atomic (ctr is a kernel argument: the branch is wave-uniform).  Two's-complement
// adds: negative contributions (the LPs the presolve settles, subtracted from the reference count) wrap as they should.
__device__ __forceinline__ void ctr_add(unsigned long long* ctr, int v, bool leader) {
    if (ctr) {
        int s = leader ? v : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        // one of PLP_CTR_SLOTS words, 64 B apart, picked by the workgroup: a single word took every tile of a launch
        // through one L2 atomic unit (measured: +17 % on the bench kernel); the reader sums the slots
        if ((threadIdx.x & 63) == 0)
            atomicAdd(ctr + (size_t)(blockIdx.x & (PLP_CTR_SLOTS - 1)) * 8, (unsigned long long)(long long)s);
    }
}

#ifndef PLP_REDUCE_R_WAVES
// Waves per SIMD the register allocator must leave room for.  Measured at d=3 (100k polytopes, m=16)
// with the fast pivot path (no general engine in this kernel: 132 VGPRs unconstrained):
// 3 waves 0.362 ms, 4 waves (128 VGPRs, 12 B/lane spilled outside the pivot loops) 0.347 ms.
// d=4 would spill 148 B/lane at 4 waves; d>=5 keeps the whole dictionary of 4 rows x (d+1) only at 1-2.
#define PLP_REDUCE_R_WAVES(D) ((D) <= 3 ? 4 : ((D) <= 4 ? 3 : 1))
#endif

#ifndef PLP_REDUCE_R2MID_WAVES
#define PLP_REDUCE_R2MID_WAVES 3  // d = 5..8 on two rows per lane (plp_reduce_r2c.hip)
#endif

#ifndef PLP_R_ASYNC
#define PLP_R_ASYNC 1  // F2: every lane group walks its own LP list inside one pivot loop (0: lock-step, for A/B runs)
#endif

#ifndef PLP_R_POOL
#define PLP_R_POOL 1  // F2: the LPs the presolve left form one list per tile, any lane group takes the next (0: every group its own rows)
#endif

#ifndef PLP_R_FAST
#define PLP_R_FAST 1  // F2/F3 on SimplexR::run_fast (0: the general step(), for A/B runs)
#endif

#ifndef PLP_REDUCE_R8_MAXD
// m <= 16 and d <= this: 8 rows per lane, two lanes per polytope (32 polytopes per wavefront).  Measured at C2:
// 22 % fewer VALU instructions per polytope but 228 VGPRs = 2 waves per SIMD, and the kernel is then bound by the
// latency of the pivot's dependency chain: 0.344 ms against 0.297 ms (at 3 waves it spills: 0.444 ms).  Off.
#define PLP_REDUCE_R8_MAXD 0
#endif
#ifndef PLP_REDUCE_R8_WAVES
#define PLP_REDUCE_R8_WAVES 2
#endif

#ifndef PLP_R_DEDUPE_HALF
#define PLP_R_DEDUPE_HALF 1  // 16 rows at four per lane: every pair of rows once (0: every lane all 16 partners of its rows)
#endif

#ifndef PLP_R_PRESOLVE
#define PLP_R_PRESOLVE 2  // F2: rows answered by the presolve below skip the simplex (0: every LP on the simplex; 1: first witness only; A/B runs)
#endif

// F2 presolve ("ray certificate").  The redundancy LP of row k (:1142-1160) is
//     max a_k.x  s.t.  A x <= b  with  b_k relaxed by 0.1,      keep row k  <=>  optimum - b_k > abs_tol  (or unbounded),
// and `reduce` reads nothing of it but that verdict.  A feasible point of the relaxed LP whose objective exceeds
// b_k + abs_tol by a safety margin therefore settles the verdict "keep" without a simplex run (what a presolve does; an
// unbounded LP is "keep" as well, so boundedness does not matter).  The witness: from the Chebyshev centre xc (strictly
// inside, slack s_i = b_i - a_i.xc > 0) along a_k to the point  x* = xc + t a_k,  t = (s_k + tau) / |a_k|^2,  i.e. tau
// beyond row k's own plane.  x* is feasible iff  t (a_i.a_k) <= s_i  for every other live row i, and
// tau <= 0.1 keeps it inside the relaxed row k.  Division-free:  (s_k + tau) (a_i.a_k) <= s_i |a_k|^2.
// tau = abs_tol + 1e-9 (1 + |b_k| + s_k): nine orders of magnitude above the rounding of either side.
// Measured on random H-polytopes: 61 % of the F2 LPs at (16,3), 63 % at (32,6), 68 % at (64,8), 77 % at (64,16) are
// settled here, for ~7 VALU instructions per pair of rows; the other rows (redundant ones, and facets whose foot point
// lies outside the facet) run the simplex as before.  `nlp` still counts every LP the reference issues.
// The rows settled here get the reference's in-place round trip h[k] = (h[k] + 0.1) - 0.1 (:1149-1151) applied to their
// slot in LDS right away (owner lane), all others in the order of their LPs as before: an LP of row k therefore sees
// settled rows k' > k already rounded through the round trip (<= 1 ulp of b + 0.1 in a right-hand side; no verdict
// depends on it outside exact ties, on which no two LP codes agree).
// Rows zeroed in LDS (never present, removed by the dedupe or the prefilter) pass every test: 0 <= 0.
// Returns my rows' bits (bit k: row row0 + k is settled as "keep").
// LS: distance (in doubles) between consecutive elements of the three arrays (1: a polytope's rows are contiguous;
// 16: the polytope-interleaved tile of plp_reduce_lane.hip)
// BETA: the third array holds the slacks  beta_i = max(b_i - a_i.xc, 0)  themselves instead of s_i = a_i.xc
template <int D, int R, int LS = 1, bool BETA = false>
__device__ __forceinline__ unsigned f2_presolve(const double* myA, double* myb, const double* myan, int row0, int m_loop,
                                                unsigned cand, double abs_tol) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (compiler: the owner lanes' stores above come first)
    double ak[R][D], skt[R], gkk[R];
    int jb[R];  // a row that blocks the ray of candidate k (the last one found; nearly always there is only one)
    unsigned ok = cand;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        double g = 0.0;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            ak[k][kk] = myA[((row0 + k) * D + kk) * LS];
            g = fma(ak[k][kk], ak[k][kk], g);
        }
        const double bk = myb[(row0 + k) * LS];
        const double sk = BETA ? myan[(row0 + k) * LS] : fmax(bk - myan[(row0 + k) * LS], 0.0);
        const double tau = abs_tol + 1e-9 * (1.0 + fabs(bk) + sk);
        skt[k] = sk + tau;
        gkk[k] = g;
        jb[k] = 0;
        if (!((g > 0.0) & (tau < 0.05))) { ok &= ~(1u << k); cand &= ~(1u << k); }  // (NaN-safe: such a row is never settled)
    }
    for (int i = 0; i < m_loop; ++i) {
        double ai[D];
#pragma unroll
        for (int kk = 0; kk < D; ++kk) ai[kk] = myA[(i * D + kk) * LS];
        const double si = BETA ? myan[(i) * LS] : fmax(myb[(i) * LS] - myan[(i) * LS], 0.0);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            double gik = 0.0;
#pragma unroll
            for (int kk = 0; kk < D; ++kk) gik = fma(ai[kk], ak[k][kk], gik);
            const bool fine = (i == row0 + k) | (skt[k] * gik <= si * gkk[k]);
            ok = fine ? ok : (ok & ~(1u << k));
            jb[k] = fine ? jb[k] : i;
        }
    }
#if PLP_R_PRESOLVE >= 2
    // Second witness for the candidates whose ray is blocked (the foot point of xc on row k's plane lies outside the
    // facet): up to the blocking row j, then along its plane in the direction of a_k projected onto it,
    //     x* = xc + t1 a_k + t2 d,   t1 = s_j / (a_j.a_k) (a hair less),   d = a_k - rho a_j,  rho = a_j.a_k / |a_j|^2,
    // with t2 such that a_k.x* = b_k + tau.  Feasible iff  t1 (a_i.a_k) + t2 (a_i.d) = (t1 + t2) (a_i.a_k) - t2 rho (a_i.a_j)
    // <= s_i  for every other live row.  Two candidates at a time (register budget of the bench kernel).
    // At most H = 2 blocked candidates per lane get the second witness (the first two; a lane rarely has more, and a row
    // that goes without simply runs the simplex): ONE pass over the rows for both instead of a pass per pair of my rows.
    unsigned ok2 = cand & ~ok;
    constexpr int H = R >= 2 ? 2 : 1;
    {
        int kq[H];
        {
            unsigned rem = ok2;
            unsigned pick = 0u;
#pragma unroll
            for (int q = 0; q < H; ++q) {
                kq[q] = rem ? __ffs((int)rem) - 1 : (q ? kq[q ? q - 1 : 0] : 0);   // (no candidate left: the previous one again)
                pick |= rem ? (1u << kq[q]) : 0u;
                rem &= rem - 1u;
            }
            ok2 &= pick;
        }
        if (__any(ok2 != 0u)) {
            double bk_[H][D], aj[H][D], c1[H], c2[H];
            int rowk[H];
#pragma unroll
            for (int q = 0; q < H; ++q) {
                const int k = kq[q];
                int j = 0;
                double sk_t = 0.0, gk = 0.0;
#pragma unroll
                for (int kx = 0; kx < R; ++kx) {  // (k is data here: select instead of indexing the register arrays)
                    j = (kx == k) ? jb[kx] : j;
                    sk_t = (kx == k) ? skt[kx] : sk_t;
                    gk = (kx == k) ? gkk[kx] : gk;
                }
                rowk[q] = row0 + k;
                double gjk = 0.0, gjj = 0.0;
#pragma unroll
                for (int kk = 0; kk < D; ++kk) {
                    bk_[q][kk] = myA[((row0 + k) * D + kk) * LS];
                    aj[q][kk] = myA[(j * D + kk) * LS];
                    gjk = fma(aj[q][kk], bk_[q][kk], gjk);
                    gjj = fma(aj[q][kk], aj[q][kk], gjj);
                }
                const double sj = BETA ? myan[(j) * LS] : fmax(myb[(j) * LS] - myan[(j) * LS], 0.0);
                const double rho = gjk / gjj;
                const double t1 = (sj / gjk) * (1.0 - 0x1p-40);
                const double akd = fma(-rho, gjk, gk);        // a_k.d = |a_k|^2 - (a_j.a_k)^2 / |a_j|^2 >= 0
                const double t2 = fma(-t1, gk, sk_t) / akd;   // (s_k + tau - t1 |a_k|^2) / a_k.d
                c1[q] = t1 + t2;
                c2[q] = t2 * rho;
                // (a blocked ray has a_j.a_k > 0 and t1 |a_k|^2 < s_k + tau; anything else -- parallel rows, NaN -- is left to the simplex)
                if (!((gjk > 0.0) & (akd > 1e-12 * gk) & (t2 >= 0.0) & (t2 < 1e300))) ok2 &= ~(1u << k);
            }
            for (int i = 0; i < m_loop; ++i) {
                double ai[D];
#pragma unroll
                for (int kk = 0; kk < D; ++kk) ai[kk] = myA[(i * D + kk) * LS];
                const double si = BETA ? myan[(i) * LS] : fmax(myb[(i) * LS] - myan[(i) * LS], 0.0);
#pragma unroll
                for (int q = 0; q < H; ++q) {
                    double gik = 0.0, gij = 0.0;
#pragma unroll
                    for (int kk = 0; kk < D; ++kk) {
                        gik = fma(ai[kk], bk_[q][kk], gik);
                        gij = fma(ai[kk], aj[q][kk], gij);
                    }
                    const double lhs = fma(-c2[q], gij, c1[q] * gik);
                    const bool fine = (i == rowk[q]) | (lhs <= si);
                    ok2 = fine ? ok2 : (ok2 & ~(1u << kq[q]));
                }
            }
        }
    }
    ok |= ok2;
#endif
#pragma unroll
    for (int k = 0; k < R; ++k)
        if ((ok >> k) & 1u) myb[(row0 + k) * LS] = (myb[(row0 + k) * LS] + 0.1) - 0.1;  // (:1149-1151), see above
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return ok;
}

// One tile = the NG = RBLOCK / GS polytopes starting at polytope `tile` (the body of reduce_r_kernel; a device function
// so that reduce_r_mix_kernel can give the last tiles of a launch a different shape).
// LAZY (GS = 64, R = 1: one polytope per wavefront): F1 on the one-LP-per-wavefront engine (its LDS block sits behind the
// tile's arrays), the F3 / F2 LPs on plp_lazy.hpp -- no dictionary is carried.
// SPLIT (small batches): ONE polytope per wavefront, its INDEPENDENT LPs spread over the RBLOCK / GS lane groups -- every
// group runs F1 and the dedupe on the same rows (identical results, nothing to exchange), then group g solves the g-th
// of the 2d box LPs and the g-th surviving row's redundancy LP (round-robin when there are more LPs than groups).  The
// in-place h[k] +- 0.1 round trip becomes a rule (rows that had their LP before k: (b + 0.1) - 0.1; row k: b + 0.1), the
// results meet in LDS.  Same engine, same arithmetic per LP: outputs bitwise equal to the batch form; a polytope takes
// ~12 pivot times instead of ~60 (latency), at ~3x the instruction slots (throughput): for batches that cannot fill
// the chip anyway.
template <int D, int GS, int R, bool LAZY = false, bool SPLIT = false, bool WDENSE = false>
__device__ __forceinline__ void reduce_r_tile(
    const long long tile, long long B, int m_max, const double* __restrict__ Ag, const double* __restrict__ bg,
    const int* __restrict__ mrows, double abs_tol, int force_retry, unsigned long long* __restrict__ keep_out,
    int* __restrict__ flags_out, double* __restrict__ r_out, double* __restrict__ xc_out,
    int* __restrict__ nlp_out, unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ retry_word,
    unsigned long long epoch) {
    constexpr unsigned RMASK = (1u << R) - 1u;
    constexpr int RSH = R == 8 ? 3 : (R == 4 ? 2 : (R == 2 ? 1 : 0));  // log2(R)
    static_assert(R == 1 || R == 2 || R == 4 || R == 8, "rows per lane");
    static_assert(!LAZY || (GS == 64 && R == 1 && RBLOCK == 64), "lazy LPs: one polytope per wavefront and workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int gs = GS;
    const Grp g(gs);
    constexpr int NGRP = RBLOCK / gs;             // lane groups per workgroup
    constexpr int NG = SPLIT ? 1 : NGRP;          // polytopes per tile
    constexpr int rows = gs * R;  // row slots per polytope
    const int grp = threadIdx.x / gs;             // my lane group
    const int gib = SPLIT ? 0 : grp;              // my polytope inside the tile
    static_assert(!(SPLIT && LAZY), "one or the other");
    static_assert(!SPLIT || RBLOCK == 64, "SPLIT: the lane groups of ONE wavefront share a polytope (same-value LDS writes in program order)");
    const int row0 = g.gl * R;  // my first row
    double* sA = reinterpret_cast<double*>(smem_raw);  // [NG][rows][D]
    double* sb = sA + (size_t)NG * rows * D;            // [NG][rows]
    double* san = sb + (size_t)NG * rows;               // [NG][rows]
    double* myA = sA + (size_t)gib * rows * D;
    double* myb = sb + (size_t)gib * rows;
    double* myan = san + (size_t)gib * rows;
    double* lzrho = san + (size_t)NG * rows;  // (LAZY) F1's block of the one-LP-per-wavefront engine
    // (SPLIT) where the groups' results meet: box values [2D], kept-row mask, flags (bit 0: an LP failed, bit 1: retry)
    double* sval = san + (size_t)NG * rows;
    unsigned long long* skeep = reinterpret_cast<unsigned long long*>(sval + 2 * D);
    unsigned* sflag = reinterpret_cast<unsigned*>(skeep + 1);
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);

    {  // one tile per workgroup (no values kept live across a tile loop)
        const int ntile = (B - tile) < NG ? (int)(B - tile) : NG;
        __syncthreads();
        {
            const int rowsz = m_max * D;
            const double* src = Ag + tile * rowsz;
            for (int idx = threadIdx.x; idx < ntile * rowsz; idx += RBLOCK) {
                const int p = idx / rowsz, rem = idx - p * rowsz;
                sA[(size_t)p * rows * D + rem] = src[idx];
            }
            const double* srcb = bg + tile * m_max;
            for (int idx = threadIdx.x; idx < ntile * m_max; idx += RBLOCK) {
                const int p = idx / m_max, row = idx - p * m_max;
                sb[p * rows + row] = srcb[idx];
            }
        }
        if constexpr (SPLIT) {
            if (threadIdx.x == 0) { *skeep = 0ull; *sflag = 0u; }  // (visible after the barrier that follows F1)
        }
        __syncthreads();
        const long long pg = tile + gib;
        const bool valid = gib < ntile;
        const int m = valid ? (mrows ? mrows[pg] : m_max) : 0;
        double xc[D];
        double rr = 0.0;
        bool ball, fulldim, f1open = false;
        uint64_t live = 0ull;
        unsigned has = 0u;
        // an F2/F3 LP needed Bland's rule: the whole polytope goes to the general kernel
        // (force_retry: test hook, PLP_REDUCE_RETRY_ALL=1 sends every polytope through that second pass)
        bool retry = force_retry != 0;
        // ---------------------------------------------------------------- F1: Chebyshev ball
        if constexpr (LAZY) {
            // one polytope per wavefront: F1 on the one-LP-per-wavefront engine (plp_wide.hpp: the entering column is
            // wave-uniform, the row a register vector); set-up and read-out as in the branch below
            constexpr int NC = D + 1;
            wide::WideShared<NC>& sh = *reinterpret_cast<wide::WideShared<NC>*>(lzrho);
            static_assert(sizeof(wide::WideShared<NC>) <= lazy::lds_bytes<D>(), "F1's LDS block");
            const int lane = g.lane;
            const bool h = valid & (lane < m) & (m <= rows);
            has = h ? 1u : 0u;
            typename wide::RowVec<NC>::type Tv = (typename wide::RowVec<NC>::type)(0.0);
            double T16 = 0.0;
            double nrm2 = 0.0;
            bool finite = true;
#pragma unroll
            for (int kk = 0; kk < D; ++kk) {
                const double v = h ? myA[row0 * D + kk] : 0.0;
                ROW_SET(kk, v);
                nrm2 = nrm2 + v * v;
                finite = finite & isfinite(v);
            }
            const double bk = h ? myb[row0] : 0.0;
            finite = finite & isfinite(bk);
            const double nrm = sqrt(nrm2);
            myan[row0] = 1.0 / nrm;
            const bool zero = !(nrm > 0.0);
            bool rowact = h & !zero;
            ROW_SET(D, rowact ? nrm : 0.0);
            double beta = rowact ? bk : 0.0;
            int rowvar = NC + lane, rowneg = 0;
            if (lane <= NC) {
                sh.cost[lane] = lane == D ? -1.0 : 0.0;
                sh.cv[lane] = (lane + 1) << 1;
            }
            const bool infeasible0 = __ballot(h & zero & (bk < -TOL_FEAS)) != 0;
            const bool bad = (__ballot(!finite) != 0) | (m > rows);
            __syncthreads();
            int st1, it1 = 0;
            if (!valid | bad) st1 = ST_NUM;
            else if (infeasible0) st1 = ST_INFEAS;
            else st1 = wide::wide_run<NC>(lane, m, Tv, T16, beta, rowvar, rowneg, rowact, sh, NC, true, bk / nrm, it1);
            const double mine = rowneg ? -beta : beta;
#pragma unroll
            for (int j = 0; j <= D; ++j) {
                const uint64_t ob = __ballot(rowvar == j);
                const double xj = ob ? wide::uniform_lane(mine, __ffsll((long long)ob) - 1) : 0.0;
                if (j < D) xc[j < D ? j : 0] = xj; else rr = xj;
            }
            ball = (st1 == ST_OPT) & (rr >= 0.0);  // cheby_ball: status 0 and r >= 0 (:1289-1293)
            fulldim = ball & (rr > abs_tol);
            f1open = valid & (st1 != ST_OPT) & (st1 != ST_INFEAS);   // (RF_F1OPEN, plp_common.hpp)
        } else
        {
#if PLP_R_FAST
            SimplexR<D + 1, R, false, true> S;  // forced first pivot handed to run_fast
            double qi[R];
#else
            SimplexR<D + 1, R, true> S;
#endif
            S.reset(D + 1, m, row0);
            unsigned actb = 0u;
            bool inf0 = false, finite = true;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const bool h = valid & (row0 + k < m) & (m <= rows);
                has |= h ? (1u << k) : 0u;
                double nrm2 = 0.0;
#pragma unroll
                for (int kk = 0; kk < D; ++kk) {
                    const double v = h ? myA[(row0 + k) * D + kk] : 0.0;
                    S.T[k][kk] = v;
                    nrm2 = nrm2 + v * v;
                    finite = finite & isfinite(v);
                }
                const double bk = h ? myb[row0 + k] : 0.0;
                finite = finite & isfinite(bk);
                const double nrm = sqrt(nrm2);
                myan[row0 + k] = 1.0 / nrm;
                const bool zero = !(nrm > 0.0);
                const bool on = h & !zero;
                S.T[k][D] = on ? nrm : 0.0;
                S.beta[k] = on ? bk : 0.0;
#if PLP_R_FAST
                qi[k] = bk / nrm;
#else
                S.init_q[k] = bk / nrm;
#endif
                actb |= on ? (1u << k) : 0u;
                inf0 = inf0 | (h & zero & (bk < -TOL_FEAS));
            }
            S.ract = actb;
            const bool infeasible0 = grp_ballot(inf0, g) != 0;
            const bool bad = (grp_ballot(!finite, g) != 0) | (m > rows);
            S.cost[D] = -1.0;
#if PLP_R_FAST
            S.mode = M_P2;
#else
            S.init_elig = actb;
            S.mode = M_INIT;
            S.init_col = D;
            S.mode_after_init = M_P2;
#endif
            if (!valid | bad) { S.mode = M_DONE; S.status = ST_NUM; }
            else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }
#if PLP_R_FAST
            S.template run_fast<GS, true>(g, qi, actb);
            retry = retry | (valid & (S.status == ST_RETRY));
#ifdef PLP_STAGE_STATS
            {
                const int wm = stat_wave_max(valid ? S.iters : 0);
                if (threadIdx.x == 0) { PLP_STAT_ADD(0, 1); PLP_STAT_ADD(1, wm); }
                if (valid & (g.gl == 0)) { PLP_STAT_ADD(2, S.iters); PLP_STAT_ADD(11, 1); }
            }
#endif
#else
            S.run(g);
#endif
            const bool ok = S.status == ST_OPT;
            f1open = valid & !ok & (S.status != ST_INFEAS);   // (RF_F1OPEN, plp_common.hpp)
#pragma unroll
            for (int j = 0; j <= D; ++j) {
                bool found;
                const double mine = S.x_of(j, found);
                const uint64_t ob = grp_ballot(found, g);
                const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
                const double xj = ob ? v : 0.0;
                if (j < D) xc[j < D ? j : 0] = xj; else rr = xj;
            }
            ball = ok & (rr >= 0.0);  // cheby_ball: status 0 and r >= 0 (:1289-1293)
            fulldim = ball & (rr > abs_tol);
        }
        // the Chebyshev ball is final: r and xc go out now (eight registers less to carry through the LP stages)
        if (valid & (g.gl == 0) & (!SPLIT || grp == 0)) {
            r_out[tile + gib] = ball ? rr : 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) xc_out[(tile + gib) * D + k] = ball ? xc[k] : qnan;
        }
        __syncthreads();  // 1/||a|| of every row is in LDS
        // ---------------------------------------------------------------- dedupe (:1094-1110)
        // (rows are re-read from LDS: the register file limits the occupancy of this kernel, LDS is idle)
        if constexpr (!SPLIT && !LAZY && rows == 16 && R == 4 && PLP_R_DEDUPE_HALF) {
            // Sixteen row slots, four per lane: every unordered pair {i, j} once instead of twice.  Row i meets its partners
            // j = i + 1 .. i + 8 (mod 16): the 8 cyclic distances cover all 120 pairs (distance 8 twice: harmless), the
            // dot product is the same number whichever of the two rows' owners forms it (products commute, same order
            // of additions), and ONE evaluation settles both rows' verdicts; the lanes' 16-bit masks of removed rows are
            // OR-ed over the group.  32 pairs per lane instead of 64.
            unsigned remmask = 0u;
            double ni[R][D], bin_[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double an_i = myan[row0 + k];
#pragma unroll
                for (int kk = 0; kk < D; ++kk) ni[k][kk] = myA[(row0 + k) * D + kk] * an_i;
                bin_[k] = myb[row0 + k] * an_i;
            }
            // (unrolled by two: fully unrolled the loop is 0.5 % faster and the kernel spills 56 B per lane instead of 28)
#pragma unroll 2
            for (int t = 1; t <= 8; ++t) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int i = row0 + k;
                    const int j = (i + t) & 15;
                    const double an_j = myan[j];
                    double dot = 0.0;
#pragma unroll
                    for (int kk = 0; kk < D; ++kk) dot = dot + ni[k][kk] * (myA[j * D + kk] * an_j);
                    const double bjn = myb[j] * an_j;
                    const bool par = valid & (m <= rows) & (i < m) & (j < m) & (dot > 1.0 - abs_tol);
                    // the reference's rule for the pair (lo, hi), lo < hi (:1104-1109): b_lo < b_hi removes hi, else lo
                    const bool i_lo = i < j;
                    const double blo = i_lo ? bin_[k] : bjn, bhi = i_lo ? bjn : bin_[k];
                    const int lo = i_lo ? i : j, hi = i_lo ? j : i;
                    const int gone = (blo < bhi) ? hi : lo;
                    remmask |= par ? (1u << gone) : 0u;
                }
            }
            remmask |= (unsigned)__shfl_xor((int)remmask, 1, 64);
            remmask |= (unsigned)__shfl_xor((int)remmask, 2, 64);
            const unsigned removed = (remmask >> row0) & RMASK;
#pragma unroll
            for (int k = 0; k < R; ++k)
                live |= spread_rows<R, GS>(grp_ballot((((has & ~removed) >> k) & 1u) != 0u, g)) << k;
        } else {
            unsigned removed = 0u;
            double ni[R][D], bin_[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double an_i = myan[row0 + k];
#pragma unroll
                for (int kk = 0; kk < D; ++kk) ni[k][kk] = myA[(row0 + k) * D + kk] * an_i;
                bin_[k] = myb[row0 + k] * an_i;
            }
            for (int j = 0; j < m_max; ++j) {
                const bool jrow = valid & (j < m);
                const double an_j = myan[j];
                double nj[D];
#pragma unroll
                for (int kk = 0; kk < D; ++kk) nj[kk] = myA[j * D + kk] * an_j;
                const double bjn = myb[j] * an_j;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    double dot = 0.0;
#pragma unroll
                    for (int kk = 0; kk < D; ++kk) dot = dot + ni[k][kk] * nj[kk];
                    const int i = row0 + k;
                    const bool par = (((has >> k) & 1u) != 0u) & jrow & (j != i) & (dot > 1.0 - abs_tol);
                    const bool rem = par & ((i < j) ? !(bin_[k] < bjn) : (bjn < bin_[k]));
                    removed |= rem ? (1u << k) : 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < R; ++k)
                live |= spread_rows<R, GS>(grp_ballot((((has & ~removed) >> k) & 1u) != 0u, g)) << k;
        }
        // dictionary translated to the Chebyshev centre: beta_i = b_i - a_i.xc.  s_i = a_i.xc replaces
        // 1/||a_i|| in LDS (only the owner lane touches its rows' slots from here on).
        bool off = false;   // a row the centre violates (centre_off, plp_common.hpp)
        const double xs = centre_scale<D>(xc);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            double sk = 0.0;
#pragma unroll
            for (int kk = 0; kk < D; ++kk)
                sk = fma(((has >> k) & 1u) ? myA[(row0 + k) * D + kk] : 0.0, ball ? xc[kk] : 0.0, sk);
            const double bk = ((has >> k) & 1u) ? myb[row0 + k] : 0.0;
            off = off | ((((has >> k) & 1u) != 0u) & centre_off(bk - sk, myan[row0 + k], bk, xs));
            myan[row0 + k] = sk;
        }
        if (ball & (grp_ballot(off, g) != 0)) {   // F1 "optimal" outside the polytope: nothing below may start from it
            ball = false; fulldim = false; f1open = true;
            if (valid & (g.gl == 0) & (!SPLIT || grp == 0)) {
                r_out[tile + gib] = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) xc_out[(tile + gib) * D + k] = qnan;
            }
        }
        // Rows that dropped out (never present, or removed by the dedupe) are zeroed in LDS -- A, b and
        // s -- so that the LP set-ups below load their rows without masking.  Only the owner lane writes
        // a row's slots and only the owner lane reads them afterwards (live rows, which other lanes read
        // as F2 objectives, are never written): no cross-lane visibility is relied upon.
        auto zero_dead = [&](unsigned alive) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (!((alive >> k) & 1u)) {
#pragma unroll
                    for (int kk = 0; kk < D; ++kk) myA[(row0 + k) * D + kk] = 0.0;
                    myb[row0 + k] = 0.0;
                    myan[row0 + k] = 0.0;
                }
            }
        };
        zero_dead(((unsigned)(live >> row0) & RMASK));
        int flags = fulldim ? 0 : (RF_EMPTY | (f1open ? RF_F1OPEN : 0));
        int nlp = 1;
        uint64_t keep = 0ull;
        int stage = 0;  // 0 done, 1 needs the box, 2 needs the redundancy LPs
        if (fulldim) {
            const int neq = __popcll(live);
            if (neq <= D + 1) { flags = RF_EARLY; keep = live; }
            else stage = (neq > 3 * D) ? 1 : 2;
        }
#ifdef PLP_DEBUG_SKIP_LPS
        stage = 0;  // debug timing build: F1 + dedupe only
#endif
        // ---------------------------------------------------------------- F3: bounding box (:1367-1409)
        if (__any(stage == 1)) {
            const bool go = stage == 1;
            const unsigned lloc = ((unsigned)(live >> row0) & RMASK);
            double s1[R], s2[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { s1[k] = 0.0; s2[k] = 0.0; }
            bool lpfail = false;
            double lbk = 0.0;
            // (batch form, R > 1) the box of coordinate kx waits in lane kx % GS of the group, slot kx / GS, and the
            // prefilter sums are formed after the last LP: 2R running sums per lane carried through the LP loops were
            // what the register allocator spilled around this stage (11 dwords per lane, 19 MB of scratch traffic per
            // C2 launch); same operations in the same order, so the sums are bitwise what they were
            constexpr int NSLOT = (D + GS - 1) / GS;
            double blo[NSLOT], bhi[NSLOT];
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) { blo[q] = 0.0; bhi[q] = 0.0; }
            constexpr bool BOX_IN_LANES = !SPLIT && !LAZY && R > 1;
            if constexpr (SPLIT) {
                for (int round = 0; round * NGRP < 2 * D; ++round) {  // group g: LP g, g + NGRP, ...
                    const int itq = round * NGRP + grp;
                    const bool mine = go & (itq < 2 * D);
                    const int it = itq < 2 * D ? itq : 0;
                    const int kx = it >> 1;
                    const bool up = it & 1;
                    double xck = 0.0;
                    SimplexR<D, R, false, false> S;
                    S.reset(D, __popcll(live), row0);
#pragma unroll
                    for (int kk = 0; kk < D; ++kk) {
                        xck = (kk == kx) ? xc[kk] : xck;
                        S.cost[kk] = (kk == kx) ? (up ? -1.0 : 1.0) : 0.0;
                    }
#pragma unroll
                    for (int k = 0; k < R; ++k) {
#pragma unroll
                        for (int kk = 0; kk < D; ++kk) S.T[k][kk] = myA[(row0 + k) * D + kk];
                        S.beta[k] = fmax(myb[row0 + k] - myan[row0 + k], 0.0);  // 0 for the zeroed rows
                    }
                    S.ract = lloc;
                    S.mode = mine ? M_P2 : M_DONE;
                    S.template run_fast<GS>(g);
                    double val;
                    unsigned fl = 0u;
                    if (S.status == ST_OPT) val = up ? (xck + S.negz) : (xck - S.negz);
                    else if (S.status == ST_UNBND) val = up ? pinf : -pinf;
                    else { val = qnan; fl = 1u; }
                    if (S.status == ST_RETRY) fl |= 2u;
                    if (mine & (g.gl == 0)) {
                        sval[it] = val;
                        if (fl) atomicOr(sflag, fl);
                    }
                }
                __syncthreads();
                for (int kx = 0; kx < D; ++kx) {  // prefilter sums, accumulated in k order (:1131-1134)
                    const double lo = sval[2 * kx], hi = sval[2 * kx + 1];
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const double aik = myA[(row0 + k) * D + kx];
                        const double pa = (aik > 0.0 ? 1.0 : 0.0) * aik;
                        s1[k] = s1[k] + pa * (hi - lo);
                        s2[k] = s2[k] + aik * lo;
                    }
                }
                const unsigned fl = *sflag;
                lpfail = go & ((fl & 1u) != 0u);
                retry = retry | (go & ((fl & 2u) != 0u));
            } else
            for (int it = 0; it < 2 * D; ++it) {  // lower_0, upper_0, lower_1, upper_1, ...
                const int kx = it >> 1;
                const bool up = it & 1;
                double xck = 0.0;
#pragma unroll
                for (int kk = 0; kk < D; ++kk) xck = (kk == kx) ? xc[kk] : xck;
                struct { int status; double negz; } S;
                if constexpr (LAZY) {
                    S.status = ST_NUM;
                    S.negz = 0.0;
                    if (__builtin_amdgcn_readfirstlane((int)go)) {  // (wave-uniform: one polytope per wavefront)
                        if constexpr (WDENSE)
                            S.status = wide::solve_dense<D>(g.lane, __popcll(live), myA, (g.lane == kx) ? (up ? -1.0 : 1.0) : 0.0,
                                                            fmax(myb[row0] - myan[row0], 0.0), (lloc & 1u) != 0u, S.negz,
                                                            *reinterpret_cast<wide::WideShared<D>*>(lzrho));
                        else
                            S.status = lazy::solve<D>(g.lane, __popcll(live), myA, (g.lane == kx) ? (up ? -1.0 : 1.0) : 0.0,
                                                      fmax(myb[row0] - myan[row0], 0.0), (lloc & 1u) != 0u, S.negz);
                    }
                    retry = retry | (go & (S.status == ST_RETRY));
                }
                SimplexR<D, R, false, false> S_;
                if constexpr (!LAZY) {
                S_.reset(D, __popcll(live), row0);
#pragma unroll
                for (int kk = 0; kk < D; ++kk) S_.cost[kk] = (kk == kx) ? (up ? -1.0 : 1.0) : 0.0;
#pragma unroll
                for (int k = 0; k < R; ++k) {
#pragma unroll
                    for (int kk = 0; kk < D; ++kk) S_.T[k][kk] = myA[(row0 + k) * D + kk];
                    S_.beta[k] = fmax(myb[row0 + k] - myan[row0 + k], 0.0);  // 0 for the zeroed rows
                }
                S_.ract = lloc;
                S_.mode = go ? M_P2 : M_DONE;
#if PLP_R_FAST
                S_.template run_fast<GS>(g);
                retry = retry | (go & (S_.status == ST_RETRY));
#ifdef PLP_STAGE_STATS
                {
                    const int wm = stat_wave_max(go ? S_.iters : 0);
                    if (threadIdx.x == 0) { PLP_STAT_ADD(3, wm); PLP_STAT_ADD(10, 1); }
                    if (go & (g.gl == 0)) { PLP_STAT_ADD(4, S_.iters); PLP_STAT_ADD(5, 1); }
                }
#endif
#else
                S_.run(g);
#endif
                S.status = S_.status;
                S.negz = S_.negz;
                }
                // zeta = c.x' = -negz ; x_k = xc_k + x'_k ; lower: c = +e_k, upper: c = -e_k
                double val;
                if (S.status == ST_OPT) val = up ? (xck + S.negz) : (xck - S.negz);
                else if (S.status == ST_UNBND) val = up ? pinf : -pinf;
                else { val = qnan; lpfail = lpfail | go; }
                if (!up) {
                    lbk = val;
                } else if constexpr (BOX_IN_LANES) {
                    const bool mine = g.gl == (kx % GS);
#pragma unroll
                    for (int q = 0; q < NSLOT; ++q) {
                        const bool here = mine & (q == kx / GS);
                        blo[q] = here ? lbk : blo[q];
                        bhi[q] = here ? val : bhi[q];
                    }
                } else {  // prefilter sums, accumulated in k order (:1131-1134)
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const double aik = myA[(row0 + k) * D + kx];
                        const double pa = (aik > 0.0 ? 1.0 : 0.0) * aik;
                        s1[k] = s1[k] + pa * (val - lbk);
                        s2[k] = s2[k] + aik * lbk;
                    }
                }
            }
            if constexpr (BOX_IN_LANES) {
#pragma unroll
                for (int kx = 0; kx < D; ++kx) {  // prefilter sums, accumulated in k order (:1131-1134)
                    const double lo = bcast(blo[kx / GS], g.gbase + kx % GS);
                    const double hi = bcast(bhi[kx / GS], g.gbase + kx % GS);
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const double aik = myA[(row0 + k) * D + kx];
                        const double pa = (aik > 0.0 ? 1.0 : 0.0) * aik;
                        s1[k] = s1[k] + pa * (hi - lo);
                        s2[k] = s2[k] + aik * lo;
                    }
                }
            }
            uint64_t outb = 0ull;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const bool out = go & (((lloc >> k) & 1u) != 0u) & ((s1[k] - (myb[row0 + k] - s2[k])) < -1e-4);
                outb |= spread_rows<R, GS>(grp_ballot(out, g)) << k;
            }
            if (go) {
                live = live & ~outb;
                zero_dead(((unsigned)(live >> row0) & RMASK));
                nlp += 2 * D;
                if (lpfail) flags |= RF_LPFAIL;
                if (__popcll(live) <= D + 1) { flags |= RF_EARLY; keep = live; stage = 0; }
                else stage = 2;
            }
        }
        // ---------------------------------------------------------------- F2: redundancy LPs (:1142-1160)
        if constexpr (SPLIT) {
            if (__any(stage == 2)) {
                const bool go2 = stage == 2;
                const unsigned lloc = ((unsigned)(live >> row0) & RMASK);
                const int nlive = __popcll(live);
                if (go2) nlp += nlive;
                for (int round = 0; round * NGRP < rows; ++round) {  // group g: the g-th surviving row, g + NGRP-th, ...
                    if (!__any(go2 & (round * NGRP < nlive))) break;
                    const int idx = round * NGRP + grp;
                    const bool mine = go2 & (idx < nlive);
                    uint64_t t = live;
                    for (int i = 0; i < rows; ++i) t = (i < idx) ? (t & (t - 1ull)) : t;
                    const int kr = (mine & (t != 0ull)) ? __ffsll((long long)t) - 1 : 0;
                    SimplexR<D, R, false, false> S;
                    S.reset(D, nlive, row0);
                    double cxc = 0.0;
#pragma unroll
                    for (int kk = 0; kk < D; ++kk) {
                        const double ck = -myA[kr * D + kk];  // f = -A[k,:]  (:1145)
                        S.cost[kk] = ck;
                        cxc = fma(ck, xc[kk], cxc);
                    }
#pragma unroll
                    for (int k = 0; k < R; ++k) {
#pragma unroll
                        for (int kk = 0; kk < D; ++kk) S.T[k][kk] = myA[(row0 + k) * D + kk];
                        // h as the reference holds it when row kr has its turn (:1149-1151): rows before it carry the round trip
                        const int rw = row0 + k;
                        const double h0 = myb[rw];
                        const double hp = h0 + 0.1;
                        const bool lv = ((lloc >> k) & 1u) != 0u;
                        const double hh = (lv & (rw < kr)) ? hp - 0.1 : ((lv & (rw == kr)) ? hp : h0);
                        S.beta[k] = fmax(hh - myan[rw], 0.0);  // 0 for the zeroed rows
                    }
                    S.ract = lloc;
                    S.mode = mine ? M_P2 : M_DONE;
                    S.template run_fast<GS>(g);
                    const double fun = cxc - S.negz;  // c.xc + zeta, zeta = -negz
                    const double hk = (myb[kr] + 0.1) - 0.1;
                    const double obj = -fun - hk;  // (:1156)
                    const bool keepk = mine & (((S.status == ST_OPT) & (obj > abs_tol)) | (S.status == ST_UNBND));
                    if (g.gl == 0) {
                        if (keepk) atomicOr(skeep, 1ull << kr);
                        if (mine & (S.status == ST_RETRY)) atomicOr(sflag, 2u);
                    }
                }
                __syncthreads();
                if (go2) {
                    keep = *skeep;
                    retry = retry | ((*sflag & 2u) != 0u);
                    flags |= RF_MINREP;
                }
            }
        } else if constexpr (LAZY) {
            // one polytope per wavefront: its rows one after the other, each LP on plp_lazy.hpp (nothing to set up but
            // the cost vector: lane j holds -A[k][j]; c.xc = -(a_k.xc) = -s_k exactly, the two FMA chains mirror each other)
            if (__builtin_amdgcn_readfirstlane(stage) == 2) {  // (wave-uniform, and said so: the LP loops stay scalar)
                const unsigned lloc = ((unsigned)(live >> row0) & RMASK);
                uint64_t todo = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(live >> 32)) << 32) |
                                (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)live);
                nlp += __popcll(live);
#if PLP_R_PRESOLVE
                {   // rows the presolve settles as "keep" need no LP (lane i = row i: the ballot is the row mask)
                    const uint64_t cert = __ballot((f2_presolve<D, 1>(myA, myb, myan, row0, m_max, lloc & 1u, abs_tol) & 1u) != 0u);
                    keep |= cert;
                    ctr_add(ctr, -__popcll(cert), g.gl == 0);
                    todo &= ~(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(cert >> 32)) << 32) |
                              (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cert));
                }
#endif
                while (todo != 0ull) {
                    const int kr = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    const double ck = g.lane < D ? -myA[kr * D + (g.lane < D ? g.lane : 0)] : 0.0;  // f = -A[k,:]  (:1145)
                    const double cxc = -myan[kr];
                    const bool owner = kr == row0;
                    if (owner) myb[kr] = myb[kr] + 0.1;  // h[k] += 0.1 in place (:1149), undone below (:1151)
                    double negz2 = 0.0;
                    int st2;
                    if constexpr (WDENSE)
                        st2 = wide::solve_dense<D>(g.lane, __popcll(live), myA, ck, fmax(myb[row0] - myan[row0], 0.0),
                                                   (lloc & 1u) != 0u, negz2, *reinterpret_cast<wide::WideShared<D>*>(lzrho));
                    else
                        st2 = lazy::solve<D>(g.lane, __popcll(live), myA, ck, fmax(myb[row0] - myan[row0], 0.0),
                                             (lloc & 1u) != 0u, negz2);
                    retry = retry | (st2 == ST_RETRY);
                    const double fun = cxc - negz2;  // c.xc + zeta, zeta = -negz
                    double hk_own = 0.0;
                    if (owner) { hk_own = myb[kr] - 0.1; myb[kr] = hk_own; }
                    const double hk = bcast(hk_own, kr);
                    const double obj = -fun - hk;  // (:1156)
                    const bool keepk = ((st2 == ST_OPT) & (obj > abs_tol)) | (st2 == ST_UNBND);
                    keep |= keepk ? (1ull << kr) : 0ull;
                }
                flags |= RF_MINREP;
            }
        } else {
#if PLP_R_ASYNC && PLP_R_FAST
        // The 16 polytopes of a wavefront need different numbers of LPs (rows that survived the dedupe and
        // the prefilter) and their LPs different numbers of pivots; in lock-step every LP costs the wave the
        // maximum over its polytopes (measured: 3.9 pivots against a mean of 2.3).  Here every group walks its
        // own list: a group whose LP has ended collects the result and sets up its next row inside the pivot
        // loop (one exec-masked block per iteration), so an iteration retires one pivot of EVERY busy group.
        if (__any(stage == 2)) {
            const unsigned lloc = ((unsigned)(live >> row0) & RMASK);
            uint64_t todo = (stage == 2) ? live : 0ull;
            if (stage == 2) nlp += __popcll(live);
#if PLP_R_PRESOLVE
            {   // rows the ray presolve settles as "keep" need no simplex run
                const unsigned okb = f2_presolve<D, R>(myA, myb, myan, row0, m_max, (stage == 2) ? lloc : 0u, abs_tol);
                uint64_t cert = 0ull;
#pragma unroll
                for (int k = 0; k < R; ++k) cert |= spread_rows<R, GS>(grp_ballot(((okb >> k) & 1u) != 0u, g)) << k;
                keep |= cert;
                todo &= ~cert;
                ctr_add(ctr, -__popcll(cert), g.gl == 0);
            }
#endif
            SimplexR<D, R, false, false> S;
            S.reset(D, __popcll(live), row0);
            S.mode = M_DONE;
            S.status = ST_OPT;
            bool busy = false;
            int kr = 0, e = -1, chi = 0;
            double cxc = 0.0, best = 0.0;
#if PLP_R_POOL
            if constexpr (rows <= 32 && PLP_R_PRESOLVE) {
                // ---- pooled: the LPs the presolve left are few (2.1 per polytope at C2) and unevenly spread -- with every
                // group on its own list the wavefront waits for the group with most left (15.9 iterations for a mean of
                // 6.6 pivots).  Here the LPs of all polytopes of the tile form ONE list (polytope order, then row order)
                // and any lane group takes the next one: the rows are in LDS anyway, and the LPs of a polytope are
                // independent once the in-place h[k] +- 0.1 round trip (:1149-1151) is written as a rule (an unsettled
                // row that had its turn before row k carries (b + 0.1) - 0.1, row k itself b + 0.1; settled rows carry the
                // round trip in LDS already, see f2_presolve) -- the rule reduce_split_kernel uses.  Verdicts travel back
                // as a bit per list position, OR-ed over the groups at the end.
                // (registers are what this kernel runs out of: the masks are kept as 32-bit words, whatever can be
                // recomputed from them at the end of a round is, and nlp / flags go out before the loop)
                const unsigned todo32 = (unsigned)todo;                       // unsettled live rows of MY polytope
                const unsigned live32 = (stage == 2) ? (unsigned)live : 0u;   // (the settled ones: live32 & ~todo32)
                const int n_g = __popc(todo32);
                int total = 0, nmax = 0;
                for (int p = 0; p < NGRP; ++p) {
                    const int np = __builtin_amdgcn_readlane(n_g, p * GS);
                    total += np;
                    nmax = np > nmax ? np : nmax;
                }
                const int lane = g.lane;
                bool pool_retry = false;
                for (int rb = 0; rb < total; rb += 64) {
                    const int rend = total < rb + 64 ? total : rb + 64;
                    // position t = rb + lane of the list -> (polytope, row)
                    int ent = -1;
                    {
                        const int t = rb + lane;
                        int tp = 0, toff = 0, run = 0;
                        unsigned ttd = 0u;
                        for (int p = 0; p < NGRP; ++p) {
                            const int np = __builtin_amdgcn_readlane(n_g, p * GS);
                            const unsigned tdp = (unsigned)__builtin_amdgcn_readlane((int)todo32, p * GS);
                            const bool at = t >= run;
                            tp = at ? p : tp;
                            ttd = at ? tdp : ttd;
                            toff = at ? run : toff;
                            run += np;
                        }
                        const int rank = t - toff;
                        for (int i = 0; i < nmax; ++i) ttd = (i < rank) ? (ttd & (ttd - 1u)) : ttd;
                        ent = (t < total) ? ((tp << 8) | (__ffs((int)ttd) - 1)) : -1;
                    }
                    int next = rb;             // wave-uniform
                    uint64_t res = 0ull;       // bit (t - rb): the LP at position t says "keep" (every lane of the group that solved it)
                    int cur_t = 0;
                    int cur_p = gib;           // the polytope whose LP my group is solving
                    for (;;) {
                        const bool fin = busy & (S.mode == M_DONE);
                        const bool want = fin | !busy;
                        const uint64_t wb = __ballot(want & (g.gl == 0));
                        const int tsk = next + __popcll(wb & ((1ull << g.gbase) - 1ull));
                        const bool start = want & (tsk < rend);
                        {
                            const int adv = next + __popcll(wb);
                            next = adv < rend ? adv : rend;
                        }
                        if (__any(fin | start)) {   // (wave-uniform: every lane is active for the cross-lane reads below)
                            const int entv = __builtin_amdgcn_ds_bpermute((tsk & 63) << 2, ent);
                            const int tp = start ? (entv >> 8) : grp;
                            const unsigned lv = (unsigned)__builtin_amdgcn_ds_bpermute((tp * GS) << 2, (int)live32);
                            const unsigned ct = lv & ~(unsigned)__builtin_amdgcn_ds_bpermute((tp * GS) << 2, (int)todo32);
                            if (fin) {
                                pool_retry = pool_retry | (S.status == ST_RETRY);
                                const double fun = cxc - S.negz;               // c.xc + zeta, zeta = -negz
                                const double hk = (sb[cur_p * rows + kr] + 0.1) - 0.1;   // h[k] after its round trip (:1149-1151)
                                const double obj = -fun - hk;                  // (:1156)
                                const bool keepk = ((S.status == ST_OPT) & (obj > abs_tol)) | (S.status == ST_UNBND);
                                res |= keepk ? (1ull << (cur_t - rb)) : 0ull;
                                busy = false;
                            }
                            if (start) {
                                cur_t = tsk;
                                kr = entv & 255;
                                cur_p = tp;
                                const double* pA = sA + tp * rows * D;
                                const double* pb = sb + tp * rows;
                                const double* pan = san + tp * rows;
                                S.reset(D, __popc(lv), row0);
#pragma unroll
                                for (int kk = 0; kk < D; ++kk) S.cost[kk] = -pA[kr * D + kk];  // f = -A[k,:]  (:1145)
                                cxc = -pan[kr];   // c.xc = -(a_k.xc): the mirror image of the FMA chain that made s_k
                                const unsigned unsettled = (lv & ~ct) >> row0;
#pragma unroll
                                for (int k = 0; k < R; ++k) {
#pragma unroll
                                    for (int kk = 0; kk < D; ++kk) S.T[k][kk] = pA[(row0 + k) * D + kk];
                                    const int rw = row0 + k;
                                    const double h0 = pb[rw];
                                    const double hp = h0 + 0.1;
                                    const bool un = ((unsettled >> k) & 1u) != 0u;
                                    const double hh = (un & (rw < kr)) ? hp - 0.1 : ((rw == kr) ? hp : h0);
                                    S.beta[k] = fmax(hh - pan[rw], 0.0);  // 0 for the zeroed rows
                                }
                                S.ract = (lv >> row0) & RMASK;
                                S.mode = M_P2;
                                S.status = -1;
                                S.scan_enter(e, best, chi);
                                if (e < 0) { S.status = ST_OPT; S.mode = M_DONE; }
                                busy = true;
                            }
                        }
                        if (!__any(busy)) break;
                        if ((S.mode != M_DONE) & (S.ndeg >= BLAND_AFTER)) { S.status = ST_RETRY; S.mode = M_DONE; }
                        S.template pivot_core<GS, 0>(g, e, best, chi, nullptr, 0u);
                    }
                    // verdicts of this round to every lane, then each owner picks those of its rows
                    {
                        unsigned rlo = (unsigned)res, rhi = (unsigned)(res >> 32);
#pragma unroll
                        for (int o = GS; o < 64; o <<= 1) {
                            rlo |= (unsigned)__shfl_xor((int)rlo, o, 64);
                            rhi |= (unsigned)__shfl_xor((int)rhi, o, 64);
                        }
                        const uint64_t all = ((uint64_t)rhi << 32) | rlo;
                        int off_g = 0;   // position of my polytope's first LP in the list
                        for (int p = 0, run = 0; p < NGRP; ++p) {
                            off_g = (grp == p) ? run : off_g;
                            run += __builtin_amdgcn_readlane(n_g, p * GS);
                        }
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            const int rw = row0 + k;
                            const int t = off_g + __popc(todo32 & ((1u << rw) - 1u)) - rb;
                            const bool mine = (((todo32 >> rw) & 1u) != 0u) & (t >= 0) & (t < 64);
                            const bool kept = mine & (((all >> (t & 63)) & 1ull) != 0ull);
                            keep |= spread_rows<R, GS>(grp_ballot(kept, g)) << k;
                        }
                    }
                }
                retry = retry | (__any(pool_retry) != 0);   // (rare: the whole tile is redone by the general kernel)
            } else
#endif
            for (;;) {
                const bool fin = busy & (S.mode == M_DONE);
                const bool start = (fin | !busy) & (todo != 0ull);
#ifdef PLP_STAGE_STATS
                if (threadIdx.x == 0) PLP_STAT_ADD(6, 1);
#endif
                if (__any(fin | start)) {
#ifdef PLP_STAGE_STATS
                    if (threadIdx.x == 0) PLP_STAT_ADD(7, 1);
                    if (fin & (g.gl == 0)) { PLP_STAT_ADD(8, S.iters); PLP_STAT_ADD(9, 1); }
#endif
                    if (fin) {
                        retry = retry | (S.status == ST_RETRY);
                        const double fun = cxc - S.negz;  // c.xc + zeta, zeta = -negz
                        // b[k] after the (+0.1, -0.1) round trip (:1149-1151): computed by the lane that owns row k
                        // and handed to the others through registers (an LDS store of one lane followed by loads of
                        // other lanes would need a fence for the compiler, which otherwise keeps an earlier load)
                        const bool owner = (kr >> RSH) == g.gl;
                        double hk_own = 0.0;
                        if (owner) { hk_own = myb[kr] - 0.1; myb[kr] = hk_own; }
                        const double hk = bcast(hk_own, g.gbase + (kr >> RSH));
                        const double obj = -fun - hk;  // (:1156)
                        const bool keepk = ((S.status == ST_OPT) & (obj > abs_tol)) | (S.status == ST_UNBND);
                        keep |= keepk ? (1ull << kr) : 0ull;
                        busy = false;
                    }
                    if (start) {
                        kr = __ffsll((long long)todo) - 1;
                        todo &= todo - 1ull;
                        S.reset(D, __popcll(live), row0);
#pragma unroll
                        for (int kk = 0; kk < D; ++kk) S.cost[kk] = -myA[kr * D + kk];  // f = -A[k,:]  (:1145)
                        cxc = -myan[kr];   // c.xc = -(a_k.xc): the mirror image of the FMA chain that made s_k
                        if ((kr >> RSH) == g.gl) myb[kr] = myb[kr] + 0.1;  // h[k] += 0.1 in place (:1149)
#pragma unroll
                        for (int k = 0; k < R; ++k) {
#pragma unroll
                            for (int kk = 0; kk < D; ++kk) S.T[k][kk] = myA[(row0 + k) * D + kk];
                            S.beta[k] = fmax(myb[row0 + k] - myan[row0 + k], 0.0);  // 0 for the zeroed rows
                        }
                        S.ract = lloc;
                        S.mode = M_P2;
                        S.status = -1;
                        S.scan_enter(e, best, chi);
                        if (e < 0) { S.status = ST_OPT; S.mode = M_DONE; }
                        busy = true;
                    }
                }
                if (!__any(busy)) break;
                if ((S.mode != M_DONE) & (S.ndeg >= BLAND_AFTER)) { S.status = ST_RETRY; S.mode = M_DONE; }
                S.template pivot_core<GS, 0>(g, e, best, chi, nullptr, 0u);
            }
            if (stage == 2) flags |= RF_MINREP;
        }
#else
        if (__any(stage == 2)) {
            const unsigned lloc = ((unsigned)(live >> row0) & RMASK);
            uint64_t todo = (stage == 2) ? live : 0ull;
            if (stage == 2) nlp += __popcll(live);
            while (__any(todo != 0ull)) {
                const bool go = todo != 0ull;
                const int kr = go ? __ffsll((long long)todo) - 1 : 0;
                todo &= todo - 1ull;
                SimplexR<D, R, false, false> S;
                S.reset(D, __popcll(live), row0);
                double cxc = 0.0;
#pragma unroll
                for (int kk = 0; kk < D; ++kk) {
                    const double ck = -myA[kr * D + kk];  // f = -A[k,:]  (:1145)
                    S.cost[kk] = ck;
                    cxc = fma(ck, xc[kk], cxc);
                }
                // h[k] += 0.1 in place, as the reference does (:1149); undone after the LP (:1151), so rows
                // k' < k carry the (+0.1, -0.1) round trip into the later LPs
                const bool owner = go & ((kr >> RSH) == g.gl);
                if (owner) myb[kr] = myb[kr] + 0.1;
#pragma unroll
                for (int k = 0; k < R; ++k) {
#pragma unroll
                    for (int kk = 0; kk < D; ++kk) S.T[k][kk] = myA[(row0 + k) * D + kk];
                    S.beta[k] = fmax(myb[row0 + k] - myan[row0 + k], 0.0);  // 0 for the zeroed rows
                }
                S.ract = lloc;
                S.mode = go ? M_P2 : M_DONE;
#if PLP_R_FAST
                S.template run_fast<GS>(g);
                retry = retry | (go & (S.status == ST_RETRY));
#else
                S.run(g);
#endif
                const double fun = cxc - S.negz;  // c.xc + zeta, zeta = -negz
                // b[k] after the round trip: computed by the lane that owns row k and handed to the others
                // through registers (an LDS store of one lane followed by loads of other lanes would need a
                // fence for the compiler, which otherwise keeps an earlier load)
                double hk_own = 0.0;
                if (owner) { hk_own = myb[kr] - 0.1; myb[kr] = hk_own; }
                const double hk = bcast(hk_own, g.gbase + (kr >> RSH));
                const double obj = -fun - hk;     // (:1156)
                const bool keepk = go & (((S.status == ST_OPT) & (obj > abs_tol)) | (S.status == ST_UNBND));
                keep |= keepk ? (1ull << kr) : 0ull;
            }
            if (stage == 2) flags |= RF_MINREP;
        }
#endif
        }
        // ---------------------------------------------------------------- results
        if (valid & (g.gl == 0) & (!SPLIT || grp == 0)) {
            keep_out[pg] = keep;
            flags_out[pg] = retry ? (int)RF_RETRY : flags;
            nlp_out[pg] = nlp;
        }
        ctr_add(ctr, nlp, valid & (g.gl == 0) & (!SPLIT || grp == 0));   // every LP the reference issues, less the presolved ones
        // a polytope handed to the general kernel: say so in the call's word, so that the second pass -- which normally
        // finds nothing -- can leave on ONE load instead of sweeping the flags of the whole batch
        if (retry_word) {
            if (__any(retry & valid)) {
                if ((threadIdx.x & 63) == 0) atomicMax(retry_word, epoch);
            }
        }
    }
}

// One polytope of up to 64 rows per wavefront, F3 / F2 on plp_lazy.hpp (d = 9..16, see there).
#ifndef PLP_REDUCE_LAZY_WAVES
#define PLP_REDUCE_LAZY_WAVES 3
#endif
#ifndef PLP_REDUCE_WDENSE_MAXD
// one polytope per wavefront: F3 / F2 on the dense one-LP-per-wavefront engine up to this d, without a stored dictionary
// beyond (measured, scripts/debug/wdense_ab.py, 64 rows, B = 20 000, ms dense / lazy: d = 8 1.39 / 2.30, 12 1.66 / 1.90,
// 13 1.58 / 1.65, 14 1.48 / 1.48, 15 1.48 / 1.38, 16 1.60 / 1.33)
#define PLP_REDUCE_WDENSE_MAXD 13
#endif
#ifndef PLP_REDUCE_WDENSE_WAVES
#define PLP_REDUCE_WDENSE_WAVES 4
#endif
#ifndef PLP_REDUCE_WDENSE_WAVES8
#define PLP_REDUCE_WDENSE_WAVES8 6   // d <= 8 (89 VGPRs at d = 8 unconstrained: the presolve; 78 under the bound, nothing spills)
#endif
template <int D>
__global__ __launch_bounds__(RBLOCK, PLP_REDUCE_LAZY_WAVES) void reduce_lazy_kernel(
    long long B, int m_max, const double* __restrict__ Ag, const double* __restrict__ bg,
    const int* __restrict__ mrows, double abs_tol, int force_retry, unsigned long long* __restrict__ keep_out,
    int* __restrict__ flags_out, double* __restrict__ r_out, double* __restrict__ xc_out,
    int* __restrict__ nlp_out, unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ retry_word,
    unsigned long long epoch) {
    reduce_r_tile<D, 64, 1, true>((long long)blockIdx.x, B, m_max, Ag, bg, mrows, abs_tol, force_retry, keep_out, flags_out,
                                  r_out, xc_out, nlp_out, ctr, retry_word, epoch);
}

// The same with the F3 / F2 LPs on the one-LP-per-wavefront DENSE engine (plp_wide.hpp: wide::solve_dense): the
// dictionary is carried, the pivot column and row are wave-uniform.
template <int D>
__global__ __launch_bounds__(RBLOCK, ((D) <= 8 ? PLP_REDUCE_WDENSE_WAVES8 : PLP_REDUCE_WDENSE_WAVES)) void reduce_wdense_kernel(
    long long B, int m_max, const double* __restrict__ Ag, const double* __restrict__ bg,
    const int* __restrict__ mrows, double abs_tol, int force_retry, unsigned long long* __restrict__ keep_out,
    int* __restrict__ flags_out, double* __restrict__ r_out, double* __restrict__ xc_out,
    int* __restrict__ nlp_out, unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ retry_word,
    unsigned long long epoch) {
    reduce_r_tile<D, 64, 1, true, false, true>((long long)blockIdx.x, B, m_max, Ag, bg, mrows, abs_tol, force_retry, keep_out,
                                               flags_out, r_out, xc_out, nlp_out, ctr, retry_word, epoch);
}

// f2_presolve<D, 1> with the two passes over the rows cut into NW ranges, one per wavefront of reduce_wsplit_kernel (lane =
// candidate row in EVERY wavefront): the verdict of a row is the AND of the ranges' verdicts, its blocking row the last one
// found in row order -- the same bits as the one-wavefront function.  Nothing is written to the rows here (the caller
// applies the round trip of the settled rows after its next barrier).  smask: 2 NW words, sjb: NW x 64 ints in LDS.
template <int D, int NW>
__device__ __forceinline__ bool f2_presolve_ws(const double* myA, const double* myb, const double* myan, const int lane,
                                               const int m_loop, bool cand, const double abs_tol, const int w,
                                               unsigned long long* smask, int* sjb) {
    double ak[D];
    double g = 0.0;
#pragma unroll
    for (int kk = 0; kk < D; ++kk) {
        ak[kk] = myA[lane * D + kk];
        g = fma(ak[kk], ak[kk], g);
    }
    const double bk = myb[lane];
    const double sk = fmax(bk - myan[lane], 0.0);
    const double tau = abs_tol + 1e-9 * (1.0 + fabs(bk) + sk);
    const double skt = sk + tau;
    if (!((g > 0.0) & (tau < 0.05))) cand = false;  // (NaN-safe: such a row is never settled)
    const int iw = (m_loop + NW - 1) / NW;
    const int i0 = w * iw;
    const int i1 = i0 + iw < m_loop ? i0 + iw : m_loop;
    bool ok = cand;
    int jb = -1;
    for (int i = i0; i < i1; ++i) {
        double ai[D];
#pragma unroll
        for (int kk = 0; kk < D; ++kk) ai[kk] = myA[i * D + kk];
        const double si = fmax(myb[i] - myan[i], 0.0);
        double gik = 0.0;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) gik = fma(ai[kk], ak[kk], gik);
        const bool fine = (i == lane) | (skt * gik <= si * g);
        ok = ok & fine;
        jb = fine ? jb : i;
    }
    {
        const uint64_t okm = __ballot(ok);
        if (lane == 0) smask[w] = okm;
        sjb[w * 64 + lane] = jb;
    }
    __syncthreads();
    bool okall = cand;
    int j = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        okall = okall & (((smask[q] >> lane) & 1ull) != 0ull);
        const int jq = sjb[q * 64 + lane];
        j = jq >= 0 ? jq : j;
    }
#if PLP_R_PRESOLVE >= 2
    bool ok2 = cand & !okall;
    if (__any(ok2)) {   // (the same in every wavefront: the barrier below is met by all or by none)
        double aj[D];
        double gjk = 0.0, gjj = 0.0;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            aj[kk] = myA[j * D + kk];
            gjk = fma(aj[kk], ak[kk], gjk);
            gjj = fma(aj[kk], aj[kk], gjj);
        }
        const double sj = fmax(myb[j] - myan[j], 0.0);
        const double rho = gjk / gjj;
        const double t1 = (sj / gjk) * (1.0 - 0x1p-40);
        const double akd = fma(-rho, gjk, g);
        const double t2 = fma(-t1, g, skt) / akd;
        const double c1 = t1 + t2;
        const double c2 = t2 * rho;
        if (!((gjk > 0.0) & (akd > 1e-12 * g) & (t2 >= 0.0) & (t2 < 1e300))) ok2 = false;
        for (int i = i0; i < i1; ++i) {
            double ai[D];
#pragma unroll
            for (int kk = 0; kk < D; ++kk) ai[kk] = myA[i * D + kk];
            const double si = fmax(myb[i] - myan[i], 0.0);
            double gik = 0.0, gij = 0.0;
#pragma unroll
            for (int kk = 0; kk < D; ++kk) {
                gik = fma(ai[kk], ak[kk], gik);
                gij = fma(ai[kk], aj[kk], gij);
            }
            const double lhs = fma(-c2, gij, c1 * gik);
            const bool fine = (i == lane) | (lhs <= si);
            ok2 = ok2 & fine;
        }
        const uint64_t m2 = __ballot(ok2);
        if (lane == 0) smask[NW + w] = m2;
        __syncthreads();
        bool all2 = true;
#pragma unroll
        for (int q = 0; q < NW; ++q) all2 = all2 & (((smask[NW + q] >> lane) & 1ull) != 0ull);
        okall = okall | (cand & !okall & all2);
    }
#endif
    return okall;
}

// ---- the same polytope-per-workgroup pipeline with the INDEPENDENT LPs of a polytope spread over NW wavefronts.
// At the batch sizes where one wavefront per polytope cannot fill the chip (5 000 polytopes are 4.9 wavefronts per SIMD,
// all resident from the start: the launch is a drain, 2-3 waves per SIMD on average, profiles/r04/r04e_wide_counters.json)
// a workgroup of NW wavefronts takes one polytope: wavefront 0 runs F1 (while the others run the dedupe, its partner
// loop cut into NW - 1 ranges), the prefilter and the presolve (lane = row, as reduce_wdense_kernel), and wavefront w takes every NW-th of the 2d box LPs and of the
// redundancy LPs the presolve left.  Each LP is
// wide::solve_dense on the rows in LDS exactly as in reduce_wdense_kernel; the in-place h[k] +- 0.1 round trip
// (:1149-1151) becomes the rule reduce_split_kernel uses (an unsettled row that had its turn before row k carries
// (b + 0.1) - 0.1, row k itself b + 0.1), so the outputs are bit for bit those of the one-wavefront form.
#ifndef PLP_REDUCE_WSPLIT_MAXB
// Measured (scripts/debug/wsplit_sweep.py, ms with one / two / four wavefronts per polytope):
//   (64,8)   B = 1  0.192 / 0.116 / 0.085    250  0.230 / 0.142 / 0.100    1 000  0.251 / 0.162 / 0.131    2 000  0.292 / 0.212 / 0.222
//            5 000  0.444 / 0.417 / 0.432    8 000  0.658 / 0.597 / 0.640    16 000  1.127 / 1.077 / 1.218
//   (48,6)   250  0.159 / 0.096 / 0.070    5 000  0.301 / 0.253 / 0.254    16 000  0.678 / 0.620 / 0.681
//   (64,12)  250  0.267 / 0.161 / 0.120    5 000  0.580 / 0.495 / 0.543    16 000  1.336 / 1.319 / 1.591
// two wavefronts per polytope up to here, four up to PLP_REDUCE_WSPLIT_MAXB4
#define PLP_REDUCE_WSPLIT_MAXB 16000
#endif
#ifndef PLP_REDUCE_WSPLIT_MAXB4
#define PLP_REDUCE_WSPLIT_MAXB4 2000   // four wavefronts per polytope up to here
#endif
template <int D>
__host__ __device__ constexpr size_t wsplit_block_bytes() { return (sizeof(wide::WideShared<D + 1>) + 15) & ~(size_t)15; }
template <int D, int NW>
static inline size_t reduce_wsplit_smem_bytes() {
    return (size_t)64 * (D + 2) * 8 + (size_t)(D + 2 + 2 * D) * 8 + 8 * 8 + 8 * 4 + 64 * 4 + NW * wsplit_block_bytes<D>() +
           (size_t)2 * NW * 8 + (size_t)NW * 64 * 4;  // + the presolve's masks and blocking rows
}
// (d <= 8: the presolve is the register peak -- 94 VGPRs at d = 8 --; held to 80, six waves per SIMD, nothing spills: 78.
//  From d = 9 on the same bound sends the 16-wide row vector to scratch: unconstrained there, 92..130)
#ifndef PLP_REDUCE_WSPLIT_WAVES8
#define PLP_REDUCE_WSPLIT_WAVES8 6
#endif
#define PLP_REDUCE_WSPLIT_WAVES(D) ((D) <= 8 ? PLP_REDUCE_WSPLIT_WAVES8 : 1)
template <int D, int NW, bool WDENSE = true>
__global__ __launch_bounds__(64 * NW, PLP_REDUCE_WSPLIT_WAVES(D)) void reduce_wsplit_kernel(
    long long B, int m_max, const double* __restrict__ Ag, const double* __restrict__ bg,
    const int* __restrict__ mrows, double abs_tol, int force_retry, unsigned long long* __restrict__ keep_out,
    int* __restrict__ flags_out, double* __restrict__ r_out, double* __restrict__ xc_out,
    int* __restrict__ nlp_out, unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ retry_word,
    unsigned long long epoch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int rows = 64;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long pg = blockIdx.x;
    double* myA = reinterpret_cast<double*>(smem_raw);   // [64][D]
    double* myb = myA + rows * D;                         // [64]
    double* myan = myb + rows;                            // [64]  1/||a_i||, then a_i.xc
    double* sxc = myan + rows;                            // [D + 2]  xc, r
    double* sval = sxc + (D + 2);                         // [2 D]    box values
    unsigned long long* s64 = reinterpret_cast<unsigned long long*>(sval + 2 * D);  // removed, kept, todo, live
    unsigned* s32 = reinterpret_cast<unsigned*>(s64 + 8);  // flags (1: an LP failed, 2: retry), next box LP, next F2 LP, ball bits
    int* slist = reinterpret_cast<int*>(s32 + 8);          // [64] the rows whose LP runs, in row order
    unsigned char* shb = reinterpret_cast<unsigned char*>(slist + 64) + (size_t)w * wsplit_block_bytes<D>();
    unsigned long long* smask = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(slist + 64) +
                                                                      (size_t)NW * wsplit_block_bytes<D>());  // [2 NW]
    int* sjb = reinterpret_cast<int*>(smask + 2 * NW);                                                        // [NW][64]
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    {
        const int rowsz = m_max * D;
        const double* src = Ag + pg * rowsz;
        for (int idx = threadIdx.x; idx < rowsz; idx += 64 * NW) myA[idx] = src[idx];
        const double* srcb = bg + pg * m_max;
        for (int idx = threadIdx.x; idx < m_max; idx += 64 * NW) myb[idx] = srcb[idx];
        if (threadIdx.x < 8) { s64[threadIdx.x] = 0ull; s32[threadIdx.x] = 0u; }
    }
    __syncthreads();
    const int m = mrows ? mrows[pg] : m_max;
    const bool has = (lane < m) & (m <= rows);
    bool retry = force_retry != 0;
    // ---------------------------------------------------------------- F1: Chebyshev ball (wavefront 0) while the others dedupe
    constexpr int NC1 = D + 1;
    wide::WideShared<NC1>& sh1 = *reinterpret_cast<wide::WideShared<NC1>*>(shb);
    typename wide::RowVec<NC1>::type Tv = (typename wide::RowVec<NC1>::type)(0.0);
    double T16 = 0.0, beta1 = 0.0, q01 = 0.0;
    int rowvar1 = NC1 + lane, rowneg1 = 0;
    bool rowact1 = false, infeasible0 = false, bad1 = false;
    if (w == 0) {
        double nrm2 = 0.0;
        bool finite = true;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            const double v = has ? myA[lane * D + kk] : 0.0;
            ROW_SET(kk, v);
            nrm2 = nrm2 + v * v;
            finite = finite & isfinite(v);
        }
        const double bk = has ? myb[lane] : 0.0;
        finite = finite & isfinite(bk);
        const double nrm = sqrt(nrm2);
        myan[lane] = 1.0 / nrm;
        const bool zero = !(nrm > 0.0);
        rowact1 = has & !zero;
        ROW_SET(D, rowact1 ? nrm : 0.0);
        beta1 = rowact1 ? bk : 0.0;
        q01 = bk / nrm;
        if (lane <= NC1) {
            sh1.cost[lane] = lane == D ? -1.0 : 0.0;
            sh1.cv[lane] = (lane + 1) << 1;
        }
        infeasible0 = __ballot(has & zero & (bk < -TOL_FEAS)) != 0;
        bad1 = (__ballot(!finite) != 0) | (m > rows);
    }
    __syncthreads();  // 1/||a|| of every row
    if (w == 0) {
        int st1, it1 = 0;
        if (bad1) st1 = ST_NUM;
        else if (infeasible0) st1 = ST_INFEAS;
        else st1 = wide::wide_run<NC1>(lane, m, Tv, T16, beta1, rowvar1, rowneg1, rowact1, sh1, NC1, true, q01, it1);
        const double mine = rowneg1 ? -beta1 : beta1;
        double xc1[D], rr1 = 0.0;
#pragma unroll
        for (int j = 0; j <= D; ++j) {
            const uint64_t ob = __ballot(rowvar1 == j);
            const double xj = ob ? wide::uniform_lane(mine, __ffsll((long long)ob) - 1) : 0.0;
            if (j < D) xc1[j < D ? j : 0] = xj; else rr1 = xj;
        }
        bool ball1 = (st1 == ST_OPT) & (rr1 >= 0.0);  // cheby_ball: status 0 and r >= 0 (:1289-1293)
        bool full1 = ball1 & (rr1 > abs_tol);
        {   // a centre that violates a row (centre_off, plp_common.hpp) is no centre: RF_F1OPEN
            double sk = 0.0;
#pragma unroll
            for (int kk = 0; kk < D; ++kk) sk = fma(has ? myA[lane * D + kk] : 0.0, xc1[kk], sk);
            const double bk = has ? myb[lane] : 0.0;
            if (ball1 & (__ballot(has & centre_off(bk - sk, myan[lane], bk, centre_scale<D>(xc1))) != 0)) {
                ball1 = false; full1 = false; st1 = ST_NUM;
            }
        }
        if (lane == 0) {
            r_out[pg] = ball1 ? rr1 : 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) { xc_out[pg * D + k] = ball1 ? xc1[k] : qnan; sxc[k] = xc1[k]; }
            sxc[D] = rr1;
            s32[3] = (ball1 ? 1u : 0u) | (full1 ? 2u : 0u) | (((st1 != ST_OPT) & (st1 != ST_INFEAS)) ? 4u : 0u);   // bit 2: RF_F1OPEN
        }
    } else {
        // ------------------------------------------------------------ dedupe (:1094-1110): the partners j in NW - 1 ranges
        bool removed = false;
        double ni[D];
        const double an_i = myan[lane];
#pragma unroll
        for (int kk = 0; kk < D; ++kk) ni[kk] = myA[lane * D + kk] * an_i;
        const double bin_ = myb[lane] * an_i;
        const int jw = (m_max + NW - 2) / (NW - 1);
        const int j1 = w * jw < m_max ? w * jw : m_max;
        for (int j = (w - 1) * jw; j < j1; ++j) {
            const bool jrow = j < m;
            const double an_j = myan[j];
            double dot = 0.0;
#pragma unroll
            for (int kk = 0; kk < D; ++kk) dot = dot + ni[kk] * (myA[j * D + kk] * an_j);
            const double bjn = myb[j] * an_j;
            const bool par = has & jrow & (j != lane) & (dot > 1.0 - abs_tol);
            const bool rem = par & ((lane < j) ? !(bin_ < bjn) : (bjn < bin_));
            removed = removed | rem;
        }
        const uint64_t rb = __ballot(removed);
        if ((lane == 0) & (rb != 0ull)) atomicOr(&s64[0], (unsigned long long)rb);
    }
    __syncthreads();  // the ball, the removed rows
    double xc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) xc[k] = sxc[k];
    const bool ball = (s32[3] & 1u) != 0u, fulldim = (s32[3] & 2u) != 0u;
    uint64_t live = __ballot(has) & ~(uint64_t)s64[0];
    auto zero_dead = [&](bool alive) {   // (wavefront 0, lane = row) rows that dropped out are zeroed: A, b and s
        if (!alive) {
#pragma unroll
            for (int kk = 0; kk < D; ++kk) myA[lane * D + kk] = 0.0;
            myb[lane] = 0.0;
            myan[lane] = 0.0;
        }
    };
    if (w == 0) {   // the dictionary translated to the Chebyshev centre: s_i = a_i.xc replaces 1/||a_i||
        double sk = 0.0;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) sk = fma(has ? myA[lane * D + kk] : 0.0, ball ? xc[kk] : 0.0, sk);
        myan[lane] = sk;
        zero_dead(((live >> lane) & 1ull) != 0ull);
    }
    int flags = fulldim ? 0 : (RF_EMPTY | ((s32[3] & 4u) ? RF_F1OPEN : 0));
    int nlp = 1;
    uint64_t keep = 0ull;
    int stage = 0;
    if (fulldim) {
        const int neq = __popcll(live);
        if (neq <= D + 1) { flags = RF_EARLY; keep = live; }
        else stage = (neq > 3 * D) ? 1 : 2;
    }
    __syncthreads();
    wide::WideShared<D>& shd = *reinterpret_cast<wide::WideShared<D>*>(shb);
    // ---------------------------------------------------------------- F3: bounding box (:1367-1409), any wavefront the next LP
    if (stage == 1) {
        const bool act = ((live >> lane) & 1ull) != 0ull;
        const int nlive = __popcll(live);
        for (int it = w; it < 2 * D; it += NW) {   // (w is a scalar: the loop stays wave-uniform)
            const int kx = it >> 1;
            const bool up = it & 1;
            double xck = 0.0;
#pragma unroll
            for (int kk = 0; kk < D; ++kk) xck = (kk == kx) ? xc[kk] : xck;
            double negz = 0.0;
            int st;
            if constexpr (WDENSE)
                st = wide::solve_dense<D>(lane, nlive, myA, (lane == kx) ? (up ? -1.0 : 1.0) : 0.0,
                                          fmax(myb[lane] - myan[lane], 0.0), act, negz, shd);
            else   // (d beyond PLP_REDUCE_WDENSE_MAXD: the LPs without a stored dictionary, plp_lazy.hpp)
                st = lazy::solve<D>(lane, nlive, myA, (lane == kx) ? (up ? -1.0 : 1.0) : 0.0,
                                    fmax(myb[lane] - myan[lane], 0.0), act, negz);
            double val;
            unsigned fl = 0u;
            if (st == ST_OPT) val = up ? (xck + negz) : (xck - negz);
            else if (st == ST_UNBND) val = up ? pinf : -pinf;
            else { val = qnan; fl = 1u; }
            if (st == ST_RETRY) fl |= 2u;
            if (lane == 0) {
                sval[it] = val;
                if (fl) atomicOr(&s32[0], fl);
            }
        }
        __syncthreads();
        if (w == 0) {   // prefilter sums, accumulated in k order (:1131-1134)
            double s1 = 0.0, s2 = 0.0;
            for (int kx = 0; kx < D; ++kx) {
                const double lo = sval[2 * kx], hi = sval[2 * kx + 1];
                const double aik = myA[lane * D + kx];
                const double pa = (aik > 0.0 ? 1.0 : 0.0) * aik;
                s1 = s1 + pa * (hi - lo);
                s2 = s2 + aik * lo;
            }
            const bool out = act & ((s1 - (myb[lane] - s2)) < -1e-4);
            const uint64_t l2 = live & ~__ballot(out);
            zero_dead(((l2 >> lane) & 1ull) != 0ull);
            if (lane == 0) s64[3] = l2;
        }
        __syncthreads();
        live = s64[3];
        const unsigned fl = s32[0];
        retry = retry | ((fl & 2u) != 0u);
        nlp += 2 * D;
        if (fl & 1u) flags |= RF_LPFAIL;
        if (__popcll(live) <= D + 1) { flags |= RF_EARLY; keep = live; stage = 0; }
        else stage = 2;
    }
    // ---------------------------------------------------------------- F2: redundancy LPs (:1142-1160)
    if (stage == 2) {
        const bool act = ((live >> lane) & 1ull) != 0ull;
        const int nlive = __popcll(live);
        nlp += nlive;
        uint64_t todo = live;
#if PLP_R_PRESOLVE
        // rows the presolve settles as "keep" need no LP: every wavefront takes a range of the rows to test against
        const bool settled = f2_presolve_ws<D, NW>(myA, myb, myan, lane, m_max, act, abs_tol, w, smask, sjb);
        const uint64_t cert = __ballot(settled);   // (lane i = row i: the ballot is the row mask, the same in every wavefront)
        todo &= ~cert;
        if (w == 0) {
            ctr_add(ctr, -__popcll(cert), lane == 0);
            if ((lane == 0) & (cert != 0ull)) atomicOr(&s64[1], (unsigned long long)cert);
            if (settled) myb[lane] = (myb[lane] + 0.1) - 0.1;  // (:1149-1151), as f2_presolve leaves the settled rows
        }
#endif
        if ((w == 0) && ((todo >> lane) & 1ull)) slist[__popcll(todo & ((1ull << lane) - 1ull))] = lane;
        __syncthreads();
        const int ntodo = __popcll(todo);
        const double h0 = myb[lane];
        const double sl = myan[lane];
        const bool mytodo = ((todo >> lane) & 1ull) != 0ull;
        for (int idx = w; idx < ntodo; idx += NW) {
            const int kr = __builtin_amdgcn_readfirstlane(slist[idx]);
            const double ck = lane < D ? -myA[kr * D + (lane < D ? lane : 0)] : 0.0;  // f = -A[k,:]  (:1145)
            const double cxc = -myan[kr];
            // h as the reference holds it when row kr has its turn (:1149-1151)
            const double hp = h0 + 0.1;
            const double hh = (mytodo & (lane < kr)) ? hp - 0.1 : ((lane == kr) ? hp : h0);
            double negz2 = 0.0;
            int st2;
            if constexpr (WDENSE) st2 = wide::solve_dense<D>(lane, nlive, myA, ck, fmax(hh - sl, 0.0), act, negz2, shd);
            else st2 = lazy::solve<D>(lane, nlive, myA, ck, fmax(hh - sl, 0.0), act, negz2);
            const double fun = cxc - negz2;  // c.xc + zeta, zeta = -negz
            const double hk = (myb[kr] + 0.1) - 0.1;
            const double obj = -fun - hk;    // (:1156)
            const bool keepk = ((st2 == ST_OPT) & (obj > abs_tol)) | (st2 == ST_UNBND);
            if (lane == 0) {
                if (keepk) atomicOr(&s64[1], 1ull << kr);
                if (st2 == ST_RETRY) atomicOr(&s32[0], 2u);
            }
        }
        __syncthreads();
        keep = s64[1];
        retry = retry | ((s32[0] & 2u) != 0u);
        flags |= RF_MINREP;
    }
    // ---------------------------------------------------------------- results
    if (w == 0) {
        if (lane == 0) {
            keep_out[pg] = keep;
            flags_out[pg] = retry ? (int)RF_RETRY : flags;
            nlp_out[pg] = nlp;
        }
        ctr_add(ctr, nlp, lane == 0);
        if (retry_word && retry && lane == 0) atomicMax(retry_word, epoch);
    }
}

template <int D>
static int launch_reduce_lazy(long long B, int m_max, const double* A, const double* b, const int* mrows, double abs_tol,
                              unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st) {
    const size_t smem = reduce_r_smem_bytes(64, D, 1) + lazy::lds_bytes<D>();
    if (B > 2147483647ll) return 2;
    const char* fr = getenv("PLP_REDUCE_RETRY_ALL");
    // PLP_REDUCE_WDENSE=0 / 1: F3 / F2 without / with a stored dictionary (A/B)
    const char* wd = getenv("PLP_REDUCE_WDENSE");
    const bool dense = wd ? wd[0] == '1' : (D <= PLP_REDUCE_WDENSE_MAXD);
    // 3: complete -- the dense LPs carry Bland's rule inside, no polytope is handed to a second pass (the caller then
    // launches none: at small batches the idle second launch was 5 % of the call)
    const int done = (dense && !(fr && fr[0] == '1')) ? 3 : 0;
    {
        // batches that leave the chip part empty at one wavefront per polytope: NW wavefronts per polytope
        // (PLP_REDUCE_WSPLIT=0: never, 2 / 4: always with that many; PLP_REDUCE_WSPLIT_MAXB / _MAXB4: the largest batches that take them)
        const char* ws = getenv("PLP_REDUCE_WSPLIT");
        const char* wb = getenv("PLP_REDUCE_WSPLIT_MAXB");
        const char* wb4 = getenv("PLP_REDUCE_WSPLIT_MAXB4");
        // (without a stored dictionary, d = 14..16: (64,16) B = 250 0.228 / 0.143 / 0.112 ms with one / two / four wavefronts,
        // 1 000 0.248 / 0.170 / 0.201, 3 000 0.355 / 0.346 / 0.437, 8 000 0.704 / 0.703 / 0.951)
        const long long maxb = wb ? atoll(wb) : (dense ? PLP_REDUCE_WSPLIT_MAXB : 3000);
        // (with the presolve over the wavefronts: four ahead of two up to 12 000 polytopes at d <= 8 -- (64,8) 5 000 0.379 / 0.402,
        // 12 000 0.803 / 0.810, 16 000 1.051 / 1.026 -- and up to ~3 000 at d = 9..13: (64,12) 3 000 0.338 / 0.366, 5 000 0.502 / 0.495)
        const long long maxb4 = wb4 ? atoll(wb4) : (dense ? (D <= 8 ? 12000 : PLP_REDUCE_WSPLIT_MAXB4) : 500);
        const int fi = (fr && fr[0] == '1') ? 1 : 0;
        int nw = 0;
        if (B >= 1 && !(ws && ws[0] == '0')) {
            if ((ws && ws[0] == '4') || (!ws && B <= maxb4)) nw = 4;
            else if ((ws && ws[0] == '2') || (!ws && B <= maxb)) nw = 2;
        }
#define PLP_WSPLIT_LAUNCH(NW_, DENSE_)                                                                                          \
        {                                                                                                                       \
            const size_t smem_ws = reduce_wsplit_smem_bytes<D, NW_>();                                                          \
            hipLaunchKernelGGL((reduce_wsplit_kernel<D, NW_, DENSE_>), dim3((unsigned)B), dim3(64 * NW_), smem_ws, st, B, m_max, A, b, \
                               mrows, abs_tol, fi, keep, flags, r, xc, nlp, t_reduce_ctr, t_reduce_retry, t_reduce_epoch);        \
            return done;                                                                                                        \
        }
        if constexpr (D <= PLP_REDUCE_WDENSE_MAXD) {
            if (dense && nw == 4) PLP_WSPLIT_LAUNCH(4, true)
            if (dense && nw == 2) PLP_WSPLIT_LAUNCH(2, true)
        } else {
            if (!dense && nw == 4) PLP_WSPLIT_LAUNCH(4, false)
            if (!dense && nw == 2) PLP_WSPLIT_LAUNCH(2, false)
        }
#undef PLP_WSPLIT_LAUNCH
    }
    if (dense)
        hipLaunchKernelGGL((reduce_wdense_kernel<D>), dim3((unsigned)(B < 1 ? 1 : B)), dim3(RBLOCK), smem, st, B, m_max, A, b,
                           mrows, abs_tol, (fr && fr[0] == '1') ? 1 : 0, keep, flags, r, xc, nlp, t_reduce_ctr, t_reduce_retry, t_reduce_epoch);
    else
        hipLaunchKernelGGL((reduce_lazy_kernel<D>), dim3((unsigned)(B < 1 ? 1 : B)), dim3(RBLOCK), smem, st, B, m_max, A, b,
                           mrows, abs_tol, (fr && fr[0] == '1') ? 1 : 0, keep, flags, r, xc, nlp, t_reduce_ctr, t_reduce_retry, t_reduce_epoch);
    return done;
}

// Small batches: one polytope per wavefront, its LPs spread over the lane groups (reduce_r_tile, SPLIT).
template <int D, int GS, int R = RR>
__global__ __launch_bounds__(RBLOCK, PLP_REDUCE_R_WAVES(D)) void reduce_split_kernel(
    long long B, int m_max, const double* __restrict__ Ag, const double* __restrict__ bg,
    const int* __restrict__ mrows, double abs_tol, int force_retry, unsigned long long* __restrict__ keep_out,
    int* __restrict__ flags_out, double* __restrict__ r_out, double* __restrict__ xc_out,
    int* __restrict__ nlp_out, unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ retry_word,
    unsigned long long epoch) {
    reduce_r_tile<D, GS, R, false, true>((long long)blockIdx.x, B, m_max, Ag, bg, mrows, abs_tol, force_retry, keep_out,
                                         flags_out, r_out, xc_out, nlp_out, ctr, retry_word, epoch);
}

// Batches up to this size take the latency form.  Measured (device time per call, batch form -> latency form): (16,3)
// B = 1: 49 -> 21 us, 256: 77 -> 25, 4096: 81 -> 50, 16384: 90 -> 153; (32,6) 256: 274 -> 79, 4096: 306 -> 202;
// (64,8) 256: 652 -> 257, 4096: 896 -> 728; (16,8) 256: 80 -> 75, 4096: 82 -> 111 (16 rows at d >= 7 gain nothing:
// the 2d box LPs already fill the 16 groups).  ~20 us of every figure are the launches of a call.
#ifndef PLP_REDUCE_SPLIT_MAXB
// (round 3, with the F2 presolve and d = 5..8 on two rows per lane beyond this size: (32,6) B = 2048: 0.121 ms here vs 0.181,
// B = 4096: 0.222 vs 0.187; (64,8) B = 1024: 0.302 vs 0.366, B = 2000: 0.426 vs 0.404)
#define PLP_REDUCE_SPLIT_MAXB(D, GS) \
    ((((GS) == 4 && (D) >= 7) || (D) > 8 || ((D) >= 5 && (GS) == 16)) ? 1024 : ((D) >= 5 ? 2048 : 4096))  // (d > 8: four groups only; (32,12) B = 4096: 154 -> 201 us)
#endif

template <int D, int GS, int R = RR>
__global__ __launch_bounds__(RBLOCK, (R == 8 ? PLP_REDUCE_R8_WAVES : (R == 2 && D <= 8 ? PLP_REDUCE_R2MID_WAVES : PLP_REDUCE_R_WAVES(D)))) void reduce_r_kernel(
    long long B, int m_max, const double* __restrict__ Ag, const double* __restrict__ bg,
    const int* __restrict__ mrows, double abs_tol, int force_retry, unsigned long long* __restrict__ keep_out,
    int* __restrict__ flags_out, double* __restrict__ r_out, double* __restrict__ xc_out,
    int* __restrict__ nlp_out, unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ retry_word,
    unsigned long long epoch) {
    reduce_r_tile<D, GS, R>((long long)blockIdx.x * (RBLOCK / GS), B, m_max, Ag, bg, mrows, abs_tol, force_retry, keep_out,
                            flags_out, r_out, xc_out, nlp_out, ctr, retry_word, epoch);
}

// Polytopes of up to 16 rows: the first `nbig` workgroups take tiles of 16 polytopes (4 lanes x 4 rows each), the rest
// tiles of 8 (8 lanes x 2 rows each), which finish in about half the time.  The launch drains over one tile lifetime
// (the wavefronts that started last run on while the CUs empty: 213 us at steady state, 277 us for one C2 launch); with
// short tiles dispatched last that window shrinks.
template <int D>
__global__ __launch_bounds__(RBLOCK, PLP_REDUCE_R_WAVES(D)) void reduce_r_mix_kernel(
    int nbig, long long B, int m_max, const double* __restrict__ Ag, const double* __restrict__ bg,
    const int* __restrict__ mrows, double abs_tol, int force_retry, unsigned long long* __restrict__ keep_out,
    int* __restrict__ flags_out, double* __restrict__ r_out, double* __restrict__ xc_out,
    int* __restrict__ nlp_out, unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ retry_word,
    unsigned long long epoch) {
    if ((int)blockIdx.x < nbig)
        reduce_r_tile<D, 4, 4>((long long)blockIdx.x * (RBLOCK / 4), B, m_max, Ag, bg, mrows, abs_tol, force_retry, keep_out,
                               flags_out, r_out, xc_out, nlp_out, ctr, retry_word, epoch);
    else
        reduce_r_tile<D, 8, 2>((long long)nbig * (RBLOCK / 4) + (long long)((int)blockIdx.x - nbig) * (RBLOCK / 8), B, m_max,
                               Ag, bg, mrows, abs_tol, force_retry, keep_out, flags_out, r_out, xc_out, nlp_out, ctr, retry_word, epoch);
}

template <int D, int GS, int R = RR>
static int launch_reduce_r_dg(long long B, int m_max, const double* A, const double* b, const int* mrows,
                             double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                             hipStream_t st) {
    const size_t smem = reduce_r_smem_bytes(GS, D, R);
    const long long NG = RBLOCK / GS;
    long long blocks = (B + NG - 1) / NG;
    if (blocks > 2147483647ll) return 2;  // grid.x limit (never reached for realistic batches)
    if (smem > 48 * 1024)  // 64 rows x d>=5: up to 82 KB of the CU's 160 KB
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(reduce_r_kernel<D, GS, R>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (blocks < 1) blocks = 1;
    const char* fr = getenv("PLP_REDUCE_RETRY_ALL");
    if constexpr ((R == 4 && D <= 8) || (R == 2 && D > 8)) {
        // small batches: one polytope per wavefront, LPs in parallel (PLP_REDUCE_SPLIT=0 / 1: never / always)
        const char* sp = getenv("PLP_REDUCE_SPLIT");
        if ((sp && sp[0] == '1') || (!(sp && sp[0] == '0') && B <= PLP_REDUCE_SPLIT_MAXB(D, GS))) {
            const size_t sm1 = (((size_t)GS * R * (D + 2) + 2 * D + 2) * 8 + 15) & ~(size_t)15;
            hipLaunchKernelGGL((reduce_split_kernel<D, GS, R>), dim3((unsigned)(B < 1 ? 1 : B)), dim3(RBLOCK), sm1, st, B, m_max,
                               A, b, mrows, abs_tol, (fr && fr[0] == '1') ? 1 : 0, keep, flags, r, xc, nlp, t_reduce_ctr, t_reduce_retry, t_reduce_epoch);
            return 0;
        }
    }
    if constexpr (GS == 4 && R == 4 && D <= 4) {
        // more tiles than the chip holds at once (4096 wavefront slots): the last 1/16 of the tiles (at most 1024) are
        // split into half-size ones.  Measured at C2 (6250 tiles): 0.2765 ms without, 0.2579-0.2609 ms with 2/64 .. 9/64
        // of the batch in half-size tiles (a flat optimum), 0.27-0.29 ms beyond 10/64.  PLP_REDUCE_MIX=k: k/64 (0: off).
        // A third class of quarter-size tiles (16 lanes x 1 row) behind the half-size ones was measured too: no gain
        // (0.2557-0.2602 ms for the last 2/256 .. 12/256 of the batch), not kept.
        const char* mx = getenv("PLP_REDUCE_MIX");
        // (round 4: 1/32 of the tiles instead of 1/16 -- the optimum is flat between 2/64 and 8/64: 0.1908 / 0.1919 ms)
        long long tail_tiles = blocks / 32 < 1024 ? blocks / 32 : 1024;
        if (mx) tail_tiles = blocks * atoi(mx) / 64;
        // medium batches (fewer full tiles than half the chip's wavefront slots): half-size tiles only -- twice the
        // wavefronts, each done in about half the time.  PLP_REDUCE_HALF=0 / 1: never / whenever blocks <= 4096 (A/B).
        const char* hf = getenv("PLP_REDUCE_HALF");
        if (!mx && blocks <= 4096 && ((hf && hf[0] == '1') || (!(hf && hf[0] == '0') && blocks <= 2048))) {
            const long long nsmall = (B + NG / 2 - 1) / (NG / 2);
            const size_t smem2 = reduce_r_smem_bytes(8, D, 2);
            hipLaunchKernelGGL((reduce_r_mix_kernel<D>), dim3((unsigned)nsmall), dim3(RBLOCK), smem > smem2 ? smem : smem2, st,
                               0, B, m_max, A, b, mrows, abs_tol, (fr && fr[0] == '1') ? 1 : 0, keep, flags, r, xc, nlp, t_reduce_ctr, t_reduce_retry, t_reduce_epoch);
            return 0;
        }
        if (tail_tiles > 0 && tail_tiles < blocks && blocks > 4096) {
            long long nbig = blocks - tail_tiles;
            const long long rest = B - nbig * NG;
            const long long nsmall = (rest + NG / 2 - 1) / (NG / 2);
            const size_t smem2 = reduce_r_smem_bytes(8, D, 2);
            hipLaunchKernelGGL((reduce_r_mix_kernel<D>), dim3((unsigned)(nbig + nsmall)), dim3(RBLOCK),
                               smem > smem2 ? smem : smem2, st, (int)nbig, B, m_max, A, b, mrows, abs_tol,
                               (fr && fr[0] == '1') ? 1 : 0, keep, flags, r, xc, nlp, t_reduce_ctr, t_reduce_retry, t_reduce_epoch);
            return 0;
        }
    }
    hipLaunchKernelGGL((reduce_r_kernel<D, GS, R>), dim3((unsigned)blocks), dim3(RBLOCK), smem, st, B, m_max, A, b, mrows,
                       abs_tol, (fr && fr[0] == '1') ? 1 : 0, keep, flags, r, xc, nlp, t_reduce_ctr, t_reduce_retry, t_reduce_epoch);
    return 0;
}

}  // namespace plp
