// plp_points.hip -- point kernels (gfx950): containment and quickhull outside-set assignment.
//
//   contains_kernel<D,PPL> : Polytope.contains / Region.contains
//                            (polytope/polytope.py:206-218, :732-746)
//                            test = A.dot(X) - b[:,None] < abs_tol ; all(test, axis=0) ; OR over polytopes
//   assign_kernel<D>       : quickhull distance() / first-facet assignment / get_furthest
//                            (polytope/quickhull.py:117-121, :224-245, :311-336, :87-102)
//
// contains: one lane owns PPL points (coordinates in VGPRs, X is [d][N] like the reference's
// column vectors, so loads are fully coalesced); the polytope rows are wave-uniform and come
// through the scalar cache (s_load), each a_ik feeding PPL v_fma_f64.  FP64 FMA-bound:
// 2 m d flops per (point, polytope).
// assign: one lane owns one point ([N][d] rows as in quickhull); facets are staged in LDS;
// per-facet furthest point by LDS u64 max -> global u64 max, then an arg-min-index pass so
// that the FIRST maximum wins as in Facet.get_furthest.
#include <stdlib.h>

#include "plp_kernels.hpp"

namespace plp {

// (s - b_i) < tol  <=>  s < thr_i for EVERY double s: fl(s - b) is monotone in s, so the predicate holds on a
// down-set of the ordered doubles and fails at +inf (inf - b is +inf or NaN); thr_i is the smallest double at which it
// fails, found by bisection over the order-preserving integer image of the doubles (64 evaluations of the reference's
// own expression).  NaN in b_i or tol: never true, thr_i = NaN.  One subtraction less per (row, point) in the kernel.
__device__ __forceinline__ unsigned long long ord_key(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord_val(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}
__global__ __launch_bounds__(BLOCK) void contains_thr_kernel(long long n, const double* __restrict__ b, double tol,
                                                             double* __restrict__ thr) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double bi = b[i];
    if (isnan(bi) || isnan(tol)) { thr[i] = __longlong_as_double(0x7ff8000000000000ll); return; }
    // invariant: the predicate holds below lo (or lo is the first double), fails at hi
    unsigned long long lo = ord_key(-__longlong_as_double(0x7ff0000000000000ll));
    unsigned long long hi = ord_key(__longlong_as_double(0x7ff0000000000000ll));
    if (!((ord_val(lo) - bi) < tol)) { thr[i] = ord_val(lo); return; }  // fails everywhere
    while (hi - lo > 1ull) {  // holds at lo, fails at hi
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        if ((ord_val(mid) - bi) < tol) lo = mid; else hi = mid;
    }
    thr[i] = ord_val(hi);
}

template <int D, int PPL, bool THR>
__global__ __launch_bounds__(BLOCK) void contains_kernel(int P, int m_max, const double* __restrict__ A,
                                                         const double* __restrict__ b,
                                                         const int* __restrict__ mrows, long long N,
                                                         const double* __restrict__ X, double tol, int mode,
                                                         unsigned char* __restrict__ out) {
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long q0 = (long long)blockIdx.x * BLOCK + threadIdx.x; q0 < N; q0 += stride * PPL) {
        double x[PPL][D];
        bool inb[PPL];
#pragma unroll
        for (int t = 0; t < PPL; ++t) {
            const long long q = q0 + t * stride;
            inb[t] = q < N;
#pragma unroll
            for (int k = 0; k < D; ++k) x[t][k] = inb[t] ? X[(long long)k * N + q] : 0.0;
        }
        // Per-point verdicts are kept as wave-wide lane masks in SGPRs (a v_cmp writes one directly and
        // the running AND / OR are scalar instructions); as `bool`s they are carried through the row loop
        // in VGPRs and cost three extra VALU instructions per point and row.
        unsigned long long any_m[PPL];
#pragma unroll
        for (int t = 0; t < PPL; ++t) any_m[t] = 0ull;
        // blockIdx.y selects a contiguous chunk of the polytopes (more waves in flight than one
        // pass over the points alone would give)
        const int pchunk = (P + (int)gridDim.y - 1) / (int)gridDim.y;
        const int p_lo = (int)blockIdx.y * pchunk;
        const int p_hi = (p_lo + pchunk < P) ? p_lo + pchunk : P;
        const unsigned long long me = 1ull << (threadIdx.x & 63);
        for (int p = p_lo; p < p_hi; ++p) {
            const int m = mrows ? mrows[p] : m_max;
            const double* Ap = A + (size_t)p * m_max * D;
            const double* bp = b + (size_t)p * m_max;
            unsigned long long ok_m[PPL];
#pragma unroll
            for (int t = 0; t < PPL; ++t) ok_m[t] = ~0ull;
            // (requesting row i + 1 before row i's FMA chains -- wait, request, compute: scalar loads return out of order, so
            // the only wait is "all of them" -- was built in round 3 and measured: 34.7 ms against 33.8 at C3, the copies of
            // the staged row cost more than the wait the other four or five wavefronts of the SIMD already cover; not kept)
            for (int i = 0; i < m; ++i) {  // rows are wave-uniform: scalar loads, SGPR operands
                double ar[D];
#pragma unroll
                for (int k = 0; k < D; ++k) ar[k] = Ap[i * D + k];
                const double bi = bp[i];
#pragma unroll
                for (int t = 0; t < PPL; ++t) {
                    double s = ar[0] * x[t][0];
#pragma unroll
                    for (int k = 1; k < D; ++k) s = fma(ar[k], x[t][k], s);
                    // THR: `b` holds the thresholds of contains_thr_kernel
                    if constexpr (THR) ok_m[t] &= __ballot(s < bi);
                    else ok_m[t] &= __ballot((s - bi) < tol);
                }
            }
            if (mode == 1) {
#pragma unroll
                for (int t = 0; t < PPL; ++t) {
                    const long long q = q0 + t * stride;
                    if (inb[t]) out[(size_t)p * N + q] = (ok_m[t] & me) ? 1 : 0;
                }
            } else {
#pragma unroll
                for (int t = 0; t < PPL; ++t) any_m[t] |= ok_m[t];
            }
        }
        if (mode == 0) {
            // Region.contains = OR over polytopes: `out` was zeroed by the launcher and every chunk
            // that found the point inside stores the same value 1 (a benign same-value race)
#pragma unroll
            for (int t = 0; t < PPL; ++t) {
                const long long q = q0 + t * stride;
                if (inb[t] && (any_m[t] & me)) out[q] = 1;
            }
        }
    }
}

template <int D>
static void launch_contains_d(int P, int m_max, const double* A, const double* b, const int* mrows, long long N,
                              const double* X, double tol, int mode, unsigned char* out, double* thr, hipStream_t st) {
#ifdef PLP_CONTAINS_PPL
    constexpr int PPL = PLP_CONTAINS_PPL;  // (A/B builds)
#else
    // points per lane: every row fetched through the scalar cache feeds PPL FMA chains.  Measured (1M points x 2000
    // polytopes of 16 rows, ms per call, PPL = 2 / 4 / 6 / 8): d = 2 6.1 / 4.1 / 3.5 / 3.4, d = 3 6.9 / 4.9 / 4.6 / 4.3,
    // d = 4 6.9 / 5.6 / 5.3 / 5.3, d = 6 9.4 / 7.3 / 7.2 / 7.3, d = 8 11.2 / 9.1 / 8.8 / 9.4 (C3 itself: 35.7 ms with 4,
    // 33.9 with 6, 34.1 with 8)
    // (d > 8, 40-row polytopes, PPL = 2 / 3 / 4: d = 9 19.8 / 16.2 / 15.8, d = 10 22.3 / 18.7 / 19.9, d = 12 26.8 / 26.0 / 25.4,
    // d = 16 39.9 / 41.6 / 43.7)
    constexpr int PPL = (D <= 3) ? 8 : ((D <= 8) ? 6 : (D == 9 ? 4 : (D <= 12 ? 3 : 2)));
#endif
    long long blocks = (N + (long long)BLOCK * PPL - 1) / ((long long)BLOCK * PPL);
    if (blocks > 256ll * 32) blocks = 256ll * 32;
    if (blocks < 1) blocks = 1;
    // aim at >= 16 wavefronts per SIMD over the launch (256 CUs x 4 SIMDs) so that the tail is short
    // and the scalar-load latency of the row fetches hides behind other waves; >= 32 polytopes per chunk
    long long chunks = (16ll * 1024 + blocks * 4 - 1) / (blocks * 4);
    if (chunks > (P + 31) / 32) chunks = (P + 31) / 32;
    if (chunks < 1) chunks = 1;
    if (mode == 0) (void)hipMemsetAsync(out, 0, (size_t)N, st);
    if (thr && P > 0 && m_max > 0) {  // per-row thresholds first (P * m_max values: microseconds)
        const long long n = (long long)P * m_max;
        hipLaunchKernelGGL(contains_thr_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, n, b, tol,
                           thr);
        hipLaunchKernelGGL((contains_kernel<D, PPL, true>), dim3((unsigned)blocks, (unsigned)chunks), dim3(BLOCK), 0, st,
                           P, m_max, A, thr, mrows, N, X, tol, mode, out);
        return;
    }
    hipLaunchKernelGGL((contains_kernel<D, PPL, false>), dim3((unsigned)blocks, (unsigned)chunks), dim3(BLOCK), 0, st, P,
                       m_max, A, b, mrows, N, X, tol, mode, out);
}

#define PLP_CASE_C(K) case K: launch_contains_d<K>(P, m_max, A, b, mrows, N, X, abs_tol, mode, out, thr, st); break;

size_t contains_scratch_bytes(int P, int m_max) { return ((size_t)P * (size_t)m_max * 8 + 255) & ~(size_t)255; }

int launch_contains(int P, int m_max, int d, const double* A, const double* b, const int* mrows, long long N,
                    const double* X, double abs_tol, int mode, unsigned char* out, void* scratch, hipStream_t st) {
    if (d < 1 || d > MAX_D || m_max < 0 || P < 0 || N < 0) return 2;
    if (N == 0) return 0;
    // (No matrix-core form: the exact v_mfma_f64_16x16x4_f64 path of round 2 took 78 ms against 35 ms here and was
    // removed in round 3 -- scripts/microbench/mfma_valu_coissue.hip shows that the f64 matrix and vector pipes of gfx950
    // do not add up: wavefronts of both kinds on every SIMD reach 41-56 TFLOP/s, the vector kind alone 60.6)
    // thresholds live in the context's scratch buffer (P * m_max doubles; PLP_CONTAINS_THR=0: the subtraction stays in
    // the kernel, for A/B runs)
    const char* th = getenv("PLP_CONTAINS_THR");
    double* thr = (scratch && !(th && th[0] == '0')) ? reinterpret_cast<double*>(scratch) : nullptr;
    switch (d) {
        PLP_CASE_C(1) PLP_CASE_C(2) PLP_CASE_C(3) PLP_CASE_C(4) PLP_CASE_C(5) PLP_CASE_C(6)
        PLP_CASE_C(7) PLP_CASE_C(8) PLP_CASE_C(9) PLP_CASE_C(10) PLP_CASE_C(11) PLP_CASE_C(12)
        PLP_CASE_C(13) PLP_CASE_C(14) PLP_CASE_C(15) PLP_CASE_C(16)
        default: return 2;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
constexpr int FCHUNK = 256;  // facets staged in LDS at a time

constexpr int SMAX_CAP = 8192;  // per-facet maxima kept in LDS for the first SMAX_CAP facets

template <int D>
__global__ __launch_bounds__(BLOCK) void assign_kernel(long long N, const double* __restrict__ X, int F,
                                                       const double* __restrict__ normals,
                                                       const double* __restrict__ offsets, double tol,
                                                       int* __restrict__ fop_out, double* __restrict__ dist_out,
                                                       unsigned long long* __restrict__ maxbits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* sn = reinterpret_cast<double*>(smem_raw);                    // [FCHUNK][D]
    double* so = sn + FCHUNK * D;                                        // [FCHUNK]
    unsigned long long* smax = reinterpret_cast<unsigned long long*>(so + FCHUNK);  // [min(F, SMAX_CAP)]
    const int FS = F < SMAX_CAP ? F : SMAX_CAP;
    const bool single = F <= FCHUNK;  // all facets fit one staging: stage them once per block
    for (int idx = threadIdx.x; idx < FS; idx += BLOCK) smax[idx] = 0ull;
    if (single) {
        for (int idx = threadIdx.x; idx < F * D; idx += BLOCK) sn[idx] = normals[idx];
        for (int idx = threadIdx.x; idx < F; idx += BLOCK) so[idx] = offsets[idx];
    }
    __syncthreads();
    const long long stride = (long long)gridDim.x * BLOCK;
    const long long nloop = (N + stride - 1) / stride;
    for (long long it = 0; it < nloop; ++it) {
        const long long q = it * stride + (long long)blockIdx.x * BLOCK + threadIdx.x;
        const bool inb = q < N;
        double x[D];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = inb ? X[q * D + k] : 0.0;
        int fop = -1;
        double dd = 0.0;
        for (int f0 = 0; f0 < F; f0 += FCHUNK) {
            const int fc = (F - f0) < FCHUNK ? (F - f0) : FCHUNK;
            if (!single) {
                __syncthreads();
                for (int idx = threadIdx.x; idx < fc * D; idx += BLOCK) sn[idx] = normals[(size_t)f0 * D + idx];
                for (int idx = threadIdx.x; idx < fc; idx += BLOCK) so[idx] = offsets[f0 + idx];
                __syncthreads();
            }
            if (!__all(fop >= 0 || !inb)) {
                for (int f = 0; f < fc; ++f) {
                    const double dist = np_dot<D>(sn + f * D, x) - so[f];  // sum(n*p) - d  (quickhull.py:121), numpy's order
                    if (inb && fop < 0 && dist > tol) { fop = f0 + f; dd = dist; }
                }
            }
        }
        // furthest point per facet: LDS max over everything this block sees, flushed once at the end
        if (fop >= 0) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(dd);
            if (fop < FS) atomicMax(&smax[fop], bits); else atomicMax(&maxbits[fop], bits);
        }
        if (inb) { fop_out[q] = fop; dist_out[q] = dd; }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < FS; idx += BLOCK)
        if (smax[idx] != 0ull) atomicMax(&maxbits[idx], smax[idx]);  // one global atomic per (block, facet)
}

__global__ __launch_bounds__(BLOCK) void argmax_kernel(long long N, const int* __restrict__ fop,
                                                       const double* __restrict__ dist,
                                                       const unsigned long long* __restrict__ maxbits,
                                                       unsigned long long* __restrict__ argmax) {
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long q = (long long)blockIdx.x * BLOCK + threadIdx.x; q < N; q += stride) {
        const int f = fop[q];
        if (f >= 0 && (unsigned long long)__double_as_longlong(dist[q]) == maxbits[f])
            atomicMin(&argmax[f], (unsigned long long)q);  // first maximum wins (quickhull.py:97-100, strict '<')
    }
}

// maxd = 0.0 and argmax = -1 (all ones: the identity of the unsigned atomicMin) for "no point"
__global__ void assign_init_kernel(int F, unsigned long long* maxbits, unsigned long long* argmax) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F) { maxbits[f] = 0ull; argmax[f] = ~0ull; }
}

// ---- few facets (F <= ASSIGN_SMALL_F): HBM-bound.  The kernel above ends every workgroup with one global u64 max per
// facet -- a thousand workgroups on nine addresses, which the L2 atomic units take one at a time -- and needs a second
// pass over all points for "first maximum wins".  Here a workgroup settles both in LDS (max of the distance bits, then
// the lowest point index among its points that attain it) and writes ONE (max, index) pair per facet to a partials table;
// a tiny second kernel folds the table.  No global atomics, no initialisation launch, no arg-max pass over N.
// Measured at C5 (1 M points, d = 8, F = 9; device time per call, 50 calls back to back): general kernel 31.9 us ->
// 21.5 us with two points per lane (one: 23.3, four: 21.7).  Also built and measured, not kept: every lane fetching one
// quarter of four consecutive points (a quad reads 64 contiguous bytes per load instead of four 16-byte pieces 64 bytes
// apart) with a 4 x 4 transpose over the quad by DPP quad_perm moves -- 24.0 us: as in round 3 (LDS transpose), the
// strided row reads are not what holds the kernel back.
constexpr int ASSIGN_SMALL_F = 64;

// FB: a tag only (16: F <= 16, 64: beyond) -- the two regimes of this kernel (HBM-bound / VALU-bound) get their own rows
// in a kernel trace.
// Round 5, measured and NOT kept: the fold inside this kernel (every workgroup publishes its pairs, draws a ticket, the
// last one folds the table -- one launch per call).  The release that must precede the ticket is a device-scope fence, and
// on this part (eight XCDs, an L2 each) that is an L2 write-back per workgroup with the kernel's 12 MB of fresh fop / dist
// lines behind it: 233 us per call against 23.4 us for the two launches.
template <int D, int PPT, int FB>
__global__ __launch_bounds__(BLOCK) void assign_small_kernel(long long N, const double* __restrict__ X, int F,
                                                             const double* __restrict__ normals,
                                                             const double* __restrict__ offsets, double tol,
                                                             int* __restrict__ fop_out, double* __restrict__ dist_out,
                                                             unsigned long long* __restrict__ part, long long nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* sn = reinterpret_cast<double*>(smem_raw);                    // [F][D]
    double* so = sn + (size_t)F * D;                                     // [F]
    unsigned long long* smax = reinterpret_cast<unsigned long long*>(so + F);  // [F]
    unsigned long long* sarg = smax + F;                                 // [F]
    for (int idx = threadIdx.x; idx < F * D; idx += BLOCK) sn[idx] = normals[idx];
    for (int idx = threadIdx.x; idx < F; idx += BLOCK) { so[idx] = offsets[idx]; smax[idx] = 0ull; sarg[idx] = ~0ull; }
    const long long base = (long long)blockIdx.x * (BLOCK * PPT);
    double x[PPT][D];
    long long q[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        q[p] = base + (long long)p * BLOCK + threadIdx.x;
#pragma unroll
        for (int k = 0; k < D; ++k) x[p][k] = q[p] < N ? X[q[p] * D + k] : 0.0;
    }
    __syncthreads();
    int fop[PPT];
    double dd[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) { fop[p] = -1; dd[p] = 0.0; }
    for (int f = 0; f < F; ++f) {
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            const double dist = np_dot<D>(sn + f * D, x[p]) - so[f];   // sum(n*p) - d  (quickhull.py:121), numpy's order
            const bool take = (q[p] < N) & (fop[p] < 0) & (dist > tol);  // the FIRST facet in list order (:224-245)
            fop[p] = take ? f : fop[p];
            dd[p] = take ? dist : dd[p];
        }
    }
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        if (q[p] < N) { fop_out[q[p]] = fop[p]; dist_out[q[p]] = dd[p]; }
        if (fop[p] >= 0) atomicMax(&smax[fop[p]], (unsigned long long)__double_as_longlong(dd[p]));  // dd > tol >= 0
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PPT; ++p)   // get_furthest keeps the first maximum (strict '<', :97-100): the lowest index
        if (fop[p] >= 0 && (unsigned long long)__double_as_longlong(dd[p]) == smax[fop[p]])
            atomicMin(&sarg[fop[p]], (unsigned long long)q[p]);
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += BLOCK) {
        part[((size_t)f * nblk + blockIdx.x) * 2] = smax[f];
        part[((size_t)f * nblk + blockIdx.x) * 2 + 1] = sarg[f];
    }
}

// one workgroup per facet folds the workgroups' (max bits, lowest index) pairs: larger distance wins, then lower index
__global__ __launch_bounds__(BLOCK) void assign_finish_kernel(long long nblk, const unsigned long long* __restrict__ part,
                                                              unsigned long long* __restrict__ maxbits,
                                                              unsigned long long* __restrict__ argmax) {
    __shared__ unsigned long long sbits[BLOCK / 64], sidx[BLOCK / 64];
    const int f = blockIdx.x;
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(part) + (size_t)f * nblk;
    unsigned long long bits = 0ull, idx = ~0ull;
    for (long long b = threadIdx.x; b < nblk; b += BLOCK) {
        const ulonglong2 v = src[b];
        const bool better = (v.x > bits) | ((v.x == bits) & (v.y < idx));
        bits = better ? v.x : bits;
        idx = better ? v.y : idx;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long v = __shfl_xor(bits, o, 64), i = __shfl_xor(idx, o, 64);
        const bool better = (v > bits) | ((v == bits) & (i < idx));
        bits = better ? v : bits;
        idx = better ? i : idx;
    }
    if ((threadIdx.x & 63) == 0) { sbits[threadIdx.x >> 6] = bits; sidx[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < BLOCK / 64; ++w) {
            const bool better = (sbits[w] > bits) | ((sbits[w] == bits) & (sidx[w] < idx));
            bits = better ? sbits[w] : bits;
            idx = better ? sidx[w] : idx;
        }
        maxbits[f] = bits;               // 0.0 and -1 for "no point"
        argmax[f] = bits ? idx : ~0ull;
    }
}

size_t assign_scratch_bytes(long long N, int F) {
    if (F > ASSIGN_SMALL_F || N < 1) return 0;
    const long long nblk = (N + BLOCK - 1) / BLOCK;   // (the PPT = 1 grid: the largest)
    return (size_t)F * nblk * 16;
}

template <int D, int PPT>
static void launch_assign_small(long long N, const double* X, int F, const double* normals, const double* offsets,
                                double tol, int* fop, double* dist, long long* argmax, double* maxd, void* scratch,
                                hipStream_t st) {
    const long long nblk = (N + (long long)BLOCK * PPT - 1) / ((long long)BLOCK * PPT);
    const size_t smem = ((size_t)F * (D + 1) + 2 * (size_t)F) * 8;
    unsigned long long* part = static_cast<unsigned long long*>(scratch);
    if (F <= 16)
        hipLaunchKernelGGL((assign_small_kernel<D, PPT, 16>), dim3((unsigned)nblk), dim3(BLOCK), smem, st, N, X, F, normals,
                           offsets, tol, fop, dist, part, nblk);
    else
        hipLaunchKernelGGL((assign_small_kernel<D, PPT, 64>), dim3((unsigned)nblk), dim3(BLOCK), smem, st, N, X, F, normals,
                           offsets, tol, fop, dist, part, nblk);
    hipLaunchKernelGGL(assign_finish_kernel, dim3((unsigned)F), dim3(BLOCK), 0, st, nblk, part,
                       reinterpret_cast<unsigned long long*>(maxd), reinterpret_cast<unsigned long long*>(argmax));
}

template <int D>
static void launch_assign_d(long long N, const double* X, int F, const double* normals, const double* offsets,
                            double tol, int* fop, double* dist, long long* argmax, double* maxd, void* scratch,
                            size_t scratch_bytes, hipStream_t st) {
    // PLP_ASSIGN_SMALL=0: the general kernel for every F (A/B, tests); PLP_ASSIGN_PPT=1|2|4: points per lane
    const char* sm = getenv("PLP_ASSIGN_SMALL");
    if (!(sm && sm[0] == '0') && F <= ASSIGN_SMALL_F && N >= 1 && scratch && scratch_bytes >= assign_scratch_bytes(N, F)) {
        const char* pp = getenv("PLP_ASSIGN_PPT");
        const int ppt = pp ? atoi(pp) : (N >= 262144 ? 2 : 1);
        if constexpr (D <= 8) {
            if (ppt >= 4) {
                launch_assign_small<D, 4>(N, X, F, normals, offsets, tol, fop, dist, argmax, maxd, scratch, st);
                return;
            }
        }
        if (ppt >= 2) launch_assign_small<D, 2>(N, X, F, normals, offsets, tol, fop, dist, argmax, maxd, scratch, st);
        else launch_assign_small<D, 1>(N, X, F, normals, offsets, tol, fop, dist, argmax, maxd, scratch, st);
        return;
    }
    long long blocks = (N + BLOCK - 1) / BLOCK;
    // few blocks when there are few facets (HBM/atomic bound: one global atomic per (block, facet));
    // more when the per-point facet scan dominates (VALU bound from F ~ 32 on)
    long long cap = 256ll * 4 * (F <= 16 ? 1 : (F <= 128 ? F / 16 : 8));
    // (requesting the next pass's point before this pass's facet scan was measured: 29.8 us against 27.8 us per launch
    // at C5 / F = 9 under rocprofv3 -- no gain, not kept; 512 .. 8192 workgroups instead of this cap: equal or slower;
    // round 3: the pass's 256 points fetched as one contiguous block, 16 bytes per lane, and transposed through LDS
    // instead of every lane reading its own row: 39.2 us against 36.1 us per call -- the strided row reads are not
    // what holds this kernel back, not kept)
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    unsigned long long* mb = reinterpret_cast<unsigned long long*>(maxd);
    unsigned long long* am = reinterpret_cast<unsigned long long*>(argmax);
    const size_t smem = ((size_t)FCHUNK * (D + 1) + (size_t)(F < SMAX_CAP ? F : SMAX_CAP)) * 8;
    hipLaunchKernelGGL(assign_init_kernel, dim3((F + 255) / 256), dim3(256), 0, st, F, mb, am);
    hipLaunchKernelGGL(assign_kernel<D>, dim3((unsigned)blocks), dim3(BLOCK), smem, st, N, X, F, normals, offsets, tol,
                       fop, dist, mb);
    long long blocks2 = (N + BLOCK - 1) / BLOCK;
    if (blocks2 > 256ll * 8) blocks2 = 256ll * 8;
    if (blocks2 < 1) blocks2 = 1;
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)blocks2), dim3(BLOCK), 0, st, N, fop, dist, mb, am);
}

#define PLP_CASE_A(K) case K: launch_assign_d<K>(N, X, F, normals, offsets, abs_tol, fop, dist, argmax, maxd, scratch, scratch_bytes, st); break;

int launch_assign(long long N, int d, const double* X, int F, const double* normals, const double* offsets,
                  double abs_tol, int* fop, double* dist, long long* argmax, double* maxd, void* scratch,
                  size_t scratch_bytes, hipStream_t st) {
    if (d < 1 || d > MAX_D || F < 1 || N < 0 || !(abs_tol >= 0.0)) return 2;
    switch (d) {
        PLP_CASE_A(1) PLP_CASE_A(2) PLP_CASE_A(3) PLP_CASE_A(4) PLP_CASE_A(5) PLP_CASE_A(6)
        PLP_CASE_A(7) PLP_CASE_A(8) PLP_CASE_A(9) PLP_CASE_A(10) PLP_CASE_A(11) PLP_CASE_A(12)
        PLP_CASE_A(13) PLP_CASE_A(14) PLP_CASE_A(15) PLP_CASE_A(16)
        default: return 2;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
}  // namespace plp

#include "plp_wave.hpp"
namespace plp {

// primitive self-test: lane l contributes d = ((l*37)%64 - 20) * 0.5 and u = (l*29)%61;
// out_d[l] / out_u[l] = group minima, out_u[64+l] = group ballot(l%3==0) low bits,
// out_d[64+l] = value broadcast from lane gbase + (l*7)%gs
__global__ void selftest_kernel(int gs, double* out_d, unsigned* out_u) {
    const Grp g(gs);
    const int l = g.lane;
    const double d = (double)((l * 37) % 64 - 20) * 0.5;
    const unsigned u = (unsigned)((l * 29) % 61);
    out_d[l] = grp_min(d, gs);
    out_u[l] = grp_min(u, gs);
    out_u[64 + l] = (unsigned)(grp_ballot(l % 3 == 0, g) & 0xffffffffull);
    out_d[64 + l] = bcast(d, g.gbase + (l * 7) % gs);
}

int launch_selftest(int gs, double* out_d, unsigned* out_u, hipStream_t st) {
    if (gs != 8 && gs != 16 && gs != 32 && gs != 64) return 2;
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, gs, out_d, out_u);
    return 0;
}

}  // namespace plp
