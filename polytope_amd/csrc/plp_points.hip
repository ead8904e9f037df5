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
#include "plp_kernels.hpp"

namespace plp {

template <int D, int PPL>
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
        bool any_in[PPL];
#pragma unroll
        for (int t = 0; t < PPL; ++t) any_in[t] = false;
        for (int p = 0; p < P; ++p) {
            const int m = mrows ? mrows[p] : m_max;
            const double* Ap = A + (size_t)p * m_max * D;
            const double* bp = b + (size_t)p * m_max;
            bool ok[PPL];
#pragma unroll
            for (int t = 0; t < PPL; ++t) ok[t] = true;
            for (int i = 0; i < m; ++i) {
                double ar[D];
#pragma unroll
                for (int k = 0; k < D; ++k) ar[k] = Ap[i * D + k];
                const double bi = bp[i];
#pragma unroll
                for (int t = 0; t < PPL; ++t) {
                    double s = ar[0] * x[t][0];
#pragma unroll
                    for (int k = 1; k < D; ++k) s = fma(ar[k], x[t][k], s);
                    ok[t] = ok[t] && ((s - bi) < tol);
                }
            }
            if (mode == 1) {
#pragma unroll
                for (int t = 0; t < PPL; ++t) {
                    const long long q = q0 + t * stride;
                    if (inb[t]) out[(size_t)p * N + q] = ok[t] ? 1 : 0;
                }
            } else {
#pragma unroll
                for (int t = 0; t < PPL; ++t) any_in[t] = any_in[t] || ok[t];
            }
        }
        if (mode == 0) {
#pragma unroll
            for (int t = 0; t < PPL; ++t) {
                const long long q = q0 + t * stride;
                if (inb[t]) out[q] = any_in[t] ? 1 : 0;
            }
        }
    }
}

template <int D>
static void launch_contains_d(int P, int m_max, const double* A, const double* b, const int* mrows, long long N,
                              const double* X, double tol, int mode, unsigned char* out, hipStream_t st) {
    constexpr int PPL = (D <= 8) ? 4 : 2;
    long long blocks = (N + (long long)BLOCK * PPL - 1) / ((long long)BLOCK * PPL);
    if (blocks > 256ll * 32) blocks = 256ll * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((contains_kernel<D, PPL>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, P, m_max, A, b, mrows,
                       N, X, tol, mode, out);
}

#define PLP_CASE_C(K) case K: launch_contains_d<K>(P, m_max, A, b, mrows, N, X, abs_tol, mode, out, st); break;

int launch_contains(int P, int m_max, int d, const double* A, const double* b, const int* mrows, long long N,
                    const double* X, double abs_tol, int mode, unsigned char* out, hipStream_t st) {
    if (d < 1 || d > MAX_D || m_max < 0 || P < 0 || N < 0) return 2;
    if (N == 0) return 0;
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

template <int D>
__global__ __launch_bounds__(BLOCK) void assign_kernel(long long N, const double* __restrict__ X, int F,
                                                       const double* __restrict__ normals,
                                                       const double* __restrict__ offsets, double tol,
                                                       int* __restrict__ fop_out, double* __restrict__ dist_out,
                                                       unsigned long long* __restrict__ maxbits) {
    __shared__ double sn[FCHUNK * D];
    __shared__ double so[FCHUNK];
    __shared__ unsigned long long smax[FCHUNK];
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
            __syncthreads();
            for (int idx = threadIdx.x; idx < fc * D; idx += BLOCK) sn[idx] = normals[(size_t)f0 * D + idx];
            for (int idx = threadIdx.x; idx < fc; idx += BLOCK) { so[idx] = offsets[f0 + idx]; smax[idx] = 0ull; }
            __syncthreads();
            if (!__all(fop >= 0 || !inb)) {
                for (int f = 0; f < fc; ++f) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) s = s + sn[f * D + k] * x[k];  // sum(n*p)  (quickhull.py:121)
                    const double dist = s - so[f];
                    if (inb && fop < 0 && dist > tol) { fop = f0 + f; dd = dist; }
                }
            }
            // furthest point of each facet of this chunk: block max, then one global atomic per facet
            if (fop >= f0 && fop < f0 + fc) atomicMax(&smax[fop - f0], (unsigned long long)__double_as_longlong(dd));
            __syncthreads();
            for (int idx = threadIdx.x; idx < fc; idx += BLOCK)
                if (smax[idx] != 0ull) atomicMax(&maxbits[f0 + idx], smax[idx]);
        }
        if (inb) { fop_out[q] = fop; dist_out[q] = dd; }
    }
}

__global__ __launch_bounds__(BLOCK) void argmax_kernel(long long N, const int* __restrict__ fop,
                                                       const double* __restrict__ dist,
                                                       const unsigned long long* __restrict__ maxbits,
                                                       long long* __restrict__ argmax) {
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long q = (long long)blockIdx.x * BLOCK + threadIdx.x; q < N; q += stride) {
        const int f = fop[q];
        if (f >= 0 && (unsigned long long)__double_as_longlong(dist[q]) == maxbits[f])
            atomicMin(&argmax[f], q);  // first maximum wins (quickhull.py:97-100, strict '<')
    }
}

__global__ void assign_init_kernel(int F, unsigned long long* maxbits, long long* argmax) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F) { maxbits[f] = 0ull; argmax[f] = 0x7fffffffffffffffll; }
}

__global__ void assign_fini_kernel(int F, double* maxd, long long* argmax) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F) {
        if (argmax[f] == 0x7fffffffffffffffll) { argmax[f] = -1; maxd[f] = -__longlong_as_double(0x7ff0000000000000ll); }
    }
}

size_t assign_scratch_bytes(long long, int) { return 0; }

template <int D>
static void launch_assign_d(long long N, const double* X, int F, const double* normals, const double* offsets,
                            double tol, int* fop, double* dist, long long* argmax, double* maxd, hipStream_t st) {
    long long blocks = (N + BLOCK - 1) / BLOCK;
    if (blocks > 256ll * 16) blocks = 256ll * 16;
    if (blocks < 1) blocks = 1;
    unsigned long long* mb = reinterpret_cast<unsigned long long*>(maxd);
    hipLaunchKernelGGL(assign_init_kernel, dim3((F + 255) / 256), dim3(256), 0, st, F, mb, argmax);
    hipLaunchKernelGGL(assign_kernel<D>, dim3((unsigned)blocks), dim3(BLOCK), 0, st, N, X, F, normals, offsets, tol,
                       fop, dist, mb);
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, st, N, fop, dist, mb, argmax);
    hipLaunchKernelGGL(assign_fini_kernel, dim3((F + 255) / 256), dim3(256), 0, st, F, maxd, argmax);
}

#define PLP_CASE_A(K) case K: launch_assign_d<K>(N, X, F, normals, offsets, abs_tol, fop, dist, argmax, maxd, st); break;

int launch_assign(long long N, int d, const double* X, int F, const double* normals, const double* offsets,
                  double abs_tol, int* fop, double* dist, long long* argmax, double* maxd, void*, size_t,
                  hipStream_t st) {
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
