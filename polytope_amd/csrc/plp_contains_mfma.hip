// plp_contains_mfma.hip -- containment on the matrix cores (gfx950): Polytope.contains / Region.contains
// (polytope/polytope.py:206-218, :732-746)  all_i( a_i.x - b_i < tol )  as the dense contraction [A | -b] . [X; 1].
//
// v_mfma_f64_16x16x4_f64 multiplies a 16 x 4 tile of [A | -b | 0] (16 rows of one polytope, 4 of the d + 1 padded
// columns) with a 4 x 16 tile of [X; 1; 0] (16 points) into a 16 x 16 accumulator: KS = ceil((d + 1) / 4) instructions
// give a_i.x - b_i for 16 rows x 16 points, and what is left for the vector ALU is one subtraction and two compares
// per value (contains_kernel spends 6 FMAs + a subtraction + a compare per value on it and sits at its FMA ceiling).
// A wavefront keeps the coordinates of T = 8 tiles of 16 points in registers for the whole polytope loop, so one
// 512-byte load of a polytope's operand tile feeds 8 x KS matrix instructions.
//
// The booleans must be the reference's, whose value is the k-ordered sum a_i0 x_0 + ... rounded step by step, then
// "- b_i", then "< tol" -- not the matrix core's 8-term accumulation.  Both are within 2^-49 * S of the exact value
// (S = sum |a_ik x_k| + |b_i|), so they can only disagree when the matrix-core value lies within tau = 2^-47 * S_max of
// tol; such values (rare: none for points in general position, every point that sits exactly on a facet) are
// recomputed with the reference's operation order on the vector ALU (the code of contains_kernel) before the verdict
// is taken.  The verdicts are therefore bit-identical to contains_kernel's, not merely close (tests: g4 boundary
// points, PLP_CONTAINS_MFMA=0/1 over random batches).
#include <stdlib.h>

#include "plp_kernels.hpp"

namespace plp {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int MF_T = 8;  // point tiles (of 16 points) per wavefront

// Operand tiles of the polytopes: Aext[p][rt][s][64]: lane l of k-step s of row tile rt holds column 4 s + l / 16 of
// row 16 rt + l % 16 of [A_p | -b_p | 0]; rows past m_p: a = 0, -b = -inf (always inside).  coef[p] = (max_i ||a_i||_1,
// max_i |b_i|) for the error bound.
__global__ __launch_bounds__(256) void contains_pack_kernel(int P, int m_max, int d, int RT, int KS,
                                                            const double* __restrict__ A, const double* __restrict__ b,
                                                            const int* __restrict__ mrows, double* __restrict__ Aext,
                                                            double* __restrict__ coef) {
    const int p = blockIdx.x;
    if (p >= P) return;
    const int m = mrows ? mrows[p] : m_max;
    const double ninf = -__longlong_as_double(0x7ff0000000000000ll);
    for (int idx = threadIdx.x; idx < RT * KS * 64; idx += 256) {
        const int l = idx & 63, s = (idx >> 6) % KS, rt = (idx >> 6) / KS;
        const int row = 16 * rt + (l & 15), k = 4 * s + (l >> 4);
        double v = 0.0;
        if (row < m) v = k < d ? A[((size_t)p * m_max + row) * d + k] : (k == d ? -b[(size_t)p * m_max + row] : 0.0);
        else v = k == d ? ninf : 0.0;
        Aext[(size_t)p * RT * KS * 64 + idx] = v;
    }
    __shared__ double s1[256], s2[256];
    double l1 = 0.0, bm = 0.0;
    for (int row = threadIdx.x; row < m; row += 256) {
        double t = 0.0;
        for (int k = 0; k < d; ++k) t += fabs(A[((size_t)p * m_max + row) * d + k]);
        l1 = fmax(l1, t);
        bm = fmax(bm, fabs(b[(size_t)p * m_max + row]));
    }
    s1[threadIdx.x] = l1;
    s2[threadIdx.x] = bm;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            s1[threadIdx.x] = fmax(s1[threadIdx.x], s1[threadIdx.x + o]);
            s2[threadIdx.x] = fmax(s2[threadIdx.x], s2[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { coef[2 * p] = s1[0]; coef[2 * p + 1] = s2[0]; }
}

template <int KS>
__global__ __launch_bounds__(256) void contains_mfma_kernel(int P, int m_max, int d, int RT, const double* __restrict__ A,
                                                            const double* __restrict__ b, const int* __restrict__ mrows,
                                                            const double* __restrict__ Aext,
                                                            const double* __restrict__ coef, long long N,
                                                            const double* __restrict__ X, double tol, int mode,
                                                            unsigned char* __restrict__ out) {
    constexpr int T = MF_T;
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long q0 = wave * (16 * T);
    if (q0 >= N) return;
    const int n = lane & 15, kq = lane >> 4;
    // ---- my part of the point tiles: B[t][s] = [X; 1; 0][4 s + kq][q0 + 16 t + n]
    double Bx[T][KS];
    double xm = 0.0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const long long q = q0 + 16 * t + n;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 4 * s + kq;
            double v = 0.0;
            if (k < d) v = q < N ? X[(long long)k * N + q] : 0.0;
            else if (k == d) v = 1.0;
            Bx[t][s] = v;
            if (k < d) xm = fmax(xm, fabs(v));
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) xm = fmax(xm, __shfl_xor(xm, o, 64));
    const int pchunk = (P + (int)gridDim.y - 1) / (int)gridDim.y;
    const int p_lo = (int)blockIdx.y * pchunk;
    const int p_hi = (p_lo + pchunk < P) ? p_lo + pchunk : P;
    unsigned any16[T];
#pragma unroll
    for (int t = 0; t < T; ++t) any16[t] = 0u;
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    for (int p = p_lo; p < p_hi; ++p) {
        // |matrix-core value - reference value| <= 2^-49 S, S <= ||a_i||_1 max|x| + |b_i|: values within tau of tol are redone
        const double tau = 0x1p-47 * (coef[2 * p] * xm + coef[2 * p + 1] + fabs(tol));
        unsigned long long okm[T];
#pragma unroll
        for (int t = 0; t < T; ++t) okm[t] = ~0ull;
        for (int rt = 0; rt < RT; ++rt) {
            double a[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) a[s] = Aext[((size_t)(p * RT + rt) * KS + s) * 64 + lane];
            bool ok[T];
            bool near = false;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], Bx[t][s], acc, 0, 0, 0);
                bool in = true;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double tt = acc[i] - tol;
                    in = in & (tt < 0.0);
                    near = near | (fabs(tt) <= tau);
                }
                ok[t] = in;
            }
            if (__any(near)) {
                // the reference's operation order for the rows / points this lane holds (row 16 rt + 4 kq + i, point n of
                // every tile): the code of contains_kernel
                const int m = mrows ? mrows[p] : m_max;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const long long q = q0 + 16 * t + n;
                    bool in = true;
                    for (int i = 0; i < 4; ++i) {
                        const int row = 16 * rt + 4 * kq + i;
                        if (row < m && q < N) {
                            const double* ar = A + ((size_t)p * m_max + row) * d;
                            double s_ = ar[0] * X[q];
                            for (int k = 1; k < d; ++k) s_ = fma(ar[k], X[(long long)k * N + q], s_);
                            in = in & ((s_ - b[(size_t)p * m_max + row]) < tol);
                        }
                    }
                    ok[t] = in;
                }
            }
#pragma unroll
            for (int t = 0; t < T; ++t) okm[t] &= __ballot(ok[t]);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            // point n is inside iff the lanes n, n + 16, n + 32, n + 48 (its 16 rows, four per lane) all agree
            const unsigned long long mm = okm[t];
            const unsigned m16 = (unsigned)(mm & (mm >> 16) & (mm >> 32) & (mm >> 48)) & 0xffffu;
            if (mode == 1) {
                const long long q = q0 + 16 * t + n;
                if (kq == 0 && q < N) out[(size_t)p * N + q] = (m16 >> n) & 1u;
            } else {
                any16[t] |= m16;
            }
        }
    }
    (void)pinf;
    if (mode == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const long long q = q0 + 16 * t + n;
            if (kq == 0 && q < N && ((any16[t] >> n) & 1u)) out[q] = 1;  // same-value stores from the polytope chunks
        }
    }
}

size_t contains_mfma_scratch_bytes(int P, int m_max, int d) {
    const int RT = (m_max + 15) / 16, KS = (d + 1 + 3) / 4;
    return ((size_t)P * RT * KS * 64 + 2 * (size_t)P) * 8 + 256;
}

template <int KS>
static void launch_mfma_ks(int P, int m_max, int d, int RT, const double* A, const double* b, const int* mrows,
                           const double* Aext, const double* coef, long long N, const double* X, double tol, int mode,
                           unsigned char* out, hipStream_t st) {
    const long long waves = (N + 16 * MF_T - 1) / (16 * MF_T);
    long long blocks = (waves + 3) / 4;
    // >= 8 wavefronts per SIMD over the launch; at least 16 polytopes per chunk
    long long chunks = (8ll * 1024 + waves - 1) / waves;
    if (chunks > (P + 15) / 16) chunks = (P + 15) / 16;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL((contains_mfma_kernel<KS>), dim3((unsigned)blocks, (unsigned)chunks), dim3(256), 0, st, P, m_max, d,
                       RT, A, b, mrows, Aext, coef, N, X, tol, mode, out);
}

// returns 1 when it does not apply (caller uses contains_kernel)
int launch_contains_mfma(int P, int m_max, int d, const double* A, const double* b, const int* mrows, long long N,
                         const double* X, double abs_tol, int mode, unsigned char* out, void* scratch, hipStream_t st) {
    if (!scratch || P < 1 || m_max < 1 || d < 1 || d > MAX_D) return 1;
    const int RT = (m_max + 15) / 16, KS = (d + 1 + 3) / 4;
    double* Aext = static_cast<double*>(scratch);
    double* coef = Aext + (size_t)P * RT * KS * 64;
    hipLaunchKernelGGL(contains_pack_kernel, dim3((unsigned)P), dim3(256), 0, st, P, m_max, d, RT, KS, A, b, mrows, Aext, coef);
    if (mode == 0) (void)hipMemsetAsync(out, 0, (size_t)N, st);
    switch (KS) {
        case 1: launch_mfma_ks<1>(P, m_max, d, RT, A, b, mrows, Aext, coef, N, X, abs_tol, mode, out, st); break;
        case 2: launch_mfma_ks<2>(P, m_max, d, RT, A, b, mrows, Aext, coef, N, X, abs_tol, mode, out, st); break;
        case 3: launch_mfma_ks<3>(P, m_max, d, RT, A, b, mrows, Aext, coef, N, X, abs_tol, mode, out, st); break;
        case 4: launch_mfma_ks<4>(P, m_max, d, RT, A, b, mrows, Aext, coef, N, X, abs_tol, mode, out, st); break;
        case 5: launch_mfma_ks<5>(P, m_max, d, RT, A, b, mrows, Aext, coef, N, X, abs_tol, mode, out, st); break;
        default: return 1;
    }
    return 0;
}

}  // namespace plp
