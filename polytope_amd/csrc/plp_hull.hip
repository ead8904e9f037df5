// plp_hull.hip -- quickhull outside-set maintenance on resident points (gfx950).
//
// Reference behaviour restated (polytope/quickhull.py:248-345, one iteration of the main loop):
//   * the outside points of every visible facet are pooled (:273-283),
//   * each pooled point goes to the FIRST new facet, in creation order, whose signed distance
//     n.p - d0 exceeds abs_tol (:311-336, distance() at :117-121); points beyond no new facet are
//     dropped for good,
//   * a facet with outside points later yields its furthest one, first maximum winning (:87-102).
// The reference rebuilds Python lists of point objects for this; here the N points never move:
// they stay in HBM as [N][d] rows with one int32 owner (facet id, -1 = inside the hull) and one
// distance each, and an iteration is one pass over the owners:
//   hull_reassign_kernel<D>: lane q reads owner[q] (4 B); only if that facet is flagged dead does it
//     load its point, scan the new facets (staged in LDS) and store the new owner/distance; per new
//     facet the block keeps max-distance bits and a population count in LDS and flushes them with
//     one global atomic per (block, facet).
//   hull_argmax_kernel: the lowest point index attaining each new facet's maximum.
// HBM traffic per iteration: 4 N bytes (owners) + 8 (d+1) bytes per pooled point + 12 per moved
// point; the facet table (n_new * 8 (d+1) bytes) is read once per block.  At N = 1e6 an iteration is
// ~4 MB, i.e. launch-latency bound: the host keeps the facet graph (hundreds of facets) and this
// kernel keeps the points.
// Ties: the reference's "first maximum" refers to the order of its pooled list; here it is the
// lowest point index.  The two differ only for exactly equal distances.
#include "plp_kernels.hpp"

namespace plp {

constexpr int HF_CHUNK = 256;   // new facets staged in LDS at a time
constexpr int HS_CAP = 2048;    // per-facet maxima / counts kept in LDS for the first HS_CAP new facets

template <int D>
__global__ __launch_bounds__(BLOCK) void hull_reassign_kernel(
    long long N, const double* __restrict__ X, int* __restrict__ owner, double* __restrict__ dist,
    const unsigned char* __restrict__ dead, int new_id0, int n_new, const double* __restrict__ normals,
    const double* __restrict__ offsets, double tol, unsigned long long* __restrict__ maxbits,
    unsigned long long* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* sn = reinterpret_cast<double*>(smem_raw);  // [HF_CHUNK][D]
    double* so = sn + HF_CHUNK * D;                    // [HF_CHUNK]
    unsigned long long* smax = reinterpret_cast<unsigned long long*>(so + HF_CHUNK);  // [FS]
    const int FS = n_new < HS_CAP ? n_new : HS_CAP;
    unsigned* scnt = reinterpret_cast<unsigned*>(smax + FS);  // [FS]
    const bool single = n_new <= HF_CHUNK;
    for (int idx = threadIdx.x; idx < FS; idx += BLOCK) { smax[idx] = 0ull; scnt[idx] = 0u; }
    if (single) {
        for (int idx = threadIdx.x; idx < n_new * D; idx += BLOCK) sn[idx] = normals[idx];
        for (int idx = threadIdx.x; idx < n_new; idx += BLOCK) so[idx] = offsets[idx];
    }
    __syncthreads();
    const long long stride = (long long)gridDim.x * BLOCK;
    const long long nloop = (N + stride - 1) / stride;
    for (long long it = 0; it < nloop; ++it) {
        const long long q = it * stride + (long long)blockIdx.x * BLOCK + threadIdx.x;
        const int own = q < N ? owner[q] : -1;
        const bool pooled = own >= 0 && own < new_id0 && dead[own] != 0;
        if (single && !__any(pooled)) continue;  // wave-uniform: nothing of this wave is pooled
        double x[D];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = pooled ? X[q * D + k] : 0.0;
        int slot = -1;
        double dd = 0.0;
        for (int f0 = 0; f0 < n_new; f0 += HF_CHUNK) {
            const int fc = (n_new - f0) < HF_CHUNK ? (n_new - f0) : HF_CHUNK;
            if (!single) {
                __syncthreads();
                for (int idx = threadIdx.x; idx < fc * D; idx += BLOCK) sn[idx] = normals[(size_t)f0 * D + idx];
                for (int idx = threadIdx.x; idx < fc; idx += BLOCK) so[idx] = offsets[f0 + idx];
                __syncthreads();
            }
            if (__any(pooled && slot < 0)) {
                for (int f = 0; f < fc; ++f) {
                    const double dv = np_dot<D>(sn + f * D, x) - so[f];  // sum(n*p) - d  (quickhull.py:121), numpy's order
                    if (pooled && slot < 0 && dv > tol) { slot = f0 + f; dd = dv; }
                }
            }
        }
        if (pooled) {
            owner[q] = slot >= 0 ? new_id0 + slot : -1;
            dist[q] = dd;
            if (slot >= 0) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(dd);  // dd > tol >= 0
                if (slot < FS) { atomicMax(&smax[slot], bits); atomicAdd(&scnt[slot], 1u); }
                else { atomicMax(&maxbits[slot], bits); atomicAdd(&count[slot], 1ull); }
            }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < FS; idx += BLOCK)
        if (scnt[idx] != 0u) { atomicMax(&maxbits[idx], smax[idx]); atomicAdd(&count[idx], (unsigned long long)scnt[idx]); }
}

__global__ __launch_bounds__(BLOCK) void hull_argmax_kernel(long long N, const int* __restrict__ owner,
                                                            const double* __restrict__ dist, int new_id0, int n_new,
                                                            const unsigned long long* __restrict__ maxbits,
                                                            unsigned long long* __restrict__ argmax) {
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long q = (long long)blockIdx.x * BLOCK + threadIdx.x; q < N; q += stride) {
        const int slot = owner[q] - new_id0;
        if (slot >= 0 && slot < n_new && (unsigned long long)__double_as_longlong(dist[q]) == maxbits[slot])
            atomicMin(&argmax[slot], (unsigned long long)q);
    }
}

// maxd = 0.0, count = 0, argmax = -1 (all ones: the identity of the unsigned atomicMin)
__global__ void hull_init_kernel(int n_new, unsigned long long* maxbits, unsigned long long* count,
                                 unsigned long long* argmax) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_new) { maxbits[f] = 0ull; count[f] = 0ull; argmax[f] = ~0ull; }
}

__global__ void hull_mark_kernel(int n, const int* __restrict__ ids, unsigned char* __restrict__ dead) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dead[ids[i]] = 1;
}

__global__ void hull_drop_kernel(long long n, const long long* __restrict__ idx, int* __restrict__ owner) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) owner[idx[i]] = -1;
}

template <int D>
static void launch_hull_d(long long N, const double* X, int* owner, double* dist, const unsigned char* dead,
                          int new_id0, int n_new, const double* normals, const double* offsets, double tol,
                          long long* argmax, double* maxd, long long* count, hipStream_t st) {
    unsigned long long* mb = reinterpret_cast<unsigned long long*>(maxd);
    unsigned long long* cn = reinterpret_cast<unsigned long long*>(count);
    unsigned long long* am = reinterpret_cast<unsigned long long*>(argmax);
    hipLaunchKernelGGL(hull_init_kernel, dim3((n_new + 255) / 256), dim3(256), 0, st, n_new, mb, cn, am);
    if (N == 0) return;
    long long blocks = (N + BLOCK - 1) / BLOCK;
    if (blocks > 256ll * 4) blocks = 256ll * 4;  // more workgroups only add contention on the per-facet global atomics
    const int FS = n_new < HS_CAP ? n_new : HS_CAP;
    const size_t smem = (size_t)HF_CHUNK * (D + 1) * 8 + (size_t)FS * 12 + 16;
    hipLaunchKernelGGL(hull_reassign_kernel<D>, dim3((unsigned)blocks), dim3(BLOCK), smem, st, N, X, owner, dist, dead,
                       new_id0, n_new, normals, offsets, tol, mb, cn);
    hipLaunchKernelGGL(hull_argmax_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, st, N, owner, dist, new_id0, n_new,
                       mb, am);
}

#define PLP_CASE_H(K) \
    case K: launch_hull_d<K>(N, X, owner, dist, dead, new_id0, n_new, normals, offsets, abs_tol, argmax, maxd, count, st); break;

int launch_hull_reassign(long long N, int d, const double* X, int* owner, double* dist, const unsigned char* dead,
                         int new_id0, int n_new, const double* normals, const double* offsets, double abs_tol,
                         long long* argmax, double* maxd, long long* count, hipStream_t st) {
    if (d < 1 || d > MAX_D || n_new < 1 || N < 0 || new_id0 < 0 || !(abs_tol >= 0.0)) return 2;
    switch (d) {
        PLP_CASE_H(1) PLP_CASE_H(2) PLP_CASE_H(3) PLP_CASE_H(4) PLP_CASE_H(5) PLP_CASE_H(6)
        PLP_CASE_H(7) PLP_CASE_H(8) PLP_CASE_H(9) PLP_CASE_H(10) PLP_CASE_H(11) PLP_CASE_H(12)
        PLP_CASE_H(13) PLP_CASE_H(14) PLP_CASE_H(15) PLP_CASE_H(16)
        default: return 2;
    }
    return 0;
}

void launch_hull_mark(int n, const int* ids, unsigned char* dead, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(hull_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, ids, dead);
}

void launch_hull_drop(long long n, const long long* idx, int* owner, hipStream_t st) {
    if (n > 0)
        hipLaunchKernelGGL(hull_drop_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, idx, owner);
}

}  // namespace plp
