// plp_rdiff.hip -- Chebyshev LPs on ROW SUBSETS of one resident constraint table (gfx950): the LPs of region_diff's
// search (polytope/polytope.py:2148-2152, :2212-2224, :2272-2274).
//
// Every LP of that search is the Chebyshev ball of {A[rows] x <= B[rows]} for an index list `rows` into ONE table of
// m + 2M rows (the minuend's rows, the subtrahends' new rows, and their negations).  The reference builds a Polytope
// per index list; here the table stays in HBM/L2 and a batch is described by index lists only (CSR: off[nlp + 1],
// rows[]), which the lane groups gather while loading -- nothing but 4-byte indices crosses PCIe per search step.
//
//   cheby_gather_r_kernel<D> : lists of up to 16 / 32 / 64 rows (three size classes, one launch) on the four-rows-per-lane engine (d <= 8)
//   (lists beyond 64 rows, and d > 8: cheby_gather_lds_kernel in plp_lds.hip, dictionary in LDS)
//
// out[p] = the radius as cheby_ball reads it (:1289-1297): x[-1] if the LP is optimal with r >= 0, 0 if optimal with
// r < 0, NaN if the LP ended with any other status (the reference reads 0 there; the search must know the difference:
// a cell without a verdict is solved again below, a cell SOLVED as empty is not).
#include <stdlib.h>

#include "plp_cheby_r_impl.hpp"
#include "plp_wide.hpp"

namespace plp {

// one lane group = one LP of the size class GS (lists of up to 4 * GS rows); q = position inside the class
template <int D, int GS, int R = RowsPerLane<D>::value>
__device__ __forceinline__ void gather_body(long long q0, long long nlp, const int* __restrict__ off,
                                            const int* __restrict__ rows, const int* __restrict__ sel,
                                            const double* __restrict__ A, const double* __restrict__ b,
                                            double* __restrict__ out, int force_retry) {
    const Grp g(GS);
    constexpr int gpb = RBLK / GS;
    const int gib = threadIdx.x / GS;
    const int row0 = g.gl * R;
    const long long q = q0 * gpb + gib;
    const bool valid = q < nlp;
    const int p = valid ? sel[q] : 0;          // position of this LP in the batch (lists are grouped by size class)
    const int o = valid ? off[p] : 0;
    const int m = valid ? off[p + 1] - o : 0;
    double x[D + 1];
    const int st = cheby_r_solve<D, GS, R>(
        g, valid, m, row0, [&](int rr, int kk) { return A[(long long)rows[o + rr] * D + kk]; },
        [&](int rr) { return b[rows[o + rr]]; }, x, force_retry);
    if (valid & (g.gl == 0)) out[p] = st != ST_OPT ? __builtin_nan("") : (x[D] >= 0.0 ? x[D] : 0.0);
}

// All three size classes in ONE launch (the search pays per launch, not per LP): workgroups [0, nb0) take the n0 lists
// of up to 16 rows, [nb0, nb0 + nb1) the n1 lists of up to 32, the rest the n2 lists of up to 64.
template <int D>
__global__ __launch_bounds__(RBLK, (D <= 4 ? 3 : 1)) void cheby_gather_r_kernel(int nb0, int nb1, long long n0, long long n1,
                                                                                 long long n2, const int* __restrict__ off,
                                                                                 const int* __restrict__ rows,
                                                                                 const int* __restrict__ sel,
                                                                                 const double* __restrict__ A,
                                                                                 const double* __restrict__ b,
                                                                                 double* __restrict__ out, int force_retry) {
    const int bid = blockIdx.x;
    if (bid < nb0) gather_body<D, 4>(bid, n0, off, rows, sel, A, b, out, force_retry);
    else if (bid < nb0 + nb1) gather_body<D, 8>(bid - nb0, n1, off, rows, sel + n0, A, b, out, force_retry);
    else gather_body<D, 16>(bid - nb0 - nb1, n2, off, rows, sel + n0 + n1, A, b, out, force_retry);
}

// The same with ONE row per lane (groups of 16 / 32 / 64 lanes): a pivot costs the wavefront about half the instructions,
// so a lone wavefront finishes its LP sooner.  A search batch is a few dozen to a few hundred LPs -- far fewer
// wavefronts than the chip holds either way -- and the search waits for the slowest of them: latency, not throughput.
template <int D>
__global__ __launch_bounds__(RBLK, (D <= 4 ? 3 : 1)) void cheby_gather_r1_kernel(int nb0, int nb1, long long n0, long long n1,
                                                                                  long long n2, const int* __restrict__ off,
                                                                                  const int* __restrict__ rows,
                                                                                  const int* __restrict__ sel,
                                                                                  const double* __restrict__ A,
                                                                                  const double* __restrict__ b,
                                                                                  double* __restrict__ out, int force_retry) {
    const int bid = blockIdx.x;
    if (bid < nb0) gather_body<D, 16, 1>(bid, n0, off, rows, sel, A, b, out, force_retry);
    else if (bid < nb0 + nb1) gather_body<D, 32, 1>(bid - nb0, n1, off, rows, sel + n0, A, b, out, force_retry);
    else gather_body<D, 64, 1>(bid - nb0 - nb1, n2, off, rows, sel + n0 + n1, A, b, out, force_retry);
}

// After the LP kernels of a batch (same stream): copy the n radii to host-mapped memory and raise the batch's sequence
// number there; the host spins on that word instead of paying a stream synchronisation per batch.
__global__ __launch_bounds__(256) void rdiff_publish_kernel(long long n, const double* __restrict__ src,
                                                            double* __restrict__ host_out,
                                                            unsigned long long* __restrict__ host_flag,
                                                            unsigned long long seq) {
    for (long long k = threadIdx.x; k < n; k += 256) host_out[k] = src[k];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int D>
static int launch_gather_d(long long n0, long long n1, long long n2, const int* off, const int* rows, const int* sel,
                           const double* A, const double* b, double* out, hipStream_t st) {
    // small batches (the search's): one row per lane; PLP_RDIFF_R1=0 / 1: never / always (A/B)
    const char* r1 = getenv("PLP_RDIFF_R1");
    const bool lowlat = r1 ? r1[0] == '1' : (n0 + n1 + n2 <= 2048);
    if (lowlat) {
        const long long b0 = (n0 + RBLK / 16 - 1) / (RBLK / 16), b1 = (n1 + RBLK / 32 - 1) / (RBLK / 32),
                        b2 = (n2 + RBLK / 64 - 1) / (RBLK / 64);
        if (b0 + b1 + b2 < 1) return 0;
        hipLaunchKernelGGL((cheby_gather_r1_kernel<D>), dim3((unsigned)(b0 + b1 + b2)), dim3(RBLK), 0, st, (int)b0, (int)b1, n0,
                           n1, n2, off, rows, sel, A, b, out, force_retry_env());
        return 0;
    }
    const long long nb0 = (n0 + RBLK / 4 - 1) / (RBLK / 4), nb1 = (n1 + RBLK / 8 - 1) / (RBLK / 8),
                    nb2 = (n2 + RBLK / 16 - 1) / (RBLK / 16);
    const long long blocks = nb0 + nb1 + nb2;
    if (blocks < 1) return 0;
    if (blocks > 2147483647ll) return 2;
    hipLaunchKernelGGL((cheby_gather_r_kernel<D>), dim3((unsigned)blocks), dim3(RBLK), 0, st, (int)nb0, (int)nb1, n0, n1, n2,
                       off, rows, sel, A, b, out, force_retry_env());
    return 0;
}

// LPs sel[0 .. n0 + n1 + n2) of the batch: n0 with at most 16 rows, then n1 with at most 32, then n2 with at most 64;
// d <= 8; returns 1 when it does not apply
int launch_cheby_gather_r(int d, long long n0, long long n1, long long n2, const int* off, const int* rows, const int* sel,
                          const double* A, const double* b, double* out, hipStream_t st) {
    switch (d) {
        case 1: return launch_gather_d<1>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 2: return launch_gather_d<2>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 3: return launch_gather_d<3>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 4: return launch_gather_d<4>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 5: return launch_gather_d<5>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 6: return launch_gather_d<6>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 7: return launch_gather_d<7>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 8: return launch_gather_d<8>(n0, n1, n2, off, rows, sel, A, b, out, st);
        default: return 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The LP server of a search (d <= 4): ONE launch stays resident for the whole search and takes batch after batch from a
// host-mapped mailbox, so a visited node costs neither a kernel launch, nor the dispatch gap to a publish kernel, nor a
// chain of dependent reads across PCIe, nor a fence -- what remained of a node after round 3 (launches 4 ms + device wait
// 11 ms of a 23 ms search at config 4: ~15 us of launch, dispatch and index chasing around ~6 us of pivots).
//   mailbox: ONE 64-bit word  [batch number : 28 | n2 : 12 | n1 : 12 | n0 : 12]  (a single store on the host, a single
//            load on the device: the list counts cannot be seen apart from the batch they belong to)
//   records, by size class (<= 16 / 32 / 64 rows), fixed stride:   [position in the batch | length | rows ...]
//   results: 16 bytes per LP, [radius | mailbox word], stored straight into host memory, the word after the radius was
//            acknowledged -- the host knows a radius is there when the word beside it is the batch's (no completion
//            counter, no system-scope fence on the device:
//            a first version had both and lost to the launches it replaced, 18.5 ms against 15.6 ms of waiting).
// Every wait is bounded: after `idle_polls` empty polls of the mailbox workgroup 0 retires the server (a device word the
// others look at between polls; the host relaunches it when the next batch comes), and a host that wants it gone writes
// RD_EXIT.
constexpr unsigned long long RD_EXIT = ~0ull;

template <int D, int GS>
__device__ __forceinline__ void server_slot(int slot, int n, const int* __restrict__ rec, int cap,
                                            const double* __restrict__ A, const double* __restrict__ b,
                                            ulonglong2* __restrict__ out_host, unsigned long long word, int force_retry) {
    const Grp g(GS);
    constexpr int gpb = 64 / GS;
    const int q = slot * gpb + (int)threadIdx.x / GS;
    const bool valid = q < n;
    const int* r = rec + (size_t)(valid ? q : 0) * (cap + 2);
    // The record is read with system-scope loads: a plain load may be served from a cache line this resident kernel
    // fetched for an EARLIER batch (nothing invalidates the caches between batches; a first version read stale rows).
    // One read across PCIe per lane, all in flight together.
    int p = 0, m = 0, myrow = 0;
    if (valid) {
        p = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        m = __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        myrow = __hip_atomic_load(r + 2 + (g.gl < cap ? g.gl : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        myrow = g.gl < m ? myrow : 0;
    }
    double x[D + 1];
    const int st = cheby_r_solve<D, GS, 1>(
        g, valid, m, g.gl, [&](int, int kk) { return A[(long long)myrow * D + kk]; }, [&](int) { return b[myrow]; }, x,
        force_retry);
    if (valid & (g.gl == 0)) {
        const double rad = st != ST_OPT ? __builtin_nan("") : (x[D] >= 0.0 ? x[D] : 0.0);
        // Written through to system memory at once (system-scope stores: a plain store to host memory may sit in the L2
        // until the kernel ends -- and this kernel does not end).  The radius first; the word beside it only after the
        // radius store has been acknowledged (one 16-byte store arrived torn: new word, old radius).
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(out_host + p);
        __hip_atomic_store(dst, (unsigned long long)__double_as_longlong(rad), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(dst + 1, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The same batch entry on the one-LP-per-wavefront engine (plp_wide.hpp: wave-uniform pivot column and row, the body of
// cheby_gather_w_kernel): a list of any length up to 64 rows takes a wavefront of its own.  Measured on the launch path at
// d = 4 (PLP_RDIFF_WIDE_MIND=4): 9.7 ms of device time for the 751 batches of config 4 against 10.9 ms on the lane groups.
template <int D>
__device__ __forceinline__ void server_slot_w(int q, const int* __restrict__ rec, int cap, const double* __restrict__ A,
                                              const double* __restrict__ b, ulonglong2* __restrict__ out_host,
                                              unsigned long long word) {
    using namespace wide;
    constexpr int NC = D + 1;
    __shared__ WideShared<NC> sh;
    const int lane = threadIdx.x;
    const int* r = rec + (size_t)q * (cap + 2);
    const int p = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int m = __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int myrow = __hip_atomic_load(r + 2 + (lane < cap ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const bool has = lane < m;
    const long long row = has ? myrow : 0;
    typename RowVec<NC>::type Tv = (typename RowVec<NC>::type)(0.0);
    double T16 = 0.0;
    double nrm2 = 0.0;
    bool finite = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double v = has ? A[row * D + k] : 0.0;
        ROW_SET(k, v);
        nrm2 = nrm2 + v * v;
        finite = finite & isfinite(v);
    }
    const double bi = has ? b[row] : 0.0;
    finite = finite & isfinite(bi);
    const double nrm = sqrt(nrm2);
    const bool zero = !(nrm > 0.0);
    bool rowact = has & !zero;
    ROW_SET(D, rowact ? nrm : 0.0);
    double beta = rowact ? bi : 0.0;
    int rowvar = NC + lane, rowneg = 0;
    wave_sync();   // (the last LP's reads of the block are done)
    if (lane <= NC) {
        sh.cost[lane] = lane == D ? -1.0 : 0.0;
        sh.cv[lane] = (lane + 1) << 1;
    }
    const bool infeasible0 = __ballot(has & zero & (bi < -TOL_FEAS)) != 0;
    const bool bad = (__ballot(!finite) != 0) | (m > 64);
    wave_sync();
    int st, iters = 0;
    if (bad) st = ST_NUM;
    else if (infeasible0) st = ST_INFEAS;
    else st = wide_run<NC>(lane, m, Tv, T16, beta, rowvar, rowneg, rowact, sh, NC, true, bi / nrm, iters);
    const double mine = rowneg ? -beta : beta;
    const uint64_t ob = __ballot(rowvar == D);
    const double rv = ob ? uniform_lane(mine, __ffsll((long long)ob) - 1) : 0.0;
    if (lane == 0) {
        const double rad = st != ST_OPT ? __builtin_nan("") : (rv >= 0.0 ? rv : 0.0);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(out_host + p);
        __hip_atomic_store(dst, (unsigned long long)__double_as_longlong(rad), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(dst + 1, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int D>
__global__ __launch_bounds__(64) void rdiff_server_kernel(const unsigned long long* __restrict__ mail_host,
                                                          const int* __restrict__ rec_host,
                                                          ulonglong2* __restrict__ out_host,
                                                          unsigned long long* __restrict__ alive_host,
                                                          unsigned long long* __restrict__ dstate,
                                                          const double* __restrict__ A, const double* __restrict__ b,
                                                          unsigned long long last_word, unsigned idle_polls, int force_retry, int wide_engine) {
    const int wg = blockIdx.x, G = gridDim.x, lane = threadIdx.x;
    unsigned long long seen = last_word;
    unsigned polls = 0;
    for (;;) {
        // Only workgroup 0 polls the mailbox across PCIe and republishes it in device memory, where the others look: with
        // every workgroup polling the host the reads queue up behind each other (scripts/microbench/mailbox_latency.hip:
        // a round trip of 1.7 us with up to 16 pollers, 6.5 us with 64 -- and the record reads wait in the same queue).
        unsigned long long w = 0ull, e = 0ull;
        if (lane == 0) {
            if (wg == 0) {
                w = __hip_atomic_load(mail_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (w != seen && w != RD_EXIT) __hip_atomic_store(dstate + 1, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                w = __hip_atomic_load(dstate + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                e = __hip_atomic_load(dstate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        w = __shfl(w, 0, 64);
        e = __shfl(e, 0, 64);
        if (w == RD_EXIT || e == RD_EXIT) break;
        if (w == seen) {
            // (workgroup 0 decides for all; the others carry a generous bound of their own so that no wait is unbounded)
            if (++polls > (wg == 0 ? idle_polls : idle_polls * 64u + 65536u)) break;
            if (wg != 0) __builtin_amdgcn_s_sleep(1);
            continue;
        }
        polls = 0;
        seen = w;
        const int n0 = (int)(w & 0xfffull), n1 = (int)((w >> 12) & 0xfffull), n2 = (int)((w >> 24) & 0xfffull);
        // slots: 4 lists of <= 16 rows, 2 of <= 32 or 1 of <= 64 per wavefront
        const int s0 = (n0 + 3) >> 2, s1 = (n1 + 1) >> 1, s2 = n2;
        const int total = s0 + s1 + s2;
        const int* rec0 = rec_host;
        const int* rec1 = rec0 + (size_t)n0 * 18;
        const int* rec2 = rec1 + (size_t)n1 * 34;
        if (wide_engine) {   // one LP per wavefront, whatever its length
            for (int s = wg; s < n0 + n1 + n2; s += G) {
                if (s < n0) server_slot_w<D>(s, rec0, 16, A, b, out_host, w);
                else if (s < n0 + n1) server_slot_w<D>(s - n0, rec1, 32, A, b, out_host, w);
                else server_slot_w<D>(s - n0 - n1, rec2, 64, A, b, out_host, w);
            }
        } else
        for (int s = wg; s < total; s += G) {
            if (s < s0) server_slot<D, 16>(s, n0, rec0, 16, A, b, out_host, w, force_retry);
            else if (s < s0 + s1) server_slot<D, 32>(s - s0, n1, rec1, 32, A, b, out_host, w, force_retry);
            else server_slot<D, 64>(s - s0 - s1, n2, rec2, 64, A, b, out_host, w, force_retry);
        }
    }
    if (wg == 0 && lane == 0) {
        __hip_atomic_store(dstate, RD_EXIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // the others follow
        __hip_atomic_store(alive_host, 0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);         // retired
    }
}

// returns 1 when there is no server for this dimension
int launch_rdiff_server(int d, int nwg, const unsigned long long* mail, const int* rec, void* out, unsigned long long* alive,
                        unsigned long long* dstate, const double* A, const double* b, unsigned long long last_word,
                        unsigned idle_polls, hipStream_t st) {
    // PLP_RDIFF_SERVER_WIDE=0: the entries on the lane-group kernels' engine (round 4's first form)
    const char* we = getenv("PLP_RDIFF_SERVER_WIDE");
    const int wide_engine = (we && we[0] == '0') ? 0 : 1;
#define PLP_SRV(K)                                                                                                      \
    case K:                                                                                                             \
        hipLaunchKernelGGL((rdiff_server_kernel<K>), dim3((unsigned)nwg), dim3(64), 0, st, mail, rec,                    \
                           static_cast<ulonglong2*>(out), alive, dstate, A, b, last_word, idle_polls, force_retry_env(), wide_engine); \
        return 0;
    switch (d) {
        PLP_SRV(1) PLP_SRV(2) PLP_SRV(3) PLP_SRV(4)
        default: return 1;
    }
#undef PLP_SRV
}

void launch_rdiff_publish(long long n, const double* src, double* host_out, unsigned long long* host_flag,
                          unsigned long long seq, hipStream_t st) {
    hipLaunchKernelGGL(rdiff_publish_kernel, dim3(1), dim3(256), 0, st, n, src, host_out, host_flag, seq);
}

}  // namespace plp
