// plp_quickhull_dev.hip -- the main loop of Quickhull (polytope/quickhull.py:224-345) with the FACET GRAPH ON THE DEVICE.
//
// plp_quickhull_host.hip keeps the facet graph in host memory and pays one device round trip per iteration while many
// points are outside the hull; the long tail (fewer than 32 768 outside points: thousands of iterations that move a
// handful of points each and make tens to hundreds of facets) used to continue on host lists.  Here that tail is ONE
// persistent kernel: a single workgroup of 1024 threads owns the whole state in device memory -- facet table (normals,
// offsets, vertices, neighbour lists), the FIFO of facets with outside points, the outside points themselves (compacted:
// at most a few ten thousand) -- and runs iteration after iteration until the queue is empty.  The host gets the facets
// back at the end; nothing crosses PCIe in between.
//
// Every step of an iteration keeps the reference's ORDER (which decides the order of the output rows) while working in
// parallel:
//   * visibility (:254-270) is a breadth-first search; a level is evaluated by all threads at once (distance of the apex
//     to every facet of the level, numpy's summation order) and the next level is the list of not-yet-discovered
//     neighbours in (parent position, neighbour slot) order, duplicates resolved in favour of the first -- the order a
//     sequential queue produces;
//   * the horizon (:284-304) is the compaction, in (visible position, neighbour slot) order, of the pairs (visible facet,
//     non-visible neighbour); one thread per pair forms the new facet's vertex list and solves its (d+1) x (d+1)
//     hyperplane system (the LU with partial pivoting of plp_quickhull_host.hip, operation for operation);
//   * links among the new facets (:305-310) through an open-addressing hash table keyed by an order-independent hash of
//     the sub-ridge, sets compared before linking, partners sorted ascending;
//   * the neighbour list of every facet behind the horizon becomes (old entries that are not visible, order kept) + (its new
//     neighbours in creation order);
//   * the pooled points (:273-283, :311-336) go to the first new facet, in creation order, beyond which they lie
//     (k-ordered products and sums, as hull_reassign_kernel), the furthest point per facet is the one with the largest
//     distance, lowest point index among equals.
// Bitwise the facets of the host loop (PLP_QH_DEVICE_TAIL=0), in the same order: tests/test_quickhull.py.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../include/plp.h"
#include "plp_kernels.hpp"
#include "plp_quickhull_dev.hpp"

namespace plp {

namespace {

constexpr int QT = 1024;            // threads of the one workgroup
constexpr int QH_CH = 512;          // new facets staged in LDS at a time (reassignment)
constexpr size_t QH_LDS_BYTES = 128 * 1024;   // dynamic LDS of the workgroup: the facet chunk, or the hyperplane systems being eliminated
constexpr int QH_OK = 0, QH_GROW = 1, QH_SINGULAR = 2, QH_NBR_OVERFLOW = 3, QH_SCRATCH = 4, QH_IDENT = 5;

struct QhDev {
    int d, cap, capn, scr, htcap, M;
    double tol;
    const double* X;          // [N][d] translated points (the hull session's)
    const int* opt;           // [M] point id of compact entry ci (ascending)
    int* oown;                // [M] owner facet, -1 = inside / taken
    double* odist;            // [M]
    double* FN;               // [cap][d]
    double* FO;               // [cap]
    int* FV;                  // [cap][d] vertex point ids
    int* NB;                  // [cap][capn]
    int* NBN;                 // [cap]
    unsigned char* LIVE;      // [cap]
    unsigned char* INP;       // [cap] in the pending queue
    int* CNT;                 // [cap] outside points
    int* FAR;                 // [cap] compact index of the furthest outside point
    unsigned long long* FKEY; // [cap] bits of the largest distance (reassignment)
    int* PQ;                  // [cap] pending FIFO (every facet enters at most once)
    int* MARK;                // [cap] == stamp: discovered by the current search
    int* VISM;                // [cap] == stamp: visible from the current apex
    int* KEY;                 // [cap] lowest candidate key of the current level (INT_MAX otherwise)
    int* TMPN;                // [cap] new neighbours collected for a facet behind the horizon
    int* TMP;                 // [cap][capn]
    int* VL;                  // [scr] visible list
    int* QA;                  // [scr] search level
    int* QB;                  // [scr]
    int* FLG;                 // [scr] per level entry: visible?
    int* H1;                  // [scr] horizon pair j: position of its visible facet in VL
    int* H2;                  // [scr] ... neighbour slot
    int* OUTER;               // [scr] ... the facet behind the ridge
    int* AFF;                 // [scr] facets behind the horizon (each once)
    unsigned long long* HT;   // [htcap] sub-ridge table: id + 1 (0 = empty)
    unsigned long long* HH;   // [htcap] hash of the entry
    long long* tim;           // [16] (PLP_QH_DEV_TIMING) clock ticks per phase
    int* ctrl;                // [16] 0 head, 1 tail, 2 facets, 3 status, 4 iterations, 5 outside points, 6 facets made here, 7 detail
};

__device__ __forceinline__ unsigned long long mix_id(long long v) {
    unsigned long long x = (unsigned long long)v + 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// numpy's add.reduce over n doubles (plp_quickhull_host.hip: np_sum)
__device__ __forceinline__ double np_sum_dev(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// np.sum(n * p) for a run-time d <= 16: the products, then numpy's order of additions (np_sum_dev)
__device__ __forceinline__ double np_dot_dev(const double* n, const double* x, int d) {
    double prod[16];
    for (int c = 0; c < d; ++c) prod[c] = n[c] * x[c];
    return np_sum_dev(prod, d);
}

// exclusive prefix sum over the workgroup; *total = the sum
__device__ __forceinline__ int block_scan(int v, int* total, int* sh /* [17] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) sh[w] = x;
    __syncthreads();
    if (w == 0) {
        int s = lane < QT / 64 ? sh[lane] : 0;
#pragma unroll
        for (int o = 1; o < QT / 64; o <<= 1) {
            const int y = __shfl_up(s, o, 64);
            if (lane >= o) s += y;
        }
        if (lane < QT / 64) sh[lane] = s;
    }
    __syncthreads();
    const int base = w ? sh[w - 1] : 0;
    *total = sh[QT / 64 - 1];
    __syncthreads();
    return base + x - v;
}

// unit outward normal and offset of the facet through the d points v[] (plp_quickhull_host.hip: hyperplane / own_solve,
// the same operations in the same order); returns 1 when the system is singular
// (M: (d+1)^2 doubles, rhs: d+1 doubles of the calling thread's own -- in LDS: as a private array they live in scratch
// memory and the elimination ran at its latency)
__device__ int hyperplane_dev(const QhDev& S, const int* v, double* n_out, double* off_out, double* M, double* rhs) {
    const int d = S.d, n = d + 1;
    double prod[16];
    for (int i = 0; i < n * n; ++i) M[i] = 0.0;
    for (int i = 0; i < n; ++i) rhs[i] = 0.0;
    for (int r = 0; r < d; ++r) {
        for (int c = 0; c < d; ++c) M[c * n + r] = S.X[(long long)v[r] * d + c];
        M[d * n + r] = 1.0;
    }
    M[d * n + d] = -1.0;
    rhs[d] = 1.0;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(M[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(M[k * n + i]) > best) { best = fabs(M[k * n + i]); piv = i; }
        if (best == 0.0) return 1;
        if (piv != k) {
            for (int j = 0; j < n; ++j) { const double t = M[j * n + k]; M[j * n + k] = M[j * n + piv]; M[j * n + piv] = t; }
            const double t = rhs[k]; rhs[k] = rhs[piv]; rhs[piv] = t;
        }
        const double inv = 1.0 / M[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double l = M[k * n + i] * inv;
            M[k * n + i] = l;
            for (int j = k + 1; j < n; ++j) M[j * n + i] -= l * M[j * n + k];
            rhs[i] -= l * rhs[k];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        for (int j = k + 1; j < n; ++j) rhs[k] -= M[j * n + k] * rhs[j];
        rhs[k] /= M[k * n + k];
    }
    for (int c = 0; c < d; ++c) prod[c] = rhs[c] * rhs[c];
    const double mult = sqrt(np_sum_dev(prod, d));
    for (int c = 0; c < d; ++c) n_out[c] = rhs[c] / mult;
    const double dd = rhs[d] / mult;
    for (int c = 0; c < d; ++c) prod[c] = n_out[c] * S.X[(long long)v[0] * d + c];
    if (np_sum_dev(prod, d) < 0.0)
        for (int c = 0; c < d; ++c) n_out[c] = -n_out[c];
    *off_out = -dd;
    return 0;
}

// same sub-ridge?  (facet ja without its ridge vertex oa) == (facet jc without oc), as sets of d - 2 point ids
__device__ bool same_sub_dev(const QhDev& S, int s0, int ja, int oa, int jc, int oc) {
    const int d = S.d;
    const int* va = S.FV + (long long)(s0 + ja) * d + 1;
    const int* vc = S.FV + (long long)(s0 + jc) * d + 1;
    for (int t = 0; t < d - 1; ++t) {
        if (t == oa) continue;
        bool found = false;
        for (int u = 0; u < d - 1; ++u) found = found || (u != oc && vc[u] == va[t]);
        if (!found) return false;
    }
    for (int u = 0; u < d - 1; ++u) {
        if (u == oc) continue;
        bool found = false;
        for (int t = 0; t < d - 1; ++t) found = found || (t != oa && va[t] == vc[u]);
        if (!found) return false;
    }
    return true;
}

#ifdef PLP_QH_DEV_TIMING
#define QH_LAP(i) do { __syncthreads(); if (tid == 0) { const long long t_ = wall_clock64(); S.tim[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define QH_LAP(i) do { } while (0)
#endif

__global__ __launch_bounds__(QT) void qh_tail_kernel(QhDev S) {
    __shared__ int sh[20];
    extern __shared__ __attribute__((aligned(16))) unsigned char qh_smem[];
    double* s_fn = reinterpret_cast<double*>(qh_smem);          // [QH_CH][d] new facets of a chunk (reassignment)
    double* s_fo = s_fn + QH_CH * S.d;                            // [QH_CH]
    __shared__ int s_facet, s_ci, s_nq, s_nvis, s_stop, s_naff;
    const int tid = threadIdx.x;
    const int d = S.d, capn = S.capn;
    int* ctrl = S.ctrl;
#ifdef PLP_QH_DEV_TIMING
    long long tlast = wall_clock64();
#endif
    for (;;) {
        // ---------------------------------------------------------------- the next facet with outside points (FIFO)
        if (tid == 0) {
            int head = ctrl[0];
            const int tail = ctrl[1];
            while (head < tail && !S.INP[S.PQ[head]]) ++head;
            ctrl[0] = head;
            s_stop = head >= tail;
            if (!s_stop) {
                const int f = S.PQ[head];
                s_facet = f;
                s_ci = S.FAR[f];
            }
        }
        __syncthreads();
        if (s_stop) break;
        const int facet = s_facet, ci = s_ci;
        const int stamp = ctrl[4] + 1;
        const long long p = S.opt[ci];           // the apex: its furthest point (get_furthest takes it out of the set, :87-102)
        const double* xp = S.X + p * d;
        // ---------------------------------------------------------------- visible set, breadth first (:254-270)
        if (tid == 0) {
            S.VL[0] = facet;
            S.MARK[facet] = stamp;
            S.VISM[facet] = stamp;
            s_nvis = 1;
            const int nn = S.NBN[facet];
            for (int s = 0; s < nn; ++s) { const int nb = S.NB[(long long)facet * capn + s]; S.QA[s] = nb; S.MARK[nb] = stamp; }
            s_nq = nn;
        }
        __syncthreads();
        int* qa = S.QA;
        int* qb = S.QB;
        for (;;) {
            const int nq = s_nq;
            if (nq == 0) break;
            if (nq > S.scr) { if (tid == 0) ctrl[3] = QH_SCRATCH; __syncthreads(); return; }
            // is every facet of the level visible from p?  distance() (:117-121): numpy's sum of the products
            for (int base = 0; base < nq; base += QT) {
                const int i = base + tid;
                int vis = 0;
                if (i < nq) {
                    const int f = qa[i];
                    double prod[16];
                    for (int c = 0; c < d; ++c) prod[c] = S.FN[(long long)f * d + c] * xp[c];
                    vis = (np_sum_dev(prod, d) - S.FO[f]) > S.tol ? 1 : 0;
                    S.FLG[i] = vis;
                    if (vis) S.VISM[f] = stamp;
                }
                int tot;
                const int pos = block_scan(vis, &tot, sh);
                if (vis) S.VL[s_nvis + pos] = qa[i];
                __syncthreads();
                if (tid == 0) s_nvis += tot;
                __syncthreads();
            }
            if (s_nvis > S.scr) { if (tid == 0) ctrl[3] = QH_SCRATCH; __syncthreads(); return; }
            // next level: the undiscovered neighbours of the visible ones, first discoverer in (position, slot) order wins
            const long long ncand = (long long)nq * capn;
            for (long long c = tid; c < ncand; c += QT) {
                const int i = (int)(c / capn), s = (int)(c - (long long)i * capn);
                if (!S.FLG[i]) continue;
                const int f = qa[i];
                if (s >= S.NBN[f]) continue;
                const int nn = S.NB[(long long)f * capn + s];
                if (S.MARK[nn] != stamp) atomicMin(&S.KEY[nn], (int)c);
            }
            __syncthreads();
            int nnext = 0;
            for (long long base = 0; base < ncand; base += QT) {
                const long long c = base + tid;
                int take = 0, nn = -1;
                if (c < ncand) {
                    const int i = (int)(c / capn), s = (int)(c - (long long)i * capn);
                    if (S.FLG[i]) {
                        const int f = qa[i];
                        if (s < S.NBN[f]) {
                            nn = S.NB[(long long)f * capn + s];
                            take = (S.MARK[nn] != stamp && S.KEY[nn] == (int)c) ? 1 : 0;
                        }
                    }
                }
                int tot;
                const int pos = block_scan(take, &tot, sh);
                if (take && nnext + pos < S.scr) qb[nnext + pos] = nn;
                nnext += tot;
            }
            __syncthreads();
            if (nnext > S.scr) { if (tid == 0) ctrl[3] = QH_SCRATCH; __syncthreads(); return; }
            for (int i = tid; i < nnext; i += QT) { const int nn = qb[i]; S.MARK[nn] = stamp; S.KEY[nn] = 0x7fffffff; }
            if (tid == 0) s_nq = nnext;
            __syncthreads();
            int* t = qa; qa = qb; qb = t;
        }
        const int nvis = s_nvis;
        QH_LAP(0);
        // ---------------------------------------------------------------- horizon: (visible facet, non-visible neighbour) pairs in order (:284-304)
        int k = 0;
        {
            const long long ncand = (long long)nvis * capn;
            for (long long base = 0; base < ncand; base += QT) {
                const long long c = base + tid;
                int take = 0, vi = 0, s = 0;
                if (c < ncand) {
                    vi = (int)(c / capn);
                    s = (int)(c - (long long)vi * capn);
                    const int f1 = S.VL[vi];
                    if (s < S.NBN[f1]) take = S.VISM[S.NB[(long long)f1 * capn + s]] != stamp ? 1 : 0;
                }
                int tot;
                const int pos = block_scan(take, &tot, sh);
                if (take && k + pos < S.scr) { S.H1[k + pos] = vi; S.H2[k + pos] = s; }
                k += tot;
            }
        }
        __syncthreads();
        if (k > S.scr) { if (tid == 0) ctrl[3] = QH_SCRATCH; __syncthreads(); return; }
        QH_LAP(1);
        const int s0 = ctrl[2];
        if (s0 + k > S.cap || (long long)k * (d - 1) * 2 > S.htcap) {
            // no room for the new facets: leave with the state untouched (the marks are stamps) -- the host grows the arrays
            if (tid == 0) { ctrl[3] = QH_GROW; ctrl[7] = k; }
            __syncthreads();
            return;
        }
        // ---- from here on the iteration modifies the state
        if (tid == 0) {
            S.oown[ci] = -1;                 // the apex leaves its facet's outside set
            S.CNT[facet] -= 1;
            ctrl[5] -= 1;
        }
        // new facets: vertices (the apex + the ridge), hyperplane; the facet behind the ridge
        for (int j = tid; j < k; j += QT) {
            const int f1 = S.VL[S.H1[j]];
            const int f2 = S.NB[(long long)f1 * capn + S.H2[j]];
            const int* v1 = S.FV + (long long)f1 * d;
            const int* v2 = S.FV + (long long)f2 * d;
            int skip = -1;
            for (int ii = 0; ii < d; ++ii) {
                bool found = false;
                for (int jj = 0; jj < d; ++jj) found = found || (v2[jj] == v1[ii]);
                if (!found) { skip = ii; break; }
            }
            int* nv = S.FV + (long long)(s0 + j) * d;
            if (skip < 0) { ctrl[3] = QH_IDENT; skip = 0; }
            nv[0] = (int)p;
            int w = 1;
            for (int ii = 0; ii < d; ++ii) if (ii != skip) nv[w++] = v1[ii];
            S.OUTER[j] = f2;
            S.LIVE[s0 + j] = 1;
            S.INP[s0 + j] = 0;
            S.CNT[s0 + j] = 0;
            S.FAR[s0 + j] = -1;
            S.FKEY[s0 + j] = 0ull;
            S.MARK[s0 + j] = 0;
            S.VISM[s0 + j] = 0;
            S.KEY[s0 + j] = 0x7fffffff;
            S.TMPN[s0 + j] = 0;
            // the facet behind the ridge collects its new neighbour (put in creation order below)
            const int t = atomicAdd(&S.TMPN[f2], 1);
            if (t < capn) S.TMP[(long long)f2 * capn + t] = j; else ctrl[3] = QH_NBR_OVERFLOW;
        }
        __syncthreads();
        {   // hyperplanes: TB systems at a time, each thread eliminating its own in LDS
            const int per = (d + 1) * (d + 1) + (d + 1);
            int TB = (int)(QH_LDS_BYTES / ((size_t)per * 8));
            TB = TB > 256 ? 256 : TB;
            for (int jb = 0; jb < k; jb += TB) {
                const int j = jb + tid;
                if (tid < TB && j < k) {
                    double* M = reinterpret_cast<double*>(qh_smem) + (size_t)tid * per;
                    if (hyperplane_dev(S, S.FV + (long long)(s0 + j) * d, S.FN + (long long)(s0 + j) * d, S.FO + (s0 + j), M,
                                       M + (d + 1) * (d + 1)))
                        ctrl[3] = QH_SINGULAR;
                }
            }
        }
        QH_LAP(2);
        // sub-ridge table
        const int nsub = k * (d - 1);
        int tsz = 16;
        while (tsz < 2 * nsub) tsz <<= 1;
        for (int i = tid; i < tsz; i += QT) S.HT[i] = 0ull;
        if (tid == 0) s_naff = 0;
        __syncthreads();
        if (ctrl[3] != QH_OK) return;
        for (int id = tid; id < nsub; id += QT) {
            const int j = id / (d - 1), omit = id - j * (d - 1);
            const int* v = S.FV + (long long)(s0 + j) * d + 1;
            unsigned long long hs = 0;
            for (int t = 0; t < d - 1; ++t) hs += mix_id(v[t]);
            const unsigned long long h = hs - mix_id(v[omit]);
            int slot = (int)((h ^ (h >> 32)) & (unsigned long long)(tsz - 1));
            for (;;) {
                const unsigned long long old = atomicCAS(&S.HT[slot], 0ull, (unsigned long long)(id + 1));
                if (old == 0ull) { S.HH[slot] = h; break; }
                slot = (slot + 1) & (tsz - 1);
            }
        }
        // the facets behind the horizon, each once
        for (int j = tid; j < k; j += QT) {
            const int f2 = S.OUTER[j];
            if (S.TMP[(long long)f2 * capn] == j) { const int a = atomicAdd(&s_naff, 1); S.AFF[a] = f2; }   // (slot 0 belongs to exactly one j)
        }
        __syncthreads();
        QH_LAP(3);
        // ---- neighbour lists of the new facets: the facet behind the ridge, then the linked new facets ascending (:305-310)
        for (int j = tid; j < k; j += QT) {
            const int* v = S.FV + (long long)(s0 + j) * d + 1;
            unsigned long long hs = 0;
            for (int t = 0; t < d - 1; ++t) hs += mix_id(v[t]);
            int* nb = S.NB + (long long)(s0 + j) * capn;
            int n = 1;                      // slot 0: the facet behind the ridge
            for (int omit = 0; omit < d - 1; ++omit) {
                const unsigned long long h = hs - mix_id(v[omit]);
                int slot = (int)((h ^ (h >> 32)) & (unsigned long long)(tsz - 1));
                for (;;) {
                    const unsigned long long e = S.HT[slot];
                    if (e == 0ull) break;
                    if (S.HH[slot] == h) {
                        const int id2 = (int)e - 1;
                        const int jc = id2 / (d - 1), oc = id2 - jc * (d - 1);
                        if (jc != j && same_sub_dev(S, s0, j, omit, jc, oc)) {
                            // keep the partners ascending and each once (a handful of entries: insertion in place)
                            const int val = s0 + jc;
                            int pos = n;
                            bool dup = false;
                            for (int a = 1; a < n; ++a) {
                                if (nb[a] == val) { dup = true; break; }
                                if (nb[a] > val) { pos = a; break; }
                            }
                            if (!dup) {
                                if (n < capn) {
                                    for (int a = n; a > pos; --a) nb[a] = nb[a - 1];
                                    nb[pos] = val;
                                    ++n;
                                } else {
                                    ctrl[3] = QH_NBR_OVERFLOW;
                                }
                            }
                        }
                    }
                    slot = (slot + 1) & (tsz - 1);
                }
            }
            nb[0] = S.OUTER[j];
            S.NBN[s0 + j] = n;
        }
        QH_LAP(4);
        // ---- neighbour lists behind the horizon: what was there and is not visible (order kept), then the new ones in creation order
        const int naff = s_naff;
        for (int a = tid; a < naff; a += QT) {
            const int f2 = S.AFF[a];
            int* nb = S.NB + (long long)f2 * capn;
            int* tm = S.TMP + (long long)f2 * capn;
            int nt = S.TMPN[f2];
            if (nt > capn) nt = capn;
            for (int x = 1; x < nt; ++x) {
                const int v = tm[x];
                int y = x - 1;
                while (y >= 0 && tm[y] > v) { tm[y + 1] = tm[y]; --y; }
                tm[y + 1] = v;
            }
            int n = 0;
            const int nold = S.NBN[f2];
            for (int s = 0; s < nold; ++s) { const int x = nb[s]; if (S.VISM[x] != stamp) nb[n++] = x; }
            for (int x = 0; x < nt; ++x) { if (n < capn) nb[n++] = s0 + tm[x]; else ctrl[3] = QH_NBR_OVERFLOW; }
            S.NBN[f2] = n;
            S.TMPN[f2] = 0;
        }
        __syncthreads();
        if (ctrl[3] != QH_OK) return;
        QH_LAP(5);
        // ---------------------------------------------------------------- the pooled points go to the new facets (:273-283, :311-336)
        {
            // the pooled entries, compacted (QA is free between searches); owner -2 = not placed yet
            int np_ = 0;
            for (int base = 0; base < S.M; base += QT) {
                const int c = base + tid;
                int pooled = 0;
                if (c < S.M) { const int own = S.oown[c]; pooled = (own >= 0 && S.VISM[own] == stamp) ? 1 : 0; }
                int tot;
                const int pos = block_scan(pooled, &tot, sh);
                if (pooled) { S.QA[np_ + pos] = c; S.oown[c] = -2; }
                np_ += tot;
            }
            __syncthreads();
            // chunks of new facets in LDS; a wavefront takes a point and its lanes the facets of the chunk, 64 at a time: the
            // first facet in creation order beyond which the point lies (k-ordered products and sums) is the lowest set bit
            const int wave = tid >> 6, lane = tid & 63;
            for (int fb = 0; fb < k; fb += QH_CH) {
                const int fc = k - fb < QH_CH ? k - fb : QH_CH;
                __syncthreads();
                for (int i = tid; i < fc * d; i += QT) s_fn[i] = S.FN[(long long)(s0 + fb) * d + i];
                for (int i = tid; i < fc; i += QT) s_fo[i] = S.FO[s0 + fb + i];
                __syncthreads();
                for (int pi = wave; pi < np_; pi += QT / 64) {
                    const int c = S.QA[pi];
                    if (S.oown[c] != -2) continue;       // placed by an earlier chunk (wave-uniform)
                    const double* x = S.X + (long long)S.opt[c] * d;
                    for (int sub = 0; sub < fc; sub += 64) {
                        const int f = sub + lane;
                        double dist = 0.0;
                        bool hit = false;
                        if (f < fc) {
                            const double* nn = s_fn + f * d;
                            dist = np_dot_dev(nn, x, d) - s_fo[f];   // sum(n*p) - d  (quickhull.py:121), numpy's order
                            hit = dist > S.tol;
                        }
                        const unsigned long long hb = __ballot(hit);
                        if (hb) {
                            const int first = __ffsll((long long)hb) - 1;
                            if (lane == first) {
                                const int to = s0 + fb + sub + first;
                                S.oown[c] = to;
                                S.odist[c] = dist;
                                atomicAdd(&S.CNT[to], 1);
                                atomicMax(&S.FKEY[to], (unsigned long long)__double_as_longlong(dist));
                            }
                            break;
                        }
                    }
                }
            }
            __syncthreads();
            // points beyond no new facet are inside the hull for good
            for (int pi = tid; pi < np_; pi += QT) {
                const int c = S.QA[pi];
                if (S.oown[c] == -2) { S.oown[c] = -1; atomicSub(&ctrl[5], 1); }
            }
        }
        __syncthreads();
        QH_LAP(6);
        // the furthest point of each new facet: largest distance, lowest point index among equals (compact entries ascend with it)
        for (int j = tid; j < k; j += QT) S.FAR[s0 + j] = 0x7fffffff;
        __syncthreads();
        for (int c = tid; c < S.M; c += QT) {
            const int own = S.oown[c];
            if (own >= s0 && (unsigned long long)__double_as_longlong(S.odist[c]) == S.FKEY[own]) atomicMin(&S.FAR[own], c);
        }
        __syncthreads();
        // new facets with outside points join the queue in creation order
        {
            int tail = ctrl[1];
            for (int base = 0; base < k; base += QT) {
                const int j = base + tid;
                const int has = (j < k && S.CNT[s0 + j] > 0) ? 1 : 0;
                int tot;
                const int pos = block_scan(has, &tot, sh);
                if (has) { S.PQ[tail + pos] = s0 + j; S.INP[s0 + j] = 1; }
                tail += tot;
            }
            // the visible facets retire (:337-344)
            for (int i = tid; i < nvis; i += QT) {
                const int f = S.VL[i];
                S.LIVE[f] = 0;
                S.INP[f] = 0;
                S.NBN[f] = 0;
            }
            __syncthreads();
            if (tid == 0) {
                ctrl[1] = tail;
                ctrl[2] = s0 + k;
                ctrl[4] = stamp;
                ctrl[6] += k;
            }
        }
        __syncthreads();
        QH_LAP(7);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Host side: lay the state out in ONE device block (parked in the context between hulls), run the kernel, grow and resume
// when the facet table fills up.
static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

int qh_tail_run(QhTailHost& H, void* (*get_block)(void*, size_t), void* user, hipStream_t st, char* err, size_t errn) {
    const int d = H.d;
    const int F0 = (int)H.FO.size();
    const int M = (int)H.opt.size();
    long long cap = 1 << 16;
    while (cap < 8ll * F0 + 64ll * M) cap <<= 1;
    if (const char* e = getenv("PLP_QH_DEV_CAP")) cap = atoll(e) > F0 + 64 ? atoll(e) : F0 + 64;   // (tests: force the grow path)
    std::vector<int> ctrl(16, 0);
    int nf = F0;
    for (;;) {
        if (cap > (1ll << 28)) { snprintf(err, errn, "quickhull: more than 2^28 facets"); return PLP_EUNSUPPORTED; }
        const int scr = (int)cap;
        long long htcap = 1 << 16;
        while (htcap < 4 * cap) htcap <<= 1;
        // layout
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
        const size_t oFN = take((size_t)cap * d * 8), oFO = take((size_t)cap * 8), oFV = take((size_t)cap * d * 4),
                     oNB = take((size_t)cap * H.capn * 4), oNBN = take((size_t)cap * 4), oLIVE = take(cap), oINP = take(cap),
                     oCNT = take((size_t)cap * 4), oFAR = take((size_t)cap * 4), oFKEY = take((size_t)cap * 8),
                     oPQ = take((size_t)cap * 4), oMARK = take((size_t)cap * 4), oVISM = take((size_t)cap * 4),
                     oKEY = take((size_t)cap * 4), oTMPN = take((size_t)cap * 4), oTMP = take((size_t)cap * H.capn * 4),
                     oVL = take((size_t)scr * 4), oQA = take((size_t)scr * 4), oQB = take((size_t)scr * 4),
                     oFLG = take((size_t)scr * 4), oH1 = take((size_t)scr * 4), oH2 = take((size_t)scr * 4),
                     oOUT = take((size_t)scr * 4), oAFF = take((size_t)scr * 4), oHT = take((size_t)htcap * 8),
                     oHH = take((size_t)htcap * 8), oOPT = take((size_t)M * 4 + 4), oOWN = take((size_t)M * 4 + 4),
                     oDST = take((size_t)M * 8 + 8), oCTRL = take(64), oTIM = take(128);
        char* blk = static_cast<char*>(get_block(user, off));
        if (!blk) { snprintf(err, errn, "quickhull: no device memory for %zu bytes of facet tables", off); return PLP_EHIP; }
        QhDev S;
        S.d = d; S.cap = (int)cap; S.capn = H.capn; S.scr = scr; S.htcap = (int)htcap; S.M = M; S.tol = H.tol; S.X = H.Xdev;
        S.FN = (double*)(blk + oFN); S.FO = (double*)(blk + oFO); S.FV = (int*)(blk + oFV); S.NB = (int*)(blk + oNB);
        S.NBN = (int*)(blk + oNBN); S.LIVE = (unsigned char*)(blk + oLIVE); S.INP = (unsigned char*)(blk + oINP);
        S.CNT = (int*)(blk + oCNT); S.FAR = (int*)(blk + oFAR); S.FKEY = (unsigned long long*)(blk + oFKEY);
        S.PQ = (int*)(blk + oPQ); S.MARK = (int*)(blk + oMARK); S.VISM = (int*)(blk + oVISM); S.KEY = (int*)(blk + oKEY);
        S.TMPN = (int*)(blk + oTMPN); S.TMP = (int*)(blk + oTMP); S.VL = (int*)(blk + oVL); S.QA = (int*)(blk + oQA);
        S.QB = (int*)(blk + oQB); S.FLG = (int*)(blk + oFLG); S.H1 = (int*)(blk + oH1); S.H2 = (int*)(blk + oH2);
        S.OUTER = (int*)(blk + oOUT); S.AFF = (int*)(blk + oAFF); S.HT = (unsigned long long*)(blk + oHT);
        S.HH = (unsigned long long*)(blk + oHH); S.opt = (const int*)(blk + oOPT); S.oown = (int*)(blk + oOWN);
        S.odist = (double*)(blk + oDST); S.ctrl = (int*)(blk + oCTRL); S.tim = (long long*)(blk + oTIM);
#define QH_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(err, errn, "quickhull: %s: %s", #x, hipGetErrorString(e_)); return PLP_EHIP; } } while (0)
        // upload the state (first round: from the host graph; after a grow: from what the last round downloaded into H)
        QH_TRY(hipMemcpyAsync(S.FN, H.FN.data(), (size_t)nf * d * 8, hipMemcpyHostToDevice, st));
        QH_TRY(hipMemcpyAsync(S.FO, H.FO.data(), (size_t)nf * 8, hipMemcpyHostToDevice, st));
        QH_TRY(hipMemcpyAsync(S.FV, H.FV.data(), (size_t)nf * d * 4, hipMemcpyHostToDevice, st));
        QH_TRY(hipMemcpyAsync(S.NB, H.NB.data(), (size_t)nf * H.capn * 4, hipMemcpyHostToDevice, st));
        QH_TRY(hipMemcpyAsync(S.NBN, H.NBN.data(), (size_t)nf * 4, hipMemcpyHostToDevice, st));
        QH_TRY(hipMemcpyAsync(S.LIVE, H.LIVE.data(), (size_t)nf, hipMemcpyHostToDevice, st));
        QH_TRY(hipMemcpyAsync(S.INP, H.INP.data(), (size_t)nf, hipMemcpyHostToDevice, st));
        QH_TRY(hipMemcpyAsync(S.CNT, H.CNT.data(), (size_t)nf * 4, hipMemcpyHostToDevice, st));
        QH_TRY(hipMemcpyAsync(S.FAR, H.FAR.data(), (size_t)nf * 4, hipMemcpyHostToDevice, st));
        if (!H.PQ.empty()) QH_TRY(hipMemcpyAsync(S.PQ, H.PQ.data(), H.PQ.size() * 4, hipMemcpyHostToDevice, st));
        if (M) {
            QH_TRY(hipMemcpyAsync((void*)S.opt, H.opt.data(), (size_t)M * 4, hipMemcpyHostToDevice, st));
            QH_TRY(hipMemcpyAsync(S.oown, H.oown.data(), (size_t)M * 4, hipMemcpyHostToDevice, st));
            QH_TRY(hipMemcpyAsync(S.odist, H.odist.data(), (size_t)M * 8, hipMemcpyHostToDevice, st));
        }
        QH_TRY(hipMemsetAsync(S.FKEY, 0, (size_t)cap * 8, st));
        QH_TRY(hipMemsetAsync(S.MARK, 0, (size_t)cap * 4, st));
        QH_TRY(hipMemsetAsync(S.VISM, 0, (size_t)cap * 4, st));
        QH_TRY(hipMemsetAsync(S.KEY, 0x7f, (size_t)cap * 4, st));     // 0x7f7f7f7f: above every candidate key
        QH_TRY(hipMemsetAsync(S.TMPN, 0, (size_t)cap * 4, st));
        ctrl[0] = 0;
        ctrl[1] = (int)H.PQ.size();
        ctrl[2] = nf;
        ctrl[3] = QH_OK;
        ctrl[5] = (int)H.total_outside;
        ctrl[6] = 0;
        QH_TRY(hipMemcpyAsync(S.ctrl, ctrl.data(), 64, hipMemcpyHostToDevice, st));
        QH_TRY(hipMemsetAsync(S.tim, 0, 128, st));
        const size_t smem = QH_LDS_BYTES;   // (>= QH_CH * (d + 1) * 8 for d <= 16)
        if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qh_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(qh_tail_kernel, dim3(1), dim3(QT), smem, st, S);
        QH_TRY(hipGetLastError());
        QH_TRY(hipMemcpyAsync(ctrl.data(), S.ctrl, 64, hipMemcpyDeviceToHost, st));
        QH_TRY(hipStreamSynchronize(st));
        const int status = ctrl[3];
#ifdef PLP_QH_DEV_TIMING
        {
            long long tim[16];
            (void)hipMemcpy(tim, S.tim, 128, hipMemcpyDeviceToHost);
            fprintf(stderr, "qh_tail_kernel: %d iterations; us per phase (100 MHz clock): bfs %.0f horizon %.0f facets+lu %.0f table %.0f newlists %.0f outerlists %.0f reassign %.0f far+queue %.0f\n",
                    ctrl[4], tim[0] / 100.0, tim[1] / 100.0, tim[2] / 100.0, tim[3] / 100.0, tim[4] / 100.0, tim[5] / 100.0, tim[6] / 100.0, tim[7] / 100.0);
        }
#endif
        H.iterations += ctrl[4];
        H.facets_made += ctrl[6];
        nf = ctrl[2];
        // what the caller (or the next round) needs
        H.FN.resize((size_t)nf * d); H.FO.resize(nf); H.FV.resize((size_t)nf * d); H.LIVE.resize(nf);
        QH_TRY(hipMemcpyAsync(H.FN.data(), S.FN, (size_t)nf * d * 8, hipMemcpyDeviceToHost, st));
        QH_TRY(hipMemcpyAsync(H.FO.data(), S.FO, (size_t)nf * 8, hipMemcpyDeviceToHost, st));
        QH_TRY(hipMemcpyAsync(H.FV.data(), S.FV, (size_t)nf * d * 4, hipMemcpyDeviceToHost, st));
        QH_TRY(hipMemcpyAsync(H.LIVE.data(), S.LIVE, (size_t)nf, hipMemcpyDeviceToHost, st));
        if (status == QH_GROW) {   // everything else too: the next round uploads it into larger tables
            H.NB.resize((size_t)nf * H.capn); H.NBN.resize(nf); H.INP.resize(nf); H.CNT.resize(nf); H.FAR.resize(nf);
            QH_TRY(hipMemcpyAsync(H.NB.data(), S.NB, (size_t)nf * H.capn * 4, hipMemcpyDeviceToHost, st));
            QH_TRY(hipMemcpyAsync(H.NBN.data(), S.NBN, (size_t)nf * 4, hipMemcpyDeviceToHost, st));
            QH_TRY(hipMemcpyAsync(H.INP.data(), S.INP, (size_t)nf, hipMemcpyDeviceToHost, st));
            QH_TRY(hipMemcpyAsync(H.CNT.data(), S.CNT, (size_t)nf * 4, hipMemcpyDeviceToHost, st));
            QH_TRY(hipMemcpyAsync(H.FAR.data(), S.FAR, (size_t)nf * 4, hipMemcpyDeviceToHost, st));
            const int head = ctrl[0], tail = ctrl[1];
            std::vector<int> pq(tail > head ? tail - head : 0);
            if (!pq.empty()) QH_TRY(hipMemcpyAsync(pq.data(), S.PQ + head, pq.size() * 4, hipMemcpyDeviceToHost, st));
            if (M) {
                QH_TRY(hipMemcpyAsync(H.oown.data(), S.oown, (size_t)M * 4, hipMemcpyDeviceToHost, st));
                QH_TRY(hipMemcpyAsync(H.odist.data(), S.odist, (size_t)M * 8, hipMemcpyDeviceToHost, st));
            }
            QH_TRY(hipStreamSynchronize(st));
            H.PQ.swap(pq);
            H.total_outside = ctrl[5];
            const long long need = (long long)nf + ctrl[7];
            while (cap < 2 * need) cap <<= 1;
            ctrl[4] = 0;
            continue;
        }
        QH_TRY(hipStreamSynchronize(st));
#undef QH_TRY
        if (status == QH_OK) return PLP_OK;
        if (status == QH_SINGULAR) { snprintf(err, errn, "Singular matrix"); return PLP_EINVAL; }
        if (status == QH_IDENT) { snprintf(err, errn, "quickhull: neighbouring facets with identical vertices"); return PLP_EINVAL; }
        snprintf(err, errn, "quickhull: device facet graph: %s", status == QH_NBR_OVERFLOW ? "a facet has more neighbours than the table holds (degenerate input)"
                                                                                                 : "scratch list overflow");
        return PLP_EUNSUPPORTED;
    }
}

}  // namespace plp
