// plp_quickhull_host.hip -- the main loop of Quickhull (polytope/quickhull.py:224-345) as native host code over the
// device-resident outside sets (plp_hull_*, plp_hull.hip).
//
// The points never leave the device; what the loop itself needs is small -- the facet graph of a few thousand facets:
// pick the first facet with outside points, take its furthest point (one number per facet, returned by the last
// reassignment), search the facets visible from it breadth-first over the neighbour lists (:254-270), form one new
// facet per horizon ridge (:284-304), link the new facets among themselves (:305-310), hand the pooled points to them
// (ONE plp_hull_reassign call) and retire the visible facets (:337-344).  In Python that bookkeeping cost ~100 us per
// new facet and dominated every hull from d = 4 on; here it is a few microseconds per iteration.
//
// Order is the reference's (FIFO of facets with outside points, breadth-first visibility, new facets in (visible
// facet, neighbour) order), and so is the arithmetic of what decides it: distances n.p - d0 summed in numpy's order,
// facet hyperplanes from the reference's (d+1) x (d+1) linear system (:66-85).  The caller may pass LAPACK's dgesv
// (what numpy.linalg.solve runs); with it the rows come out bit-identical to the reference's (fixture g8).  Without,
// an own LU with partial pivoting is used (same rows to ~1e-15).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory_resource>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/plp.h"

namespace {

typedef void (*dgesv_fn)(int* n, int* nrhs, double* a, int* lda, int* ipiv, double* b, int* ldb, int* info);

// numpy's add.reduce over a contiguous run of n doubles (pairwise_sum: plain loop below 8 elements, eight running
// sums above; the run lengths here never reach the 128-element blocking)
double np_sum(const double* a, int n) {
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

void own_solve(int n, double* a /* column major */, double* b, int* info) {
    *info = 0;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(a[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(a[k * n + i]) > best) { best = fabs(a[k * n + i]); piv = i; }
        if (best == 0.0) { *info = k + 1; return; }
        if (piv != k) {
            for (int j = 0; j < n; ++j) { const double t = a[j * n + k]; a[j * n + k] = a[j * n + piv]; a[j * n + piv] = t; }
            const double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        const double inv = 1.0 / a[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double l = a[k * n + i] * inv;
            a[k * n + i] = l;
            for (int j = k + 1; j < n; ++j) a[j * n + i] -= l * a[j * n + k];
            b[i] -= l * b[k];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        for (int j = k + 1; j < n; ++j) b[k] -= a[j * n + k] * b[j];
        b[k] /= a[k * n + k];
    }
}

// A few host threads for the per-facet work of one iteration (k independent hyperplane systems): they sleep on a
// condition variable between iterations and are only woken for iterations with enough new facets to pay for it.
class ParFor {
public:
    explicit ParFor(int nthreads) {
        for (int t = 0; t < nthreads; ++t) th_.emplace_back([this] { worker(); });
    }
    ~ParFor() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int threads() const { return (int)th_.size(); }
    // fn(i) for i in [0, n), in blocks of `grain`; the calling thread takes part; returns when all are done
    template <typename F>
    void run(int n, int grain, const F& fn) {
        if (th_.empty() || n <= grain) { for (int i = 0; i < n; ++i) fn(i); return; }
        {
            // no worker is inside work() while the job is replaced: they enter it only after registering under the mutex
            std::unique_lock<std::mutex> lk(mu_);
            while (active_.load(std::memory_order_acquire) != 0) { lk.unlock(); std::this_thread::yield(); lk.lock(); }
            call_ = [&fn](int i) { fn(i); };
            n_ = n; grain_ = grain;
            next_.store(0, std::memory_order_relaxed);
            pending_.store((n + grain - 1) / grain, std::memory_order_release);
            ++gen_;
        }
        cv_.notify_all();
        work();
        while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    }

private:
    void work() {
        for (;;) {
            const int b = next_.fetch_add(grain_, std::memory_order_relaxed);
            if (b >= n_) return;
            const int e = b + grain_ < n_ ? b + grain_ : n_;
            for (int i = b; i < e; ++i) call_(i);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    void worker() {
        unsigned seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                active_.fetch_add(1, std::memory_order_acq_rel);
            }
            work();
            active_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
    unsigned gen_ = 0;
    std::function<void(int)> call_;
    int n_ = 0, grain_ = 1;
    std::atomic<int> next_{0}, pending_{0}, active_{0};
};

struct Hull {
    int d;
    const double* X;  // [N][d], translated
    std::vector<double> FN, FO;
    std::vector<int64_t> verts;                       // [F][d]
    std::pmr::monotonic_buffer_resource arena;        // neighbour lists: bump allocation, released with the hull
    std::vector<std::pmr::vector<int>> nbrs;
    std::vector<int64_t> cnt, far;
    std::vector<int32_t> fid;
    std::vector<char> live, in_pending;
    std::deque<int> pending;
    dgesv_fn solve;
    ParFor* par = nullptr;
    bool use_lapack = true;   // dgesv for the hyperplanes (small inputs: bit-identical to the reference), else the own LU
    int par_min = 512;        // ... and from which the host threads are woken (PLP_QH_PARMIN)

    const int64_t* vt(int f) const { return &verts[(size_t)f * d]; }
    // unit outward normal and offset of the facet through the d points v[] (reference Facet.__init__, :61-85):
    // solve [V 1; 0 -1] [x; s] = [0; 1], n = x / |x|, offset = -s / |x|.  (Re-entrant: called from several threads.)
    int hyperplane(const int64_t* v, double* n_out, double* off_out, bool lapack) const {
        const int n = d + 1;
        double M[17 * 17], rhs[17], prod[16];
        int ipiv[17];
        for (int i = 0; i < n * n; ++i) M[i] = 0.0;
        for (int i = 0; i < n; ++i) rhs[i] = 0.0;
        for (int r = 0; r < d; ++r) {
            for (int c = 0; c < d; ++c) M[(size_t)c * n + r] = X[v[r] * d + c];
            M[(size_t)d * n + r] = 1.0;
        }
        M[(size_t)d * n + d] = -1.0;
        rhs[d] = 1.0;
        int info = 0;
        if (solve && lapack) {
            int nn = n, one = 1;
            solve(&nn, &one, M, &nn, ipiv, rhs, &nn, &info);
        } else {
            own_solve(n, M, rhs, &info);
        }
        if (info != 0) return 1;  // numpy.linalg.solve raises LinAlgError("Singular matrix")
        for (int c = 0; c < d; ++c) prod[c] = rhs[c] * rhs[c];
        const double mult = sqrt(np_sum(prod, d));
        for (int c = 0; c < d; ++c) n_out[c] = rhs[c] / mult;
        const double dd = rhs[d] / mult;
        for (int c = 0; c < d; ++c) prod[c] = n_out[c] * X[v[0] * d + c];
        if (np_sum(prod, d) < 0.0)
            for (int c = 0; c < d; ++c) n_out[c] = -n_out[c];
        *off_out = -dd;
        return 0;
    }
    // k new facets with the vertex lists nv[k][d]; their hyperplanes are independent: one parallel sweep
    int add_facets(const int64_t* nv, int k, int* s0_out) {
        const int s0 = (int)cnt.size();
        FN.resize((size_t)(s0 + k) * d);
        FO.resize(s0 + k);
        verts.insert(verts.end(), nv, nv + (size_t)k * d);
        // Hulls of fewer than `lapack_below` points: LAPACK's dgesv when the caller handed it in (what numpy.linalg.solve
        // runs: rows bit-identical to the reference's, fixture g8 -- these are the sizes the reference itself can compute).
        // Larger inputs: the own LU with partial pivoting (same hyperplanes to ~1e-15).  dgesv through scipy's OpenBLAS
        // costs 0.25-1.5 us per (d+1) x (d+1) system depending on what its thread pool is doing (measured: the 56 612
        // systems of a 100 000-point hull in d = 5 took 14 ms in one run and 88 ms in the next), concurrent calls
        // serialise inside it (430 000 systems of d = 6: 0.17 s on one thread, 0.56-0.73 s on eight); the own LU takes
        // ~0.15 us and spreads over the host threads when an iteration makes enough facets to pay for waking them.
        // The rule depends on N and k alone, so the rows do not depend on the number of threads.
        const bool lapack = use_lapack;
        std::atomic<int> bad{0};
        auto one = [&](int j) {
            if (hyperplane(&verts[(size_t)(s0 + j) * d], &FN[(size_t)(s0 + j) * d], &FO[s0 + j], lapack))
                bad.store(1, std::memory_order_relaxed);
        };
        if (par && !lapack && k >= par_min) par->run(k, 32, one);
        else for (int j = 0; j < k; ++j) one(j);
        if (bad.load()) return 1;
        for (int j = 0; j < k; ++j) {
            nbrs.emplace_back(&arena);
            nbrs.back().reserve(d + 1);
            cnt.push_back(0);
            far.push_back(-1);
            fid.push_back(-1);
            live.push_back(1);
            in_pending.push_back(0);
        }
        *s0_out = s0;
        return 0;
    }
    void set_pending(int f) {
        if (!in_pending[f]) { in_pending[f] = 1; pending.push_back(f); }
    }
};

// order-independent hash of a set of point indices: sum of mixed ids (a sub-ridge = the ridge minus one vertex)
inline uint64_t mix_id(int64_t v) {
    uint64_t x = (uint64_t)v + 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

thread_local char g_qh_err[256] = "";

}  // namespace

struct plp_qh_result {
    int d;
    std::vector<double> normals, offsets;   // live facets in creation order
    std::vector<int64_t> verts;             // d point indices per live facet
    long long iterations = 0, facets_made = 0;
};

extern "C" {

const char* plp_quickhull_last_error(void) { return g_qh_err; }

int plp_quickhull_run(plp_ctx* ctx, int64_t N, int d, const double* X0, const int64_t* simplex, double abs_tol,
                      void* lapack_dgesv, plp_qh_result** out) {
    if (!ctx || !X0 || !simplex || !out || N < d + 1 || d < 1 || d > 16) {
        snprintf(g_qh_err, sizeof(g_qh_err), "plp_quickhull_run: bad arguments");
        return PLP_EINVAL;
    }
    *out = nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    Hull H;
    H.d = d;
    H.X = X0;
    H.solve = reinterpret_cast<dgesv_fn>(lapack_dgesv);
    plp_hull* sess = nullptr;
    int rc = plp_hull_create(ctx, N, d, X0, &sess);
    if (rc) return rc;
    auto bail = [&](int code, const char* msg) {
        snprintf(g_qh_err, sizeof(g_qh_err), "%s", msg);
        plp_hull_destroy(sess);
        return code;
    };
    // ---- start simplex: facet i omits simplex point i; all facets are neighbours (:215-222)
    int nthreads = 0;
    {
        const unsigned hc = std::thread::hardware_concurrency();
        nthreads = hc > 16 ? 7 : (hc > 2 ? (int)hc / 2 - 1 : 0);
        if (const char* e = getenv("PLP_QH_THREADS")) nthreads = atoi(e) - 1;
        if (nthreads < 0) nthreads = 0;
        if (nthreads > 31) nthreads = 31;
    }
    {
        long long lapack_below = 4096;   // points (PLP_QH_LAPACK_BELOW, A/B)
        if (const char* e = getenv("PLP_QH_LAPACK_BELOW")) lapack_below = atoll(e);
        H.use_lapack = N < lapack_below;
    }
    if (const char* e = getenv("PLP_QH_PARMIN")) H.par_min = atoi(e);
    std::unique_ptr<ParFor> pool;   // (created when the first iteration with many new facets comes along)
    std::vector<int64_t> lists;      // vertex lists of the new facets of one iteration, [k][d]
    for (int i = 0; i <= d; ++i)
        for (int j = 0; j <= d; ++j) if (j != i) lists.push_back(simplex[j]);
    int s0 = 0;
    if (H.add_facets(lists.data(), d + 1, &s0)) return bail(PLP_EINVAL, "Singular matrix");
    for (int i = 0; i <= d; ++i)
        for (int j = i + 1; j <= d; ++j) { H.nbrs[i].push_back(j); H.nbrs[j].push_back(i); }
    std::vector<int64_t> am, cn;
    std::vector<double> mx;
    auto hand_out = [&](const std::vector<int32_t>& dead, int f0, int k) -> int {
        am.resize(k); cn.resize(k); mx.resize(k);
        int32_t id0 = 0;
        int r = plp_hull_reassign(sess, (int)dead.size(), dead.data(), k, &H.FN[(size_t)f0 * d], &H.FO[f0], abs_tol, &id0,
                                  am.data(), mx.data(), cn.data());
        if (r) return r;
        for (int j = 0; j < k; ++j) {
            const int f = f0 + j;
            H.fid[f] = id0 + j;
            H.cnt[f] = cn[j];
            H.far[f] = am[j];
            if (cn[j] > 0) H.set_pending(f);
        }
        return 0;
    };
    rc = plp_hull_drop(sess, d + 1, simplex);   // the simplex' own points are not candidates (:186)
    if (rc) { plp_hull_destroy(sess); return rc; }
    rc = hand_out(std::vector<int32_t>{0}, s0, d + 1);   // facet id 0 owns every point initially
    if (rc) { plp_hull_destroy(sess); return rc; }
    long long iterations = 0, tail_iterations = 0;
    const bool stats = getenv("PLP_QH_STATS") != nullptr;
    if (stats) fprintf(stderr, "plp_quickhull_run: session + start simplex + first assignment %.4f s\n",
                       std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    double t_untimed = 0.0;
    auto t_iter_end = std::chrono::steady_clock::now();
    double t_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0 visible set, 1 horizon lists, 2 hyperplanes, 3 links, 4 hand-out, 5 retire, 6 drop
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto lap = [&](int i, std::chrono::steady_clock::time_point& t0) {
        if (stats) { const auto t1 = now(); t_sec[i] += std::chrono::duration<double>(t1 - t0).count(); t0 = t1; }
    };
    // ---- the long tail on the host.  Late iterations move a handful of points each; a device round trip (~50 us)
    // per iteration then costs more than the arithmetic.  Once fewer than HOST_TAIL points are outside the hull the
    // owners and distances are downloaded once and the outside sets continue as host lists, with the device
    // kernel's arithmetic (products summed in numpy's order, first facet with distance > abs_tol, furthest point with
    // the lowest index among equals).
    long long host_tail = 32768;
    if (const char* e = getenv("PLP_QH_HOST_TAIL")) host_tail = atoll(e);
    bool host_mode = false;
    std::vector<std::vector<int64_t>> outside;
    std::vector<double> dist_of;
    long long total_outside = 0;
    for (size_t f = 0; f < H.cnt.size(); ++f) total_outside += H.cnt[f];
    std::vector<int> hsub_head, hsub_next, link_cnt, link_off, link_fill, link_buf;
    std::vector<uint64_t> hsub_key;
    auto to_host = [&]() -> int {
        std::vector<int32_t> owner(N);
        dist_of.resize(N);
        int r = plp_hull_read(sess, owner.data(), dist_of.data());
        if (r) return r;
        int32_t max_id = 0;
        for (int32_t v : H.fid) max_id = v > max_id ? v : max_id;
        std::vector<int> slot_of(max_id + 1, -1);
        for (size_t f = 0; f < H.fid.size(); ++f) if (H.fid[f] >= 0 && H.live[f]) slot_of[H.fid[f]] = (int)f;
        outside.assign(H.cnt.size(), std::vector<int64_t>());
        for (int64_t q = 0; q < N; ++q) {
            const int32_t o = owner[q];
            if (o > 0 && o <= max_id && slot_of[o] >= 0) outside[slot_of[o]].push_back(q);
        }
        host_mode = true;
        return 0;
    };
    auto host_drop = [&](int f, int64_t q) {
        std::vector<int64_t>& l = outside[f];
        for (size_t t = 0; t < l.size(); ++t) if (l[t] == q) { l[t] = l.back(); l.pop_back(); break; }
    };
    auto host_hand_out = [&](const std::vector<int>& dead_slots, int f0, int k) {
        outside.resize(H.cnt.size());
        std::vector<double> best(k, 0.0);
        for (int j = 0; j < k; ++j) { H.cnt[f0 + j] = 0; H.far[f0 + j] = -1; }
        for (int f : dead_slots) {
            for (int64_t q : outside[f]) {
                for (int j = 0; j < k; ++j) {
                    const double* nn = &H.FN[(size_t)(f0 + j) * d];
                    double prod[16];
                    for (int c = 0; c < d; ++c) prod[c] = nn[c] * X0[q * d + c];
                    const double dist = np_sum(prod, d) - H.FO[f0 + j];   // sum(n*p) - d  (quickhull.py:121), numpy's order
                    if (dist > abs_tol) {
                        outside[f0 + j].push_back(q);
                        dist_of[q] = dist;
                        H.cnt[f0 + j] += 1;
                        if (H.far[f0 + j] < 0 || dist > best[j] || (dist == best[j] && q < H.far[f0 + j])) {
                            best[j] = dist;
                            H.far[f0 + j] = q;
                        }
                        break;
                    }
                }
            }
            outside[f].clear();
        }
        for (int j = 0; j < k; ++j) if (H.cnt[f0 + j] > 0) H.set_pending(f0 + j);
    };
    // (Rounds 3-4 also carried this tail with the facet graph ON THE DEVICE -- one persistent kernel, bitwise the same facets.
    // It never won: an iteration is ~15 dependent phases on ONE compute unit, each paying workgroup barriers and ~1 us per
    // dependent global access, where the host core does the same iteration in 10-170 us.  Measured once more in round 5 on
    // many-facet hulls (points on a sphere, d = 3..7, up to 1.5 M facets: scripts/debug/qh_device_graph.py at commit 3b66ebf)
    // -- 0.95x .. 4.3x the host graph's time -- and removed.  DESIGN.md section 4.4.)
    std::vector<char> in_visible, seen, queued;
    std::vector<int> visible, outer, touched;
    std::vector<double> prod(d);
    for (;;) {
        while (!H.pending.empty() && !H.in_pending[H.pending.front()]) H.pending.pop_front();
        if (H.pending.empty()) break;
        const int facet = H.pending.front();
        const int64_t p = H.far[facet];
        if (!host_mode && total_outside < host_tail) {
            auto th = now();
            rc = to_host();
            if (rc) { plp_hull_destroy(sess); return rc; }
            lap(7, th);
        }
        auto tq = now();
        if (stats) t_untimed += std::chrono::duration<double>(tq - t_iter_end).count();
        // get_furthest() takes the point out of the facet's outside set (:87-102)
        if (host_mode) host_drop(facet, p);
        else {
            rc = plp_hull_drop(sess, 1, &p);
            if (rc) { plp_hull_destroy(sess); return rc; }
        }
        H.cnt[facet] -= 1;
        total_outside -= 1;
        ++iterations;
        lap(6, tq);
        // distance of p to a facet, with distance()'s arithmetic (:117-121); evaluated for the facets the search reaches
        const int nf = (int)H.cnt.size();
        auto is_vis = [&](int f) {
            for (int c = 0; c < d; ++c) prod[c] = H.FN[(size_t)f * d + c] * X0[p * d + c];
            return (np_sum(prod.data(), d) - H.FO[f]) > abs_tol;
        };
        // ---- visible set: breadth-first over neighbours with distance > abs_tol (:254-270)
        if ((int)in_visible.size() < nf) { in_visible.resize(nf, 0); seen.resize(nf, 0); queued.resize(nf, 0); }
        for (int f : touched) { in_visible[f] = 0; seen[f] = 0; queued[f] = 0; }   // undo the marks of the last search
        touched.clear();
        visible.assign(1, facet);
        in_visible[facet] = 1; seen[facet] = 1;
        touched.push_back(facet);
        std::deque<int> queue;
        for (int nb : H.nbrs[facet]) { queue.push_back(nb); queued[nb] = 1; touched.push_back(nb); }
        while (!queue.empty()) {
            const int nb = queue.front();
            queue.pop_front();
            queued[nb] = 0;
            seen[nb] = 1;
            if (is_vis(nb)) {
                visible.push_back(nb);
                in_visible[nb] = 1;
                for (int nn : H.nbrs[nb])
                    if (!seen[nn] && !queued[nn]) { queue.push_back(nn); queued[nn] = 1; touched.push_back(nn); }
            }
        }
        lap(0, tq);
        // ---- horizon: one new facet per (visible facet, non-visible neighbour) (:284-304)
        lists.clear();
        outer.clear();
        for (int f1 : visible) {
            const int64_t* v1 = H.vt(f1);
            for (int f2 : H.nbrs[f1]) {
                if (in_visible[f2]) continue;
                const int64_t* v2 = H.vt(f2);
                int skip = -1;
                for (int ii = 0; ii < d; ++ii) {
                    bool found = false;
                    for (int jj = 0; jj < d; ++jj) found = found || (v2[jj] == v1[ii]);
                    if (!found) { skip = ii; break; }
                }
                if (skip < 0) return bail(PLP_EINVAL, "quickhull: neighbouring facets with identical vertices");
                lists.push_back(p);
                for (int ii = 0; ii < d; ++ii) if (ii != skip) lists.push_back(v1[ii]);
                outer.push_back(f2);
            }
        }
        const int k = (int)outer.size();
        lap(1, tq);
        if (!pool && nthreads > 0 && k >= H.par_min) { pool.reset(new ParFor(nthreads)); H.par = pool.get(); }
        if (H.add_facets(lists.data(), k, &s0)) return bail(PLP_EINVAL, "Singular matrix");
        lap(2, tq);
        for (int j = 0; j < k; ++j) { H.nbrs[s0 + j].push_back(outer[j]); H.nbrs[outer[j]].push_back(s0 + j); }
        // ---- links among the new facets: two of them share p and d-2 ridge vertices (:305-310).  The reference keys a
        // dict by frozenset(sub-ridge); here an order-independent hash of the sub-ridge (sum of mixed ids: the ridge's sum
        // minus the omitted vertex) finds the candidates and the vertex sets are compared before two facets are linked.
        if (d >= 2 && k > 1) {
            const int nsub = k * (d - 1);
            int tsz = 16;
            while (tsz < 2 * nsub) tsz <<= 1;
            hsub_head.assign(tsz, -1);
            hsub_next.resize(nsub);
            hsub_key.resize(nsub);
            link_cnt.assign(k + 1, 0);
            // same sub-ridge?  a = (facet ja, omitted position oa), c likewise; both sets have d - 2 elements
            auto same_sub = [&](int ja, int oa, int jc, int oc) {
                const int64_t* va = H.vt(s0 + ja) + 1;
                const int64_t* vc = H.vt(s0 + jc) + 1;
                for (int t = 0; t < d - 1; ++t) {
                    if (t == oa) continue;
                    bool found = false;
                    for (int u = 0; u < d - 1; ++u) found = found || (u != oc && vc[u] == va[t]);
                    if (!found) return false;
                }
                for (int u = 0; u < d - 1; ++u) {   // (and the other way round: sets, as frozenset compares them)
                    if (u == oc) continue;
                    bool found = false;
                    for (int t = 0; t < d - 1; ++t) found = found || (t != oa && va[t] == vc[u]);
                    if (!found) return false;
                }
                return true;
            };
            link_buf.clear();   // pairs (a, c), both directions
            for (int j = 0; j < k; ++j) {
                const int64_t* v = H.vt(s0 + j) + 1;   // the ridge (v[-1] = p)
                uint64_t hs = 0;
                for (int t = 0; t < d - 1; ++t) hs += mix_id(v[t]);
                for (int omit = 0; omit < d - 1; ++omit) {
                    const int id = j * (d - 1) + omit;
                    const uint64_t h = hs - mix_id(v[omit]);
                    hsub_key[id] = h;
                    const int slot = (int)((h ^ (h >> 32)) & (uint64_t)(tsz - 1));
                    for (int e = hsub_head[slot]; e >= 0; e = hsub_next[e]) {
                        if (hsub_key[e] != h) continue;
                        const int jc = e / (d - 1), oc = e - jc * (d - 1);
                        if (jc == j || !same_sub(j, omit, jc, oc)) continue;
                        link_buf.push_back(j); link_buf.push_back(jc);
                        link_buf.push_back(jc); link_buf.push_back(j);
                        link_cnt[j + 1]++; link_cnt[jc + 1]++;
                    }
                    hsub_next[id] = hsub_head[slot];
                    hsub_head[slot] = id;
                }
            }
            // per facet: its partners ascending, each once (the reference's sorted set of linked facets)
            link_off.assign(k + 1, 0);
            for (int j = 0; j < k; ++j) link_off[j + 1] = link_off[j] + link_cnt[j + 1];
            link_fill.assign(link_off.begin(), link_off.end() - 1);
            std::vector<int>& part = hsub_next;   // (reused as the bucketed partner array)
            part.resize(link_off[k] > nsub ? link_off[k] : nsub);
            for (size_t t = 0; t + 1 < link_buf.size(); t += 2) part[link_fill[link_buf[t]]++] = link_buf[t + 1];
            for (int j = 0; j < k; ++j) {
                int* b0 = part.data() + link_off[j];
                int* b1 = part.data() + link_off[j + 1];
                std::sort(b0, b1);
                b1 = std::unique(b0, b1);
                for (int* q = b0; q < b1; ++q) H.nbrs[s0 + j].push_back(s0 + *q);
            }
        }
        lap(3, tq);
        // ---- hand the pooled points to the new facets, retire the visible ones (:311-344)
        long long pooled = 0;
        std::vector<int32_t> dead;
        std::vector<int> dead_slots;
        for (int f : visible) { pooled += H.cnt[f]; if (H.cnt[f] > 0) { dead.push_back(H.fid[f]); dead_slots.push_back(f); } }
        if (pooled > 0 && k > 0) {
            if (host_mode) host_hand_out(dead_slots, s0, k);
            else {
                rc = hand_out(dead, s0, k);
                if (rc) { plp_hull_destroy(sess); return rc; }
            }
            long long kept = 0;
            for (int j = 0; j < k; ++j) kept += H.cnt[s0 + j];
            total_outside += kept - pooled;   // pooled points that are inside every new facet leave the outside sets
        }
        lap(4, tq);
        for (int f1 : visible) {
            for (int f2 : H.nbrs[f1]) {
                auto& l = H.nbrs[f2];
                for (size_t t = 0; t < l.size(); ++t) if (l[t] == f1) { l.erase(l.begin() + t); break; }
            }
            H.in_pending[f1] = 0;
            H.live[f1] = 0;
            H.nbrs[f1].clear();
        }
        lap(5, tq);
        t_iter_end = tq;
    }
    if (stats)
        fprintf(stderr, "plp_quickhull_run: N=%lld d=%d %lld iterations %zu facets made; seconds: visible %.4f horizon %.4f hyperplanes %.4f "
                "links %.4f hand-out %.4f retire %.4f drop %.4f to-host / device tail %.4f (%lld iterations on the device) between-iterations %.4f; whole call so far %.4f\n", (long long)N, d, iterations, H.cnt.size(), t_sec[0], t_sec[1],
                t_sec[2], t_sec[3], t_sec[4], t_sec[5], t_sec[6], t_sec[7], tail_iterations, t_untimed, std::chrono::duration<double>(now() - t_begin).count());
    plp_hull_destroy(sess);
    plp_qh_result* res = new plp_qh_result();
    res->d = d;
    res->iterations = iterations;
    res->facets_made = (long long)H.cnt.size();
    for (size_t f = 0; f < H.cnt.size(); ++f) {
        if (!H.live[f]) continue;
        res->normals.insert(res->normals.end(), &H.FN[f * d], &H.FN[f * d] + d);
        res->offsets.push_back(H.FO[f]);
        res->verts.insert(res->verts.end(), H.vt((int)f), H.vt((int)f) + d);
    }
    *out = res;
    return PLP_OK;
}

int plp_qh_result_sizes(const plp_qh_result* r, int64_t* n_facets, int64_t* iterations, int64_t* facets_made) {
    if (!r) return PLP_EINVAL;
    if (n_facets) *n_facets = (int64_t)r->offsets.size();
    if (iterations) *iterations = r->iterations;
    if (facets_made) *facets_made = r->facets_made;
    return PLP_OK;
}

int plp_qh_result_copy(const plp_qh_result* r, double* normals, double* offsets, int64_t* verts) {
    if (!r || !normals || !offsets || !verts) return PLP_EINVAL;
    if (!r->offsets.empty()) {
        memcpy(normals, r->normals.data(), r->normals.size() * 8);
        memcpy(offsets, r->offsets.data(), r->offsets.size() * 8);
        memcpy(verts, r->verts.data(), r->verts.size() * 8);
    }
    return PLP_OK;
}

int plp_qh_result_free(plp_qh_result* r) {
    delete r;
    return PLP_OK;
}

}  // extern "C"
