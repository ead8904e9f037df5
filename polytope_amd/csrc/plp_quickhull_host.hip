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
#include <deque>
#include <map>
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

struct Hull {
    int d;
    const double* X;  // [N][d], translated
    std::vector<double> FN, FO;
    std::vector<std::vector<int64_t>> verts;
    std::vector<std::vector<int>> nbrs;
    std::vector<int64_t> cnt, far;
    std::vector<int32_t> fid;
    std::vector<char> live, in_pending;
    std::deque<int> pending;
    dgesv_fn solve;
    std::vector<double> M, rhs, prod;
    std::vector<int> ipiv;

    // unit outward normal and offset of the facet through the d points v[] (reference Facet.__init__, :61-85):
    // solve [V 1; 0 -1] [x; s] = [0; 1], n = x / |x|, offset = -s / |x|
    int hyperplane(const int64_t* v, double* n_out, double* off_out) {
        const int n = d + 1;
        M.assign((size_t)n * n, 0.0);
        rhs.assign(n, 0.0);
        for (int r = 0; r < d; ++r) {
            for (int c = 0; c < d; ++c) M[(size_t)c * n + r] = X[v[r] * d + c];
            M[(size_t)d * n + r] = 1.0;
        }
        M[(size_t)d * n + d] = -1.0;
        rhs[d] = 1.0;
        int info = 0;
        if (solve) {
            int nn = n, one = 1;
            ipiv.resize(n);
            solve(&nn, &one, M.data(), &nn, ipiv.data(), rhs.data(), &nn, &info);
        } else {
            own_solve(n, M.data(), rhs.data(), &info);
        }
        if (info != 0) return 1;  // numpy.linalg.solve raises LinAlgError("Singular matrix")
        prod.resize(d);
        for (int c = 0; c < d; ++c) prod[c] = rhs[c] * rhs[c];
        const double mult = sqrt(np_sum(prod.data(), d));
        for (int c = 0; c < d; ++c) n_out[c] = rhs[c] / mult;
        const double dd = rhs[d] / mult;
        for (int c = 0; c < d; ++c) prod[c] = n_out[c] * X[v[0] * d + c];
        if (np_sum(prod.data(), d) < 0.0)
            for (int c = 0; c < d; ++c) n_out[c] = -n_out[c];
        *off_out = -dd;
        return 0;
    }
    int add_facets(const std::vector<std::vector<int64_t>>& lists, int* s0_out) {
        const int s0 = (int)verts.size();
        FN.resize((size_t)(s0 + lists.size()) * d);
        FO.resize(s0 + lists.size());
        for (size_t k = 0; k < lists.size(); ++k) {
            if (hyperplane(lists[k].data(), &FN[(size_t)(s0 + k) * d], &FO[s0 + k])) return 1;
            verts.push_back(lists[k]);
            nbrs.emplace_back();
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
    std::vector<std::vector<int64_t>> lists;
    for (int i = 0; i <= d; ++i) {
        std::vector<int64_t> v;
        for (int j = 0; j <= d; ++j) if (j != i) v.push_back(simplex[j]);
        lists.push_back(v);
    }
    int s0 = 0;
    if (H.add_facets(lists, &s0)) return bail(PLP_EINVAL, "Singular matrix");
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
    long long iterations = 0;
    // ---- the long tail on the host.  Late iterations move a handful of points each; a device round trip (~50 us)
    // per iteration then costs more than the arithmetic.  Once fewer than HOST_TAIL points are outside the hull the
    // owners and distances are downloaded once and the outside sets continue as host lists, with the device
    // kernel's arithmetic (k-ordered products and sums, first facet with distance > abs_tol, furthest point with
    // the lowest index among equals).
    long long host_tail = 32768;
    if (const char* e = getenv("PLP_QH_HOST_TAIL")) host_tail = atoll(e);
    bool host_mode = false;
    std::vector<std::vector<int64_t>> outside;
    std::vector<double> dist_of;
    long long total_outside = 0;
    for (size_t f = 0; f < H.cnt.size(); ++f) total_outside += H.cnt[f];
    auto to_host = [&]() -> int {
        std::vector<int32_t> owner(N);
        dist_of.resize(N);
        int r = plp_hull_read(sess, owner.data(), dist_of.data());
        if (r) return r;
        int32_t max_id = 0;
        for (int32_t v : H.fid) max_id = v > max_id ? v : max_id;
        std::vector<int> slot_of(max_id + 1, -1);
        for (size_t f = 0; f < H.fid.size(); ++f) if (H.fid[f] >= 0 && H.live[f]) slot_of[H.fid[f]] = (int)f;
        outside.assign(H.verts.size(), std::vector<int64_t>());
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
        outside.resize(H.verts.size());
        std::vector<double> best(k, 0.0);
        for (int j = 0; j < k; ++j) { H.cnt[f0 + j] = 0; H.far[f0 + j] = -1; }
        for (int f : dead_slots) {
            for (int64_t q : outside[f]) {
                for (int j = 0; j < k; ++j) {
                    const double* nn = &H.FN[(size_t)(f0 + j) * d];
                    double sdot = nn[0] * X0[q * d];
                    for (int c = 1; c < d; ++c) sdot = sdot + nn[c] * X0[q * d + c];
                    const double dist = sdot - H.FO[f0 + j];
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
    std::vector<char> in_visible, seen, queued;
    std::vector<int> visible, outer, touched;
    std::vector<double> prod(d);
    for (;;) {
        while (!H.pending.empty() && !H.in_pending[H.pending.front()]) H.pending.pop_front();
        if (H.pending.empty()) break;
        const int facet = H.pending.front();
        const int64_t p = H.far[facet];
        if (!host_mode && total_outside < host_tail) {
            rc = to_host();
            if (rc) { plp_hull_destroy(sess); return rc; }
        }
        // get_furthest() takes the point out of the facet's outside set (:87-102)
        if (host_mode) host_drop(facet, p);
        else {
            rc = plp_hull_drop(sess, 1, &p);
            if (rc) { plp_hull_destroy(sess); return rc; }
        }
        H.cnt[facet] -= 1;
        total_outside -= 1;
        ++iterations;
        // distance of p to a facet, with distance()'s arithmetic (:117-121); evaluated for the facets the search reaches
        const int nf = (int)H.verts.size();
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
        // ---- horizon: one new facet per (visible facet, non-visible neighbour) (:284-304)
        lists.clear();
        outer.clear();
        for (int f1 : visible) {
            const std::vector<int64_t>& v1 = H.verts[f1];
            for (int f2 : H.nbrs[f1]) {
                if (in_visible[f2]) continue;
                const std::vector<int64_t>& v2 = H.verts[f2];
                int skip = -1;
                for (int ii = 0; ii < d; ++ii) {
                    bool found = false;
                    for (int jj = 0; jj < d; ++jj) found = found || (v2[jj] == v1[ii]);
                    if (!found) { skip = ii; break; }
                }
                if (skip < 0) return bail(PLP_EINVAL, "quickhull: neighbouring facets with identical vertices");
                std::vector<int64_t> nv;
                nv.push_back(p);
                for (int ii = 0; ii < d; ++ii) if (ii != skip) nv.push_back(v1[ii]);
                lists.push_back(nv);
                outer.push_back(f2);
            }
        }
        const int k = (int)lists.size();
        if (H.add_facets(lists, &s0)) return bail(PLP_EINVAL, "Singular matrix");
        for (int j = 0; j < k; ++j) { H.nbrs[s0 + j].push_back(outer[j]); H.nbrs[outer[j]].push_back(s0 + j); }
        // ---- links among the new facets: two of them share p and d-2 ridge vertices (:305-310)
        {
            std::map<std::vector<int64_t>, std::vector<int>> by_sub;
            std::vector<int64_t> sub;
            for (int j = 0; j < k; ++j) {
                const std::vector<int64_t>& v = H.verts[s0 + j];   // v[0] = p, v[1..] = ridge
                for (int omit = 0; omit < d - 1; ++omit) {
                    sub.clear();
                    for (int t = 0; t < d - 1; ++t) if (t != omit) sub.push_back(v[1 + t]);
                    std::sort(sub.begin(), sub.end());
                    sub.erase(std::unique(sub.begin(), sub.end()), sub.end());   // (a frozenset)
                    by_sub[sub].push_back(j);
                }
            }
            std::vector<std::vector<int>> links(k);
            for (auto& kv : by_sub) {
                const std::vector<int>& g = kv.second;
                if (g.size() > 1)
                    for (int a : g) for (int c : g) if (a != c) links[a].push_back(c);
            }
            for (int j = 0; j < k; ++j) {
                std::sort(links[j].begin(), links[j].end());
                links[j].erase(std::unique(links[j].begin(), links[j].end()), links[j].end());
                for (int c : links[j]) H.nbrs[s0 + j].push_back(s0 + c);
            }
        }
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
        for (int f1 : visible) {
            for (int f2 : H.nbrs[f1]) {
                std::vector<int>& l = H.nbrs[f2];
                for (size_t t = 0; t < l.size(); ++t) if (l[t] == f1) { l.erase(l.begin() + t); break; }
            }
            H.in_pending[f1] = 0;
            H.live[f1] = 0;
            H.nbrs[f1].clear();
        }
    }
    plp_hull_destroy(sess);
    plp_qh_result* res = new plp_qh_result();
    res->d = d;
    res->iterations = iterations;
    res->facets_made = (long long)H.verts.size();
    for (size_t f = 0; f < H.verts.size(); ++f) {
        if (!H.live[f]) continue;
        res->normals.insert(res->normals.end(), &H.FN[f * d], &H.FN[f * d] + d);
        res->offsets.push_back(H.FO[f]);
        res->verts.insert(res->verts.end(), H.verts[f].begin(), H.verts[f].end());
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
