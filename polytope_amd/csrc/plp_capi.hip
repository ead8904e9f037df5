// plp_capi.hip -- extern "C" boundary of libplp_hip.so (declared in include/plp.h).
// Host-pointer entry points stage through a grow-only device scratch arena owned by the
// context, call the *_dev entry point on the context's stream and synchronise.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <algorithm>
#include <initializer_list>
#include <string>
#include <utility>
#include <unordered_map>
#include <vector>

#include "../../include/plp.h"
#include "plp_kernels.hpp"
#include "plp_stage.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(PLP_EHIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

}  // namespace

namespace plp {
size_t verify_scratch_bytes(long long nlp, int m_max);
bool verify_enabled();
int launch_verify_lp(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
                     double* x, double* fun, int* status, void* scratch, int parity, hipStream_t st);
int launch_verify_cheby(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* r,
                        double* xc, int* status, void* scratch, int parity, hipStream_t st);
int launch_verify_box(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* lb, double* ub,
                      int* status, const signed char* basis8, const double* centre, const double* xfin, void* scratch,
                      int parity, hipStream_t st);
}  // namespace plp

struct plp_ctx {
    int device;
    hipStream_t stream;
    char* arena;
    size_t arena_bytes;
    char* pin;  // pinned host mirror of the first SMALL_XFER bytes of the arena (small calls: one copy each way)
    // region_diff search (kept across calls: a pinned allocation costs more than a small search)
    char* rd_pin = nullptr;      // host-mapped block [index block | radii | sequence word]
    char* rd_pin_dev = nullptr;
    double* rd_out = nullptr;    // radii of a batch (device)
    double* rd_tab = nullptr;    // the constraint table A | b (device)
    size_t rd_tab_bytes = 0;
    unsigned long long rd_seq = 0;
    // the search's resident LP server (plp_rdiff.hip: rdiff_server_kernel): host-mapped mailbox / records / results block,
    // its device view, the device-side state words, the sequence number of the last batch
    char* rd_srv = nullptr;
    char* rd_srv_dev = nullptr;
    unsigned long long* rd_srv_state = nullptr;
    unsigned long long rd_srv_seq = 0;    // batches issued so far
    unsigned long long rd_srv_word = 0;   // mailbox word of the last batch that was answered
    unsigned long long rd_srv_init[4] = {0, 0, 0, 0};
    // containment: per-row thresholds of the comparison form (plp_points.hip), a grow-only buffer
    void* mf_buf = nullptr;
    size_t mf_bytes = 0;
    hipEvent_t mf_ev = nullptr;  // recorded after every launch that uses mf_buf: the next user (any stream) waits on it
    bool mf_used = false;
    // device / pinned buffers of the last quickhull session that ended (plp_hull_destroy parks them here, plp_hull_create
    // takes them when they are large enough): hipMalloc / hipFree of five buffers cost more than a 100 000-point hull
    struct HullSpare {
        double* X = nullptr; int32_t* owner = nullptr; double* dist = nullptr; uint8_t* dead = nullptr;
        char* io = nullptr; char* pin = nullptr;
        size_t X_bytes = 0, owner_bytes = 0, dist_bytes = 0, dead_cap = 0, io_bytes = 0;
        bool full = false;
    } hull_spare;
    // large host-pointer batches (plp_stage.hpp): staging threads, pinned staging buffer, copy stream, one event per chunk
    plp::StagePool* pool = nullptr;
    char* stage = nullptr;
    size_t stage_bytes = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t stage_ev[16] = {};
    int stage_nev = 0;
    bool check_finite = false;  // plp_ctx_set_check_finite
    // plp_reduce_counters: device word the fused reduce kernels add their simplex-run count to (lazily allocated; the
    // kernels get nullptr until the first plp_reduce_counters call of the context, and then it costs one atomic per tile)
    unsigned long long* reduce_ctr = nullptr;
    // fused reduce: one word per call in flight (a ring of 64) that the fast kernels raise to the call's number when they
    // hand a polytope to the general kernel, so that its second pass can leave on one load (plp_reduce.hip)
    unsigned long long* retry_ring = nullptr;
    unsigned long long reduce_epoch = 0;
    // plp_assign_dev (few facets): the workgroups' (max, index) partials, one grow-only buffer PER STREAM -- calls on
    // different streams never share one, so nothing has to order them (a handful of streams per context in practice;
    // beyond 16 the table is emptied after a device synchronisation)
    struct StreamBuf { void* p = nullptr; size_t bytes = 0; unsigned calls = 0; };
    std::unordered_map<void*, StreamBuf> as_scratch;
    // the verifier behind the LP / Chebyshev / bounding-box batches (plp_verify.hip): fail list + the careful engine's
    // dictionaries, one grow-only buffer per stream like as_scratch; bounding boxes: the engines' bases and centres
    std::unordered_map<void*, StreamBuf> vf_scratch;
    std::unordered_map<void*, StreamBuf> vf_basis;
};

namespace {

struct Arena {
    plp_ctx* ctx;
    size_t off;
    explicit Arena(plp_ctx* c) : ctx(c), off(0) {}
    template <typename T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(ctx->arena + off);
        off += (count * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

size_t pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

// a grow-only device buffer of this stream (see plp_ctx::as_scratch); nullptr: allocation failed
void* stream_buf(std::unordered_map<void*, plp_ctx::StreamBuf>& table, void* stream, size_t need) {
    if (table.size() >= 16 && !table.count(stream)) {
        (void)hipDeviceSynchronize();
        for (auto& kv : table) if (kv.second.p) (void)hipFree(kv.second.p);
        table.clear();
    }
    plp_ctx::StreamBuf& sb = table[stream];
    if (need > sb.bytes) {
        if (sb.p) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(sb.p); }  // (its last user ran on this stream)
        sb.p = nullptr;
        sb.bytes = 0;
        if (hipMalloc(&sb.p, need + need / 4) == hipSuccess) {
            sb.bytes = need + need / 4;
            sb.calls = 0;
            (void)hipMemsetAsync(sb.p, 0, 256, (hipStream_t)stream);   // (the verifier's list counters start at zero)
        } else {
            (void)hipGetLastError();
        }
    }
    return sb.p;
}

int ensure_arena(plp_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->arena_bytes) return PLP_OK;
    if (ctx->arena) HIP_TRY(hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_bytes = 0;
    size_t want = bytes + bytes / 4;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->arena), want));
    ctx->arena_bytes = want;
    return PLP_OK;
}

// Host-pointer calls whose buffers fit SMALL_XFER move them through the pinned mirror: the inputs are
// gathered into it and cross PCIe as ONE copy, likewise the outputs.  A pageable hipMemcpyAsync of a few
// hundred bytes costs 10-20 us, and the set operations issue hundreds of small batches (region_diff: one
// per search level), so six copies per call were most of such a call.  Large calls copy each array directly.
constexpr size_t SMALL_XFER = 1u << 20;

struct Span {
    void* dev;
    const void* host_in;  // copy_in source (or nullptr)
    void* host_out;       // copy_out destination (or nullptr)
    size_t bytes;
};

int ensure_pin(plp_ctx* ctx) {
    if (ctx->pin) return PLP_OK;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->pin), SMALL_XFER, hipHostMallocDefault));
    return PLP_OK;
}

bool fits_small(plp_ctx* ctx, std::initializer_list<Span> spans) {
    for (const Span& sp : spans) {
        if (!sp.bytes) continue;
        const size_t end = (size_t)(static_cast<char*>(sp.dev) - ctx->arena) + sp.bytes;
        if (end > SMALL_XFER) return false;
    }
    return true;
}

int copy_in(plp_ctx* ctx, hipStream_t st, std::initializer_list<Span> spans) {
    if (fits_small(ctx, spans) && ensure_pin(ctx) == PLP_OK) {
        size_t lo = SMALL_XFER, hi = 0;
        for (const Span& sp : spans) {
            if (!sp.bytes || !sp.host_in) continue;
            const size_t off = (size_t)(static_cast<char*>(sp.dev) - ctx->arena);
            memcpy(ctx->pin + off, sp.host_in, sp.bytes);
            lo = off < lo ? off : lo;
            hi = off + sp.bytes > hi ? off + sp.bytes : hi;
        }
        if (hi > lo) HIP_TRY(hipMemcpyAsync(ctx->arena + lo, ctx->pin + lo, hi - lo, hipMemcpyHostToDevice, st));
        return PLP_OK;
    }
    for (const Span& sp : spans)
        if (sp.bytes && sp.host_in) HIP_TRY(hipMemcpyAsync(sp.dev, sp.host_in, sp.bytes, hipMemcpyHostToDevice, st));
    return PLP_OK;
}

// Large host-pointer batch: the per-unit input arrays go to the device chunk by chunk (plp_stage.hpp) and `launch(lo, hi)`
// enqueues the kernels of units [lo, hi) on `st` behind the arrival of their chunk.  *staged = false: not applicable
// (small batch, PLP_STAGE=0, or a resource could not be had) and nothing was done -- the caller copies as before.
struct StageArray {
    const void* host;
    void* dev;
    size_t unit_bytes;
    bool f64 = false;  // doubles (checked for inf / nan when the context asks for it)
};

const char* const NONFINITE_MSG = "input must not contain values inf, nan, or None";

// plp_ctx_set_check_finite, inputs that are not staged chunk-wise: one pass over each array
int finite_or_fail(plp_ctx* ctx, std::initializer_list<std::pair<const double*, size_t>> arrays) {
    if (!ctx->check_finite) return PLP_OK;
    for (const auto& a : arrays)
        if (a.first && a.second && plp::any_nonfinite_f64(reinterpret_cast<const char*>(a.first), a.second * 8))
            return fail(PLP_ENONFINITE, "%s", NONFINITE_MSG);
    return PLP_OK;
}

// The arrays' device regions (neighbours in the arena, `blk` .. `blk + blk_bytes`) are used as ONE block in which every
// chunk's pieces sit back to back -- the layout of the staging buffer -- so that a chunk crosses PCIe as one copy;
// `launch(lo, hi, ptrs)` gets the device address of each array's rows lo.. (ptrs[i] for arrays[i], NULL where host is).
template <typename F>
int staged_run(plp_ctx* ctx, hipStream_t st, int64_t B, int64_t align, std::initializer_list<StageArray> arrays, char* blk,
               size_t blk_bytes, F launch, bool* staged) {
    *staged = false;
    size_t unit = 0;
    for (const StageArray& a : arrays)
        if (a.host) unit += a.unit_bytes;
    const size_t total = unit * (size_t)B;
    const char* off = getenv("PLP_STAGE");
    if ((off && off[0] == '0') || total < (8u << 20) || B < 4 * align || total > blk_bytes || arrays.size() > 8) return PLP_OK;
    if (!ctx->pool) {
        const char* nt = getenv("PLP_STAGE_THREADS");
        unsigned hw = std::thread::hardware_concurrency();
        int n = nt ? atoi(nt) : (hw >= 16 ? 7 : (hw >= 4 ? (int)hw / 2 - 1 : 1));
        if (n < 1) n = 1;
        if (n > 32) n = 32;
        try {
            ctx->pool = new plp::StagePool(n);
        } catch (...) {  // no threads to be had: the caller copies as before
            ctx->pool = nullptr;
            return PLP_OK;
        }
    }
    if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        ctx->copy_stream = nullptr;
        return PLP_OK;
    }
    while (ctx->stage_nev < 16) {
        if (hipEventCreateWithFlags(&ctx->stage_ev[ctx->stage_nev], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return PLP_OK;
        }
        ++ctx->stage_nev;
    }
    if (total > ctx->stage_bytes) {
        if (ctx->stage) (void)hipHostFree(ctx->stage);
        ctx->stage = nullptr;
        ctx->stage_bytes = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(&ctx->stage), total + total / 4, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return PLP_OK;
        }
        ctx->stage_bytes = total + total / 4;
    }
    int64_t nch = (int64_t)(total / (4u << 20));
    nch = nch < 2 ? 2 : (nch > 16 ? 16 : nch);
    int64_t per = (B + nch - 1) / nch;
    per = (per + align - 1) / align * align;
    nch = (B + per - 1) / per;
    std::vector<std::vector<plp::StagePiece>> chunks;
    try {
        chunks.resize((size_t)nch);
        for (auto& c : chunks) c.reserve(arrays.size());
    } catch (...) {
        return PLP_OK;
    }
    size_t so = 0;  // (unit sizes are multiples of 4 and chunk lengths multiples of `align` >= 16: every piece 8-byte aligned)
    for (int64_t c = 0; c < nch; ++c) {
        const int64_t lo = c * per, hi = lo + per < B ? lo + per : B;
        for (const StageArray& a : arrays) {
            if (!a.host) continue;
            const size_t bytes = (size_t)(hi - lo) * a.unit_bytes;
            chunks[(size_t)c].push_back({static_cast<const char*>(a.host) + (size_t)lo * a.unit_bytes, ctx->stage + so, blk + so,
                                         bytes, a.f64 && ctx->check_finite});
            so += bytes;
        }
    }
    // the copy stream must not overwrite device inputs an earlier call on `st` may still be reading
    HIP_TRY(hipEventRecord(ctx->stage_ev[15], st));
    HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->stage_ev[15], 0));
    const bool timing = getenv("PLP_STAGE_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    ctx->pool->start(chunks);
    *staged = true;
    int rc = PLP_OK;
    for (int64_t c = 0; c < nch && rc == PLP_OK; ++c) {
        ctx->pool->wait((int)c);
        if (timing) fprintf(stderr, "[stage] chunk %d staged at %.0f us\n", (int)c, us());
        if (ctx->pool->nonfinite()) break;  // (set only when the context checks its inputs)
        void* ptrs[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        {
            const std::vector<plp::StagePiece>& pc = chunks[(size_t)c];
            size_t bytes = 0, k = 0, i = 0;
            for (const StageArray& a : arrays) {
                if (a.host) { ptrs[i] = pc[k].dev; bytes += pc[k].bytes; ++k; }
                ++i;
            }
            const hipError_t e = hipMemcpyAsync(pc[0].dev, pc[0].dst, bytes, hipMemcpyHostToDevice, ctx->copy_stream);
            if (e != hipSuccess) rc = fail(PLP_EHIP, "staged upload: %s", hipGetErrorString(e));
        }
        if (rc == PLP_OK && (hipEventRecord(ctx->stage_ev[c % 15], ctx->copy_stream) != hipSuccess ||
                             hipStreamWaitEvent(st, ctx->stage_ev[c % 15], 0) != hipSuccess))
            rc = fail(PLP_EHIP, "staged upload: event");
        const int64_t lo = c * per, hi = lo + per < B ? lo + per : B;
        if (rc == PLP_OK) rc = launch(lo, hi, ptrs);
    }
    ctx->pool->finish();
    if (timing) {
        fprintf(stderr, "[stage] all enqueued at %.0f us\n", us());
        (void)hipStreamSynchronize(ctx->copy_stream);
        fprintf(stderr, "[stage] copies done at %.0f us\n", us());
        (void)hipStreamSynchronize(st);
        fprintf(stderr, "[stage] kernels done at %.0f us (%d chunks, %zu bytes)\n", us(), (int)nch, total);
    }
    if (rc == PLP_OK && ctx->pool->nonfinite()) rc = fail(PLP_ENONFINITE, "%s", NONFINITE_MSG);
    if (rc != PLP_OK) {  // nothing of this call may still be in flight when the caller sees the error
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(st);
    }
    return rc;
}

// copies the outputs to the host and synchronises the stream
int copy_out(plp_ctx* ctx, hipStream_t st, std::initializer_list<Span> spans) {
    // after a staged upload (plp_stage.hpp): the outputs (neighbours in the arena) come back as ONE copy into the pinned
    // staging buffer and the staging threads hand them out -- five pageable D2H copies of a C2 batch cost 1.5 ms
    if (ctx->pool && ctx->stage) {
        size_t lo = ~(size_t)0, hi = 0;
        for (const Span& sp : spans) {
            if (!sp.bytes || !sp.host_out) continue;
            const size_t off = (size_t)(static_cast<char*>(sp.dev) - ctx->arena);
            lo = off < lo ? off : lo;
            hi = off + sp.bytes > hi ? off + sp.bytes : hi;
        }
        if (hi > lo && hi - lo >= SMALL_XFER && hi - lo <= ctx->stage_bytes) {
            HIP_TRY(hipMemcpyAsync(ctx->stage, ctx->arena + lo, hi - lo, hipMemcpyDeviceToHost, st));
            std::vector<std::vector<plp::StagePiece>> one(1);
            for (const Span& sp : spans)
                if (sp.bytes && sp.host_out)
                    one[0].push_back({ctx->stage + ((size_t)(static_cast<char*>(sp.dev) - ctx->arena) - lo),
                                      static_cast<char*>(sp.host_out), nullptr, sp.bytes});
            HIP_TRY(hipStreamSynchronize(st));
            ctx->pool->start(one);
            ctx->pool->wait(0);
            ctx->pool->finish();
            return PLP_OK;
        }
    }
    if (fits_small(ctx, spans) && ensure_pin(ctx) == PLP_OK) {
        size_t lo = SMALL_XFER, hi = 0;
        for (const Span& sp : spans) {
            if (!sp.bytes || !sp.host_out) continue;
            const size_t off = (size_t)(static_cast<char*>(sp.dev) - ctx->arena);
            lo = off < lo ? off : lo;
            hi = off + sp.bytes > hi ? off + sp.bytes : hi;
        }
        if (hi > lo) HIP_TRY(hipMemcpyAsync(ctx->pin + lo, ctx->arena + lo, hi - lo, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (const Span& sp : spans)
            if (sp.bytes && sp.host_out)
                memcpy(sp.host_out, ctx->pin + (size_t)(static_cast<char*>(sp.dev) - ctx->arena), sp.bytes);
        return PLP_OK;
    }
    for (const Span& sp : spans)
        if (sp.bytes && sp.host_out) HIP_TRY(hipMemcpyAsync(sp.host_out, sp.dev, sp.bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PLP_OK;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PLP_EHIP, "%s launch: %s", what, hipGetErrorString(e));
    return PLP_OK;
}

bool is_gfx950(int dev) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

}  // namespace

extern "C" {

int plp_version(void) { return 300; }  // 200: round 2 (LPs beyond 64 rows, plp_region_diff_search, plp_quickhull_run); 210: plp_ctx_set_check_finite, plp_bbox_batch for d <= 16; 300: round 3 (plp_reduce_wide_batch, F2 presolve)

int plp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int k = 0;
    for (int i = 0; i < n; ++i) k += is_gfx950(i) ? 1 : 0;
    return k;
}

const char* plp_last_error(void) { return g_err; }

int plp_ctx_create(int device, plp_ctx** out) {
    if (!out) return fail(PLP_EINVAL, "plp_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(PLP_ENODEVICE, "no HIP device visible");
    }
    if (device < 0 || device >= n) return fail(PLP_EINVAL, "device %d out of range [0,%d)", device, n);
    if (!is_gfx950(device)) return fail(PLP_ENODEVICE, "device %d is not gfx950 (MI355X)", device);
    HIP_TRY(hipSetDevice(device));
    plp_ctx* ctx = new plp_ctx();
    ctx->device = device;
    ctx->arena = nullptr;
    ctx->arena_bytes = 0;
    ctx->pin = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fail(PLP_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return PLP_OK;
}

int plp_ctx_set_check_finite(plp_ctx* ctx, int on) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    ctx->check_finite = on != 0;
    return PLP_OK;
}

int plp_ctx_destroy(plp_ctx* ctx) {
    if (!ctx) return PLP_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->rd_pin) (void)hipHostFree(ctx->rd_pin);
    if (ctx->rd_srv) (void)hipHostFree(ctx->rd_srv);
    if (ctx->rd_srv_state) (void)hipFree(ctx->rd_srv_state);
    if (ctx->rd_out) (void)hipFree(ctx->rd_out);
    if (ctx->rd_tab) (void)hipFree(ctx->rd_tab);
    if (ctx->mf_buf) (void)hipFree(ctx->mf_buf);
    if (ctx->reduce_ctr) (void)hipFree(ctx->reduce_ctr);
    if (ctx->retry_ring) (void)hipFree(ctx->retry_ring);
    for (auto& kv : ctx->as_scratch) if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& kv : ctx->vf_scratch) if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& kv : ctx->vf_basis) if (kv.second.p) (void)hipFree(kv.second.p);
    if (ctx->mf_ev) (void)hipEventDestroy(ctx->mf_ev);
    if (ctx->hull_spare.full) {
        (void)hipFree(ctx->hull_spare.X); (void)hipFree(ctx->hull_spare.owner); (void)hipFree(ctx->hull_spare.dist);
        if (ctx->hull_spare.dead) (void)hipFree(ctx->hull_spare.dead);
        if (ctx->hull_spare.io) (void)hipFree(ctx->hull_spare.io);
        if (ctx->hull_spare.pin) (void)hipHostFree(ctx->hull_spare.pin);
    }
    delete ctx->pool;
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    for (int i = 0; i < ctx->stage_nev; ++i) (void)hipEventDestroy(ctx->stage_ev[i]);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PLP_OK;
}

int plp_ctx_synchronize(plp_ctx* ctx, void* stream) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return PLP_OK;
}

// Every answer of the LP engines passes the verifier before it leaves the library (plp_verify.hip; PLP_VERIFY=0: A/B)
static int verify_lp_answers(plp_ctx* ctx, hipStream_t st, int64_t B, int m_max, int n, const double* c, const double* G,
                             const double* h, const int32_t* m, double* x, double* fun, int32_t* status) {
    if (!plp::verify_enabled()) return PLP_OK;
    void* sc = stream_buf(ctx->vf_scratch, (void*)st, plp::verify_scratch_bytes(B, m_max));
    if (!sc) return fail(PLP_EHIP, "verifier: no device memory for its scratch (%zu bytes)", plp::verify_scratch_bytes(B, m_max));
    if (plp::launch_verify_lp(B, m_max, n, c, G, h, m, x, fun, status, sc, (int)(ctx->vf_scratch[(void*)st].calls++ & 1u), st))
        return fail(PLP_EUNSUPPORTED, "verifier: unsupported size");
    return check_launch("verify_x_kernel");
}
static int verify_cheby_answers(plp_ctx* ctx, hipStream_t st, int64_t B, int m_max, int d, const double* A, const double* b,
                                const int32_t* m, double* r, double* xc, int32_t* status) {
    if (!plp::verify_enabled()) return PLP_OK;
    void* sc = stream_buf(ctx->vf_scratch, (void*)st, plp::verify_scratch_bytes(B, m_max));
    if (!sc) return fail(PLP_EHIP, "verifier: no device memory for its scratch (%zu bytes)", plp::verify_scratch_bytes(B, m_max));
    if (plp::launch_verify_cheby(B, m_max, d, A, b, m, r, xc, status, sc, (int)(ctx->vf_scratch[(void*)st].calls++ & 1u), st))
        return fail(PLP_EUNSUPPORTED, "verifier: unsupported size");
    return check_launch("verify_x_kernel");
}

int plp_verify_counters(plp_ctx* ctx, void* stream, int64_t* careful_lps) {
    if (!ctx || !careful_lps) return fail(PLP_EINVAL, "NULL pointer");
    *careful_lps = 0;
    auto it = ctx->vf_scratch.find(stream);
    if (it == ctx->vf_scratch.end()) {   // (host-pointer calls run on the context's own stream)
        stream = (void*)ctx->stream;
        it = ctx->vf_scratch.find(stream);
    }
    if (it == ctx->vf_scratch.end() || !it->second.p || it->second.calls == 0) return PLP_OK;   // nothing verified on this stream yet
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    // the list counter of the last launch: the two counters (64 B apart) are used in turn, the last call took (calls - 1) & 1
    unsigned cnt = 0;
    const char* base = static_cast<const char*>(it->second.p) + (((it->second.calls - 1) & 1u) ? 64 : 0);
    HIP_TRY(hipMemcpy(&cnt, base, sizeof(cnt), hipMemcpyDeviceToHost));
    *careful_lps = (int64_t)cnt;
    return PLP_OK;
}

// ------------------------------------------------------------------------------- lp_solve
int plp_lp_solve_batch_dev(plp_ctx* ctx, void* stream, int64_t B, int m_max, int n, const double* c,
                           const double* G, const double* h, const int32_t* m, double* x, double* fun,
                           int32_t* status, int32_t* iters) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (B < 0 || m_max < 0 || n < 1) return fail(PLP_EINVAL, "bad sizes B=%lld m_max=%d n=%d", (long long)B, m_max, n);
    if (B == 0) return PLP_OK;
    if (!c || !h || !x || !fun || !status || (!G && m_max > 0)) return fail(PLP_EINVAL, "NULL pointer");
    if (n > plp::MAX_D + 1 || (m_max > plp::MAX_M && !plp::lds_lp_bytes(m_max, n + 1)))
        return fail(PLP_EUNSUPPORTED, "m_max=%d n=%d outside envelope (n<=17; rows: the dictionary must fit 160 KB of LDS)", m_max, n);
    hipStream_t st = (hipStream_t)stream;  // NULL = the HIP default stream
    if (plp::launch_lp(B, m_max, n, c, G, h, m, x, fun, status, iters, st))
        return fail(PLP_EUNSUPPORTED, "lp kernel: unsupported size");
    int rc = check_launch("lp_kernel");
    if (rc) return rc;
    return verify_lp_answers(ctx, st, B, m_max, n, c, G, h, m, x, fun, status);
}

int plp_lp_solve_batch(plp_ctx* ctx, int64_t B, int m_max, int n, const double* c, const double* G,
                       const double* h, const int32_t* m, double* x, double* fun, int32_t* status,
                       int32_t* iters) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (B < 0 || m_max < 0 || n < 1) return fail(PLP_EINVAL, "bad sizes");
    if (B == 0) return PLP_OK;
    if (!c || !h || !x || !fun || !status || (!G && m_max > 0)) return fail(PLP_EINVAL, "NULL pointer");
    if (n > plp::MAX_D + 1 || (m_max > plp::MAX_M && !plp::lds_lp_bytes(m_max, n + 1)))  // before any buffer is read
        return fail(PLP_EUNSUPPORTED, "m_max=%d n=%d outside envelope (n<=17; rows: the dictionary must fit 160 KB of LDS)", m_max, n);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nc = (size_t)B * n, nG = (size_t)B * m_max * n, nh = (size_t)B * m_max;
    size_t need = pad(nc * 8) * 2 + pad(nG * 8) + pad(nh * 8) + pad(B * 8) + pad(B * 4) * 3 + 4096;
    int rc = ensure_arena(ctx, need);
    if (rc) return rc;
    Arena a(ctx);
    double* dc = a.take<double>(nc);
    double* dG = a.take<double>(nG ? nG : 1);
    double* dh = a.take<double>(nh ? nh : 1);
    int32_t* dm = a.take<int32_t>(B);
    double* dx = a.take<double>(nc);
    double* dfun = a.take<double>(B);
    int32_t* dst = a.take<int32_t>(B);
    int32_t* dit = a.take<int32_t>(B);
    hipStream_t st = ctx->stream;
    bool staged = false;  // large batches: chunked upload, kernels of earlier chunks running meanwhile (plp_stage.hpp)
    int more = 0;
    const size_t mn = (size_t)m_max * n;
    rc = staged_run(ctx, st, B, 64,
                    {{c, dc, (size_t)n * 8, true}, {G, dG, mn * 8, true}, {h, dh, (size_t)m_max * 8, true}, {m, dm, 4}},
                    reinterpret_cast<char*>(dc), (size_t)(reinterpret_cast<char*>(dm + B) - reinterpret_cast<char*>(dc)),
                    [&](int64_t lo, int64_t hi, void* const* q) {
                        return plp_lp_solve_batch_dev(ctx, st, hi - lo, m_max, n, static_cast<double*>(q[0]),
                                                      static_cast<double*>(q[1]), static_cast<double*>(q[2]),
                                                      static_cast<int32_t*>(q[3]), dx + (size_t)lo * n, dfun + lo, dst + lo,
                                                      dit + lo);
                    },
                    &staged);
    if (rc) return rc;
    if (!staged) {
        rc = finite_or_fail(ctx, {{c, nc}, {G, nG}, {h, nh}});
        if (rc) return rc;
        rc = copy_in(ctx, st, {{dc, c, nullptr, nc * 8}, {dG, G, nullptr, nG * 8}, {dh, h, nullptr, nh * 8},
                               {dm, m, nullptr, m ? (size_t)B * 4 : 0}});
        if (rc) return rc;
        // small batches: the fast kernels now; the general kernel only if a status, host-visible below anyway, asks for it
        if (B <= 16384 && !plp::verify_enabled()) {
            if (plp::launch_lp_phase(B, m_max, n, dc, dG, dh, m ? dm : nullptr, dx, dfun, dst, dit, st, 1, &more))
                return fail(PLP_EUNSUPPORTED, "lp kernel: unsupported size");
            rc = check_launch("lp_kernel");
        } else {  // (with the verifier behind the engines both passes are launched: it has to see final answers)
            rc = plp_lp_solve_batch_dev(ctx, st, B, m_max, n, dc, dG, dh, m ? dm : nullptr, dx, dfun, dst, dit);
        }
        if (rc) return rc;
    }
    rc = copy_out(ctx, st, {{dx, nullptr, x, nc * 8}, {dfun, nullptr, fun, (size_t)B * 8},
                            {dst, nullptr, status, (size_t)B * 4}, {dit, nullptr, iters, iters ? (size_t)B * 4 : 0}});
    if (rc || !more) return rc;
    bool again = false;
    for (int64_t k = 0; k < B && !again; ++k) again = status[k] == plp::ST_RETRY;
    if (!again) return PLP_OK;
    if (plp::launch_lp_phase(B, m_max, n, dc, dG, dh, m ? dm : nullptr, dx, dfun, dst, dit, st, 2, nullptr))
        return fail(PLP_EUNSUPPORTED, "lp kernel: unsupported size");
    rc = check_launch("lp_kernel");
    if (rc) return rc;
    return copy_out(ctx, st, {{dx, nullptr, x, nc * 8}, {dfun, nullptr, fun, (size_t)B * 8},
                              {dst, nullptr, status, (size_t)B * 4}, {dit, nullptr, iters, iters ? (size_t)B * 4 : 0}});
}

// ------------------------------------------------------------------------------- cheby
int plp_cheby_batch_dev(plp_ctx* ctx, void* stream, int64_t B, int m_max, int d, const double* A,
                        const double* b, const int32_t* m, double* r, double* xc, int32_t* status) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (B < 0 || m_max < 0 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (B == 0) return PLP_OK;
    if (!r || !xc || !status || ((!A || !b) && m_max > 0)) return fail(PLP_EINVAL, "NULL pointer");
    if (d > plp::MAX_D || (m_max > plp::MAX_M && !plp::lds_lp_bytes(m_max, d + 1)))
        return fail(PLP_EUNSUPPORTED, "m_max=%d d=%d outside envelope (d<=16; rows: the dictionary must fit 160 KB of LDS)", m_max, d);
    hipStream_t st = (hipStream_t)stream;  // NULL = the HIP default stream
    if (plp::launch_cheby(B, m_max, d, A, b, m, r, xc, status, st))
        return fail(PLP_EUNSUPPORTED, "cheby kernel: unsupported size");
    int rc = check_launch("cheby_kernel");
    if (rc) return rc;
    return verify_cheby_answers(ctx, st, B, m_max, d, A, b, m, r, xc, status);
}

int plp_cheby_batch(plp_ctx* ctx, int64_t B, int m_max, int d, const double* A, const double* b,
                    const int32_t* m, double* r, double* xc, int32_t* status) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (B < 0 || m_max < 0 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (B == 0) return PLP_OK;
    if (!r || !xc || !status || ((!A || !b) && m_max > 0)) return fail(PLP_EINVAL, "NULL pointer");
    if (d > plp::MAX_D || (m_max > plp::MAX_M && !plp::lds_lp_bytes(m_max, d + 1)))
        return fail(PLP_EUNSUPPORTED, "m_max=%d d=%d outside envelope (d<=16; rows: the dictionary must fit 160 KB of LDS)", m_max, d);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nA = (size_t)B * m_max * d, nb = (size_t)B * m_max, nx = (size_t)B * d;
    int rc = ensure_arena(ctx, pad(nA * 8) + pad(nb * 8) + pad(nx * 8) + pad(B * 8) + pad(B * 4) * 2 + 4096);
    if (rc) return rc;
    Arena a(ctx);
    double* dA = a.take<double>(nA ? nA : 1);
    double* db = a.take<double>(nb ? nb : 1);
    int32_t* dm = a.take<int32_t>(B);
    double* dr = a.take<double>(B);
    double* dxc = a.take<double>(nx);
    int32_t* dst = a.take<int32_t>(B);
    hipStream_t st = ctx->stream;
    bool staged = false;  // large batches: chunked upload, kernels of earlier chunks running meanwhile (plp_stage.hpp)
    const size_t md = (size_t)m_max * d;
    rc = staged_run(ctx, st, B, 64, {{A, dA, md * 8, true}, {b, db, (size_t)m_max * 8, true}, {m, dm, 4}},
                    reinterpret_cast<char*>(dA), (size_t)(reinterpret_cast<char*>(dm + B) - reinterpret_cast<char*>(dA)),
                    [&](int64_t lo, int64_t hi, void* const* q) {
                        return plp_cheby_batch_dev(ctx, st, hi - lo, m_max, d, static_cast<double*>(q[0]),
                                                   static_cast<double*>(q[1]), static_cast<int32_t*>(q[2]), dr + lo,
                                                   dxc + (size_t)lo * d, dst + lo);
                    },
                    &staged);
    if (rc) return rc;
    if (!staged) {
        rc = finite_or_fail(ctx, {{A, nA}, {b, nb}});
        if (rc) return rc;
        rc = copy_in(ctx, st, {{dA, A, nullptr, nA * 8}, {db, b, nullptr, nb * 8}, {dm, m, nullptr, m ? (size_t)B * 4 : 0}});
        if (rc) return rc;
        rc = plp_cheby_batch_dev(ctx, st, B, m_max, d, dA, db, m ? dm : nullptr, dr, dxc, dst);
        if (rc) return rc;
    }
    return copy_out(ctx, st, {{dr, nullptr, r, (size_t)B * 8}, {dxc, nullptr, xc, nx * 8},
                              {dst, nullptr, status, (size_t)B * 4}});
}

// ------------------------------------------------------------------------------- bounding box
int plp_bbox_batch_dev(plp_ctx* ctx, void* stream, int64_t B, int m_max, int d, const double* A, const double* b,
                       const int32_t* m, double* lb, double* ub, int32_t* status) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (B < 0 || m_max < 0 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (B == 0) return PLP_OK;
    if (!lb || !ub || !status || ((!A || !b) && m_max > 0)) return fail(PLP_EINVAL, "NULL pointer");
    if (m_max > plp::MAX_M || d > plp::MAX_D)
        return fail(PLP_EUNSUPPORTED, "m_max=%d d=%d outside envelope (m<=64, d<=16)", m_max, d);
    if (m_max < 1) return fail(PLP_EUNSUPPORTED, "bbox kernel: m_max=%d (needs m_max >= 1)", m_max);
    hipStream_t st = (hipStream_t)stream;  // NULL = the HIP default stream
    // the verifier behind the fused kernels (plp_verify.hip): they hand over each box LP's final basis + the centre, or the
    // point the LP ended on; one buffer per stream: [bases 2 d d bytes | centres d doubles | points 2 d d doubles] per polytope
    plp::BoxHandover ho{nullptr, nullptr, nullptr, 0};
    void* vsc = nullptr;
    if (plp::verify_enabled()) {
        const size_t nb8 = ((size_t)B * 2 * d * d + 255) & ~(size_t)255, nct = (size_t)B * d * 8;
        const bool lane_form = d <= 3 && m_max <= 32;   // (launch_bbox: the shapes bbox_lane_kernel may take)
        const size_t nxf = lane_form ? (size_t)B * 2 * d * d * 8 : 0;
        char* hb = static_cast<char*>(stream_buf(ctx->vf_basis, stream, nb8 + nct + nxf + 256));
        vsc = stream_buf(ctx->vf_scratch, stream, plp::verify_scratch_bytes(B * 2 * d, m_max));
        if (!hb || !vsc) return fail(PLP_EHIP, "verifier: no device memory for its scratch");
        ho.basis8 = reinterpret_cast<signed char*>(hb);
        ho.centre = reinterpret_cast<double*>(hb + nb8);
        ho.xfin = lane_form ? reinterpret_cast<double*>(hb + nb8 + nct) : nullptr;
    }
    if (plp::launch_bbox(B, m_max, d, A, b, m, lb, ub, status, st, &ho))
        return fail(PLP_EUNSUPPORTED, "bbox kernel: unsupported size");
    int rc = check_launch("bbox_r_kernel");
    if (rc || !ho.mode) return rc;
    if (plp::launch_verify_box(B, m_max, d, A, b, m, lb, ub, status, ho.mode == 1 ? ho.basis8 : nullptr,
                               ho.mode == 1 ? ho.centre : nullptr, ho.mode == 2 ? ho.xfin : nullptr, vsc,
                               (int)(ctx->vf_scratch[stream].calls++ & 1u), st))
        return fail(PLP_EUNSUPPORTED, "verifier: unsupported size");
    return check_launch("verify_box_kernel");
}

int plp_bbox_batch(plp_ctx* ctx, int64_t B, int m_max, int d, const double* A, const double* b, const int32_t* m,
                   double* lb, double* ub, int32_t* status) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (B < 0 || m_max < 0 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (B == 0) return PLP_OK;
    if (!lb || !ub || !status || ((!A || !b) && m_max > 0)) return fail(PLP_EINVAL, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nA = (size_t)B * m_max * d, nb = (size_t)B * m_max, nx = (size_t)B * d;
    int rc = ensure_arena(ctx, pad(nA * 8) + pad(nb * 8) + pad(nx * 8) * 2 + pad(B * 4) * 2 + 4096);
    if (rc) return rc;
    Arena a(ctx);
    double* dA = a.take<double>(nA ? nA : 1);
    double* db = a.take<double>(nb ? nb : 1);
    int32_t* dm = a.take<int32_t>(B);
    double* dlb = a.take<double>(nx);
    double* dub = a.take<double>(nx);
    int32_t* dst = a.take<int32_t>(B);
    hipStream_t st = ctx->stream;
    bool staged = false;  // large batches: chunked upload, kernels of earlier chunks running meanwhile (plp_stage.hpp)
    const size_t md = (size_t)m_max * d;
    rc = staged_run(ctx, st, B, 64, {{A, dA, md * 8, true}, {b, db, (size_t)m_max * 8, true}, {m, dm, 4}},
                    reinterpret_cast<char*>(dA), (size_t)(reinterpret_cast<char*>(dm + B) - reinterpret_cast<char*>(dA)),
                    [&](int64_t lo, int64_t hi, void* const* q) {
                        return plp_bbox_batch_dev(ctx, st, hi - lo, m_max, d, static_cast<double*>(q[0]),
                                                  static_cast<double*>(q[1]), static_cast<int32_t*>(q[2]),
                                                  dlb + (size_t)lo * d, dub + (size_t)lo * d, dst + lo);
                    },
                    &staged);
    if (rc) return rc;
    if (!staged) {
        rc = finite_or_fail(ctx, {{A, nA}, {b, nb}});
        if (rc) return rc;
        rc = copy_in(ctx, st, {{dA, A, nullptr, nA * 8}, {db, b, nullptr, nb * 8}, {dm, m, nullptr, m ? (size_t)B * 4 : 0}});
        if (rc) return rc;
        rc = plp_bbox_batch_dev(ctx, st, B, m_max, d, dA, db, m ? dm : nullptr, dlb, dub, dst);
        if (rc) return rc;
    }
    return copy_out(ctx, st, {{dlb, nullptr, lb, nx * 8}, {dub, nullptr, ub, nx * 8}, {dst, nullptr, status, (size_t)B * 4}});
}

// ------------------------------------------------------------------------------- reduce
int plp_reduce_batch_dev(plp_ctx* ctx, void* stream, int64_t B, int m_max, int d, const double* A,
                         const double* b, const int32_t* m, double abs_tol, uint64_t* keep, int32_t* flags,
                         double* r, double* xc, int32_t* nlp) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (B < 0 || m_max < 0 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (B == 0) return PLP_OK;
    if (!keep || !flags || !r || !xc || !nlp || ((!A || !b) && m_max > 0)) return fail(PLP_EINVAL, "NULL pointer");
    if (m_max > plp::MAX_M || d > plp::MAX_D)
        return fail(PLP_EUNSUPPORTED, "m_max=%d d=%d outside envelope (m<=64, d<=16)", m_max, d);
    hipStream_t st = (hipStream_t)stream;  // NULL = the HIP default stream
    plp::t_reduce_ctr = ctx->reduce_ctr;
    if (!ctx->retry_ring) {
        if (hipMalloc(reinterpret_cast<void**>(&ctx->retry_ring), 64 * 8) == hipSuccess) {
            if (hipMemset(ctx->retry_ring, 0, 64 * 8) != hipSuccess) { (void)hipFree(ctx->retry_ring); ctx->retry_ring = nullptr; }
        } else {
            ctx->retry_ring = nullptr;
        }
        (void)hipGetLastError();
    }
    ctx->reduce_epoch += 1;
    plp::t_reduce_retry = ctx->retry_ring ? ctx->retry_ring + (ctx->reduce_epoch & 63ull) : nullptr;
    plp::t_reduce_epoch = ctx->reduce_epoch;
    const int lrc = plp::launch_reduce(B, m_max, d, A, b, m, abs_tol, reinterpret_cast<unsigned long long*>(keep), flags, r,
                                       xc, nlp, st);
    plp::t_reduce_ctr = nullptr;
    plp::t_reduce_retry = nullptr;
    if (lrc) return fail(PLP_EUNSUPPORTED, "reduce kernel: unsupported size");
    return check_launch("reduce_kernel");
}

int plp_reduce_counters(plp_ctx* ctx, void* stream, uint64_t* simplex_runs, int reset) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    constexpr size_t CTR_BYTES = (size_t)plp::PLP_CTR_SLOTS * 64;
    if (!ctx->reduce_ctr) {   // first call: from now on the kernels launched through this context count
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->reduce_ctr), CTR_BYTES));
        HIP_TRY(hipMemsetAsync(ctx->reduce_ctr, 0, CTR_BYTES, st));
    }
    std::vector<unsigned long long> host(CTR_BYTES / 8);
    HIP_TRY(hipMemcpyAsync(host.data(), ctx->reduce_ctr, CTR_BYTES, hipMemcpyDeviceToHost, st));
    if (reset) HIP_TRY(hipMemsetAsync(ctx->reduce_ctr, 0, CTR_BYTES, st));
    HIP_TRY(hipStreamSynchronize(st));
    unsigned long long v = 0ull;   // (two's-complement partial sums: the total is what counts)
    for (int k = 0; k < plp::PLP_CTR_SLOTS; ++k) v += host[(size_t)k * 8];
    if (simplex_runs) *simplex_runs = (uint64_t)v;
    return PLP_OK;
}

int plp_reduce_batch(plp_ctx* ctx, int64_t B, int m_max, int d, const double* A, const double* b,
                     const int32_t* m, double abs_tol, uint64_t* keep, int32_t* flags, double* r, double* xc,
                     int32_t* nlp) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (B < 0 || m_max < 0 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (B == 0) return PLP_OK;
    if (!keep || !flags || !r || !xc || !nlp || ((!A || !b) && m_max > 0)) return fail(PLP_EINVAL, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nA = (size_t)B * m_max * d, nb = (size_t)B * m_max, nx = (size_t)B * d;
    int rc = ensure_arena(ctx, pad(nA * 8) + pad(nb * 8) + pad(nx * 8) + pad(B * 8) * 2 + pad(B * 4) * 3 + 4096);
    if (rc) return rc;
    Arena a(ctx);
    double* dA = a.take<double>(nA ? nA : 1);
    double* db = a.take<double>(nb ? nb : 1);
    int32_t* dm = a.take<int32_t>(B);
    uint64_t* dkeep = a.take<uint64_t>(B);
    int32_t* dfl = a.take<int32_t>(B);
    double* dr = a.take<double>(B);
    double* dxc = a.take<double>(nx);
    int32_t* dnlp = a.take<int32_t>(B);
    hipStream_t st = ctx->stream;
    // large batches: chunked upload with the kernels of earlier chunks running meanwhile (tiles hold 16 polytopes)
    bool staged = false;
    const size_t md = (size_t)m_max * d;
    rc = staged_run(ctx, st, B, 16,
                    {{A, dA, md * 8, true}, {b, db, (size_t)m_max * 8, true}, {m, dm, 4}},
                    reinterpret_cast<char*>(dA), (size_t)(reinterpret_cast<char*>(dm + B) - reinterpret_cast<char*>(dA)),
                    [&](int64_t lo, int64_t hi, void* const* q) {
                        return plp_reduce_batch_dev(ctx, st, hi - lo, m_max, d, static_cast<double*>(q[0]),
                                                    static_cast<double*>(q[1]), static_cast<int32_t*>(q[2]), abs_tol,
                                                    dkeep + lo, dfl + lo, dr + lo, dxc + (size_t)lo * d, dnlp + lo);
                    },
                    &staged);
    if (rc) return rc;
    bool two_step = false;
    if (!staged) {
        rc = finite_or_fail(ctx, {{A, nA}, {b, nb}});
        if (rc) return rc;
        rc = copy_in(ctx, st, {{dA, A, nullptr, nA * 8}, {db, b, nullptr, nb * 8}, {dm, m, nullptr, m ? (size_t)B * 4 : 0}});
        if (rc) return rc;
        // small batches: only the first launch now; the pass that redoes polytopes flagged for the general engine runs
        // when the flags, host-visible below anyway, ask for it (it is a launch that normally finds nothing to do)
        two_step = B <= 16384 && m_max <= plp::MAX_M && d <= plp::MAX_D;
        if (two_step) {
            if (plp::launch_reduce_phase(B, m_max, d, dA, db, m ? dm : nullptr, abs_tol,
                                         reinterpret_cast<unsigned long long*>(dkeep), dfl, dr, dxc, dnlp, st, 1))
                return fail(PLP_EUNSUPPORTED, "reduce kernel: unsupported size");
            rc = check_launch("reduce_kernel");
        } else {
            rc = plp_reduce_batch_dev(ctx, st, B, m_max, d, dA, db, m ? dm : nullptr, abs_tol, dkeep, dfl, dr, dxc, dnlp);
        }
        if (rc) return rc;
    }
    rc = copy_out(ctx, st, {{dkeep, nullptr, keep, (size_t)B * 8}, {dfl, nullptr, flags, (size_t)B * 4},
                            {dr, nullptr, r, (size_t)B * 8}, {dxc, nullptr, xc, nx * 8},
                            {dnlp, nullptr, nlp, (size_t)B * 4}});
    if (rc || !two_step) return rc;
    bool again = false;
    for (int64_t k = 0; k < B && !again; ++k) again = (flags[k] & plp::RF_RETRY) != 0;
    if (!again) return PLP_OK;
    if (plp::launch_reduce_phase(B, m_max, d, dA, db, m ? dm : nullptr, abs_tol, reinterpret_cast<unsigned long long*>(dkeep),
                                 dfl, dr, dxc, dnlp, st, 2))
        return fail(PLP_EUNSUPPORTED, "reduce kernel: unsupported size");
    rc = check_launch("reduce_kernel");
    if (rc) return rc;
    return copy_out(ctx, st, {{dkeep, nullptr, keep, (size_t)B * 8}, {dfl, nullptr, flags, (size_t)B * 4},
                              {dr, nullptr, r, (size_t)B * 8}, {dxc, nullptr, xc, nx * 8},
                              {dnlp, nullptr, nlp, (size_t)B * 4}});
}

// ------------------------------------------------------------------------------- reduce beyond 64 rows
int plp_reduce_wide_batch_dev(plp_ctx* ctx, void* stream, int64_t B, int m_max, int d, const double* A,
                              const double* b, const int32_t* m, double abs_tol, uint64_t* keep, int32_t* flags,
                              double* r, double* xc, int32_t* nlp) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (B < 0 || m_max < 1 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (B == 0) return PLP_OK;
    if (!keep || !flags || !r || !xc || !nlp || !A || !b) return fail(PLP_EINVAL, "NULL pointer");
    if (d > plp::MAX_D) return fail(PLP_EUNSUPPORTED, "d=%d > 16", d);
    hipStream_t st = (hipStream_t)stream;  // NULL = the HIP default stream
    if (m_max <= plp::MAX_M)  // one word per polytope: the register-resident kernels
        return plp_reduce_batch_dev(ctx, stream, B, m_max, d, A, b, m, abs_tol, keep, flags, r, xc, nlp);
    if (plp::launch_reduce_lds(B, m_max, d, A, b, m, abs_tol, reinterpret_cast<unsigned long long*>(keep), flags, r, xc,
                               nlp, st))
        return fail(PLP_EUNSUPPORTED, "reduce: a polytope of %d rows in dimension %d does not fit the LDS of a CU", m_max, d);
    return check_launch("reduce_lds_kernel");
}

int plp_reduce_wide_batch(plp_ctx* ctx, int64_t B, int m_max, int d, const double* A, const double* b,
                          const int32_t* m, double abs_tol, uint64_t* keep, int32_t* flags, double* r, double* xc,
                          int32_t* nlp) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (B < 0 || m_max < 1 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (B == 0) return PLP_OK;
    if (!keep || !flags || !r || !xc || !nlp || !A || !b) return fail(PLP_EINVAL, "NULL pointer");
    if (m_max <= plp::MAX_M) return plp_reduce_batch(ctx, B, m_max, d, A, b, m, abs_tol, keep, flags, r, xc, nlp);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t W = (size_t)(m_max + 63) / 64;
    const size_t nA = (size_t)B * m_max * d, nb = (size_t)B * m_max, nx = (size_t)B * d;
    int rc = ensure_arena(ctx, pad(nA * 8) + pad(nb * 8) + pad(nx * 8) + pad(B * W * 8) + pad(B * 8) + pad(B * 4) * 3 + 4096);
    if (rc) return rc;
    rc = finite_or_fail(ctx, {{A, nA}, {b, nb}});
    if (rc) return rc;
    Arena a(ctx);
    double* dA = a.take<double>(nA);
    double* db = a.take<double>(nb);
    int32_t* dm = a.take<int32_t>(B);
    uint64_t* dkeep = a.take<uint64_t>(B * W);
    int32_t* dfl = a.take<int32_t>(B);
    double* dr = a.take<double>(B);
    double* dxc = a.take<double>(nx);
    int32_t* dnlp = a.take<int32_t>(B);
    hipStream_t st = ctx->stream;
    rc = copy_in(ctx, st, {{dA, A, nullptr, nA * 8}, {db, b, nullptr, nb * 8}, {dm, m, nullptr, m ? (size_t)B * 4 : 0}});
    if (rc) return rc;
    rc = plp_reduce_wide_batch_dev(ctx, st, B, m_max, d, dA, db, m ? dm : nullptr, abs_tol, dkeep, dfl, dr, dxc, dnlp);
    if (rc) return rc;
    return copy_out(ctx, st, {{dkeep, nullptr, keep, (size_t)B * W * 8}, {dfl, nullptr, flags, (size_t)B * 4},
                              {dr, nullptr, r, (size_t)B * 8}, {dxc, nullptr, xc, nx * 8},
                              {dnlp, nullptr, nlp, (size_t)B * 4}});
}

// ------------------------------------------------------------------------------- contains
int plp_contains_dev(plp_ctx* ctx, void* stream, int P, int m_max, int d, const double* A, const double* b,
                     const int32_t* m, int64_t N, const double* X, double abs_tol, int mode, uint8_t* out) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (P < 0 || m_max < 0 || d < 1 || N < 0 || (mode != 0 && mode != 1)) return fail(PLP_EINVAL, "bad sizes/mode");
    if (N == 0) return PLP_OK;
    if (!X || !out || ((!A || !b) && P > 0 && m_max > 0)) return fail(PLP_EINVAL, "NULL pointer");
    if (d > plp::MAX_D) return fail(PLP_EUNSUPPORTED, "d=%d > 16", d);
    hipStream_t st = (hipStream_t)stream;  // NULL = the HIP default stream
    // per-row thresholds: a grow-only buffer of the context, handed from stream to stream in order
    void* scratch = nullptr;
    if (P > 0 && m_max > 0) {
        const size_t need = plp::contains_scratch_bytes(P, m_max);
        if (!ctx->mf_ev && hipEventCreateWithFlags(&ctx->mf_ev, hipEventDisableTiming) != hipSuccess) {
            ctx->mf_ev = nullptr;
            (void)hipGetLastError();
        }
        if (need > ctx->mf_bytes) {
            // growing: the last launch that used the old buffer must have finished before it is freed (an event of
            // the context, not the caller's stream handle, which may no longer exist)
            if (ctx->mf_buf) {
                if (ctx->mf_used && ctx->mf_ev) (void)hipEventSynchronize(ctx->mf_ev);
                else if (ctx->mf_used) (void)hipDeviceSynchronize();
                (void)hipFree(ctx->mf_buf);
            }
            ctx->mf_buf = nullptr;
            ctx->mf_bytes = 0;
            ctx->mf_used = false;
            if (hipMalloc(&ctx->mf_buf, need + need / 4) == hipSuccess) ctx->mf_bytes = need + need / 4;
            else (void)hipGetLastError();
        } else if (ctx->mf_used) {
            // the previous user of the buffer may have run on another stream: order the streams on the device, the host
            // does not wait (a no-op when it is the same stream)
            if (ctx->mf_ev) HIP_TRY(hipStreamWaitEvent(st, ctx->mf_ev, 0));
            else (void)hipDeviceSynchronize();
        }
        if (ctx->mf_buf) scratch = ctx->mf_buf;
    }
    if (plp::launch_contains(P, m_max, d, A, b, m, N, X, abs_tol, mode, out, scratch, st))
        return fail(PLP_EUNSUPPORTED, "contains kernel: unsupported size");
    int rc = check_launch("contains_kernel");
    if (scratch && rc == PLP_OK) {
        ctx->mf_used = true;
        if (ctx->mf_ev) HIP_TRY(hipEventRecord(ctx->mf_ev, st));
    }
    return rc;
}

int plp_contains(plp_ctx* ctx, int P, int m_max, int d, const double* A, const double* b, const int32_t* m,
                 int64_t N, const double* X, double abs_tol, int mode, uint8_t* out) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (P < 0 || m_max < 0 || d < 1 || N < 0 || (mode != 0 && mode != 1)) return fail(PLP_EINVAL, "bad sizes/mode");
    if (N == 0) return PLP_OK;
    if (!X || !out || ((!A || !b) && P > 0 && m_max > 0)) return fail(PLP_EINVAL, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nA = (size_t)P * m_max * d, nb = (size_t)P * m_max, nX = (size_t)N * d;
    const size_t nout = mode == 1 ? (size_t)P * N : (size_t)N;
    int rc = ensure_arena(ctx, pad(nA * 8) + pad(nb * 8) + pad(nX * 8) + pad(nout) + pad((size_t)P * 4) + 4096);
    if (rc) return rc;
    Arena a(ctx);
    double* dA = a.take<double>(nA ? nA : 1);
    double* db = a.take<double>(nb ? nb : 1);
    int32_t* dm = a.take<int32_t>(P ? P : 1);
    double* dX = a.take<double>(nX);
    uint8_t* dout = a.take<uint8_t>(nout ? nout : 1);
    hipStream_t st = ctx->stream;
    if (nA) HIP_TRY(hipMemcpyAsync(dA, A, nA * 8, hipMemcpyHostToDevice, st));
    if (nb) HIP_TRY(hipMemcpyAsync(db, b, nb * 8, hipMemcpyHostToDevice, st));
    if (m && P) HIP_TRY(hipMemcpyAsync(dm, m, (size_t)P * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dX, X, nX * 8, hipMemcpyHostToDevice, st));
    rc = plp_contains_dev(ctx, st, P, m_max, d, dA, db, m ? dm : nullptr, N, dX, abs_tol, mode, dout);
    if (rc) return rc;
    if (nout) HIP_TRY(hipMemcpyAsync(out, dout, nout, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PLP_OK;
}

// ------------------------------------------------------------------------------- assign
int plp_assign_dev(plp_ctx* ctx, void* stream, int64_t N, int d, const double* X, int F, const double* normals,
                   const double* offsets, double abs_tol, int32_t* fop, double* dist, int64_t* argmax,
                   double* maxd) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (N < 0 || d < 1 || F < 1 || !(abs_tol >= 0.0)) return fail(PLP_EINVAL, "bad sizes (need F>=1, abs_tol>=0)");
    if (!normals || !offsets || !argmax || !maxd || (N > 0 && (!X || !fop || !dist)))
        return fail(PLP_EINVAL, "NULL pointer");
    if (d > plp::MAX_D) return fail(PLP_EUNSUPPORTED, "d=%d > 16", d);
    hipStream_t st = (hipStream_t)stream;  // NULL = the HIP default stream
    void* scratch = nullptr;
    const size_t need = plp::assign_scratch_bytes(N, F);
    if (need) {
        if (ctx->as_scratch.size() >= 16 && !ctx->as_scratch.count(stream)) {
            (void)hipDeviceSynchronize();
            for (auto& kv : ctx->as_scratch) if (kv.second.p) (void)hipFree(kv.second.p);
            ctx->as_scratch.clear();
        }
        plp_ctx::StreamBuf& sb = ctx->as_scratch[stream];
        if (need > sb.bytes) {
            if (sb.p) { (void)hipStreamSynchronize(st); (void)hipFree(sb.p); }   // (its last user ran on this stream)
            sb.p = nullptr;
            sb.bytes = 0;
            if (hipMalloc(&sb.p, need + need / 2) == hipSuccess) sb.bytes = need + need / 2;
            else (void)hipGetLastError();   // no scratch: the general kernel takes the call
        }
        scratch = sb.p;
    }
    if (plp::launch_assign(N, d, X, F, normals, offsets, abs_tol, fop, dist, reinterpret_cast<long long*>(argmax),
                           maxd, scratch, scratch ? ctx->as_scratch[stream].bytes : 0, st))
        return fail(PLP_EUNSUPPORTED, "assign kernel: unsupported size");
    return check_launch("assign_kernel");
}

int plp_assign(plp_ctx* ctx, int64_t N, int d, const double* X, int F, const double* normals,
               const double* offsets, double abs_tol, int32_t* fop, double* dist, int64_t* argmax, double* maxd) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (N < 0 || d < 1 || F < 1 || !(abs_tol >= 0.0)) return fail(PLP_EINVAL, "bad sizes (need F>=1, abs_tol>=0)");
    if (!normals || !offsets || !argmax || !maxd || (N > 0 && (!X || !fop || !dist)))
        return fail(PLP_EINVAL, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nX = (size_t)N * d, nF = (size_t)F * d;
    int rc = ensure_arena(ctx, pad(nX * 8) + pad(nF * 8) + pad((size_t)F * 8) * 3 + pad((size_t)N * 4) +
                                   pad((size_t)N * 8) + 4096);
    if (rc) return rc;
    Arena a(ctx);
    double* dX = a.take<double>(nX ? nX : 1);
    double* dn = a.take<double>(nF);
    double* dof = a.take<double>(F);
    int32_t* dfop = a.take<int32_t>(N ? N : 1);
    double* ddist = a.take<double>(N ? N : 1);
    int64_t* dam = a.take<int64_t>(F);
    double* dmx = a.take<double>(F);
    hipStream_t st = ctx->stream;
    if (nX) HIP_TRY(hipMemcpyAsync(dX, X, nX * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dn, normals, nF * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dof, offsets, (size_t)F * 8, hipMemcpyHostToDevice, st));
    rc = plp_assign_dev(ctx, st, N, d, dX, F, dn, dof, abs_tol, dfop, ddist, dam, dmx);
    if (rc) return rc;
    if (N) {
        HIP_TRY(hipMemcpyAsync(fop, dfop, (size_t)N * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(dist, ddist, (size_t)N * 8, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipMemcpyAsync(argmax, dam, (size_t)F * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(maxd, dmx, (size_t)F * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PLP_OK;
}

// ------------------------------------------------------------------------------- hull session
int plp_hull_reassign_dev(plp_ctx* ctx, void* stream, int64_t N, int d, const double* X, int32_t* owner, double* dist,
                          const uint8_t* dead, int new_id0, int n_new, const double* normals, const double* offsets,
                          double abs_tol, int64_t* argmax, double* maxd, int64_t* count) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (N < 0 || d < 1 || n_new < 1 || new_id0 < 0 || !(abs_tol >= 0.0))
        return fail(PLP_EINVAL, "bad sizes (need n_new>=1, new_id0>=0, abs_tol>=0)");
    if (!normals || !offsets || !argmax || !maxd || !count || (new_id0 > 0 && !dead) ||
        (N > 0 && (!X || !owner || !dist)))
        return fail(PLP_EINVAL, "NULL pointer");
    if (d > plp::MAX_D) return fail(PLP_EUNSUPPORTED, "d=%d > 16", d);
    if (plp::launch_hull_reassign(N, d, X, owner, dist, dead, new_id0, n_new, normals, offsets, abs_tol,
                                  reinterpret_cast<long long*>(argmax), maxd, reinterpret_cast<long long*>(count),
                                  (hipStream_t)stream))
        return fail(PLP_EUNSUPPORTED, "hull kernel: unsupported size");
    return check_launch("hull_reassign_kernel");
}

}  // extern "C"

struct plp_hull {
    plp_ctx* ctx;
    int64_t N;
    int d;
    double* X;
    int32_t* owner;
    double* dist;
    size_t X_bytes, owner_bytes, dist_bytes;   // capacities (buffers may come from the context's spare set)
    uint8_t* dead;     // one byte per facet id handed out so far (grow-only)
    size_t dead_cap;
    int next_id;       // next facet id to hand out
    // Per-call staging, grow-only: one device block and its pinned host mirror with the layout
    // [dead ids | normals | offsets | argmax | maxd | count] (reassign) or [point indices] (drop), so that a
    // call is one H2D copy, its kernels and one D2H copy.  (Pageable copies of a few hundred bytes cost
    // 30-50 us each and made an iteration of quickhull 0.4 ms.)
    char* io;
    char* pin;
    size_t io_bytes;
    bool pending;      // an asynchronous drop still reads `pin`
};

namespace {

int hull_ensure(plp_hull* h, size_t ids_needed, size_t io_needed) {
    if (ids_needed > h->dead_cap) {
        size_t want = h->dead_cap ? h->dead_cap : 4096;
        while (want < ids_needed) want *= 2;
        uint8_t* nd = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&nd), want));
        HIP_TRY(hipMemsetAsync(nd, 0, want, h->ctx->stream));
        if (h->dead) {
            HIP_TRY(hipMemcpyAsync(nd, h->dead, h->dead_cap, hipMemcpyDeviceToDevice, h->ctx->stream));
            HIP_TRY(hipStreamSynchronize(h->ctx->stream));
            HIP_TRY(hipFree(h->dead));
        }
        h->dead = nd;
        h->dead_cap = want;
    }
    if (io_needed > h->io_bytes) {
        HIP_TRY(hipStreamSynchronize(h->ctx->stream));
        h->pending = false;
        if (h->io) HIP_TRY(hipFree(h->io));
        if (h->pin) HIP_TRY(hipHostFree(h->pin));
        h->io = nullptr;
        h->pin = nullptr;
        h->io_bytes = 0;
        const size_t want = io_needed * 2 + 4096;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->io), want));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin), want, hipHostMallocDefault));
        h->io_bytes = want;
    }
    return PLP_OK;
}

}  // namespace

extern "C" {

int plp_hull_destroy(plp_hull* h) {
    if (!h) return PLP_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    plp_ctx::HullSpare& sp = h->ctx->hull_spare;
    if (!sp.full && h->X && h->owner && h->dist) {   // park the buffers for the next session of this context
        sp.X = h->X; sp.owner = h->owner; sp.dist = h->dist; sp.dead = h->dead; sp.io = h->io; sp.pin = h->pin;
        sp.X_bytes = h->X_bytes; sp.owner_bytes = h->owner_bytes; sp.dist_bytes = h->dist_bytes;
        sp.dead_cap = h->dead_cap; sp.io_bytes = h->io_bytes;
        sp.full = true;
        delete h;
        return PLP_OK;
    }
    if (h->X) (void)hipFree(h->X);
    if (h->owner) (void)hipFree(h->owner);
    if (h->dist) (void)hipFree(h->dist);
    if (h->dead) (void)hipFree(h->dead);
    if (h->io) (void)hipFree(h->io);
    if (h->pin) (void)hipHostFree(h->pin);
    delete h;
    return PLP_OK;
}

int plp_hull_create(plp_ctx* ctx, int64_t N, int d, const double* X, plp_hull** out) {
    if (!out) return fail(PLP_EINVAL, "plp_hull_create: out is NULL");
    *out = nullptr;
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (N < 0 || d < 1 || (N > 0 && !X)) return fail(PLP_EINVAL, "bad sizes / NULL points");
    if (d > plp::MAX_D) return fail(PLP_EUNSUPPORTED, "d=%d > 16", d);
    HIP_TRY(hipSetDevice(ctx->device));
    plp_hull* h = new plp_hull();
    memset(h, 0, sizeof(*h));
    h->ctx = ctx;
    h->N = N;
    h->d = d;
    h->next_id = 1;  // id 0: the virtual facet owning every point (owner[] is zero-filled)
    const size_t n1 = N > 0 ? (size_t)N : 1;
    hipError_t e = hipSuccess;
    plp_ctx::HullSpare& sp = ctx->hull_spare;
    if (sp.full && sp.X_bytes >= n1 * d * 8 && sp.owner_bytes >= n1 * 4 && sp.dist_bytes >= n1 * 8) {
        h->X = sp.X; h->owner = sp.owner; h->dist = sp.dist; h->dead = sp.dead; h->io = sp.io; h->pin = sp.pin;
        h->X_bytes = sp.X_bytes; h->owner_bytes = sp.owner_bytes; h->dist_bytes = sp.dist_bytes;
        h->dead_cap = sp.dead_cap; h->io_bytes = sp.io_bytes;
        sp = plp_ctx::HullSpare();
        if (h->dead) e = hipMemsetAsync(h->dead, 0, h->dead_cap, ctx->stream);   // no facet id is dead yet
    } else {
        h->X_bytes = n1 * d * 8; h->owner_bytes = n1 * 4; h->dist_bytes = n1 * 8;
        e = hipMalloc(reinterpret_cast<void**>(&h->X), h->X_bytes);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->owner), h->owner_bytes);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->dist), h->dist_bytes);
    }
    if (e == hipSuccess) e = hipMemsetAsync(h->owner, 0, n1 * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->dist, 0, n1 * 8, ctx->stream);
    if (e == hipSuccess && N > 0) e = hipMemcpyAsync(h->X, X, (size_t)N * d * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        plp_hull_destroy(h);
        return fail(PLP_EHIP, "plp_hull_create: %s", hipGetErrorString(e));
    }
    *out = h;
    return PLP_OK;
}

// asynchronous: ordered before the next call of this session on the context's stream
int plp_hull_drop(plp_hull* h, int64_t n, const int64_t* idx) {
    if (!h) return fail(PLP_EINVAL, "hull is NULL");
    if (n < 0 || (n > 0 && !idx)) return fail(PLP_EINVAL, "bad index list");
    if (n == 0) return PLP_OK;
    for (int64_t i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= h->N) return fail(PLP_EINVAL, "point index %lld out of range", (long long)idx[i]);
    HIP_TRY(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    if (h->pending) {  // the previous drop may not have consumed the pinned block yet
        HIP_TRY(hipStreamSynchronize(st));
        h->pending = false;
    }
    int rc = hull_ensure(h, 0, (size_t)n * 8);
    if (rc) return rc;
    memcpy(h->pin, idx, (size_t)n * 8);
    HIP_TRY(hipMemcpyAsync(h->io, h->pin, (size_t)n * 8, hipMemcpyHostToDevice, st));
    plp::launch_hull_drop(n, reinterpret_cast<const long long*>(h->io), h->owner, st);
    rc = check_launch("hull_drop_kernel");
    if (rc) return rc;
    h->pending = true;
    return PLP_OK;
}

int plp_hull_reassign(plp_hull* h, int n_dead, const int32_t* dead_ids, int n_new, const double* normals,
                      const double* offsets, double abs_tol, int32_t* new_id0, int64_t* argmax, double* maxd,
                      int64_t* count) {
    if (!h) return fail(PLP_EINVAL, "hull is NULL");
    if (n_dead < 0 || n_new < 1 || !(abs_tol >= 0.0)) return fail(PLP_EINVAL, "bad sizes (need n_new>=1, abs_tol>=0)");
    if ((n_dead > 0 && !dead_ids) || !normals || !offsets || !new_id0 || !argmax || !maxd || !count)
        return fail(PLP_EINVAL, "NULL pointer");
    for (int i = 0; i < n_dead; ++i)
        if (dead_ids[i] < 0 || dead_ids[i] >= h->next_id)
            return fail(PLP_EINVAL, "dead facet id %d was never handed out", dead_ids[i]);
    if ((long long)h->next_id + n_new > 0x7fffffffll) return fail(PLP_EUNSUPPORTED, "facet ids exhausted");
    HIP_TRY(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    if (h->pending) {
        HIP_TRY(hipStreamSynchronize(st));
        h->pending = false;
    }
    const int id0 = h->next_id;
    const int d = h->d;
    const size_t b_ids = pad((size_t)(n_dead ? n_dead : 1) * 4), b_n = pad((size_t)n_new * d * 8),
                 b_f = pad((size_t)n_new * 8);
    const size_t in_bytes = b_ids + b_n + b_f, out_bytes = 3 * b_f;
    int rc = hull_ensure(h, (size_t)id0 + n_new, in_bytes + out_bytes);
    if (rc) return rc;
    memcpy(h->pin, dead_ids, (size_t)n_dead * 4);
    memcpy(h->pin + b_ids, normals, (size_t)n_new * d * 8);
    memcpy(h->pin + b_ids + b_n, offsets, (size_t)n_new * 8);
    HIP_TRY(hipMemcpyAsync(h->io, h->pin, in_bytes, hipMemcpyHostToDevice, st));
    int32_t* d_ids = reinterpret_cast<int32_t*>(h->io);
    double* d_n = reinterpret_cast<double*>(h->io + b_ids);
    double* d_o = reinterpret_cast<double*>(h->io + b_ids + b_n);
    int64_t* d_am = reinterpret_cast<int64_t*>(h->io + in_bytes);
    double* d_mx = reinterpret_cast<double*>(h->io + in_bytes + b_f);
    int64_t* d_cn = reinterpret_cast<int64_t*>(h->io + in_bytes + 2 * b_f);
    plp::launch_hull_mark(n_dead, d_ids, h->dead, st);
    rc = plp_hull_reassign_dev(h->ctx, st, h->N, d, h->X, h->owner, h->dist, h->dead, id0, n_new, d_n, d_o, abs_tol,
                               d_am, d_mx, d_cn);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->pin + in_bytes, h->io + in_bytes, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(argmax, h->pin + in_bytes, (size_t)n_new * 8);
    memcpy(maxd, h->pin + in_bytes + b_f, (size_t)n_new * 8);
    memcpy(count, h->pin + in_bytes + 2 * b_f, (size_t)n_new * 8);
    h->next_id = id0 + n_new;
    *new_id0 = id0;
    return PLP_OK;
}

int plp_hull_read(plp_hull* h, int32_t* owner, double* dist) {
    if (!h) return fail(PLP_EINVAL, "hull is NULL");
    HIP_TRY(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    if (owner && h->N) HIP_TRY(hipMemcpyAsync(owner, h->owner, (size_t)h->N * 4, hipMemcpyDeviceToHost, st));
    if (dist && h->N) HIP_TRY(hipMemcpyAsync(dist, h->dist, (size_t)h->N * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->pending = false;
    return PLP_OK;
}

}  // extern "C"

extern "C" {

// ------------------------------------------------------------------------------- adjacency
int plp_adjacent_pairs_dev(plp_ctx* ctx, void* stream, int n, int m_max, int d, const double* A, const double* b,
                           const int32_t* m, double abs_tol, uint8_t* adj) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (n < 0 || m_max < 1 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (n == 0) return PLP_OK;
    if (!A || !b || !adj) return fail(PLP_EINVAL, "NULL pointer");
    if (2 * m_max > plp::MAX_M || d > plp::MAX_D)
        return fail(PLP_EUNSUPPORTED, "m_max=%d d=%d outside envelope (2*m_max<=64, d<=16)", m_max, d);
    if (plp::launch_adjacent(n, m_max, d, A, b, m, abs_tol, abs_tol / 10, adj, 0, 0, nullptr, (hipStream_t)stream))
        return fail(PLP_EUNSUPPORTED, "adjacent kernel: unsupported size");
    return check_launch("adjacent_r_kernel");
}

int plp_overlap_pairs_dev(plp_ctx* ctx, void* stream, int n, int m_max, int d, const double* A, const double* b,
                          const int32_t* m, double abs_tol, uint8_t* out) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (n < 0 || m_max < 1 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (n == 0) return PLP_OK;
    if (!A || !b || !out) return fail(PLP_EINVAL, "NULL pointer");
    if (2 * m_max > plp::MAX_M || d > plp::MAX_D)
        return fail(PLP_EUNSUPPORTED, "m_max=%d d=%d outside envelope (2*m_max<=64, d<=16)", m_max, d);
    if (plp::launch_adjacent(n, m_max, d, A, b, m, 0.0, abs_tol, out, 0, 0, nullptr, (hipStream_t)stream))
        return fail(PLP_EUNSUPPORTED, "adjacent kernel: unsupported size");
    return check_launch("adjacent_r_kernel");
}

int plp_overlap_pairs(plp_ctx* ctx, int n, int m_max, int d, const double* A, const double* b, const int32_t* m,
                      double abs_tol, uint8_t* out) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (n < 0 || m_max < 1 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (n == 0) return PLP_OK;
    if (!A || !b || !out) return fail(PLP_EINVAL, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nA = (size_t)n * m_max * d, nb = (size_t)n * m_max, nout = (size_t)n * n;
    int rc = ensure_arena(ctx, pad(nA * 8) + pad(nb * 8) + pad((size_t)n * 4) + pad(nout) + 4096);
    if (rc) return rc;
    Arena a(ctx);
    double* dA = a.take<double>(nA);
    double* db = a.take<double>(nb);
    int32_t* dm = a.take<int32_t>(n);
    uint8_t* dout = a.take<uint8_t>(nout);
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(dA, A, nA * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(db, b, nb * 8, hipMemcpyHostToDevice, st));
    if (m) HIP_TRY(hipMemcpyAsync(dm, m, (size_t)n * 4, hipMemcpyHostToDevice, st));
    rc = plp_overlap_pairs_dev(ctx, st, n, m_max, d, dA, db, m ? dm : nullptr, abs_tol, dout);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, nout, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PLP_OK;
}

int plp_overlap_cross_dev(plp_ctx* ctx, void* stream, int n1, int n2, int m_max, int d, const double* A, const double* b,
                          const int32_t* m, double thresh, uint8_t* out) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (n1 < 0 || n2 < 0 || m_max < 1 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (n1 == 0 || n2 == 0) return PLP_OK;
    if (!A || !b || !out) return fail(PLP_EINVAL, "NULL pointer");
    if (2 * m_max > plp::MAX_M || d > plp::MAX_D)
        return fail(PLP_EUNSUPPORTED, "m_max=%d d=%d outside envelope (2*m_max<=64, d<=16)", m_max, d);
    if ((long long)n1 + n2 > 2147483647ll) return fail(PLP_EUNSUPPORTED, "too many cells");
    if (plp::launch_adjacent(n1 + n2, m_max, d, A, b, m, 0.0, thresh, nullptr, 0, (long long)n1 * n2, out,
                             (hipStream_t)stream, n1))
        return fail(PLP_EUNSUPPORTED, "adjacent kernel: unsupported size");
    return check_launch("adjacent_r_kernel");
}

int plp_overlap_cross(plp_ctx* ctx, int n1, int n2, int m_max, int d, const double* A, const double* b, const int32_t* m,
                      double thresh, uint8_t* out) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (n1 < 0 || n2 < 0 || m_max < 1 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (n1 == 0 || n2 == 0) return PLP_OK;
    if (!A || !b || !out) return fail(PLP_EINVAL, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)n1 + n2;
    const size_t nA = n * m_max * d, nb = n * m_max, nout = (size_t)n1 * n2;
    int rc = ensure_arena(ctx, pad(nA * 8) + pad(nb * 8) + pad(n * 4) + pad(nout) + 4096);
    if (rc) return rc;
    Arena a(ctx);
    double* dA = a.take<double>(nA);
    double* db = a.take<double>(nb);
    int32_t* dm = a.take<int32_t>(n);
    uint8_t* dout = a.take<uint8_t>(nout);
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(dA, A, nA * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(db, b, nb * 8, hipMemcpyHostToDevice, st));
    if (m) HIP_TRY(hipMemcpyAsync(dm, m, n * 4, hipMemcpyHostToDevice, st));
    rc = plp_overlap_cross_dev(ctx, st, n1, n2, m_max, d, dA, db, m ? dm : nullptr, thresh, dout);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, nout, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PLP_OK;
}

int plp_adjacent_pairs_range_dev(plp_ctx* ctx, void* stream, int n, int m_max, int d, const double* A,
                                 const double* b, const int32_t* m, double abs_tol, int64_t pair_lo,
                                 int64_t pair_hi, uint8_t* out) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (n < 0 || m_max < 1 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    const int64_t npairs = (int64_t)n * (n - 1) / 2;
    if (pair_lo < 0 || pair_hi > npairs || pair_lo > pair_hi)
        return fail(PLP_EINVAL, "pair range [%lld, %lld) outside [0, %lld)", (long long)pair_lo, (long long)pair_hi,
                    (long long)npairs);
    if (pair_lo == pair_hi) return PLP_OK;
    if (!A || !b || !out) return fail(PLP_EINVAL, "NULL pointer");
    if (2 * m_max > plp::MAX_M || d > plp::MAX_D)
        return fail(PLP_EUNSUPPORTED, "m_max=%d d=%d outside envelope (2*m_max<=64, d<=16)", m_max, d);
    if (plp::launch_adjacent(n, m_max, d, A, b, m, abs_tol, abs_tol / 10, nullptr, pair_lo, pair_hi, out,
                             (hipStream_t)stream))
        return fail(PLP_EUNSUPPORTED, "adjacent kernel: unsupported size");
    return check_launch("adjacent_r_kernel");
}

int plp_adjacent_pairs_range(plp_ctx* ctx, int n, int m_max, int d, const double* A, const double* b,
                             const int32_t* m, double abs_tol, int64_t pair_lo, int64_t pair_hi, uint8_t* out) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (n < 0 || m_max < 1 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (pair_lo == pair_hi) return PLP_OK;
    if (!A || !b || !out) return fail(PLP_EINVAL, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nA = (size_t)n * m_max * d, nb = (size_t)n * m_max;
    const size_t nout = pair_hi > pair_lo ? (size_t)(pair_hi - pair_lo) : 0;
    int rc = ensure_arena(ctx, pad(nA * 8) + pad(nb * 8) + pad((size_t)n * 4) + pad(nout) + 4096);
    if (rc) return rc;
    Arena a(ctx);
    double* dA = a.take<double>(nA);
    double* db = a.take<double>(nb);
    int32_t* dm = a.take<int32_t>(n);
    uint8_t* dout = a.take<uint8_t>(nout ? nout : 1);
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(dA, A, nA * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(db, b, nb * 8, hipMemcpyHostToDevice, st));
    if (m) HIP_TRY(hipMemcpyAsync(dm, m, (size_t)n * 4, hipMemcpyHostToDevice, st));
    rc = plp_adjacent_pairs_range_dev(ctx, st, n, m_max, d, dA, db, m ? dm : nullptr, abs_tol, pair_lo, pair_hi, dout);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, nout, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PLP_OK;
}

int plp_adjacent_pairs(plp_ctx* ctx, int n, int m_max, int d, const double* A, const double* b, const int32_t* m,
                       double abs_tol, uint8_t* adj) {
    if (!ctx) return fail(PLP_EINVAL, "ctx is NULL");
    if (n < 0 || m_max < 1 || d < 1) return fail(PLP_EINVAL, "bad sizes");
    if (n == 0) return PLP_OK;
    if (!A || !b || !adj) return fail(PLP_EINVAL, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nA = (size_t)n * m_max * d, nb = (size_t)n * m_max, nadj = (size_t)n * n;
    int rc = ensure_arena(ctx, pad(nA * 8) + pad(nb * 8) + pad((size_t)n * 4) + pad(nadj) + 4096);
    if (rc) return rc;
    Arena a(ctx);
    double* dA = a.take<double>(nA);
    double* db = a.take<double>(nb);
    int32_t* dm = a.take<int32_t>(n);
    uint8_t* dadj = a.take<uint8_t>(nadj);
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(dA, A, nA * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(db, b, nb * 8, hipMemcpyHostToDevice, st));
    if (m) HIP_TRY(hipMemcpyAsync(dm, m, (size_t)n * 4, hipMemcpyHostToDevice, st));
    rc = plp_adjacent_pairs_dev(ctx, st, n, m_max, d, dA, db, m ? dm : nullptr, abs_tol, dadj);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(adj, dadj, nadj, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PLP_OK;
}

int plp_selftest(plp_ctx* ctx, int group_size, double* out_d, uint32_t* out_u) {
    if (!ctx || !out_d || !out_u) return fail(PLP_EINVAL, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_arena(ctx, 128 * 8 + 128 * 4 + 1024);
    if (rc) return rc;
    Arena a(ctx);
    double* dd = a.take<double>(128);
    unsigned* du = a.take<unsigned>(128);
    if (plp::launch_selftest(group_size, dd, du, ctx->stream)) return fail(PLP_EINVAL, "group size must be 8/16/32/64");
    rc = check_launch("selftest_kernel");
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out_d, dd, 128 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_u, du, 128 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PLP_OK;
}

// ------------------------------------------------------------------------------- region_diff search
}  // extern "C"

struct plp_rdiff_result {
    std::vector<int32_t> kind;   // per leaf: 0 = piece as is (ref :2229), 1 = piece to be reduce()d (ref :2276)
    std::vector<int32_t> off;    // leaf k holds rows[off[k] .. off[k+1])
    std::vector<int32_t> rows;
    long long n_lps = 0, n_batches = 0, n_requests = 0, n_scan_miss = 0, n_node_miss = 0;
};

namespace {

// 128-bit order-dependent hash of a row list, extendable by a suffix (lists are "current rows + a few more")
struct Key {
    uint64_t a, b;
    bool operator==(const Key& o) const { return a == o.a && b == o.b; }
};
inline Key key_push(Key k, int32_t r) {
    k.a = (k.a ^ (uint64_t)(uint32_t)(r + 1)) * 0x9E3779B97F4A7C15ull;
    k.a ^= k.a >> 29;
    k.b = (k.b + (uint64_t)(uint32_t)(r + 1)) * 0xC2B2AE3D27D4EB4Full;
    k.b ^= k.b >> 31;
    return k;
}
inline Key key_of_list(const int32_t* r, size_t n) {
    Key k{0x243F6A8885A308D3ull, 0x13198A2E03707344ull};
    for (size_t i = 0; i < n; ++i) k = key_push(k, r[i]);
    return k;
}

// open-addressing table Key -> radius; grows by rehashing, clear() touches only the occupied slots
struct RadiusTable {
    std::vector<Key> keys;
    std::vector<double> vals;
    std::vector<unsigned char> used;
    std::vector<uint32_t> slots;   // occupied positions, in insertion order
    size_t count = 0, mask = 0;
    void reset(size_t cap_pow2) {
        keys.assign(cap_pow2, Key{0, 0});
        vals.assign(cap_pow2, 0.0);
        used.assign(cap_pow2, 0);
        slots.clear();
        count = 0;
        mask = cap_pow2 - 1;
    }
    void clear() {
        for (uint32_t i : slots) used[i] = 0;
        slots.clear();
        count = 0;
    }
    const double* find(const Key& k) const {
        for (size_t i = (size_t)k.a & mask;; i = (i + 1) & mask) {
            if (!used[i]) return nullptr;
            if (keys[i] == k) return &vals[i];
        }
    }
    void put(const Key& k, double v) {
        if ((count + 1) * 2 > mask) grow();
        for (size_t i = (size_t)k.a & mask;; i = (i + 1) & mask) {
            if (!used[i]) { used[i] = 1; keys[i] = k; vals[i] = v; ++count; slots.push_back((uint32_t)i); return; }
            if (keys[i] == k) { vals[i] = v; return; }
        }
    }
    void grow() {
        std::vector<Key> ok;
        std::vector<double> ov;
        ok.reserve(count);
        ov.reserve(count);
        for (uint32_t i : slots) { ok.push_back(keys[i]); ov.push_back(vals[i]); }
        reset((mask + 1) * 2);
        for (size_t t = 0; t < ok.size(); ++t) put(ok[t], ov[t]);
    }
};

// The Chebyshev radii the search asks for, keyed by the row list, filled by batches of LPs that are solved on the
// device straight from the resident table (plp_rdiff.hip).  A miss triggers ONE batch holding the missing list
// plus whatever the search will need next if the current node is not empty (speculation), so the search pays one
// launch + one synchronisation per visited node instead of one per LP (the reference) or per scan.
struct RadiusOracle {
    plp_ctx* ctx;
    int d;
    long long nrows;
    double *dA = nullptr, *dB = nullptr, *dOut = nullptr;
    char* pin = nullptr;          // host-mapped block: [index block | radii | sequence word]
    char* pin_dev = nullptr;      // the same block as the device sees it
    volatile unsigned long long* flag = nullptr;
    unsigned long long seq = 0;
    size_t cap_lp = 0, cap_rows = 0;
    RadiusTable memo, pending;    // pending: key -> position in the batch being assembled
    std::vector<int32_t> prow, poff;
    std::vector<Key> pkey;
    long long n_lps = 0, n_batches = 0;
    long long n_lps_long = 0, n_batches_long = 0;  // LPs of more than 64 rows (LDS engine) / batches that hold at least one (stats)
    double t_launch = 0.0, t_wait = 0.0;  // seconds spent enqueueing / waiting for the device (PLP_RDIFF_STATS=1 prints them)
    double t_store = 0.0;                 // ... and putting the radii of a batch into the memo
    // ---- resident LP server (d <= 4; PLP_RDIFF_SERVER=0: every batch a launch, as in round 3)
    static constexpr size_t SRV_MAXLP = 2048, SRV_OUT = 128, SRV_REC = SRV_OUT + SRV_MAXLP * 16;
    static constexpr size_t SRV_BYTES = SRV_REC + SRV_MAXLP * 66 * 4;
    static constexpr unsigned long long SRV_EXIT = ~0ull;
    bool srv_on = false, srv_running = false;
    long long n_srv_batches = 0, n_srv_starts = 0;
    volatile unsigned long long* srv_mail() { return reinterpret_cast<volatile unsigned long long*>(ctx->rd_srv); }
    volatile unsigned long long* srv_done() { return reinterpret_cast<volatile unsigned long long*>(ctx->rd_srv + 64); }
    int srv_init() {
        const char* e = getenv("PLP_RDIFF_SERVER");
        if ((e && e[0] == '0') || d > 4) return PLP_OK;
        if (!ctx->rd_srv) {
            char *hp = nullptr, *hp_dev = nullptr;
            unsigned long long* st8 = nullptr;
            hipError_t err = hipHostMalloc(reinterpret_cast<void**>(&hp), SRV_BYTES, hipHostMallocMapped | hipHostMallocCoherent);
            if (err == hipSuccess) err = hipHostGetDevicePointer(reinterpret_cast<void**>(&hp_dev), hp, 0);
            if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&st8), 64);
            if (err != hipSuccess) {
                if (hp) (void)hipHostFree(hp);
                (void)hipGetLastError();
                return PLP_OK;   // no server: the launch path serves the search
            }
            memset(hp, 0, 128);
            ctx->rd_srv = hp;
            ctx->rd_srv_dev = hp_dev;
            ctx->rd_srv_state = st8;
            ctx->rd_srv_seq = 0;
            ctx->rd_srv_word = 0;
        }
        srv_on = true;
        return PLP_OK;
    }
    // last: the sequence number of the last batch that was answered.  pending: batch last + 1 is in the mailbox already
    // (the server retired while it was on its way) -- the new server finds it there; else the mailbox says "nothing new".
    int srv_start(unsigned long long last, bool pending) {
        if (!pending) __atomic_store_n(srv_mail(), last, __ATOMIC_RELEASE);
        __atomic_store_n(srv_done() + 1, 1ull, __ATOMIC_RELEASE);
        ctx->rd_srv_init[0] = ctx->rd_srv_init[2] = ctx->rd_srv_init[3] = 0ull;   // word 0: "retire" flag
        ctx->rd_srv_init[1] = last;                                                // word 1: the mailbox as the device republishes it
        HIP_TRY(hipMemcpyAsync(ctx->rd_srv_state, ctx->rd_srv_init, 32, hipMemcpyHostToDevice, ctx->stream));
        const char* ip = getenv("PLP_RDIFF_SERVER_IDLE");
        const unsigned idle = ip ? (unsigned)atoi(ip) : 1500u;   // empty mailbox polls (~2 us each) before it retires
        const char* wg = getenv("PLP_RDIFF_SERVER_WGS");
        // workgroups (one wavefront each: 4 / 2 / 1 lists per round).  Measured at config 4 (751 batches, ~120 lists each, a
        // few of up to 2048): waiting for the device 22.4 / 16.6 / 14.1 / 12.7 / 12.3 ms with 32 / 64 / 128 / 256 / 1024
        // (the launch path: 4.3 ms of launches + 11.3 ms of waiting)
        const int nwg = wg ? atoi(wg) : 256;
        if (plp::launch_rdiff_server(d, nwg < 1 ? 1 : nwg, reinterpret_cast<const unsigned long long*>(ctx->rd_srv_dev),
                                     reinterpret_cast<const int*>(ctx->rd_srv_dev + SRV_REC),
                                     ctx->rd_srv_dev + SRV_OUT,
                                     reinterpret_cast<unsigned long long*>(ctx->rd_srv_dev + 64 + 8), ctx->rd_srv_state, dA, dB,
                                     last, idle, ctx->stream))
            return fail(PLP_EUNSUPPORTED, "region_diff: no LP server for d=%d", d);
        int rc = check_launch("rdiff_server");
        if (rc) return rc;
        srv_running = true;
        ++n_srv_starts;
        return PLP_OK;
    }
    void srv_stop() {
        if (!srv_running) return;
        __atomic_store_n(srv_mail(), SRV_EXIT, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(ctx->stream);
        srv_running = false;
    }

    int init(plp_ctx* c, int d_, long long nrows_, const double* A, const double* b) {
        ctx = c; d = d_; nrows = nrows_;
        cap_lp = 1 << 15;
        cap_rows = (size_t)cap_lp * 64;
        memo.reset(1 << 14);
        pending.reset(1 << 12);
        poff.assign(1, 0);
        const size_t blk = (2 * cap_lp + 1 + cap_rows) * 4;
        if (!ctx->rd_pin) {
            // host-mapped, coherent: the kernels read the index block and publish the radii through it (no copies, no
            // stream synchronisation per batch: the host spins on a sequence word the last kernel of the batch raises).
            // Allocated into locals and committed to the context only when all three calls succeeded.
            char *hp = nullptr, *hp_dev = nullptr;
            double* dout = nullptr;
            hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&hp), blk + cap_lp * 8 + 64,
                                         hipHostMallocMapped | hipHostMallocCoherent);
            if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&hp_dev), hp, 0);
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dout), cap_lp * 8);
            if (e != hipSuccess) {
                if (hp) (void)hipHostFree(hp);
                (void)hipGetLastError();
                return fail(PLP_EHIP, "region_diff: staging buffers: %s", hipGetErrorString(e));
            }
            *reinterpret_cast<volatile unsigned long long*>(hp + blk + cap_lp * 8) = 0ull;
            ctx->rd_pin = hp;
            ctx->rd_pin_dev = hp_dev;
            ctx->rd_out = dout;
            ctx->rd_seq = 0;
        }
        const size_t need = (size_t)nrows * (d + 1) * 8 + 64;
        if (need > ctx->rd_tab_bytes) {
            if (ctx->rd_tab) HIP_TRY(hipFree(ctx->rd_tab));
            ctx->rd_tab = nullptr;
            ctx->rd_tab_bytes = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->rd_tab), need + need / 2));
            ctx->rd_tab_bytes = need + need / 2;
        }
        pin = ctx->rd_pin;
        pin_dev = ctx->rd_pin_dev;
        dOut = ctx->rd_out;
        dA = ctx->rd_tab;
        dB = ctx->rd_tab + (size_t)nrows * d;
        flag = reinterpret_cast<volatile unsigned long long*>(pin + blk + cap_lp * 8);
        seq = ctx->rd_seq;
        HIP_TRY(hipMemcpyAsync(dA, A, (size_t)nrows * d * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(dB, b, (size_t)nrows * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return srv_init();
    }
    void release() {
        if (ctx) { srv_stop(); (void)hipStreamSynchronize(ctx->stream); ctx->rd_seq = seq; }
    }
    // queue the list base[0..nb) + suf[0..ns) (key given) for the next batch unless known or queued already
    // `required`: the search cannot continue without this radius; only speculative lists are subject to the bound
    // on one batch
    void want(const Key& k, const int32_t* base, size_t nb, const int32_t* suf, size_t ns, bool required = false) {
        if (memo.find(k) || pending.find(k)) return;
        if (!required && pkey.size() >= (1u << 16)) return;
        pending.put(k, (double)pkey.size());
        pkey.push_back(k);
        prow.insert(prow.end(), base, base + nb);
        prow.insert(prow.end(), suf, suf + ns);
        poff.push_back((int32_t)prow.size());
    }
    // solve everything queued; results go to the memo
    int flush() {
        const size_t total = pkey.size();
        size_t done = 0;
        // radii are consumed soon after they are computed.  The memo is emptied only BEFORE the first sub-batch, never
        // between two of them, so everything this call stores is there when it returns (a caller that skipped a list
        // because it was known before the call asks again: see the retry loops of the search)
        if (memo.count + total > (1u << 20)) memo.clear();
        while (done < total) {
            size_t n = 0, nr = 0;
            while (done + n < total && n < cap_lp && nr + (size_t)(poff[done + n + 1] - poff[done + n]) <= cap_rows) {
                nr += (size_t)(poff[done + n + 1] - poff[done + n]);
                ++n;
            }
            if (n == 0) return fail(PLP_EUNSUPPORTED, "region_diff: a row list of %d rows exceeds the staging buffer", poff[done + 1] - poff[done]);
            // one contiguous block per batch: [off (n + 1) | sel (n) | rows (nr)] -> ONE copy across PCIe
            int32_t* off = reinterpret_cast<int32_t*>(pin);
            int32_t* sel = off + n + 1;
            int32_t* rows = sel + n;
            const int32_t base0 = poff[done];
            memcpy(rows, prow.data() + base0, nr * 4);
            int cnt[5] = {0, 0, 0, 0, 0};
            int max_len = 0;
            for (size_t k = 0; k <= n; ++k) off[k] = poff[done + k] - base0;
            // size classes: 0 / 1 / 2 = lists of up to 16 / 32 / 64 rows on the lane-group kernels (d <= 4), 3 = beyond 64 rows
            // (LDS engine), 4 = d >= 5, up to 64 rows: one LP per wavefront (cheby_gather_w_kernel; PLP_RDIFF_WIDE=0: d = 5..8
            // on the lane groups, d >= 9 on the LDS engine, as before round 3)
            const char* rw_env = getenv("PLP_RDIFF_WIDE");
            const bool rdiff_wide = !(rw_env && rw_env[0] == '0');
            const char* rwm_env = getenv("PLP_RDIFF_WIDE_MIND");  // (A/B: smallest d that takes the one-LP-per-wavefront kernel)
            const int rdiff_wide_mind = rwm_env ? atoi(rwm_env) : 5;
            auto cls_of = [&](int len) {
                if (len > 64) return 3;
                if (d >= rdiff_wide_mind && rdiff_wide) return 4;
                return d > 8 ? 3 : (len > 32 ? 2 : (len > 16 ? 1 : 0));
            };
            for (size_t k = 0; k < n; ++k) {
                const int len = off[k + 1] - off[k];
                max_len = len > max_len ? len : max_len;
                cnt[cls_of(len)]++;
            }
            if (srv_on && n <= SRV_MAXLP && !cnt[3] && !cnt[4]) {
                // ---- the resident server takes the batch: records by size class, header, sequence word; spin on done_seq
                const auto tp0 = std::chrono::steady_clock::now();
                int32_t* rec = reinterpret_cast<int32_t*>(ctx->rd_srv + SRV_REC);
                int32_t* rc_[3] = {rec, rec + (size_t)cnt[0] * 18, rec + (size_t)cnt[0] * 18 + (size_t)cnt[1] * 34};
                static const int cap_[3] = {16, 32, 64};
                for (size_t k = 0; k < n; ++k) {
                    const int len = off[k + 1] - off[k];
                    const int c = cls_of(len);
                    int32_t* r = rc_[c];
                    r[0] = (int32_t)k;
                    r[1] = len;
                    memcpy(r + 2, prow.data() + base0 + off[k], (size_t)len * 4);
                    rc_[c] = r + cap_[c] + 2;
                }
                if (!srv_running || __atomic_load_n(srv_done() + 1, __ATOMIC_ACQUIRE) == 0ull) {
                    if (srv_running) { (void)hipStreamSynchronize(ctx->stream); srv_running = false; }   // it retired (idle)
                    int rc = srv_start(ctx->rd_srv_word, false);
                    if (rc) return rc;
                }
                // one word: [batch number : 28 | n2 : 12 | n1 : 12 | n0 : 12] (never 0, never all ones)
                const unsigned long long last_word = ctx->rd_srv_word;
                ctx->rd_srv_seq = (ctx->rd_srv_seq % 0xffffff0ull) + 1ull;
                const unsigned long long sq = (ctx->rd_srv_seq << 36) | ((unsigned long long)cnt[2] << 24) |
                                              ((unsigned long long)cnt[1] << 12) | (unsigned long long)cnt[0];
                __atomic_store_n(srv_mail(), sq, __ATOMIC_RELEASE);
                const auto tp1 = std::chrono::steady_clock::now();
                // every radius arrives beside the batch's word (one 16-byte store of the lane group that solved it)
                const volatile unsigned long long* sres = reinterpret_cast<const volatile unsigned long long*>(ctx->rd_srv + SRV_OUT);
                unsigned long long spins = 0;
                for (size_t k = 0; k < n;) {
                    if (__atomic_load_n(&sres[2 * k + 1], __ATOMIC_ACQUIRE) == sq) { ++k; continue; }
                    if ((++spins & 0x3fffull) == 0ull) {
                        if (__atomic_load_n(srv_done() + 1, __ATOMIC_ACQUIRE) == 0ull) {
                            // the server retired while the batch was on its way (or part of it did the batch and part did
                            // not): start it again, it finds the batch in the mailbox and solves it (again)
                            (void)hipStreamSynchronize(ctx->stream);
                            int rc = srv_start(last_word, true);
                            if (rc) return rc;
                        }
                        if (spins > 4000000000ull) return fail(PLP_EHIP, "region_diff: the LP server did not answer batch %llu", sq);
                    }
                }
                const auto tp2 = std::chrono::steady_clock::now();
                t_launch += std::chrono::duration<double>(tp1 - tp0).count();
                t_wait += std::chrono::duration<double>(tp2 - tp1).count();
                ctx->rd_srv_word = sq;
                const double* sout = reinterpret_cast<const double*>(ctx->rd_srv + SRV_OUT);
                for (size_t k = 0; k < n; ++k) memo.put(pkey[done + k], sout[2 * k]);
                t_store += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp2).count();
                n_lps += (long long)n;
                n_batches += 1;
                n_srv_batches += 1;
                done += n;
                continue;
            }
            srv_stop();   // (a batch the server does not take: the launches below queue behind it on the same stream)
            size_t start[5], fill[5];
            start[0] = 0;
            for (int c = 1; c < 5; ++c) start[c] = start[c - 1] + cnt[c - 1];
            for (int c = 0; c < 5; ++c) fill[c] = start[c];
            for (size_t k = 0; k < n; ++k) sel[fill[cls_of(off[k + 1] - off[k])]++] = (int32_t)k;
            hipStream_t st = ctx->stream;
            const auto tp0 = std::chrono::steady_clock::now();
            const int32_t* d_off = reinterpret_cast<const int32_t*>(pin_dev);
            const int32_t* d_sel = d_off + n + 1;
            const int32_t* d_rows = d_sel + n;
            if ((cnt[0] | cnt[1] | cnt[2]) &&
                plp::launch_cheby_gather_r(d, cnt[0], cnt[1], cnt[2], d_off, d_rows, d_sel, dA, dB, dOut, st))
                return fail(PLP_EUNSUPPORTED, "region_diff: gather kernel does not apply (d=%d)", d);
            if (cnt[3] && plp::launch_cheby_gather_lds(d, max_len, cnt[3], d_off, d_rows, d_sel + start[3], dA, dB, dOut, st))
                return fail(PLP_EUNSUPPORTED, "region_diff: a stack of %d rows does not fit the LDS engine", max_len);
            if (cnt[4] && plp::launch_cheby_gather_w(d, cnt[4], d_off, d_rows, d_sel + start[4], dA, dB, dOut, st))
                return fail(PLP_EUNSUPPORTED, "region_diff: gather kernel does not apply (d=%d)", d);
            double* out = reinterpret_cast<double*>(pin + (2 * cap_lp + 1 + cap_rows) * 4);
            double* out_dev = reinterpret_cast<double*>(pin_dev + (2 * cap_lp + 1 + cap_rows) * 4);
            ++seq;
            plp::launch_rdiff_publish((long long)n, dOut, out_dev,
                                      reinterpret_cast<unsigned long long*>(pin_dev + (2 * cap_lp + 1 + cap_rows) * 4 + cap_lp * 8),
                                      seq, st);
            int rc = check_launch("cheby_gather");
            if (rc) return rc;
            const auto tp1 = std::chrono::steady_clock::now();
            // spin on the sequence word (bounded: fall back to a stream synchronisation, which also reports errors)
            {
                unsigned long long spins = 0;
                while (__atomic_load_n(const_cast<const unsigned long long*>(flag), __ATOMIC_ACQUIRE) != seq) {
                    if (++spins > 400000000ull) { HIP_TRY(hipStreamSynchronize(st)); break; }
                }
                if (__atomic_load_n(const_cast<const unsigned long long*>(flag), __ATOMIC_ACQUIRE) != seq) {
                    HIP_TRY(hipStreamSynchronize(st));
                    if (*flag != seq) return fail(PLP_EHIP, "region_diff: batch %llu did not complete", seq);
                }
            }
            const auto tp2 = std::chrono::steady_clock::now();
            t_launch += std::chrono::duration<double>(tp1 - tp0).count();
            t_wait += std::chrono::duration<double>(tp2 - tp1).count();
            for (size_t k = 0; k < n; ++k) memo.put(pkey[done + k], out[k]);
            n_lps += (long long)n;
            n_batches += 1;
            n_lps_long += cnt[3];
            n_batches_long += cnt[3] ? 1 : 0;
            done += n;
        }
        pkey.clear();
        prow.clear();
        poff.assign(1, 0);
        pending.clear();
        return PLP_OK;
    }
};

}  // namespace

extern "C" {

int plp_region_diff_search(plp_ctx* ctx, int d, int m, int N, const int32_t* mi, const double* A, const double* b,
                           double abs_tol, plp_rdiff_result** out) {
    if (!ctx || !mi || !A || !b || !out) return fail(PLP_EINVAL, "NULL pointer");
    if (d < 1 || d > plp::MAX_D || m < 0 || N < 1) return fail(PLP_EINVAL, "bad sizes d=%d m=%d N=%d", d, m, N);
    *out = nullptr;
    long long M = 0;
    std::vector<long long> beg(N);
    for (int j = 0; j < N; ++j) {
        if (mi[j] < 1) return fail(PLP_EINVAL, "mi[%d] = %d (a cell without a new constraint covers the polytope)", j, mi[j]);
        beg[j] = m + M;
        M += mi[j];
    }
    const long long nrows = m + 2 * M;
    HIP_TRY(hipSetDevice(ctx->device));
    RadiusOracle R;
    int rc = R.init(ctx, d, nrows, A, b);
    if (rc) { R.release(); return rc; }
    plp_rdiff_result* res = new plp_rdiff_result();
    res->off.push_back(0);
    // Python's negative indexing of counter / mi / beg_mi (level == -1 reads the LAST cell, ref :2231) and of the
    // table rows (an index that went below zero after "- M" wraps to the end of A, ref :2233)
    auto at = [&](long long level) { return (int)(level < 0 ? level + N : level); };
    std::vector<int> counter(N, 0);
    std::vector<long long> idx(m);       // the reference's INDICES (may hold wrapped / out-of-range values)
    for (int i = 0; i < m; ++i) idx[i] = i;
    long long level = 0;
    std::vector<int32_t> cur, suf;        // cur = idx resolved to table rows
    bool cur_ok = true;
    auto resolve = [&]() {
        cur.resize(idx.size());
        cur_ok = true;
        for (size_t k = 0; k < idx.size(); ++k) {
            long long r = idx[k] < 0 ? idx[k] + nrows : idx[k];
            if (r < 0 || r >= nrows) { cur_ok = false; r = 0; }
            cur[k] = (int32_t)r;
        }
    };
    auto emit = [&](int kind) {
        res->kind.push_back(kind);
        res->rows.insert(res->rows.end(), cur.begin(), cur.end());
        res->off.push_back((int32_t)res->rows.size());
    };
    // open cells (counter != 0) in ascending order and the sum of the counters, kept incrementally
    std::vector<int> open_cells;
    long long sumc = 0;
    auto set_counter = [&](int L, int v) {
        const int old = counter[L];
        sumc += v - old;
        counter[L] = v;
        if (old == 0 && v != 0) open_cells.insert(std::lower_bound(open_cells.begin(), open_cells.end(), L), L);
        else if (old != 0 && v == 0) open_cells.erase(std::lower_bound(open_cells.begin(), open_cells.end(), L));
    };
    // the move after a piece was emitted (ref :2230-2245: "level - 1", then re-open the cell there) on an explicit state
    auto leaf_on = [&](std::vector<int>& cnt, std::vector<int>& open, std::vector<long long>& ix, long long& lvl,
                       long long& sc) -> bool {
        lvl = lvl - 1;
        const int nz = (int)open.size();
        for (int t = 0; t < nz; ++t) {
            const int L = at(lvl);
            if (cnt[L] <= mi[L]) {
                ix.back() -= M;
                ix.push_back(beg[L] + cnt[L] + M);
                return false;
            }
            sc -= cnt[L];
            cnt[L] = 0;
            open.erase(std::lower_bound(open.begin(), open.end(), L));
            const long long keep_n = m + sc;
            if ((long long)ix.size() > keep_n) ix.resize(keep_n < 0 ? 0 : keep_n);
            if (lvl == -1) return true;
        }
        return false;
    };
    // the "next sibling" move (ref :2246-2271) on an explicit state, so that it can also be run on a COPY to see which
    // nodes follow when the current one turns out empty; returns true when the search ends
    auto advance_on = [&](std::vector<int>& cnt, std::vector<int>& open, std::vector<long long>& ix, long long& lvl,
                          long long& sc) -> bool {
        auto setc = [&](int L, int v) {
            const int old = cnt[L];
            sc += v - old;
            cnt[L] = v;
            if (old == 0 && v != 0) open.insert(std::lower_bound(open.begin(), open.end(), L), L);
            else if (old != 0 && v == 0) open.erase(std::lower_bound(open.begin(), open.end(), L));
        };
        while (!open.empty()) {   // deepest open cell first (the reference walks nzcount backwards)
            lvl = open.back();
            setc((int)lvl, cnt[lvl] + 1);
            if (cnt[lvl] <= mi[lvl]) {
                ix.back() -= M;
                ix.push_back(beg[lvl] + cnt[lvl] + M - 1);
                return false;
            }
            setc((int)lvl, 0);
            const long long keep_n = m + sc;
            if ((long long)ix.size() > keep_n) ix.resize(keep_n < 0 ? 0 : keep_n);
            lvl = lvl - 1;
            if (lvl == -1) return true;
        }
        return false;
    };
    // ---- what earlier scans already decided.  A cell whose stack with the rows of an ANCESTOR node had radius
    // <= abs_tol / 2 cannot reach abs_tol with more rows added (the set only shrinks; LP values are exact to ~1e-12),
    // so its LP is not issued again below that node: it counts as "no hit", which is what the reference would find.
    // A frame = the rows a scan was made with + the cells that stayed alive; it is used only while the current rows
    // contain all of its rows (checked, because the reference's INDICES arithmetic does not always nest).
    struct Frame { std::vector<int32_t> rows; std::vector<int> alive; };
    std::vector<Frame> frames;
    std::vector<int> mult(nrows, 0);
    std::vector<int> all_cells(N);
    for (int j = 0; j < N; ++j) all_cells[j] = j;
    auto alive_now = [&]() -> const std::vector<int>& {  // needs `cur` resolved
        for (int32_t r : cur) mult[r]++;
        while (!frames.empty()) {
            bool sub = true;
            for (int32_t r : frames.back().rows) if (!mult[r]) { sub = false; break; }
            if (sub) break;
            frames.pop_back();
        }
        for (int32_t r : cur) mult[r]--;
        return frames.empty() ? all_cells : frames.back().alive;
    };
    // How far the speculation goes (A/B knobs; scripts/debug/rdiff_spec_sweep.sh).  Rounds 3-4 tuned it for the
    // fewest batches (SIB = 600, DEEP = 1: 751 batches of 89 724 LPs at config 4); the host pays ~50 ns per speculated list
    // (assembling + storing) and only 8 833 radii are ever read.  Measured: SIB = 100, DEEP = 0 -> 808 batches of 41 335 LPs,
    // the call 21.2-23 -> 18.8-19.2 ms; no speculation beyond the chain (SIB = 0): 985 batches, 19.8-20.6 ms.
    auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    const int spec_sib = env_int("PLP_RDIFF_SPEC_SIB", 100);        // bound on the lists of a cell's sibling chain
    const int spec_chain_scan = env_int("PLP_RDIFF_SPEC_CHAIN_SCAN", 6), spec_chain_node = env_int("PLP_RDIFF_SPEC_CHAIN_NODE", 8);
    const int spec_deep = env_int("PLP_RDIFF_SPEC_DEEP", 0);        // the first alive cell's child scan one level down
    const int spec_child = env_int("PLP_RDIFF_SPEC_CHILD", 1);      // the first child of every alive cell with a scan
    // queue, for every alive cell j >= from: the stack of the current rows with ALL its new rows (the scan, ref
    // :2212-2224) and with its first new row negated (the node the search enters when j is the first hit)
    std::vector<int32_t> child;
    // `required`: the scan lists are what the search is waiting for (queued first, exempt from the bound on a batch);
    // the first children are speculation
    auto queue_scan1 = [&](const std::vector<int>& alive, long long from, const Key& kbase, const std::vector<int32_t>& base,
                           bool required) {
        for (int j : alive) {
            if (j < from) continue;
            suf.resize(mi[j]);
            Key k = kbase;
            for (int t = 0; t < mi[j]; ++t) { suf[t] = (int32_t)(beg[j] + t); k = key_push(k, suf[t]); }
            R.want(k, base.data(), base.size(), suf.data(), suf.size(), required);
        }
        if (spec_child)
        for (int j : alive) {
            if (j < from) continue;
            const int32_t neg = (int32_t)(beg[j] + M);
            R.want(key_push(kbase, neg), base.data(), base.size(), &neg, 1);
        }
    };
    // ... and the same one level further down for the cell the scan will most likely stop at (the first one still
    // alive): when that guess is right the search descends two levels on one batch
    auto queue_scan = [&](const std::vector<int>& alive, long long from, const Key& kbase, bool required) {
        queue_scan1(alive, from, kbase, cur, required);
        for (int j : alive) {
            if (j < from) continue;
            if (j < N - 1 && spec_deep) {
                child = cur;
                child.push_back((int32_t)(beg[j] + M));
                queue_scan1(alive, (long long)j + 1, key_push(kbase, (int32_t)(beg[j] + M)), child, false);
            }
            // ... and its whole sibling chain (rows 1..c-1 of the cell kept, row c negated) with the scans they need when
            // they are not empty, as long as that stays a few hundred lists
            long long n_after = 0;
            for (int a : alive) n_after += a > j;
            if ((long long)mi[j] * (2 * n_after + 1) <= spec_sib) {
                child = cur;
                Key kc = kbase;
                for (int c = 2; c <= mi[j]; ++c) {
                    child.push_back((int32_t)(beg[j] + c - 2));
                    kc = key_push(kc, (int32_t)(beg[j] + c - 2));
                    const int32_t neg = (int32_t)(beg[j] + c - 1 + M);
                    R.want(key_push(kc, neg), child.data(), child.size(), &neg, 1);
                    if (j < N - 1) {
                        child.push_back(neg);
                        queue_scan1(alive, (long long)j + 1, key_push(kc, neg), child, false);
                        child.pop_back();
                    }
                }
            }
            break;
        }
    };
    // queue the nodes the search visits from the state (c2, o2, i2, l2, s2) on while every node turns out empty: its own
    // moves run on the copy, so the lists are exactly the ones it will form (odd turns of the INDICES arithmetic included)
    std::vector<int32_t> sim;
    auto want_state = [&](const std::vector<long long>& i2) -> bool {
        sim.resize(i2.size());
        for (size_t k = 0; k < i2.size(); ++k) {
            long long r = i2[k] < 0 ? i2[k] + nrows : i2[k];
            if (r < 0 || r >= nrows) return false;
            sim[k] = (int32_t)r;
        }
        R.want(key_of_list(sim.data(), sim.size()), sim.data(), sim.size(), nullptr, 0);
        return true;
    };
    auto queue_empty_chain = [&](std::vector<int>& c2, std::vector<int>& o2, std::vector<long long>& i2, long long l2,
                                 long long s2, int steps) {
        for (int step = 0; step < steps; ++step) {
            if (l2 == -1 || c2[at(l2)] == 0) break;               // the next move would be a scan (or the end)
            if (advance_on(c2, o2, i2, l2, s2)) break;
            if (!want_state(i2)) break;
        }
    };
    rc = PLP_OK;
    bool bad_index = false;
    const auto t_search0 = std::chrono::steady_clock::now();
    double t_spec = 0.0;   // seconds spent assembling the lists of a batch (PLP_RDIFF_STATS)
    while (level != -1 && rc == PLP_OK) {
        if (counter[at(level)] == 0) {
            // ---- scan: first cell j >= level whose stack with the current rows is full-dimensional
            resolve();
            if (!cur_ok) { bad_index = true; break; }
            const Key kbase = key_of_list(cur.data(), cur.size());
            const std::vector<int>& alive = alive_now();
            bool have_all = false;
            for (int attempt = 0; attempt < 3; ++attempt) {  // (a full memo is emptied by flush(): ask again then)
                bool miss = false;
                for (int j : alive) {
                    if (j < level) continue;
                    Key k = kbase;
                    for (int t = 0; t < mi[j]; ++t) k = key_push(k, (int32_t)(beg[j] + t));
                    if (!R.memo.find(k)) { miss = true; break; }
                }
                if (!miss) { have_all = true; break; }
                res->n_scan_miss++;
                const auto ts0 = std::chrono::steady_clock::now();
                queue_scan(alive, level, kbase, true);
                {   // the outcome "no cell hits": a piece is emitted, then the node the search re-opens and what follows it
                    std::vector<int> c2 = counter, o2 = open_cells;
                    std::vector<long long> i2 = idx;
                    long long l2 = level, s2 = sumc;
                    if (!leaf_on(c2, o2, i2, l2, s2) && want_state(i2)) queue_empty_chain(c2, o2, i2, l2, s2, spec_chain_scan);
                }
                t_spec += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
                rc = R.flush();
                if (rc) break;
            }
            if (rc) break;
            if (!have_all) {   // the last attempt flushed: everything it stored is in the memo unless the memo cannot hold one scan
                for (int j : alive) {
                    if (j < level) continue;
                    Key k = kbase;
                    for (int t = 0; t < mi[j]; ++t) k = key_push(k, (int32_t)(beg[j] + t));
                    if (!R.memo.find(k)) { rc = fail(PLP_EUNSUPPORTED, "region_diff: a scan over %d cells does not fit the radius memo", N); break; }
                }
                if (rc) break;
            }
            double Rl = 0.0;  // the reference's R after the loop: the radius of the LAST cell looked at
            Frame fr;
            fr.rows = cur;
            long long hit = -1;
            for (int j : alive) {
                if (j < level) continue;
                Key k = kbase;
                for (int t = 0; t < mi[j]; ++t) k = key_push(k, (int32_t)(beg[j] + t));
                const double Rraw = *R.memo.find(k);
                // NaN = the LP ended without a verdict (unbounded ball, iteration limit): the reference reads radius 0
                // there (ref :1294-1297) and solves the cell again at every node below, so only a cell whose LP was
                // SOLVED with a radius <= abs_tol / 2 is dropped from the scans below this node
                if (!(Rraw <= 0.5 * abs_tol)) fr.alive.push_back(j);
                const double Rj = Rraw != Rraw ? 0.0 : Rraw;
                if (hit < 0) {
                    res->n_requests++;
                    if (Rj > abs_tol) { hit = j; Rl = Rj; }
                    else if (j == N - 1) Rl = Rj;
                }
            }
            frames.push_back(std::move(fr));
            if (hit >= 0) {
                level = hit;
                set_counter((int)level, 1);
                idx.push_back(beg[level] + M);
            }
            if (Rl < abs_tol) {  // nothing left to subtract: the current rows are a piece (ref :2226-2245)
                emit(0);
                if (leaf_on(counter, open_cells, idx, level, sumc)) break;
            }
        } else {
            // ---- next sibling of the deepest open cell, closing exhausted cells on the way (ref :2246-2271)
            if (advance_on(counter, open_cells, idx, level, sumc)) break;
        }
        // ---- the node itself
        resolve();
        if (!cur_ok) { bad_index = true; break; }
        const Key knode = key_of_list(cur.data(), cur.size());
        if (!R.memo.find(knode)) {
            res->n_node_miss++;
            const auto ts0 = std::chrono::steady_clock::now();
            R.want(knode, cur.data(), cur.size(), nullptr, 0, true);
            {
                std::vector<int> c2 = counter, o2 = open_cells;
                std::vector<long long> i2 = idx;
                queue_empty_chain(c2, o2, i2, level, sumc, spec_chain_node);
            }
            // what it needs next when it is NOT empty: its scan and the first child of every cell still alive
            if (level >= 0 && level < N - 1) queue_scan(alive_now(), level + 1, knode, false);
            t_spec += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
            rc = R.flush();
            if (rc) break;
        }
        const double* pnode = R.memo.find(knode);
        if (!pnode) { rc = fail(PLP_EHIP, "region_diff: the radius of the current node is missing after its batch"); break; }
        const double rcv = *pnode != *pnode ? 0.0 : *pnode;
        res->n_requests++;
        if (rcv > abs_tol) {
            if (level == N - 1) emit(1);
            else level = level + 1;
        }
    }
    res->n_lps = R.n_lps;
    res->n_batches = R.n_batches;
    if (getenv("PLP_RDIFF_STATS"))
        fprintf(stderr, "plp_region_diff_search: %lld LPs, %lld batches (%lld scan misses, %lld node misses), %lld requests, launch %.1f ms, device wait %.1f ms; beyond 64 rows: %lld LPs in %lld batches; resident server: %lld batches, %lld starts; search loop %.1f ms of which assembling the batches' lists %.1f ms, storing results %.1f ms\n",
                R.n_lps, R.n_batches, res->n_scan_miss, res->n_node_miss, res->n_requests, R.t_launch * 1e3, R.t_wait * 1e3,
                R.n_lps_long, R.n_batches_long, R.n_srv_batches, R.n_srv_starts,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_search0).count() * 1e3, t_spec * 1e3, R.t_store * 1e3);
    R.release();
    if (rc == PLP_OK && bad_index) rc = fail(PLP_EINVAL, "region_diff: row index out of range (the reference raises IndexError here)");
    if (rc) { delete res; return rc; }
    *out = res;
    return PLP_OK;
}

int plp_rdiff_result_sizes(const plp_rdiff_result* r, int64_t* n_leaves, int64_t* n_rows, int64_t* n_lps, int64_t* n_batches) {
    if (!r) return fail(PLP_EINVAL, "NULL result");
    if (n_leaves) *n_leaves = (int64_t)r->kind.size();
    if (n_rows) *n_rows = (int64_t)r->rows.size();
    if (n_lps) *n_lps = r->n_lps;
    if (n_batches) *n_batches = r->n_batches;
    return PLP_OK;
}

int plp_rdiff_result_copy(const plp_rdiff_result* r, int32_t* kind, int32_t* off, int32_t* rows) {
    if (!r || !kind || !off || !rows) return fail(PLP_EINVAL, "NULL pointer");
    if (!r->kind.empty()) memcpy(kind, r->kind.data(), r->kind.size() * 4);
    memcpy(off, r->off.data(), r->off.size() * 4);
    if (!r->rows.empty()) memcpy(rows, r->rows.data(), r->rows.size() * 4);
    return PLP_OK;
}

int plp_rdiff_result_free(plp_rdiff_result* r) {
    delete r;
    return PLP_OK;
}

}  // extern "C"
