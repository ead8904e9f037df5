// plp_stage.hpp -- large host-pointer batches: the inputs cross PCIe in chunks while the kernels of earlier chunks run.
//
// A pageable hipMemcpyAsync is staged by the runtime on the calling thread (measured 17 GB/s for the 51 MB of a C2
// batch, and nothing else happens meanwhile: 3.0 ms per pass of which the kernel is 0.26).  Here a small pool of host
// threads copies the caller's arrays slice by slice (chunk-major, so chunk 0 is complete first) into a pinned staging
// buffer of the context; as soon as a chunk is staged its arrays go out on a copy stream, and the chunk's kernel waits
// for them on the compute stream through an event.  Upload, staging and kernels overlap; the call still returns with
// host-visible results (the caller's copy_out synchronises the compute stream).
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace plp {

struct StagePiece {   // one array of one chunk: the pool copies src -> dst on the host
    const char* src;  // upload: caller's memory; download: staging buffer
    char* dst;        // upload: staging buffer; download: caller's memory
    char* dev;        // upload: device destination of dst (unused for downloads)
    size_t bytes;
    bool check_f64 = false;  // the bytes are doubles: note inf / nan while copying (nonfinite())
};

// memcpy of doubles that also reports whether any of them is inf or nan (exponent field all ones)
inline bool copy_f64_checked(char* dst, const char* src, size_t bytes) {
    const size_t n = bytes / 8;
    unsigned long long acc = 0;
    for (size_t i = 0; i < n; ++i) {
        unsigned long long v;
        memcpy(&v, src + 8 * i, 8);
        memcpy(dst + 8 * i, &v, 8);
        acc |= (((v >> 52) & 0x7ffull) + 1ull) >> 11;  // 1 iff the exponent is 0x7ff
    }
    return acc != 0;
}

inline bool any_nonfinite_f64(const char* src, size_t bytes) {
    const size_t n = bytes / 8;
    unsigned long long acc = 0;
    for (size_t i = 0; i < n; ++i) {
        unsigned long long v;
        memcpy(&v, src + 8 * i, 8);
        acc |= (((v >> 52) & 0x7ffull) + 1ull) >> 11;
    }
    return acc != 0;
}

class StagePool {
public:
    static constexpr size_t SLICE = 256u << 10;
    explicit StagePool(int nthreads) {
        for (int t = 0; t < nthreads; ++t) th_.emplace_back([this] { worker(); });
    }
    ~StagePool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // chunks[i] = the pieces of chunk i.  Returns at once; chunk i has been copied when ready(i).
    void start(const std::vector<std::vector<StagePiece>>& chunks) {
        {
            // workers enter the item list only after registering under the mutex: with it held and nobody registered
            // the list can be rebuilt (a worker that wakes late for the previous job finds this one)
            std::lock_guard<std::mutex> lk(mu_);
            while (busy_.load(std::memory_order_acquire) != 0) sched_yield();
            items_.clear();
            const int nc = (int)chunks.size();
            left_.reset(new std::atomic<int>[nc]);
            for (int i = 0; i < nc; ++i) {
                int n = 0;
                for (const StagePiece& p : chunks[i])
                    for (size_t o = 0; o < p.bytes; o += SLICE, ++n)
                        items_.push_back({p.src + o, p.dst + o, p.bytes - o < SLICE ? p.bytes - o : SLICE, i, p.check_f64});
                left_[i].store(n, std::memory_order_relaxed);
            }
            nonfinite_.store(0, std::memory_order_relaxed);
            next_.store(0, std::memory_order_release);
            ++gen_;
        }
        cv_.notify_all();
    }
    bool ready(int chunk) const { return left_[chunk].load(std::memory_order_acquire) == 0; }
    // the calling thread helps until the chunk is staged
    void wait(int chunk) {
        while (!ready(chunk))
            if (!take_one()) sched_yield();
    }
    // every item copied and every worker out of the item list (the list may be rebuilt afterwards)
    void finish() {
        while (take_one()) {}
        while (busy_.load(std::memory_order_acquire) != 0) sched_yield();
    }
    // a checked piece of the last job held an inf or a nan (complete after finish(); per chunk after ready(chunk))
    bool nonfinite() const { return nonfinite_.load(std::memory_order_acquire) != 0; }

private:
    struct Item { const char* src; char* dst; size_t bytes; int chunk; bool check; };
    bool take_one() {
        const int k = next_.fetch_add(1, std::memory_order_relaxed);
        if (k >= (int)items_.size()) return false;
        const Item& it = items_[k];
        if (it.check) {
            if (copy_f64_checked(it.dst, it.src, it.bytes)) nonfinite_.store(1, std::memory_order_relaxed);
        } else {
            memcpy(it.dst, it.src, it.bytes);
        }
        left_[it.chunk].fetch_sub(1, std::memory_order_release);
        return true;
    }
    void worker() {
        unsigned seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                busy_.fetch_add(1, std::memory_order_acq_rel);
            }
            while (take_one()) {}
            busy_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
    unsigned gen_ = 0;
    std::vector<Item> items_;
    std::unique_ptr<std::atomic<int>[]> left_;
    std::atomic<int> next_{0};
    std::atomic<int> busy_{0};
    std::atomic<int> nonfinite_{0};
};

}  // namespace plp
