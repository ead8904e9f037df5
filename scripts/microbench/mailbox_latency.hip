// mailbox_latency.hip -- what a resident kernel pays to talk to the host through host-mapped memory (MI355X, PCIe).
//   hipcc --offload-arch=gfx950 -O3 -o mailbox_latency mailbox_latency.hip && ./mailbox_latency
// One workgroup (or NW of them) polls a mailbox word in host memory; on a new value it (A) echoes it at once,
// (B) first reads a second host word (a record), (C) stores a payload word, waits for the acknowledgement, then echoes.
// The host measures round trips of 2000 pings each.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void server(const unsigned long long* mail, const unsigned long long* rec, unsigned long long* out, int mode,
                       unsigned long long last, unsigned idle) {
    unsigned long long seen = last;
    unsigned polls = 0;
    for (;;) {
        unsigned long long w = 0;
        if (threadIdx.x == 0) w = __hip_atomic_load(mail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        w = __shfl(w, 0, 64);
        if (w == ~0ull) break;
        if (w == seen) { if (++polls > idle) break; continue; }
        polls = 0;
        seen = w;
        if (blockIdx.x != 0) continue;
        unsigned long long v = w;
        if (mode == 1 || mode == 3) v += __hip_atomic_load(rec + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) * 0ull;
        if (threadIdx.x == 0) {
            if (mode >= 2) {
                __hip_atomic_store(out + 8, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_store(out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void tiny(unsigned long long* out, unsigned long long v) {
    if (threadIdx.x == 0) __hip_atomic_store(out, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main() {
    char* h; char* d;
    CK(hipHostMalloc((void**)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d, h, 0));
    volatile unsigned long long* mail = (volatile unsigned long long*)h;
    volatile unsigned long long* out = (volatile unsigned long long*)(h + 1024);
    hipStream_t st; CK(hipStreamCreate(&st));
    const int N = 2000;
    unsigned long long seq = 0;
    for (int nw : {1, 16, 64}) for (int mode = 0; mode < 4; ++mode) {
        mail[0] = seq; out[0] = seq;
        hipLaunchKernelGGL(server, dim3(nw), dim3(64), 0, st, (const unsigned long long*)d, (const unsigned long long*)(d + 2048),
                           (unsigned long long*)(d + 1024), mode, seq, 4000000u);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            ++seq;
            __atomic_store_n((unsigned long long*)mail, seq, __ATOMIC_RELEASE);
            while (__atomic_load_n((unsigned long long*)out, __ATOMIC_ACQUIRE) != seq) {}
        }
        auto t1 = std::chrono::steady_clock::now();
        __atomic_store_n((unsigned long long*)mail, ~0ull, __ATOMIC_RELEASE);
        CK(hipStreamSynchronize(st));
        printf("resident, %2d pollers, mode %d (%s): %.2f us per round trip\n", nw, mode,
               mode == 0 ? "echo" : mode == 1 ? "host read, echo" : mode == 2 ? "store, ack, echo" : "host read, store, ack, echo",
               std::chrono::duration<double>(t1 - t0).count() / N * 1e6);
    }
    // the launch it replaces: one tiny kernel per ping, completion seen through the same kind of word
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
        ++seq;
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, (unsigned long long*)(d + 1024), seq);
        while (__atomic_load_n((unsigned long long*)out, __ATOMIC_ACQUIRE) != seq) {}
    }
    auto t1 = std::chrono::steady_clock::now();
    printf("one launch per ping: %.2f us\n", std::chrono::duration<double>(t1 - t0).count() / N * 1e6);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
        ++seq;
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, (unsigned long long*)(d + 1536), seq);
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, (unsigned long long*)(d + 1024), seq);
        while (__atomic_load_n((unsigned long long*)out, __ATOMIC_ACQUIRE) != seq) {}
    }
    t1 = std::chrono::steady_clock::now();
    printf("two dependent launches per ping: %.2f us\n", std::chrono::duration<double>(t1 - t0).count() / N * 1e6);
    return 0;
}
