// A reduction record the device writes STRAIGHT into pinned host memory, read by polling.
//
// The adaptive solvers decide on the host (accept / reject, next step size: section 2 of DESIGN.md says why the controller stays
// there) and so wait once per attempted step for 16 bytes.  As a copy + event that wait measured ~45 us of idle GPU per step
// (tools/micro/bench_gaps.sh C5: a blit kernel for the copy, the event's signal, the wake-up of hipEventSynchronize) - 10 % of a
// 0.4 ms step, a third of a README-sized solve.  Here the last kernel of the reduction stores its result through a device alias of
// a hipHostMalloc'ed (fine-grained: uncached on the device, coherent with the host) buffer that the host armed with a sentinel before
// enqueuing; the host spins on the sentinel.  The stream is queried while waiting: a fault ends the wait with an error, not a hang.
#pragma once
#include <stdint.h>
#include <string.h>
#include <chrono>

#include "common.h"

namespace ndcn {

constexpr uint64_t kRecSentinel = 0x7ff8dead0000beefull;       // a NaN payload no arithmetic produces

inline bool poll_records_enabled() {
    static const bool on = [] { const char *e = getenv("NDCN_POLL_RECORD"); return !(e && e[0] == '0'); }();
    return on;
}

inline void rec_arm(double *h, int n) {
    volatile uint64_t *p = reinterpret_cast<volatile uint64_t *>(h);
    for (int i = 0; i < n; ++i) p[i] = kRecSentinel;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
}

// wait until the n doubles at h have all been written by the device (launches enqueued on `st` after rec_arm)
inline int rec_wait(const double *h, int n, hipStream_t st) {
    const volatile uint64_t *p = reinterpret_cast<const volatile uint64_t *>(h);
    auto pending = [&] {
        for (int i = 0; i < n; ++i)
            if (p[i] == kRecSentinel) return true;
        return false;
    };
    const auto t0 = std::chrono::steady_clock::now();
    int drained_polls = 0;
    for (uint64_t spin = 1;; ++spin) {
        if (!pending()) break;
        __builtin_ia32_pause();
        if ((spin & 0xfff) == 0) {
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us > 500.0) {
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) {                  // everything enqueued has run: the record must be there
                    if (++drained_polls > 64 && pending()) {
                        set_error("reduction record never arrived in host memory (stream idle)");
                        return NDCN_EHIP;
                    }
                } else if (q != hipErrorNotReady) {
                    set_error("stream failed while waiting for a reduction record: %s", hipGetErrorString(q));
                    return NDCN_EHIP;
                }
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    return NDCN_OK;
}

}  // namespace ndcn
