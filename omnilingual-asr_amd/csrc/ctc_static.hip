// oasr_ctc_greedy_static: greedy CTC decode for callers that cannot allocate device memory (code
// under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same kernels as
// oasr_ctc_greedy (ctc.hip); the per-frame winners between the two launches go to a library-owned
// array instead of a caller workspace.
//
// The array is a file-scope __device__ variable: each GPU the library runs on holds its own copy (4 MiB)
// at its own address until the process ends.  In host code the variable's name is only a stand-in,
// never a device address: the address of the current device's copy comes from hipGetSymbolAddress,
// resolved once per device on the entry point's first call there, and that pointer is what the
// kernels receive.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two
// streams, two threads) race on it.  Nothing needs clearing between calls: the first launch writes
// every entry below batch * t_pad before the second reads it.
#include <mutex>

#include "oasr_common.h"

namespace oasr {

constexpr int64_t kCtcStaticFrames = (int64_t)1 << 20;   // batch * t_pad limit: 1 Mi frames = 4 MiB
constexpr int kMaxDevices = 64;

static __device__ __align__(256) int32_t scratch[kCtcStaticFrames];

// Device address of the current device's copy of `scratch`.
static int static_ids(int32_t** out) {
    static std::mutex mu;
    static int32_t* addr[kMaxDevices] = {};
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        set_error("ctc_greedy_static: no usable current device (%s)", hipGetErrorString(e));
        return OASR_EHIP;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!addr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            set_error("ctc_greedy_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return OASR_EHIP;
        }
        addr[dev] = static_cast<int32_t*>(p);
    }
    *out = addr[dev];
    return OASR_OK;
}

}  // namespace oasr

using namespace oasr;

extern "C" int64_t oasr_ctc_static_max_frames(void) { return kCtcStaticFrames; }

extern "C" int oasr_ctc_greedy_static(const float* logits, int64_t ld, int32_t vocab, const int32_t* lengths,
                                      int32_t batch, int64_t t_pad, int32_t blank, int32_t* tokens,
                                      int32_t* frames, int32_t* counts, void* stream) {
    int rc = ctc_check_args(logits, ld, vocab, lengths, batch, t_pad, blank, tokens, frames, counts);
    if (rc != OASR_OK) return rc;
    if ((int64_t)batch * t_pad > kCtcStaticFrames) {
        set_error("ctc_greedy_static: batch * t_pad = %lld frames exceeds the built-in scratch of %lld frames; "
                  "use oasr_ctc_greedy with a caller workspace",
                  (long long)((int64_t)batch * t_pad), (long long)kCtcStaticFrames);
        return OASR_ECAPACITY;
    }
    int32_t* ids = nullptr;
    rc = static_ids(&ids);
    if (rc != OASR_OK) return rc;
    return ctc_launch(logits, ld, vocab, lengths, batch, t_pad, blank, tokens, frames, counts, ids,
                      (hipStream_t)stream);
}
