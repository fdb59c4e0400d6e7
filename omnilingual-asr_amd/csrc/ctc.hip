// CTC output stage for gfx950: bf16 copy of the encoder states (operand of the vocabulary
// projection GEMM) and greedy decoding of the projected logits.
//
// Greedy decoding per utterance: frame-wise argmax over the vocabulary (lowest index wins a tie),
// then frames equal to their predecessor and blank frames are dropped.  Two launches:
//   1. ctc_argmax_kernel — over all batch * t_pad frames at once.  A group of 1..64 lanes (a power of
//      two that covers the vocabulary in one or more 16-byte loads per lane) owns one frame, so a
//      wavefront reads 64 consecutive float4 of the logits matrix whatever the vocabulary size, and
//      reduces (value, index) pairs with lane exchanges inside the group.  One int32 winner per frame
//      goes to the `ids` scratch array (frames past an utterance's length are written as blank, so
//      the array is fully defined after every call).
//   2. ctc_collapse_kernel — one work-group per utterance walks its ids in tiles of 2048 frames
//      (thread = 8 consecutive frames, predecessor read from the array itself), block-wide exclusive
//      scan of the survivors per tile, running total carried across tiles: no limit on t_pad.
// HBM-bound: batch * t_pad * ld * 4 bytes of logits read once; a few bytes per frame otherwise.
// Integer results: bit-exact against any correct host decode.
#include "oasr_common.h"

namespace oasr {

constexpr int kCtcThreads = 256;
constexpr int kCtcPerThread = 8;
constexpr int kCtcTile = kCtcThreads * kCtcPerThread;

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y,
                                                        int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<uint2*>(y)[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}

__global__ __launch_bounds__(kCtcThreads) void ctc_argmax_kernel(
    const float* __restrict__ logits, int64_t ld, int vocab, const int32_t* __restrict__ lengths, int t_pad,
    int64_t total_frames, int blank, int lanes_log2, int32_t* __restrict__ ids) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int group = 1 << lanes_log2, sub = lane & (group - 1);
    const int64_t wave = (int64_t)blockIdx.x * (kCtcThreads / 64) + (tid >> 6);
    const int64_t f = (wave << (6 - lanes_log2)) + (lane >> lanes_log2);     // global frame index
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    bool live = false;
    if (f < total_frames) {
        const int b = (int)(f / t_pad), t = (int)(f - (int64_t)b * t_pad);
        live = t < min(lengths[b], t_pad);
        if (live) {
            const float* row = logits + f * ld;
            for (int c = 4 * sub; c < vocab; c += 4 * group) {
                const float4 v = *reinterpret_cast<const float4*>(row + c);   // ld >= round_up(vocab, 4)
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c + q < vocab && (e[q] > best || (e[q] == best && c + q < best_i))) {
                        best = e[q];
                        best_i = c + q;
                    }
            }
        }
    }
    for (int o = group >> 1; o > 0; o >>= 1) {      // every lane of the wave takes part
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        if (ov > best || (ov == best && oi < best_i)) {
            best = ov;
            best_i = oi;
        }
    }
    // a row of NaNs decodes as blank, and so does every frame past the utterance's length
    if (sub == 0 && f < total_frames) ids[f] = (live && best_i != 0x7fffffff) ? best_i : blank;
}

__global__ __launch_bounds__(kCtcThreads) void ctc_collapse_kernel(
    const int32_t* __restrict__ ids, const int32_t* __restrict__ lengths, int t_pad, int blank,
    int32_t* __restrict__ tokens, int32_t* __restrict__ frames, int32_t* __restrict__ counts) {
    __shared__ int32_t wave_total[kCtcThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len = min(lengths[b], t_pad);
    const int32_t* row = ids + (int64_t)b * t_pad;
    int32_t* tok = tokens + (int64_t)b * t_pad;
    int32_t* frm = frames + (int64_t)b * t_pad;
    int carry = 0;                                   // survivors of the tiles already done
    for (int base = 0; base < len; base += kCtcTile) {
        const int t0 = base + tid * kCtcPerThread;
        int id[kCtcPerThread];
        int prev = (t0 > 0 && t0 < len) ? row[t0 - 1] : -1;   // -1 equals no token id
        unsigned keep = 0;
#pragma unroll
        for (int j = 0; j < kCtcPerThread; ++j) {
            id[j] = t0 + j < len ? row[t0 + j] : blank;
            if (id[j] != blank && id[j] != prev) keep |= 1u << j;
            prev = id[j];
        }
        const int kept = __popc(keep);
        int incl = kept;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        __syncthreads();                             // the previous tile's totals have been read
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        int offset = carry + incl - kept, total = 0;
#pragma unroll
        for (int w = 0; w < kCtcThreads / 64; ++w) {
            if (w < wave) offset += wave_total[w];
            total += wave_total[w];
        }
#pragma unroll
        for (int j = 0; j < kCtcPerThread; ++j)
            if (keep & (1u << j)) {
                tok[offset] = id[j];
                frm[offset] = t0 + j;
                ++offset;
            }
        carry += total;
    }
    // unused tail: -1, so the outputs are fully defined
    for (int i = carry + tid; i < t_pad; i += kCtcThreads) {
        tok[i] = -1;
        frm[i] = -1;
    }
    if (tid == 0) counts[b] = carry;
}

int ctc_check_args(const float* logits, int64_t ld, int32_t vocab, const int32_t* lengths, int32_t batch,
                   int64_t t_pad, int32_t blank, int32_t* tokens, int32_t* frames, int32_t* counts) {
    OASR_REQUIRE(logits && lengths && tokens && frames && counts, "ctc_greedy: null pointer");
    OASR_REQUIRE(batch > 0 && t_pad > 0 && t_pad < ((int64_t)1 << 30),
                 "ctc_greedy: batch and t_pad must be positive (t_pad below 2^30)");
    OASR_REQUIRE(vocab > 0 && blank >= 0 && blank < vocab, "ctc_greedy: bad vocabulary size or blank index");
    OASR_REQUIRE(ld % 4 == 0 && ld >= (vocab + 3) / 4 * 4, "ctc_greedy: ld must be a multiple of 4 and cover the vocabulary");
    OASR_REQUIRE(((uintptr_t)logits & 15) == 0, "ctc_greedy: logits must be 16-byte aligned");
    return OASR_OK;
}

int ctc_launch(const float* logits, int64_t ld, int32_t vocab, const int32_t* lengths, int32_t batch,
               int64_t t_pad, int32_t blank, int32_t* tokens, int32_t* frames, int32_t* counts,
               int32_t* ids, hipStream_t stream) {
    const int64_t total = (int64_t)batch * t_pad;
    int lanes_log2 = 0;
    while (lanes_log2 < 6 && (4 << lanes_log2) < vocab) ++lanes_log2;
    const int64_t frames_per_wg = (int64_t)(kCtcThreads / 64) << (6 - lanes_log2);
    const int64_t blocks = (total + frames_per_wg - 1) / frames_per_wg;
    OASR_REQUIRE(blocks < ((int64_t)1 << 31), "ctc_greedy: too many frames");
    hipLaunchKernelGGL(ctc_argmax_kernel, dim3((unsigned)blocks), dim3(kCtcThreads), 0, stream, logits, ld, vocab,
                       lengths, (int)t_pad, total, blank, lanes_log2, ids);
    OASR_CHECK_LAUNCH("ctc_greedy (argmax)");
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3((unsigned)batch), dim3(kCtcThreads), 0, stream, ids, lengths,
                       (int)t_pad, blank, tokens, frames, counts);
    OASR_CHECK_LAUNCH("ctc_greedy (collapse)");
    return OASR_OK;
}

}  // namespace oasr

using namespace oasr;

extern "C" int oasr_cast_bf16(const float* x, uint16_t* y, int64_t n, void* stream) {
    OASR_REQUIRE(x && y, "cast_bf16: null pointer");
    OASR_REQUIRE(n >= 0 && n % 4 == 0, "cast_bf16: n must be a non-negative multiple of 4");
    OASR_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0, "cast_bf16: misaligned pointer");
    if (n == 0) return OASR_OK;
    const int64_t n4 = n / 4, blocks = (n4 + 255) / 256;
    OASR_REQUIRE(blocks < ((int64_t)1 << 31), "cast_bf16: too many elements");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n4);
    OASR_CHECK_LAUNCH("cast_bf16");
    return OASR_OK;
}

extern "C" int64_t oasr_ctc_workspace_bytes(int32_t batch, int64_t t_pad) {
    if (batch <= 0 || t_pad <= 0) return 0;
    return (int64_t)batch * t_pad * (int64_t)sizeof(int32_t);
}

extern "C" int oasr_ctc_greedy(const float* logits, int64_t ld, int32_t vocab, const int32_t* lengths,
                               int32_t batch, int64_t t_pad, int32_t blank, int32_t* tokens,
                               int32_t* frames, int32_t* counts, void* workspace, void* stream) {
    int rc = ctc_check_args(logits, ld, vocab, lengths, batch, t_pad, blank, tokens, frames, counts);
    if (rc != OASR_OK) return rc;
    OASR_REQUIRE(workspace && ((uintptr_t)workspace & 3) == 0, "ctc_greedy: workspace missing or misaligned");
    return ctc_launch(logits, ld, vocab, lengths, batch, t_pad, blank, tokens, frames, counts,
                      static_cast<int32_t*>(workspace), (hipStream_t)stream);
}
