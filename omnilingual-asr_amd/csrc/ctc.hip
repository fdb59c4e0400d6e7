// CTC output stage for gfx950: bf16 copy of the encoder states (operand of the vocabulary
// projection GEMM) and greedy decoding of the projected logits.
//
// Greedy decoding per utterance: frame-wise argmax over the vocabulary (lowest index wins a tie),
// then frames equal to their predecessor and blank frames are dropped.  One work-group owns one
// utterance: a wavefront reads one logits row per step (coalesced, 16 bytes per lane) and reduces
// (value, index) pairs with lane exchanges; the per-frame winners stay in LDS; the survivors'
// output positions come from a block-wide exclusive scan.  HBM-bound: t_pad * ld * 4 bytes read per
// utterance, a few kilobytes written.  Integer results: bit-exact against any correct host decode.
#include "oasr_common.h"

namespace oasr {

constexpr int kCtcThreads = 256;
constexpr int kCtcMaxFrames = 8192;   // LDS: one int32 per frame

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y,
                                                        int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<uint2*>(y)[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}

__global__ __launch_bounds__(kCtcThreads) void ctc_greedy_kernel(
    const float* __restrict__ logits, int64_t ld, int vocab, const int32_t* __restrict__ lengths,
    int t_pad, int blank, int32_t* __restrict__ tokens, int32_t* __restrict__ frames,
    int32_t* __restrict__ counts) {
    __shared__ int32_t ids[kCtcMaxFrames];
    __shared__ int32_t wave_total[kCtcThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len = min(lengths[b], t_pad);
    const float* base = logits + (int64_t)b * t_pad * ld;
    // ---- frame-wise argmax: wave per frame ----------------------------------------------------------
    for (int t = wave; t < len; t += kCtcThreads / 64) {
        const float* row = base + (int64_t)t * ld;
        float best = -INFINITY;
        int best_i = 0x7fffffff;
        for (int c = 4 * lane; c < vocab; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(row + c);   // ld >= round_up(vocab, 4)
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c + q < vocab && (e[q] > best || (e[q] == best && c + q < best_i))) {
                    best = e[q];
                    best_i = c + q;
                }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(best_i, o, 64);
            if (ov > best || (ov == best && oi < best_i)) {
                best = ov;
                best_i = oi;
            }
        }
        if (lane == 0) ids[t] = best_i == 0x7fffffff ? blank : best_i;   // a row of NaNs decodes as blank
    }
    __syncthreads();
    // ---- collapse: thread owns a contiguous run of frames ---------------------------------------------
    const int per = (t_pad + kCtcThreads - 1) / kCtcThreads;
    const int t0 = tid * per, t1 = min(t0 + per, len);
    int kept = 0;
    for (int t = t0; t < t1; ++t) {
        const int id = ids[t];
        kept += (id != blank && (t == 0 || id != ids[t - 1])) ? 1 : 0;
    }
    // exclusive scan of `kept` over the block
    int incl = kept;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    int offset = incl - kept;
    for (int w = 0; w < wave; ++w) offset += wave_total[w];
    int total = 0;
    for (int w = 0; w < kCtcThreads / 64; ++w) total += wave_total[w];
    int32_t* tok = tokens + (int64_t)b * t_pad;
    int32_t* frm = frames + (int64_t)b * t_pad;
    for (int t = t0; t < t1; ++t) {
        const int id = ids[t];
        if (id != blank && (t == 0 || id != ids[t - 1])) {
            tok[offset] = id;
            frm[offset] = t;
            ++offset;
        }
    }
    // unused tail: -1, so the outputs are fully defined
    for (int i = total + tid; i < t_pad; i += kCtcThreads) {
        tok[i] = -1;
        frm[i] = -1;
    }
    if (tid == 0) counts[b] = total;
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

extern "C" int oasr_ctc_greedy(const float* logits, int64_t ld, int32_t vocab, const int32_t* lengths,
                               int32_t batch, int64_t t_pad, int32_t blank, int32_t* tokens,
                               int32_t* frames, int32_t* counts, void* stream) {
    OASR_REQUIRE(logits && lengths && tokens && frames && counts, "ctc_greedy: null pointer");
    OASR_REQUIRE(batch > 0 && t_pad > 0 && t_pad <= kCtcMaxFrames,
                 "ctc_greedy: batch must be positive and t_pad in 1..%d", kCtcMaxFrames);
    OASR_REQUIRE(vocab > 0 && blank >= 0 && blank < vocab, "ctc_greedy: bad vocabulary size or blank index");
    OASR_REQUIRE(ld % 4 == 0 && ld >= (vocab + 3) / 4 * 4, "ctc_greedy: ld must be a multiple of 4 and cover the vocabulary");
    OASR_REQUIRE(((uintptr_t)logits & 15) == 0, "ctc_greedy: logits must be 16-byte aligned");
    hipLaunchKernelGGL(ctc_greedy_kernel, dim3((unsigned)batch), dim3(kCtcThreads), 0, (hipStream_t)stream,
                       logits, ld, vocab, lengths, (int)t_pad, blank, tokens, frames, counts);
    OASR_CHECK_LAUNCH("ctc_greedy");
    return OASR_OK;
}
