// Causal depthwise convolution over time (kernel <= 32) for gfx950.
//   y[b, t, c] = sum_j w[j][c] * x[b, t - (K-1) + j, c],   rows before the utterance start read as 0.
// A thread owns two adjacent channels and kTimeTile consecutive time steps: it loads the
// (kTimeTile + K - 1) input rows once (each wave instruction is a coalesced 256-byte row segment),
// keeps them in registers as float32 and produces all its outputs from them, so every input row is
// fetched (kTimeTile + K - 1) / kTimeTile times instead of K times.  HBM-bound.
#include "oasr_common.h"

namespace oasr {

constexpr int kTimeTile = 16;
constexpr int kMaxTaps = 32;

template <int K>
__global__ __launch_bounds__(256) void dwconv_causal_kernel(const uint16_t* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            float* __restrict__ y, int64_t t_pad,
                                                            int d) {
    constexpr int kRows = kTimeTile + K - 1;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * kTimeTile;
    const int c = (blockIdx.y * 256 + threadIdx.x) * 2;
    const uint16_t* xb = x + (int64_t)b * t_pad * d + c;
    float in0[kRows], in1[kRows];
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
        const int t = t0 - (K - 1) + i;
        uint32_t v = 0;
        if (t >= 0) v = *reinterpret_cast<const uint32_t*>(xb + (int64_t)t * d);
        in0[i] = __uint_as_float(v << 16);
        in1[i] = __uint_as_float(v & 0xFFFF0000u);
    }
    float a0[kTimeTile], a1[kTimeTile];
#pragma unroll
    for (int i = 0; i < kTimeTile; ++i) {
        a0[i] = 0.f;
        a1[i] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const float2 wj = *reinterpret_cast<const float2*>(w + (int64_t)j * d + c);
#pragma unroll
        for (int i = 0; i < kTimeTile; ++i) {
            a0[i] = fmaf(wj.x, in0[i + j], a0[i]);
            a1[i] = fmaf(wj.y, in1[i + j], a1[i]);
        }
    }
    float* yb = y + ((int64_t)b * t_pad + t0) * d + c;
#pragma unroll
    for (int i = 0; i < kTimeTile; ++i)
        *reinterpret_cast<float2*>(yb + (int64_t)i * d) = make_float2(a0[i], a1[i]);
}

}  // namespace oasr

using namespace oasr;

extern "C" int oasr_dwconv_causal(const uint16_t* x, const float* w, float* y, int32_t batch,
                                  int64_t t_pad, int32_t d, int32_t kernel, void* stream) {
    OASR_REQUIRE(x && w && y, "dwconv: null pointer");
    OASR_REQUIRE(batch > 0 && batch <= 65535, "dwconv: bad batch");
    OASR_REQUIRE(t_pad > 0 && t_pad % kTimeTile == 0, "dwconv: t_pad must be a multiple of %d", kTimeTile);
    OASR_REQUIRE(d > 0 && d % 512 == 0 && d / 512 <= 65535, "dwconv: d must be a multiple of 512");
    dim3 grid((unsigned)(t_pad / kTimeTile), (unsigned)(d / 512), (unsigned)batch), blk(256);
    hipStream_t st = (hipStream_t)stream;
    switch (kernel) {
        case 31: hipLaunchKernelGGL((dwconv_causal_kernel<31>), grid, blk, 0, st, x, w, y, t_pad, d); break;
        case 15: hipLaunchKernelGGL((dwconv_causal_kernel<15>), grid, blk, 0, st, x, w, y, t_pad, d); break;
        case 7: hipLaunchKernelGGL((dwconv_causal_kernel<7>), grid, blk, 0, st, x, w, y, t_pad, d); break;
        case 3: hipLaunchKernelGGL((dwconv_causal_kernel<3>), grid, blk, 0, st, x, w, y, t_pad, d); break;
        default:
            set_error("dwconv: unsupported kernel size %d (3, 7, 15, 31)", kernel);
            return OASR_EINVAL;
    }
    OASR_CHECK_LAUNCH("dwconv");
    return OASR_OK;
}
