// Causal depthwise convolution over time (kernel <= 32) for gfx950.
//   y[b, t, c] = sum_j w[j][c] * x[b, t - (K-1) + j, c],   rows before the utterance start read as 0.
// A thread owns two adjacent channels and kTimeTile consecutive time steps: it loads the
// (kTimeTile + K - 1) input rows once (each wave instruction is a coalesced 256-byte row segment),
// keeps them in registers as float32 and produces all its outputs from them, so every input row is
// fetched (kTimeTile + K - 1) / kTimeTile times instead of K times.  HBM-bound.
//
// dwconv_ln_swish_kernel fuses the convolution with the LayerNorm + Swish that follows it in the
// conformer's convolution module: a work-group owns kTimeTile rows x all d channels, the float32
// convolution outputs cross LDS once so that each wavefront then owns whole rows, and only the
// bf16 result goes back to memory (the separate kernels wrote and re-read a float32 copy).
#include "oasr_common.h"
#include "row_norm.h"

namespace oasr {

constexpr int kTimeTile = 16;

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
    uint32_t raw[kRows];
    if (t0 >= K - 1) {   // no row before the utterance start: no test per row
        const uint16_t* xt = xb + (int64_t)(t0 - (K - 1)) * d;
#pragma unroll
        for (int i = 0; i < kRows; ++i) raw[i] = *reinterpret_cast<const uint32_t*>(xt + (int64_t)i * d);
    } else {
#pragma unroll
        for (int i = 0; i < kRows; ++i) {
            const int t = t0 - (K - 1) + i;
            raw[i] = 0;
            if (t >= 0) raw[i] = *reinterpret_cast<const uint32_t*>(xb + (int64_t)t * d);
        }
    }
    float in0[kRows], in1[kRows];
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
        in0[i] = __uint_as_float(raw[i] << 16);
        in1[i] = __uint_as_float(raw[i] & 0xFFFF0000u);
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

// x bf16 [batch * t_pad][d]; y bf16 [batch * t_pad][d] = Swish(LayerNorm(causal depthwise conv(x))).
// Work-group = d / 2 threads (thread: 2 channels x kTimeTile time steps), VPL = d / 256.
// Work items are (utterance, time tile) pairs; work-groups with equal blockIdx % 8 share an XCD
// (observed round-robin placement; locality only), so each XCD gets a contiguous run of items and
// the K - 1 input rows two neighbouring tiles share are served by one L2.
template <int K, int VPL>
__global__ __launch_bounds__(128 * VPL) void dwconv_ln_swish_kernel(
    const uint16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ ln_w,
    const float* __restrict__ ln_b, uint16_t* __restrict__ y, int t_tiles, int64_t t_pad, float eps) {
    constexpr int D = 256 * VPL, kRows = kTimeTile + K - 1, kWaves = 2 * VPL;
    __shared__ __attribute__((aligned(16))) float tile[kTimeTile * D];
    int item;
    {
        const int nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
        item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
    }
    const int b = item / t_tiles, t0 = (item % t_tiles) * kTimeTile;
    const int c = threadIdx.x * 2;
    const uint16_t* xb = x + (int64_t)b * t_pad * D + c;
    // Only the first tiles of an utterance have rows before its start: every other tile loads its
    // kRows rows without a test per row.
    uint32_t raw[kRows];
    if (t0 >= K - 1) {
        const uint16_t* xt = xb + (int64_t)(t0 - (K - 1)) * D;
#pragma unroll
        for (int i = 0; i < kRows; ++i) raw[i] = *reinterpret_cast<const uint32_t*>(xt + (int64_t)i * D);
    } else {
#pragma unroll
        for (int i = 0; i < kRows; ++i) {
            const int t = t0 - (K - 1) + i;
            raw[i] = 0;
            if (t >= 0) raw[i] = *reinterpret_cast<const uint32_t*>(xb + (int64_t)t * D);
        }
    }
    float in0[kRows], in1[kRows];
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
        in0[i] = __uint_as_float(raw[i] << 16);
        in1[i] = __uint_as_float(raw[i] & 0xFFFF0000u);
    }
    float a0[kTimeTile], a1[kTimeTile];
#pragma unroll
    for (int i = 0; i < kTimeTile; ++i) {
        a0[i] = 0.f;
        a1[i] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const float2 wj = *reinterpret_cast<const float2*>(w + (int64_t)j * D + c);
#pragma unroll
        for (int i = 0; i < kTimeTile; ++i) {
            a0[i] = fmaf(wj.x, in0[i + j], a0[i]);
            a1[i] = fmaf(wj.y, in1[i + j], a1[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < kTimeTile; ++i) *reinterpret_cast<float2*>(tile + i * D + c) = make_float2(a0[i], a1[i]);
    __syncthreads();
    // each wavefront normalises kTimeTile / kWaves whole rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int rr = 0; rr < kTimeTile / kWaves; ++rr) {
        const int row = wave * (kTimeTile / kWaves) + rr;
        float4 v[VPL];
#pragma unroll
        for (int q = 0; q < VPL; ++q) v[q] = *reinterpret_cast<const float4*>(tile + row * D + 4 * lane + 256 * q);
        float mean, rstd;
        row_stats<VPL>(v, mean, rstd, eps);
        row_affine<VPL>(v, mean, rstd, ln_w, ln_b, lane);
        uint16_t* yr = y + ((int64_t)b * t_pad + t0 + row) * D;
#pragma unroll
        for (int q = 0; q < VPL; ++q) {
            const uint2 pk = make_uint2(pack_bf16x2(swishf(v[q].x), swishf(v[q].y)),
                                        pack_bf16x2(swishf(v[q].z), swishf(v[q].w)));
            *reinterpret_cast<uint2*>(yr + 4 * lane + 256 * q) = pk;
        }
    }
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

extern "C" int oasr_dwconv_ln_swish(const uint16_t* x, const float* w, const float* ln_w,
                                    const float* ln_b, uint16_t* y, int32_t batch, int64_t t_pad,
                                    int32_t d, int32_t kernel, float eps, void* stream) {
    OASR_REQUIRE(x && w && ln_w && ln_b && y, "dwconv_ln_swish: null pointer");
    OASR_REQUIRE(batch > 0 && t_pad > 0 && t_pad % kTimeTile == 0,
                 "dwconv_ln_swish: batch must be positive and t_pad a multiple of %d", kTimeTile);
    OASR_REQUIRE(d == 512 || d == 1024 || d == 2048, "dwconv_ln_swish: d must be 512, 1024 or 2048");
    OASR_REQUIRE(kernel == 31, "dwconv_ln_swish: kernel size must be 31");
    const int64_t items = (int64_t)batch * (t_pad / kTimeTile);
    OASR_REQUIRE(items < ((int64_t)1 << 31), "dwconv_ln_swish: too many work items");
    const int t_tiles = (int)(t_pad / kTimeTile);
    dim3 grid((unsigned)items);
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
        case 512:
            hipLaunchKernelGGL((dwconv_ln_swish_kernel<31, 2>), grid, dim3(256), 0, st, x, w, ln_w, ln_b, y, t_tiles, t_pad, eps);
            break;
        case 1024:
            hipLaunchKernelGGL((dwconv_ln_swish_kernel<31, 4>), grid, dim3(512), 0, st, x, w, ln_w, ln_b, y, t_tiles, t_pad, eps);
            break;
        default:
            hipLaunchKernelGGL((dwconv_ln_swish_kernel<31, 8>), grid, dim3(1024), 0, st, x, w, ln_w, ln_b, y, t_tiles, t_pad, eps);
            break;
    }
    OASR_CHECK_LAUNCH("dwconv_ln_swish");
    return OASR_OK;
}
