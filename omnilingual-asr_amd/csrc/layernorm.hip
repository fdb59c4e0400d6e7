// Row LayerNorm kernels for gfx950.  One wavefront owns one row: the row lives in registers
// (16-byte loads, 4 columns per lane per 256-column chunk), mean and centred variance are two
// wave reductions, and the result leaves as packed bf16 (GEMM operand) or float32.
// HBM-bound: d*4 bytes in, d*2 (or d*4) bytes out per row, nothing re-read.
#include "oasr_common.h"
#include "row_norm.h"

namespace oasr {

template <int VPL, bool OUT_BF16, bool SWISH>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b,
                                                        void* y, int64_t rows,
                                                        int64_t y_stride, float eps) {
    constexpr int D = 256 * VPL;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    float4 v[VPL];
#pragma unroll
    for (int c = 0; c < VPL; ++c) v[c] = *reinterpret_cast<const float4*>(xr + 4 * lane + 256 * c);
    float mean, rstd;
    row_stats<VPL>(v, mean, rstd, eps);
    row_affine<VPL>(v, mean, rstd, w, b, lane);
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        if (SWISH) {
            v[c].x = swishf(v[c].x);
            v[c].y = swishf(v[c].y);
            v[c].z = swishf(v[c].z);
            v[c].w = swishf(v[c].w);
        }
        const int col = 4 * lane + 256 * c;
        if (OUT_BF16) {
            uint2 pk = make_uint2(pack_bf16x2(v[c].x, v[c].y), pack_bf16x2(v[c].z, v[c].w));
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(y) + row * y_stride + col) = pk;
        } else {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + row * y_stride + col) = v[c];
        }
    }
}

template <int VPL>
__global__ __launch_bounds__(256) void layernorm2_kernel(float* __restrict__ x,
                                                         const float* __restrict__ w1,
                                                         const float* __restrict__ b1,
                                                         const float* __restrict__ w2,
                                                         const float* __restrict__ b2,
                                                         uint16_t* __restrict__ y, int64_t rows,
                                                         float eps) {
    constexpr int D = 256 * VPL;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + row * D;
    float4 v[VPL];
#pragma unroll
    for (int c = 0; c < VPL; ++c) v[c] = *reinterpret_cast<const float4*>(xr + 4 * lane + 256 * c);
    float mean, rstd;
    row_stats<VPL>(v, mean, rstd, eps);
    row_affine<VPL>(v, mean, rstd, w1, b1, lane);
#pragma unroll
    for (int c = 0; c < VPL; ++c) *reinterpret_cast<float4*>(xr + 4 * lane + 256 * c) = v[c];
    row_stats<VPL>(v, mean, rstd, eps);
    row_affine<VPL>(v, mean, rstd, w2, b2, lane);
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        uint2 pk = make_uint2(pack_bf16x2(v[c].x, v[c].y), pack_bf16x2(v[c].z, v[c].w));
        *reinterpret_cast<uint2*>(y + row * D + 4 * lane + 256 * c) = pk;
    }
}

// e4m3 output for the fp8 projections: the normalised row is rounded to bf16 first (so the bytes equal
// what the bf16 kernel followed by the row quantiser produces), then scale = max|row| / 448 (1 for an
// all-zero row) and y = e4m3(row / scale).  Saves the quantiser's sweep: 1 byte out instead of 2 out,
// 2 in, 1 out per element.
template <int VPL>
__device__ __forceinline__ void store_row_fp8(float4 (&v)[VPL], uint8_t* __restrict__ yr, float* __restrict__ scale,
                                              int64_t row, int lane) {
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        const uint32_t lo = pack_bf16x2(v[c].x, v[c].y), hi = pack_bf16x2(v[c].z, v[c].w);
        v[c].x = __uint_as_float(lo << 16);
        v[c].y = __uint_as_float(lo & 0xFFFF0000u);
        v[c].z = __uint_as_float(hi << 16);
        v[c].w = __uint_as_float(hi & 0xFFFF0000u);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[c].x), fabsf(v[c].y))), fmaxf(fabsf(v[c].z), fabsf(v[c].w)));
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / 448.0f : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) scale[row] = s;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        int pk = 0;
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[c].x * inv, v[c].y * inv, pk, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[c].z * inv, v[c].w * inv, pk, true);
        *reinterpret_cast<uint32_t*>(yr + 4 * lane + 256 * c) = (uint32_t)pk;
    }
}

template <int VPL, bool SWISH>
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, uint8_t* __restrict__ y,
                                                            float* __restrict__ scale, int64_t rows, float eps) {
    constexpr int D = 256 * VPL;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    float4 v[VPL];
#pragma unroll
    for (int c = 0; c < VPL; ++c) v[c] = *reinterpret_cast<const float4*>(xr + 4 * lane + 256 * c);
    float mean, rstd;
    row_stats<VPL>(v, mean, rstd, eps);
    row_affine<VPL>(v, mean, rstd, w, b, lane);
    if (SWISH) {
#pragma unroll
        for (int c = 0; c < VPL; ++c) {
            v[c].x = swishf(v[c].x);
            v[c].y = swishf(v[c].y);
            v[c].z = swishf(v[c].z);
            v[c].w = swishf(v[c].w);
        }
    }
    store_row_fp8<VPL>(v, y + row * D, scale, row, lane);
}

template <int VPL>
__global__ __launch_bounds__(256) void layernorm2_fp8_kernel(float* __restrict__ x, const float* __restrict__ w1,
                                                             const float* __restrict__ b1, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, uint8_t* __restrict__ y,
                                                             float* __restrict__ scale, int64_t rows, float eps) {
    constexpr int D = 256 * VPL;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + row * D;
    float4 v[VPL];
#pragma unroll
    for (int c = 0; c < VPL; ++c) v[c] = *reinterpret_cast<const float4*>(xr + 4 * lane + 256 * c);
    float mean, rstd;
    row_stats<VPL>(v, mean, rstd, eps);
    row_affine<VPL>(v, mean, rstd, w1, b1, lane);
#pragma unroll
    for (int c = 0; c < VPL; ++c) *reinterpret_cast<float4*>(xr + 4 * lane + 256 * c) = v[c];
    row_stats<VPL>(v, mean, rstd, eps);
    row_affine<VPL>(v, mean, rstd, w2, b2, lane);
    store_row_fp8<VPL>(v, y + row * D, scale, row, lane);
}

// d = 160 (the stacked feature rows): 40 float4 per row on lanes 0..39; output row pitch
// y_stride >= 160 with the tail zero-filled so it can feed a GEMM whose K is padded to 192.
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void layernorm160_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ b,
                                                           void* y, int64_t rows,
                                                           int64_t y_stride, float eps) {
    constexpr int D = 160;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bool act = lane < D / 4;
    float4 v = act ? *reinterpret_cast<const float4*>(x + row * D + 4 * lane) : make_float4(0, 0, 0, 0);
    const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / D);
    float q = 0.f;
    if (act) {
        const float a = v.x - mean, bb = v.y - mean, c = v.z - mean, d = v.w - mean;
        q = (a * a + bb * bb) + (c * c + d * d);
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
    float4 o = make_float4(0, 0, 0, 0);
    if (act) {
        const float4 g = *reinterpret_cast<const float4*>(w + 4 * lane);
        const float4 h = *reinterpret_cast<const float4*>(b + 4 * lane);
        o.x = (v.x - mean) * rstd * g.x + h.x;
        o.y = (v.y - mean) * rstd * g.y + h.y;
        o.z = (v.z - mean) * rstd * g.z + h.z;
        o.w = (v.w - mean) * rstd * g.w + h.w;
    }
    const int col = 4 * lane;
    if (col < y_stride) {
        if (OUT_BF16) {
            uint2 pk = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(y) + row * y_stride + col) = pk;
        } else {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + row * y_stride + col) = o;
        }
    }
}

template <int VPL>
static int launch_ln(const float* x, const float* w, const float* b, void* y, int64_t rows,
                     int64_t y_stride, float eps, bool bf16, bool swish, hipStream_t st) {
    dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
    if (bf16 && swish)
        hipLaunchKernelGGL((layernorm_kernel<VPL, true, true>), grid, blk, 0, st, x, w, b, y, rows, y_stride, eps);
    else if (bf16)
        hipLaunchKernelGGL((layernorm_kernel<VPL, true, false>), grid, blk, 0, st, x, w, b, y, rows, y_stride, eps);
    else if (swish)
        hipLaunchKernelGGL((layernorm_kernel<VPL, false, true>), grid, blk, 0, st, x, w, b, y, rows, y_stride, eps);
    else
        hipLaunchKernelGGL((layernorm_kernel<VPL, false, false>), grid, blk, 0, st, x, w, b, y, rows, y_stride, eps);
    return 0;
}

}  // namespace oasr

using namespace oasr;

extern "C" {

int oasr_layernorm(const float* x, const float* w, const float* b, void* y, int64_t rows,
                   int32_t d, int64_t y_stride, float eps, int32_t out_bf16, int32_t swish,
                   void* stream) {
    OASR_REQUIRE(x && w && b && y, "layernorm: null pointer");
    OASR_REQUIRE(rows > 0 && rows < ((int64_t)1 << 33), "layernorm: bad row count");
    hipStream_t st = (hipStream_t)stream;
    if (d == 160) {
        OASR_REQUIRE(!swish, "layernorm: swish is not supported for d=160");
        OASR_REQUIRE(y_stride >= 160 && y_stride <= 256 && (y_stride & 3) == 0,
                     "layernorm: y_stride must be in [160,256] and a multiple of 4 for d=160");
        dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
        if (out_bf16)
            hipLaunchKernelGGL((layernorm160_kernel<true>), grid, blk, 0, st, x, w, b, y, rows, y_stride, eps);
        else
            hipLaunchKernelGGL((layernorm160_kernel<false>), grid, blk, 0, st, x, w, b, y, rows, y_stride, eps);
    } else {
        OASR_REQUIRE(y_stride >= d, "layernorm: y_stride < d");
        OASR_REQUIRE(y_stride == d, "layernorm: y_stride must equal d for d != 160");
        switch (d) {
            case 256: launch_ln<1>(x, w, b, y, rows, y_stride, eps, out_bf16, swish, st); break;
            case 512: launch_ln<2>(x, w, b, y, rows, y_stride, eps, out_bf16, swish, st); break;
            case 1024: launch_ln<4>(x, w, b, y, rows, y_stride, eps, out_bf16, swish, st); break;
            case 2048: launch_ln<8>(x, w, b, y, rows, y_stride, eps, out_bf16, swish, st); break;
            case 4096: launch_ln<16>(x, w, b, y, rows, y_stride, eps, out_bf16, swish, st); break;
            default:
                set_error("layernorm: unsupported width %d (160, 256, 512, 1024, 2048, 4096)", d);
                return OASR_EINVAL;
        }
    }
    OASR_CHECK_LAUNCH("layernorm");
    return OASR_OK;
}

int oasr_layernorm_fp8(const float* x, const float* w, const float* b, uint8_t* y, float* scale, int64_t rows,
                       int32_t d, float eps, int32_t swish, void* stream) {
    OASR_REQUIRE(x && w && b && y && scale, "layernorm_fp8: null pointer");
    OASR_REQUIRE(rows > 0 && rows < ((int64_t)1 << 33), "layernorm_fp8: bad row count");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
#define OASR_LN8(V)                                                                                              \
    do {                                                                                                         \
        if (swish)                                                                                               \
            hipLaunchKernelGGL((layernorm_fp8_kernel<V, true>), grid, blk, 0, st, x, w, b, y, scale, rows, eps); \
        else                                                                                                     \
            hipLaunchKernelGGL((layernorm_fp8_kernel<V, false>), grid, blk, 0, st, x, w, b, y, scale, rows, eps); \
    } while (0)
    switch (d) {
        case 256: OASR_LN8(1); break;
        case 512: OASR_LN8(2); break;
        case 1024: OASR_LN8(4); break;
        case 2048: OASR_LN8(8); break;
        case 4096: OASR_LN8(16); break;
        default:
            set_error("layernorm_fp8: unsupported width %d (256, 512, 1024, 2048, 4096)", d);
            return OASR_EINVAL;
    }
#undef OASR_LN8
    OASR_CHECK_LAUNCH("layernorm_fp8");
    return OASR_OK;
}

int oasr_layernorm2_fp8(float* x, const float* w1, const float* b1, const float* w2, const float* b2, uint8_t* y,
                        float* scale, int64_t rows, int32_t d, float eps, void* stream) {
    OASR_REQUIRE(x && w1 && b1 && w2 && b2 && y && scale, "layernorm2_fp8: null pointer");
    OASR_REQUIRE(rows > 0 && rows < ((int64_t)1 << 33), "layernorm2_fp8: bad row count");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
#define OASR_LN28(V) hipLaunchKernelGGL((layernorm2_fp8_kernel<V>), grid, blk, 0, st, x, w1, b1, w2, b2, y, scale, rows, eps)
    switch (d) {
        case 256: OASR_LN28(1); break;
        case 512: OASR_LN28(2); break;
        case 1024: OASR_LN28(4); break;
        case 2048: OASR_LN28(8); break;
        case 4096: OASR_LN28(16); break;
        default:
            set_error("layernorm2_fp8: unsupported width %d (256, 512, 1024, 2048, 4096)", d);
            return OASR_EINVAL;
    }
#undef OASR_LN28
    OASR_CHECK_LAUNCH("layernorm2_fp8");
    return OASR_OK;
}

int oasr_layernorm2(float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                    uint16_t* y, int64_t rows, int32_t d, float eps, void* stream) {
    OASR_REQUIRE(x && w1 && b1 && w2 && b2 && y, "layernorm2: null pointer");
    OASR_REQUIRE(rows > 0 && rows < ((int64_t)1 << 33), "layernorm2: bad row count");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
    switch (d) {
        case 256: hipLaunchKernelGGL((layernorm2_kernel<1>), grid, blk, 0, st, x, w1, b1, w2, b2, y, rows, eps); break;
        case 512: hipLaunchKernelGGL((layernorm2_kernel<2>), grid, blk, 0, st, x, w1, b1, w2, b2, y, rows, eps); break;
        case 1024: hipLaunchKernelGGL((layernorm2_kernel<4>), grid, blk, 0, st, x, w1, b1, w2, b2, y, rows, eps); break;
        case 2048: hipLaunchKernelGGL((layernorm2_kernel<8>), grid, blk, 0, st, x, w1, b1, w2, b2, y, rows, eps); break;
        case 4096: hipLaunchKernelGGL((layernorm2_kernel<16>), grid, blk, 0, st, x, w1, b1, w2, b2, y, rows, eps); break;
        default:
            set_error("layernorm2: unsupported width %d (256, 512, 1024, 2048, 4096)", d);
            return OASR_EINVAL;
    }
    OASR_CHECK_LAUNCH("layernorm2");
    return OASR_OK;
}

}  // extern "C"
