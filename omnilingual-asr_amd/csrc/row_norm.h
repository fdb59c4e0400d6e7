// LayerNorm of one row held by one wavefront in registers (4 columns per lane per 256-column
// chunk): mean and centred variance by two wave reductions, then the affine map.
#pragma once
#include "oasr_common.h"

namespace oasr {

template <int VPL>  // float4 chunks per lane; row width D = 256 * VPL
__device__ __forceinline__ void row_stats(const float4 (&v)[VPL], float& mean, float& rstd, float eps) {
    constexpr int D = 256 * VPL;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < VPL; ++c) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
    }
    rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
}

template <int VPL>
__device__ __forceinline__ void row_affine(float4 (&v)[VPL], float mean, float rstd,
                                           const float* __restrict__ w, const float* __restrict__ b,
                                           int lane) {
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        const int col = 4 * lane + 256 * c;
        const float4 g = *reinterpret_cast<const float4*>(w + col);
        const float4 h = *reinterpret_cast<const float4*>(b + col);
        v[c].x = (v[c].x - mean) * rstd * g.x + h.x;
        v[c].y = (v[c].y - mean) * rstd * g.y + h.y;
        v[c].z = (v[c].z - mean) * rstd * g.z + h.z;
        v[c].w = (v[c].w - mean) * rstd * g.w + h.w;
    }
}

}  // namespace oasr
