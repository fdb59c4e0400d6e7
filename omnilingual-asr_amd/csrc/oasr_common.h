// Shared host/device helpers for liboasr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "oasr.h"

namespace oasr {

void set_error(const char* fmt, ...);

#define OASR_REQUIRE(cond, ...)                                  \
    do {                                                         \
        if (!(cond)) {                                           \
            ::oasr::set_error(__VA_ARGS__);                      \
            return OASR_EINVAL;                                  \
        }                                                        \
    } while (0)

#define OASR_CHECK_LAUNCH(what)                                                         \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            ::oasr::set_error("%s: launch failed: %s", what, hipGetErrorString(e_));    \
            return OASR_EHIP;                                                           \
        }                                                                               \
    } while (0)

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// float -> bf16, round to nearest even.  Inputs on this path are finite (LayerNorm keeps the
// activations bounded), so the integer form is safe and cheap.
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// v_exp_f32 + v_rcp_f32 (1 ulp each): far inside the bf16 tolerance of every consumer
__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf(float x) { return x * sigmoidf(x); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Sigmoid / Swish of two values with the multiplies and the add as packed f32 instructions (v_pk_mul_f32,
// v_pk_add_f32: two elements per issue slot); the same operations per element as sigmoidf / swishf, bit for bit.
__device__ __forceinline__ f32x2 sigmoid2(f32x2 x) {
    const f32x2 t = x * (f32x2){-1.442695040888963f, -1.442695040888963f};
    const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + (f32x2){1.0f, 1.0f};
    return (f32x2){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2 swish2(f32x2 x) { return x * sigmoid2(x); }

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Greedy CTC decode (ctc.hip): argument checks shared by the entry points, and the two launches with the
// per-frame winners going through `ids` (int32 [batch * t_pad], device).
int ctc_check_args(const float* logits, int64_t ld, int32_t vocab, const int32_t* lengths, int32_t batch,
                   int64_t t_pad, int32_t blank, int32_t* tokens, int32_t* frames, int32_t* counts);
int ctc_launch(const float* logits, int64_t ld, int32_t vocab, const int32_t* lengths, int32_t batch,
               int64_t t_pad, int32_t blank, int32_t* tokens, int32_t* frames, int32_t* counts,
               int32_t* ids, hipStream_t stream);

}  // namespace oasr
