// Fused multi-head self-attention for gfx950 with clamped relative-key bias and key-length mask:
//   scores[i][j] = (q_i . k_j + q_i . E[clip(j - i, -L, R) + L]) / sqrt(64);  softmax over j < len;
//   ctx_i = sum_j p[i][j] v_j.
// Flash-style: the T x T scores never leave the CU.  A 256-thread work-group owns 128 query rows
// of one (utterance, head); each wavefront owns 32 of them and walks the keys in tiles of 64.
//
// MFMA orientation (v_mfma_f32_32x32x16_bf16): S^T = K . Q^T, so a lane holds one query column and
// 16 keys of it in registers — row max / row sum are in-register plus one exchange with lane^32,
// and the rescale factor is a per-lane scalar.  The S^T accumulator, converted to bf16, is directly
// the B operand of O^T = V^T . P^T (summing over its row index), so P never touches LDS.
// The relative-key term needs only the L+R+1 distinct values q_i . E[c]: they are computed once per
// wave by MFMA into LDS; key tiles entirely left (right) of the clamp window use one register value.
#include "oasr_common.h"

namespace oasr {

constexpr int kQBlock = 128;   // query rows per work-group (32 per wave)
constexpr int kKTile = 64;     // keys per iteration
constexpr int kHead = 64;      // head size
constexpr int kDistRows = 96;  // rows of the (zero padded) distance-embedding table
constexpr int kVtPitch = 136;  // bytes per V^T row in LDS: 64 keys * 2 B + 8 B pad (conflict-free b64)

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}

struct AttnSmem {
    __attribute__((aligned(16))) char k[kKTile * 128];          // [key][64 d] bf16, chunk-swizzled
    __attribute__((aligned(16))) char vt[kHead * kVtPitch];     // [d][key] bf16, padded rows
    __attribute__((aligned(16))) float qe[4][kDistRows * 32];   // per wave: [c][q] = q . E[c]
};

__global__ __launch_bounds__(256, 2) void attention_relkey_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dist_emb,
    const int32_t* __restrict__ lengths, uint16_t* __restrict__ ctx, int64_t t_pad, int d_model,
    int left_max, int right_max) {
    __shared__ AttnSmem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int len = lengths[b];
    const int q_blk0 = blockIdx.x * kQBlock;
    if (q_blk0 >= len) return;  // whole block is padding (uniform)
    const int64_t ld = 3 * (int64_t)d_model;
    const uint16_t* base = qkv + (int64_t)b * t_pad * ld + head * kHead;
    const int qw0 = q_blk0 + wave * 32;
    const bool wave_active = qw0 < len;
    const int n_dist = left_max + right_max + 1;

    // ---- Q fragments (B operand): lane (r, hh) holds Q[qw0 + r][16 s + 8 hh .. +8], s = 0..3 ---------
    bf16x8 qf[4];
    {
        const int qrow = min(qw0 + r, (int)t_pad - 1);
        const uint16_t* qp = base + (int64_t)qrow * ld + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    // ---- QE^T[c][q] = E[c] . Q[q]: three 32-row blocks of the distance table -------------------------
    float* qe = sm.qe[wave];
    if (wave_active) {
#pragma unroll
        for (int cb = 0; cb < kDistRows / 32; ++cb) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const uint16_t* ep = dist_emb + (cb * 32 + r) * kHead + 8 * hh;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 ef = *reinterpret_cast<const bf16x8*>(ep + 16 * s);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ef, qf[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                qe[c * 32 + r] = acc[e];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const float cs = 1.4426950408889634f * 0.125f;  // log2(e) / sqrt(64)
    const float bias_left = wave_active ? qe[0 * 32 + r] * cs : 0.f;
    const float bias_right = wave_active ? qe[(n_dist - 1) * 32 + r] * cs : 0.f;

    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        o0[e] = 0.f;
        o1[e] = 0.f;
    }
    float m_run = -INFINITY, l_run = 0.f;

    // staging roles: K as 2 x 16-byte chunks per thread; V as a key pair x 8 channels per thread
    const int kp = tid >> 3, vch = tid & 7;
    const int n_tiles = (len + kKTile - 1) / kKTile;

    uint4 kreg[2], vreg[2];
    auto load_tile = [&](int kt) {
        const int k0 = kt * kKTile;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ci = tid + 256 * i;
            const int key = min(k0 + (ci >> 3), (int)t_pad - 1);
            kreg[i] = *reinterpret_cast<const uint4*>(base + (int64_t)key * ld + d_model + (ci & 7) * 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = k0 + 2 * kp + i;
            const int krow = min(key, (int)t_pad - 1);
            uint4 v = *reinterpret_cast<const uint4*>(base + (int64_t)krow * ld + 2 * d_model + vch * 8);
            if (key >= len) v = make_uint4(0, 0, 0, 0);  // masked keys must not inject non-finite values
            vreg[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ci = tid + 256 * i;
            const int key = ci >> 3, ch = ci & 7;
            *reinterpret_cast<uint4*>(sm.k + key * 128 + ((ch ^ (key & 7)) << 4)) = kreg[i];
        }
        const uint32_t a[4] = {vreg[0].x, vreg[0].y, vreg[0].z, vreg[0].w};
        const uint32_t c[4] = {vreg[1].x, vreg[1].y, vreg[1].z, vreg[1].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // channels 8 vch + 2e and + 2e + 1, keys 2 kp and 2 kp + 1 packed into one dword each
            const uint32_t lo = (a[e] & 0xFFFFu) | (c[e] << 16);
            const uint32_t hi = (a[e] >> 16) | (c[e] & 0xFFFF0000u);
            *reinterpret_cast<uint32_t*>(sm.vt + (vch * 8 + 2 * e) * kVtPitch + kp * 4) = lo;
            *reinterpret_cast<uint32_t*>(sm.vt + (vch * 8 + 2 * e + 1) * kVtPitch + kp * 4) = hi;
        }
    };

    load_tile(0);
    for (int kt = 0; kt < n_tiles; ++kt) {
        __syncthreads();  // previous tile's LDS reads are done
        store_tile();
        __syncthreads();
        if (kt + 1 < n_tiles) load_tile(kt + 1);  // in flight during the MFMA work below
        if (!wave_active) continue;
        const int k0 = kt * kKTile;

        // ---- S^T = K . Q^T for the two 32-key halves ----------------------------------------------------
        f32x16 s0, s1;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s0[e] = 0.f;
            s1[e] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int chunk = (2 * s + hh) ^ (r & 7);
            const bf16x8 ka = *reinterpret_cast<const bf16x8*>(sm.k + r * 128 + (chunk << 4));
            const bf16x8 kb = *reinterpret_cast<const bf16x8*>(sm.k + (32 + r) * 128 + (chunk << 4));
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb, qf[s], s1, 0, 0, 0);
        }
        // ---- scale, relative-key bias, mask ----------------------------------------------------------------
        const int q = qw0 + r;
        const int dmax = k0 + kKTile - 1 - qw0, dmin = k0 - (qw0 + 31);  // wave-uniform distance range
        if (dmax <= -left_max) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s0[e] = s0[e] * cs + bias_left;
                s1[e] = s1[e] * cs + bias_left;
            }
        } else if (dmin >= right_max) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s0[e] = s0[e] * cs + bias_right;
                s1[e] = s1[e] * cs + bias_right;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const int c0 = min(max(key - q, -left_max), right_max) + left_max;
                const int c1 = min(max(key + 32 - q, -left_max), right_max) + left_max;
                s0[e] = (s0[e] + qe[c0 * 32 + r]) * cs;
                s1[e] = (s1[e] + qe[c1 * 32 + r]) * cs;
            }
        }
        if (k0 + kKTile > len) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (key >= len) s0[e] = -INFINITY;
                if (key + 32 >= len) s1[e] = -INFINITY;
            }
        }
        // ---- online softmax (base 2) -----------------------------------------------------------------------
        float tmax = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, fmaxf(s0[e], s1[e]));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s0[e] = __builtin_amdgcn_exp2f(s0[e] - m_new);
            s1[e] = __builtin_amdgcn_exp2f(s1[e] - m_new);
            psum += s0[e] + s1[e];
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            o0[e] *= alpha;
            o1[e] *= alpha;
        }
        // ---- O^T += V^T . P^T ---------------------------------------------------------------------------------
        // k-step (u, s2): P fragment = registers 8 s2 .. 8 s2 + 7 of S_u; its element j is key
        // 32 u + 16 s2 + 8 (j >> 2) + 4 hh + (j & 3), so V^T is read as two 8-byte pieces.
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const f32x16& sv = u ? s1 : s0;
                typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
                u32x4 pw;
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[j] = cvt_pk_bf16(sv[8 * s2 + 2 * j], sv[8 * s2 + 2 * j + 1]);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
                const int koff = (32 * u + 16 * s2 + 4 * hh) * 2;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char* vp = sm.vt + (32 * db + r) * kVtPitch + koff;
                    const uint2 va = *reinterpret_cast<const uint2*>(vp);
                    const uint2 vb = *reinterpret_cast<const uint2*>(vp + 16);
                    u32x4 vw = {va.x, va.y, vb.x, vb.y};
                    const bf16x8 vf = __builtin_bit_cast(bf16x8, vw);
                    if (db == 0)
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o0, 0, 0, 0);
                    else
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o1, 0, 0, 0);
                }
            }
        }
    }

    if (!wave_active) return;
    // ---- normalise and store: lane holds ctx[q][32 db + 8 g + 4 hh + {0..3}] ---------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = qw0 + r;
    if (q < (int)t_pad) {
        uint16_t* op = ctx + ((int64_t)b * t_pad + q) * d_model + head * kHead + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 pk0 = make_uint2(cvt_pk_bf16(o0[4 * g] * inv, o0[4 * g + 1] * inv),
                                   cvt_pk_bf16(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
            uint2 pk1 = make_uint2(cvt_pk_bf16(o1[4 * g] * inv, o1[4 * g + 1] * inv),
                                   cvt_pk_bf16(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
            *reinterpret_cast<uint2*>(op + 8 * g) = pk0;
            *reinterpret_cast<uint2*>(op + 32 + 8 * g) = pk1;
        }
    }
}

}  // namespace oasr

using namespace oasr;

extern "C" int oasr_attention_relkey(const uint16_t* qkv, const uint16_t* dist_emb,
                                     const int32_t* lengths, uint16_t* ctx, int32_t batch,
                                     int64_t t_pad, int32_t n_heads, int32_t left_max,
                                     int32_t right_max, void* stream) {
    OASR_REQUIRE(qkv && dist_emb && lengths && ctx, "attention: null pointer");
    OASR_REQUIRE(batch > 0 && batch <= 65535 && n_heads > 0 && n_heads <= 65535, "attention: bad batch/heads");
    OASR_REQUIRE(t_pad > 0 && t_pad % 32 == 0, "attention: t_pad must be a positive multiple of 32");
    OASR_REQUIRE(left_max >= 0 && right_max >= 0 && left_max + right_max + 1 <= kDistRows,
                 "attention: left_max + right_max + 1 must be <= %d", kDistRows);
    dim3 grid((unsigned)((t_pad + kQBlock - 1) / kQBlock), (unsigned)n_heads, (unsigned)batch);
    hipLaunchKernelGGL(attention_relkey_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv,
                       dist_emb, lengths, ctx, t_pad, n_heads * kHead, left_max, right_max);
    OASR_CHECK_LAUNCH("attention");
    return OASR_OK;
}
