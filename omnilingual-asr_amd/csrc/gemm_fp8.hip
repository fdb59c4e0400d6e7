// fp8 (OCP e4m3) MFMA GEMM with per-row / per-column scales and fused epilogues for gfx950:
//   C[m][n] = epi( sa[m] * sw[n] * sum_k A8[m][k] * W8[n][k] ).
// A8 are activations quantised per row (quant_fp8_rows_kernel below), W8 weights quantised per output
// channel on the host.  Same persistent ping-pong schedule, tile order, staging ring and barrier
// protocol as the bf16 kernel (gemm_bf16.hip; see the comments there) with bytes in place of bf16
// pairs: a ring slot holds 256 rows x 64 bytes = 64 k per operand, and one phase is one
// v_mfma_scale_f32_32x32x64_f8f6f4 step (block scales left at 1) over a wave's 128 x 64 sub-tile:
// 8 MFMAs of 64 cycles, twice the bf16 rate per clock.  Fragment: lane (r = lane & 31, g = lane >> 5)
// reads the 32 bytes k = 32 g .. 32 g + 31 of row r as two ds_read_b128; chunk c of row r sits at
// chunk c ^ ((r >> 2) & 3), which keeps those reads conflict-free.  The weight fragment is the
// MFMA's A operand, so lane (j = lane & 31, g) ends up with C[m = j][n = (e & 3) + 8 (e >> 2) + 4 g].
//
// Epilogue: every 16-row slice is bounced through a wave-private LDS slab as float32 (scaled by
// sa[m]) in two 32-column halves; in the read-back role a lane holds 4 consecutive n of one row and
// applies sw[n], bias, activation, residual there.
#include "oasr_common.h"

namespace oasr {

namespace f8 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kSliceBytes = BM * BK;       // 16 KiB per operand per ring slot
constexpr int kRing = 4;
constexpr int kEpiPitch = 144;
constexpr int kEpiSlab = 16 * kEpiPitch;
constexpr int kMaxCUs = 256;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;

#define OASR8_RAW_BARRIER()                   \
    do {                                      \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
        asm volatile("" ::: "memory");        \
    } while (0)

#define OASR8_WAVE_LDS_SYNC()                                           \
    do {                                                                \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); \
        __builtin_amdgcn_wave_barrier();                                \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); \
    } while (0)

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ void stage_slice(const char* __restrict__ src, uint32_t lane_off, int64_t piece_stride,
                                            char* lds_slice, int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const char* p = src + i * piece_stride;
        __builtin_amdgcn_global_load_lds((gbl_void*)(p + lane_off), (lds_void*)(lds_slice + (wave * 2 + i) * 1024), 16, 0, 0);
    }
}

// Column scales and bias of one output tile in the layout of the read-back role.
//   bf16 / float32 epilogues: lane (rchunk = lane & 7) owns columns 32 h + 4 rchunk .. + 3: sw[h], bv[h]
//   GLU (interleaved weights: 16 value rows then their 16 gate rows per 32): lane (q = lane & 3) owns
//   value columns 32 h + 4 q .. and gate columns 32 h + 16 + 4 q ..: sw[h] = value scales, bv[h] = gate scales
template <int EPI>
__device__ __forceinline__ void load_cols(float4 (&sw)[2], float4 (&bv)[2], const float* __restrict__ scale_w,
                                          const float* __restrict__ bias, int64_t nbase, int lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (EPI == OASR_EPI_GLU_BF16) {
            sw[h] = *reinterpret_cast<const float4*>(scale_w + nbase + h * 32 + (lane & 3) * 4);
            bv[h] = *reinterpret_cast<const float4*>(scale_w + nbase + h * 32 + 16 + (lane & 3) * 4);
        } else {
            sw[h] = *reinterpret_cast<const float4*>(scale_w + nbase + h * 32 + (lane & 7) * 4);
            bv[h] = bias ? *reinterpret_cast<const float4*>(bias + nbase + h * 32 + (lane & 7) * 4) : make_float4(0, 0, 0, 0);
        }
    }
}

template <int EPI>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[4][2], const float (&sam)[4], const float4 (&sw)[2],
                                         const float4 (&bv)[2], const float* resid, void* C, int64_t N,
                                         float alpha, int64_t mbase, int64_t nbase, int lane, char* slab) {
    const int wrow = lane & 15, whalf = (lane & 31) >> 4, g5 = lane >> 5;
    const int rrow = lane >> 3, rchunk = lane & 7;
    auto index_of = [&](int i, int h, int r2) {
        return (mbase + i * 16 + rrow + 8 * r2) * N + nbase + h * 32 + rchunk * 4;
    };
    float4 rcur[2][2], rnext[2][2];
    if (EPI == OASR_EPI_RESIDUAL_F32) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) rcur[h][r2] = *reinterpret_cast<const float4*>(resid + index_of(0, h, r2));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int im = i >> 1, hf = i & 1;
        if (EPI == OASR_EPI_RESIDUAL_F32 && i + 1 < 8) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
                    rnext[h][r2] = *reinterpret_cast<const float4*>(resid + index_of(i + 1, h, r2));
        }
        u32x2 pk16[2][2];   // bf16 epilogues: this lane's 4 outputs of (half h, row pair r2), kept until both halves exist
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // rows 16 hf .. 16 hf + 15 of m-tile im are held by the lanes with ((lane & 31) >> 4) == hf
            if (whalf == hf) {
                const float s = sam[im];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[im][h][4 * g] * s, acc[im][h][4 * g + 1] * s, acc[im][h][4 * g + 2] * s,
                               acc[im][h][4 * g + 3] * s};
                    *reinterpret_cast<f32x4*>(slab + wrow * kEpiPitch + (8 * g + 4 * g5) * 4) = v;
                }
            }
            OASR8_WAVE_LDS_SYNC();
            if (EPI == OASR_EPI_GLU_BF16) {
                const int row = lane >> 2, q = lane & 3;
                const float4 val = *reinterpret_cast<const float4*>(slab + row * kEpiPitch + q * 16);
                const float4 gate = *reinterpret_cast<const float4*>(slab + row * kEpiPitch + 64 + q * 16);
                OASR8_WAVE_LDS_SYNC();
                const u32x2 o = {cvt_pk_bf16(val.x * sw[h].x * sigmoidf(gate.x * bv[h].x),
                                             val.y * sw[h].y * sigmoidf(gate.y * bv[h].y)),
                                 cvt_pk_bf16(val.z * sw[h].z * sigmoidf(gate.z * bv[h].z),
                                             val.w * sw[h].w * sigmoidf(gate.w * bv[h].w))};
                uint16_t* cp = reinterpret_cast<uint16_t*>(C) + (mbase + i * 16 + row) * (N >> 1) + (nbase >> 1) + h * 16 + q * 4;
                __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(cp));
            } else {
                float4 v[2];
                v[0] = *reinterpret_cast<const float4*>(slab + rrow * kEpiPitch + rchunk * 16);
                v[1] = *reinterpret_cast<const float4*>(slab + (rrow + 8) * kEpiPitch + rchunk * 16);
                OASR8_WAVE_LDS_SYNC();
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    float4 o = make_float4(fmaf(v[r2].x, sw[h].x, bv[h].x), fmaf(v[r2].y, sw[h].y, bv[h].y),
                                           fmaf(v[r2].z, sw[h].z, bv[h].z), fmaf(v[r2].w, sw[h].w, bv[h].w));
                    if (EPI == OASR_EPI_BIAS_BF16 || EPI == OASR_EPI_BIAS_SWISH_BF16) {
                        if (EPI == OASR_EPI_BIAS_SWISH_BF16) {
                            const f32x2 lo = swish2((f32x2){o.x, o.y}), hi = swish2((f32x2){o.z, o.w});
                            o = make_float4(lo[0], lo[1], hi[0], hi[1]);
                        }
                        pk16[h][r2] = (u32x2){cvt_pk_bf16(o.x, o.y), cvt_pk_bf16(o.z, o.w)};
                    } else {
                        if (EPI == OASR_EPI_RESIDUAL_F32) {
                            const float4 r = rcur[h][r2];
                            o = make_float4(r.x + alpha * o.x, r.y + alpha * o.y, r.z + alpha * o.z, r.w + alpha * o.w);
                        }
                        __builtin_nontemporal_store((u32x4){__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z),
                                                            __float_as_uint(o.w)},
                                                    reinterpret_cast<u32x4*>(reinterpret_cast<float*>(C) + index_of(i, h, r2)));
                    }
                }
            }
        }
        if (EPI == OASR_EPI_BIAS_BF16 || EPI == OASR_EPI_BIAS_SWISH_BF16) {
            // A lane holds columns 4 c .. 4 c + 3 (c = lane & 7) of both 32-column halves: 8-byte pieces, which stored
            // as they are leave as half cache lines (measured: 1.21 bytes written per byte of output, and twice the
            // store instructions).  Neighbouring lanes swap one piece each, so that an even lane has columns
            // 4 c .. 4 c + 7 of the first half and the odd lane next to it the same columns of the second half:
            // 16-byte stores, 8 whole 128-byte lines per instruction.
            const bool odd = rchunk & 1;
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                const u32x2 send = odd ? pk16[0][r2] : pk16[1][r2];
                u32x2 recv;
                recv[0] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send[0], 0xB1, 0xf, 0xf, false);   // lane ^ 1
                recv[1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send[1], 0xB1, 0xf, 0xf, false);
                const u32x4 v = odd ? (u32x4){recv[0], recv[1], pk16[1][r2][0], pk16[1][r2][1]}
                                    : (u32x4){pk16[0][r2][0], pk16[0][r2][1], recv[0], recv[1]};
                uint16_t* dst = reinterpret_cast<uint16_t*>(C) + (mbase + i * 16 + rrow + 8 * r2) * N + nbase + (odd ? 32 : 0) +
                                (rchunk >> 1) * 8;
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
            }
        }
        if (EPI == OASR_EPI_RESIDUAL_F32) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) rcur[h][r2] = rnext[h][r2];
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_fp8_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                       const float* __restrict__ scale_a,
                                                       const float* __restrict__ scale_w,
                                                       const float* __restrict__ bias, const float* resid, void* C,
                                                       int64_t N, int64_t K, float alpha, int n_tiles, int w_packed) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * kRing * kSliceBytes + 8 * kEpiSlab];  // A ring | W ring | slabs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_n = (int)(N / BN);

    // Tile list of this work-group.  Work-groups with equal blockIdx % 8 share an XCD (observed
    // round-robin placement; used for L2 locality only, never for correctness).  XCD x owns a
    // contiguous range of m-tiles (all n).  Inside it tiles are ordered band by band, a band being
    // `band_w` (<= 4) n-tiles wide, m running fastest after n: the ~32 tiles in flight on an XCD are
    // then ~8 m-tiles x 4 n-tiles, whose W rows (4 x 256 x K) stay in the 4 MiB L2 for the whole
    // band while each A panel is fetched once per band.  (With n running over all of N, every
    // round of 32 tiles re-streamed the whole weight matrix: 0.93 GB fetched per 48128x4096x1024
    // launch against 0.11 GB of operands.)
    // Small problems (fewer than 2 m-tiles per XCD) split the flat tile list evenly instead.
    const int tiles_m = n_tiles / tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const bool banded = tiles_m >= 16;
    const int m_lo = (int)((int64_t)tiles_m * xcd / 8), m_cnt = (int)((int64_t)tiles_m * (xcd + 1) / 8) - m_lo;
    const int band_w = tiles_n % 4 == 0 ? 4 : (tiles_n % 3 == 0 ? 3 : (tiles_n % 2 == 0 ? 2 : 1));
    const int flat_lo = (int)((int64_t)n_tiles * xcd / 8);
    const int local_tiles = banded ? m_cnt * tiles_n : (int)((int64_t)n_tiles * (xcd + 1) / 8) - flat_lo;
    if (slot >= local_tiles) return;
    const int my_tiles = (local_tiles - slot + per_xcd - 1) / per_xcd;  // local tiles slot + i*per_xcd

    f32x16 acc[4][2], zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16;

    const int nph = (int)(K / BK);          // phases (K slices) per output tile
    const int total = my_tiles * nph;
    const int fr = lane & 31, fg = lane >> 5;
    // byte offsets of this lane's two 16-byte chunks inside a 32-row group of a slice
    const int off0 = fr * 64 + (((2 * fg) ^ ((fr >> 2) & 3)) << 4);
    const int off1 = fr * 64 + (((2 * fg + 1) ^ ((fr >> 2) & 3)) << 4);

    auto tile_origin = [&](int i, int64_t& m0, int64_t& n0) {
        const int l = slot + i * per_xcd;
        if (banded) {
            const int band = l / (m_cnt * band_w), in_band = l % (m_cnt * band_w);
            m0 = (int64_t)(m_lo + in_band / band_w) * BM;
            n0 = (int64_t)(band * band_w + in_band % band_w) * BN;
        } else {
            m0 = (int64_t)((flat_lo + l) / tiles_n) * BM;
            n0 = (int64_t)((flat_lo + l) % tiles_n) * BN;
        }
    };
    // Staging cursor (see gemm_bf16.hip): bytes instead of bf16 pairs.
    const uint32_t lane_off = (uint32_t)((lane >> 2) * K + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    const int64_t piece_stride = 16 * K;
    // packed weights (oasr_gemm_pack_w_fp8): piece (row group g, slice t) lies at (g * nph + t) KiB, lanes in order
    const uint32_t lane_off_w = w_packed ? (uint32_t)lane * 16 : lane_off;
    const int64_t piece_stride_w = w_packed ? (int64_t)nph * 1024 : piece_stride;
    const int w_step = w_packed ? 1024 : BK;
    int st_i = 0, st_t = 0;
    const char *st_a, *st_w;
    auto stage_seek = [&](int i) {
        int64_t m0, n0;
        tile_origin(i, m0, n0);
        st_a = reinterpret_cast<const char*>(A) + (m0 + wave * 32) * K;
        st_w = reinterpret_cast<const char*>(W) + (w_packed ? ((n0 + wave * 32) >> 4) * (int64_t)nph * 1024 : (n0 + wave * 32) * K);
    };
    auto stage_next = [&](int p) {
        stage_slice(st_a, lane_off, piece_stride, smem + (p & 3) * kSliceBytes, wave);
        stage_slice(st_w, lane_off_w, piece_stride_w, smem + (kRing + (p & 3)) * kSliceBytes, wave);
        st_a += BK;
        st_w += w_step;
        if (++st_t == nph) {
            st_t = 0;
            if (++st_i < my_tiles) stage_seek(st_i);
        }
    };
    stage_seek(0);

    stage_next(0);
    if (total > 1) {
        stage_next(1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    OASR8_RAW_BARRIER();
    if (wm == 1) OASR8_RAW_BARRIER();  // group B runs one barrier behind

    char* slab = smem + 2 * kRing * kSliceBytes + wave * kEpiSlab;
    int p = 0;
    float sam[4];
    float4 swv[2], bv[2];
    for (int i = 0; i < my_tiles; ++i) {
        int64_t m0, n0;
        tile_origin(i, m0, n0);
        load_cols<EPI>(swv, bv, scale_w, bias, n0 + wn * 64, lane);
#pragma unroll
        for (int im = 0; im < 4; ++im) sam[im] = scale_a[m0 + wm * 128 + im * 32 + fr];
        for (int t = 0; t < nph; ++t, ++p) {
            const char* sa = smem + (p & 3) * kSliceBytes + (wm * 128) * 64;
            const char* sb = smem + (kRing + (p & 3)) * kSliceBytes + (wn * 64) * 64;
            // ---- LOAD segment
            u32x4 w0[2], w1[2], x0[4], x1[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                w0[j] = *reinterpret_cast<const u32x4*>(sb + j * 32 * 64 + off0);
                w1[j] = *reinterpret_cast<const u32x4*>(sb + j * 32 * 64 + off1);
            }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                x0[ii] = *reinterpret_cast<const u32x4*>(sa + ii * 32 * 64 + off0);
                x1[ii] = *reinterpret_cast<const u32x4*>(sa + ii * 32 * 64 + off1);
            }
            if (p + 2 < total) {
                stage_next(p + 2);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            OASR8_RAW_BARRIER();
            // ---- MFMA segment
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const i32x8 xf = {(int)x0[ii][0], (int)x0[ii][1], (int)x0[ii][2], (int)x0[ii][3],
                                  (int)x1[ii][0], (int)x1[ii][1], (int)x1[ii][2], (int)x1[ii][3]};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const i32x8 wf = {(int)w0[j][0], (int)w0[j][1], (int)w0[j][2], (int)w0[j][3],
                                      (int)w1[j][0], (int)w1[j][1], (int)w1[j][2], (int)w1[j][3]};
                    // a tile's first slice starts from a constant zero: no accumulator clearing pass
                    acc[ii][j] = t == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf, xf, zero16, 0, 0, 0, 0, 0, 0)
                                        : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf, xf, acc[ii][j], 0, 0, 0, 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (!(wm == 1 && t == nph - 1)) OASR8_RAW_BARRIER();
        }
        epilogue<EPI>(acc, sam, swv, bv, resid, C, N, alpha, m0 + wm * 128, n0 + wn * 64, lane, slab);
        if (wm == 1) OASR8_RAW_BARRIER();
    }
    if (wm == 0) OASR8_RAW_BARRIER();  // same barrier count in both groups
}

// Per-row quantisation of bf16 activations to e4m3: scale[r] = max|x[r]| / 448 (1 for an all-zero
// row), y = round_to_e4m3(x / scale).  One wavefront per row, two sweeps over it (the second comes
// from cache).  HBM-bound: 2 bytes in, 1 byte out per element.
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const uint16_t* __restrict__ x, int64_t rows, int64_t k,
                                                             uint8_t* __restrict__ y, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t* xr = x + row * k;
    float amax = 0.f;
    for (int64_t c = 8 * lane; c < k; c += 512) {
        const uint4 v = *reinterpret_cast<const uint4*>(xr + c);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            amax = fmaxf(amax, fabsf(__uint_as_float(w[q] << 16)));
            amax = fmaxf(amax, fabsf(__uint_as_float(w[q] & 0xFFFF0000u)));
        }
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / 448.0f : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) scale[row] = s;
    uint8_t* yr = y + row * k;
    for (int64_t c = 8 * lane; c < k; c += 512) {
        const uint4 v = *reinterpret_cast<const uint4*>(xr + c);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[0] << 16) * inv, __uint_as_float(w[0] & 0xFFFF0000u) * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[1] << 16) * inv, __uint_as_float(w[1] & 0xFFFF0000u) * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[2] << 16) * inv, __uint_as_float(w[2] & 0xFFFF0000u) * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[3] << 16) * inv, __uint_as_float(w[3] & 0xFFFF0000u) * inv, hi, true);
        *reinterpret_cast<uint2*>(yr + c) = make_uint2((uint32_t)lo, (uint32_t)hi);
    }
}

// Same result with the row held in registers (k = 512 NV, NV 16-byte loads per lane, all issued before
// the first use): one sweep over memory and NV loads in flight per lane instead of one.
template <int NV>
__global__ __launch_bounds__(256) void quant_fp8_rows_reg_kernel(const uint16_t* __restrict__ x, int64_t rows,
                                                                 uint8_t* __restrict__ y, float* __restrict__ scale) {
    constexpr int64_t K = 512 * NV;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t* xr = x + row * K + 8 * lane;
    u32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xr + 512 * i));
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const uint32_t w[4] = {v[i][0], v[i][1], v[i][2], v[i][3]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            amax = fmaxf(amax, fabsf(__uint_as_float(w[q] << 16)));
            amax = fmaxf(amax, fabsf(__uint_as_float(w[q] & 0xFFFF0000u)));
        }
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / 448.0f : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) scale[row] = s;
    uint8_t* yr = y + row * K + 8 * lane;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const uint32_t w[4] = {v[i][0], v[i][1], v[i][2], v[i][3]};
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[0] << 16) * inv, __uint_as_float(w[0] & 0xFFFF0000u) * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[1] << 16) * inv, __uint_as_float(w[1] & 0xFFFF0000u) * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[2] << 16) * inv, __uint_as_float(w[2] & 0xFFFF0000u) * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[3] << 16) * inv, __uint_as_float(w[3] & 0xFFFF0000u) * inv, hi, true);
        *reinterpret_cast<uint2*>(yr + 512 * i) = make_uint2((uint32_t)lo, (uint32_t)hi);
    }
}


// Row-major e4m3 weights [n][k] -> the packed form: 1 KiB pieces of 16 rows x 64 k in the byte order a copy
// instruction places them (chunk `lane` = row lane >> 2, source chunk (lane & 3) ^ ((row >> 2) & 3)), pieces
// ordered [row group][K slice].
__global__ __launch_bounds__(256) void gemm_pack_w8_kernel(const uint8_t* __restrict__ w, uint8_t* __restrict__ out,
                                                           int64_t n, int64_t k) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // 16-byte chunk of the output
    const int64_t nph = k / BK, total = (n / 16) * nph * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int64_t piece = idx >> 6, g = piece / nph, t = piece - g * nph;
    const int row = lane >> 2, chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const u32x4 v = *reinterpret_cast<const u32x4*>(w + (g * 16 + row) * k + t * BK + chunk * 16);
    *reinterpret_cast<u32x4*>(out + idx * 16) = v;
}

}  // namespace f8
}  // namespace oasr

using namespace oasr;

extern "C" int oasr_quant_fp8_rows(const uint16_t* x, int64_t rows, int64_t k, uint8_t* y, float* scale,
                                   void* stream) {
    OASR_REQUIRE(x && y && scale, "quant_fp8_rows: null pointer");
    OASR_REQUIRE(rows > 0 && k > 0 && k % 8 == 0, "quant_fp8_rows: rows must be positive and k a positive multiple of 8");
    OASR_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0, "quant_fp8_rows: misaligned pointer");
    const int64_t blocks = (rows + 3) / 4;
    OASR_REQUIRE(blocks < ((int64_t)1 << 31), "quant_fp8_rows: too many rows");
    const dim3 grid((unsigned)blocks), blk(256);
    hipStream_t st = (hipStream_t)stream;
    switch (k) {   // widths of the encoder geometries keep the row in registers
        case 1024: hipLaunchKernelGGL(f8::quant_fp8_rows_reg_kernel<2>, grid, blk, 0, st, x, rows, y, scale); break;
        case 2048: hipLaunchKernelGGL(f8::quant_fp8_rows_reg_kernel<4>, grid, blk, 0, st, x, rows, y, scale); break;
        case 4096: hipLaunchKernelGGL(f8::quant_fp8_rows_reg_kernel<8>, grid, blk, 0, st, x, rows, y, scale); break;
        case 8192: hipLaunchKernelGGL(f8::quant_fp8_rows_reg_kernel<16>, grid, blk, 0, st, x, rows, y, scale); break;
        case 16384: hipLaunchKernelGGL(f8::quant_fp8_rows_reg_kernel<32>, grid, blk, 0, st, x, rows, y, scale); break;
        default: hipLaunchKernelGGL(f8::quant_fp8_rows_kernel, grid, blk, 0, st, x, rows, k, y, scale);
    }
    OASR_CHECK_LAUNCH("quant_fp8_rows");
    return OASR_OK;
}

static int gemm_fp8_launch(const uint8_t* a, const uint8_t* w, const float* scale_a, const float* scale_w,
                           const float* bias, const float* resid, void* c, int64_t m, int64_t n, int64_t k,
                           float alpha, int32_t epilogue, void* stream, int w_packed) {
    OASR_REQUIRE(a && w && scale_a && scale_w && c, "gemm_fp8: null pointer");
    OASR_REQUIRE(m > 0 && n > 0 && k > 0 && m % f8::BM == 0 && n % f8::BN == 0 && k % f8::BK == 0,
                 "gemm_fp8: m=%lld n=%lld must be multiples of 256 and k=%lld of 64", (long long)m, (long long)n,
                 (long long)k);
    OASR_REQUIRE((m / f8::BM) * (n / f8::BN) < ((int64_t)1 << 30), "gemm_fp8: too many tiles");
    OASR_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)c & 15) == 0 &&
                     ((uintptr_t)bias & 15) == 0 && ((uintptr_t)resid & 15) == 0 && ((uintptr_t)scale_w & 15) == 0,
                 "gemm_fp8: operands must be 16-byte aligned");
    OASR_REQUIRE(epilogue != OASR_EPI_RESIDUAL_F32 || resid != nullptr, "gemm_fp8: residual epilogue without resid");
    const int tiles = (int)((m / f8::BM) * (n / f8::BN));
    dim3 grid((unsigned)(tiles < f8::kMaxCUs ? (tiles + 7) / 8 * 8 : f8::kMaxCUs)), blk(512);
    hipStream_t st = (hipStream_t)stream;
#define OASR_LAUNCH_GEMM8(E)                                                                                    \
    hipLaunchKernelGGL((f8::gemm_fp8_kernel<E>), grid, blk, 0, st, a, w, scale_a, scale_w, bias, resid, c, n, k, \
                       alpha, tiles, w_packed)
    switch (epilogue) {
        case OASR_EPI_BIAS_BF16: OASR_LAUNCH_GEMM8(OASR_EPI_BIAS_BF16); break;
        case OASR_EPI_BIAS_SWISH_BF16: OASR_LAUNCH_GEMM8(OASR_EPI_BIAS_SWISH_BF16); break;
        case OASR_EPI_GLU_BF16: OASR_LAUNCH_GEMM8(OASR_EPI_GLU_BF16); break;
        case OASR_EPI_RESIDUAL_F32: OASR_LAUNCH_GEMM8(OASR_EPI_RESIDUAL_F32); break;
        case OASR_EPI_BIAS_F32: OASR_LAUNCH_GEMM8(OASR_EPI_BIAS_F32); break;
        default:
            set_error("gemm_fp8: unknown epilogue %d", epilogue);
            return OASR_EINVAL;
    }
#undef OASR_LAUNCH_GEMM8
    OASR_CHECK_LAUNCH("gemm_fp8");
    return OASR_OK;
}

extern "C" int oasr_gemm_fp8(const uint8_t* a, const uint8_t* w, const float* scale_a, const float* scale_w,
                             const float* bias, const float* resid, void* c, int64_t m, int64_t n, int64_t k,
                             float alpha, int32_t epilogue, void* stream) {
    return gemm_fp8_launch(a, w, scale_a, scale_w, bias, resid, c, m, n, k, alpha, epilogue, stream, 0);
}

extern "C" int oasr_gemm_fp8_pw(const uint8_t* a, const uint8_t* w_packed, const float* scale_a, const float* scale_w,
                                const float* bias, const float* resid, void* c, int64_t m, int64_t n, int64_t k,
                                float alpha, int32_t epilogue, void* stream) {
    return gemm_fp8_launch(a, w_packed, scale_a, scale_w, bias, resid, c, m, n, k, alpha, epilogue, stream, 1);
}

extern "C" int oasr_gemm_pack_w_fp8(const uint8_t* w, int64_t n, int64_t k, uint8_t* out, void* stream) {
    OASR_REQUIRE(w && out && w != out, "gemm_pack_w_fp8: null pointer or in-place call");
    OASR_REQUIRE(n > 0 && k > 0 && n % 16 == 0 && k % f8::BK == 0, "gemm_pack_w_fp8: n=%lld must be a multiple of 16 and k=%lld of 64",
                 (long long)n, (long long)k);
    OASR_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0, "gemm_pack_w_fp8: operands must be 16-byte aligned");
    const int64_t chunks = n * k / 16;
    OASR_REQUIRE(chunks / 256 + 1 < ((int64_t)1 << 31), "gemm_pack_w_fp8: matrix too large");
    hipLaunchKernelGGL(f8::gemm_pack_w8_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, out, n, k);
    OASR_CHECK_LAUNCH("gemm_pack_w_fp8");
    return OASR_OK;
}
