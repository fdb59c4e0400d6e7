// bf16 MFMA GEMM with fused epilogues for gfx950:  C[m][n] = epi( sum_k A[m][k] * W[n][k] ).
//
// Operands.  Both are K-contiguous (activations [m][k], Linear weights [n][k]), so both are staged
// the same way: K runs in slices of 32 (256 rows x 64 bytes per operand), copied global -> LDS
// directly by global_load_lds (16 bytes per lane, no VGPR round trip) into a ring of four slices.
// The LDS image is lane-linear per wave instruction (16 rows x 64 bytes), so the bank-conflict
// swizzle (16-byte chunk index XOR (-(row >> 2) & 3)) is applied to the per-lane *source* address
// and again on the fragment read (conflict-free ds_read_b128).
// Weights may also be given "packed" (oasr_gemm_pack_w_bf16): every 1 KiB piece of that LDS image (16 rows x
// one K slice, swizzle applied) stored contiguously, pieces ordered [row group][K slice].  A copy
// instruction then reads 8 whole cache lines instead of 16 half lines: 6 % off the 48128 x 4096 x 1024
// product, bit-identical results (the LDS image is the same).
//
// Tile and waves.  256 x 256 output tile, 8 wavefronts as 2 (m) x 4 (n); a wave owns 128 m x 64 n
// as 8 x 4 tiles of v_mfma_f32_16x16x32_bf16.  The weight fragment is the MFMA's A operand and the
// activation fragment its B operand, so a lane ends up holding 4 consecutive n for one m.
//
// Schedule (persistent ping-pong).  One work-group per CU walks a list of output tiles.  The 8
// waves form two groups of 4 (waves 0-3: upper 128 rows, waves 4-7: lower 128 rows); each SIMD
// hosts one wave of each group.  A phase is one K slice: a LOAD segment (12 ds_read_b128 of slice
// p, plus this wave's 4 global_load_lds of slice p+2) and an MFMA segment (32 MFMAs), separated by
// raw s_barriers.  Group B runs one barrier behind group A, so on every SIMD one wave is in its
// MFMA segment while its partner is in its LOAD segment.  The slices of consecutive output tiles
// form ONE continuous pipeline (global phase p, ring slot p & 3): the first slices of the next
// output tile are staged while the current tile's last MFMAs run, and a wave's epilogue overlaps
// its partner's MFMA segment.  Every LOAD segment carries the same staging work (4 KiB per wave):
// the copies of all waves of a segment share the CU's one address/texture path, and 8 per wave in
// one segment made that segment longer than the partner's MFMA segment.
//
//   barrier timeline (b_i = i-th barrier, L/M = load/MFMA segment of phase p):
//     A:  L0 |b0| M0 |b1| L1 |b2| M1 |b3| L2 ...
//     B:     |b0| L0 |b1| M0 |b2| L1 |b3| M1 ...
//   Slice p+2 goes to ring slot (p+2) & 3, which held slice p-2: its last reads were issued in
//   B's L(p-2), two barriers before A's L(p).  A wave leaves L(p) with at most its 4 newest copies
//   outstanding (vmcnt(4); the counter retires in order), so its part of slice p+1 has landed
//   before b_(2p) (A) / b_(2p+1) (B); slice p+1 is first read after b_(2p+1).
//
// Epilogue.  Each 16-row slice of a wave's sub-tile is bounced through a wave-private LDS slab and
// leaves as 16-byte-per-lane stores covering 8 rows x 128 contiguous bytes.  Outputs are written
// with non-temporal stores: a GEMM output here is 100-400 MB, read next by another kernel, and
// letting it allocate dirty lines in the 4 MiB L2 evicted the A/W panels the main loop re-reads
// (measured: 453 -> 388 us on 48128 x 4096 x 1024).
#include "oasr_common.h"

namespace oasr {

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int kSliceBytes = BM * BK * 2;   // 16 KiB per operand per ring slot
constexpr int kRing = 4;
constexpr int kEpiPitch = 144;            // bytes per slab row: 16-byte aligned, <= 2-way write conflicts
constexpr int kEpiSlab = 16 * kEpiPitch;  // bytes per wave
constexpr int kMaxCUs = 256;


typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define OASR_RAW_BARRIER()                    \
    do {                                      \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
        asm volatile("" ::: "memory");        \
    } while (0)

// Orders this wave's LDS writes before its later LDS reads (DS operations of one wave execute in
// order; this only pins the compiler).  The fences name the LDS address space only, so global
// loads and stores (residual reads, output stores) stay free to move and overlap.
#define OASR_WAVE_LDS_SYNC()                                    \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); \
        __builtin_amdgcn_wave_barrier();                                 \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); \
    } while (0)

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ void store_nt16(void* p, u32x4 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
}

// Stage this wave's part of one K slice of one operand (256 rows x 64 bytes): 2 global_load_lds of
// 1 KiB (16 rows each).  `src` is the wave-uniform address of the wave's first row at this k
// offset, `lane_off` the lane's constant byte offset inside a 16-row piece (row lane >> 2,
// swizzled 16-byte chunk), `piece_stride` the bytes between pieces: the address of a load is then
// scalar base + 32-bit lane offset and needs no per-load vector arithmetic.
__device__ __forceinline__ void stage_slice(const char* __restrict__ src, uint32_t lane_off, int64_t piece_stride,
                                            char* lds_slice, int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const char* p = src + i * piece_stride;
        __builtin_amdgcn_global_load_lds((gbl_void*)(p + lane_off), (lds_void*)(lds_slice + (wave * 2 + i) * 1024), 16, 0, 0);
    }
}

// Bias.  The 64 bias values of a wave's 64 output columns reach the accumulators without touching the
// vector ALU: one 4-byte-per-lane global_load_lds puts them into a wave-private LDS strip (two
// strips, alternating by tile, 256 bytes each), issued one K slice before the tile starts so that the
// in-order vmcnt wait of the tile's first LOAD segment covers it; after that wait the wave reads
// bv[j] = bias[nbase + 16 j + 4 fq ..] (the four consecutive n a lane holds in acc[.][j]) and the
// tile's first MFMAs take bv[j] as their C operand.  No bias (or the GLU epilogue): bv = 0.
__device__ __forceinline__ void bias_to_lds(const float* __restrict__ bias, int64_t nbase, int lane, char* strip) {
    __builtin_amdgcn_global_load_lds((gbl_void*)(bias + nbase + lane), (lds_void*)strip, 4, 0, 0);
}

__device__ __forceinline__ void bias_from_lds(f32x4 (&bv)[4], const char* strip, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(strip + (j * 16 + (lane >> 4) * 4) * 4);
}

// acc[i][j]: lane (fr = lane & 15, fq = lane >> 4) holds C[mbase + 16 i + fr][nbase + 16 j + 4 fq + 0..3].
template <int EPI, int MI = 8>
__device__ __forceinline__ void gemm_epilogue(const f32x4 (&acc)[MI][4],
                                              const float* resid, void* C, int64_t N, float alpha,
                                              int64_t mbase, int64_t nbase, int lane, char* slab, bool c_packed = false) {
    const int fr = lane & 15, fq = lane >> 4;
    if (EPI == OASR_EPI_BIAS_BF16 || EPI == OASR_EPI_BIAS_SWISH_BF16) {
        const int rrow = lane >> 3, rchunk = lane & 7;   // read-back role: rows rrow, rrow + 8; 16-byte chunk
        uint16_t* cbase = reinterpret_cast<uint16_t*>(C) + (mbase + rrow) * N + nbase + rchunk * 8;
        // Packed output (the layout of oasr_gemm_pack_w_bf16, for a product that is the next one's A operand):
        // a 16-row slice of the wave's 64 columns is two whole 1 KiB pieces; lane l stores position l of both,
        // i.e. row l >> 2, source chunk (l & 3) ^ (-(row >> 2) & 3).
        const int prow = lane >> 2, pchunk = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
        char* pbase = reinterpret_cast<char*>(C) + ((mbase >> 4) * (N >> 5) + (nbase >> 5)) * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[i][j];
                if (EPI == OASR_EPI_BIAS_SWISH_BF16) {
                    const f32x2 lo = swish2((f32x2){v[0], v[1]}), hi = swish2((f32x2){v[2], v[3]});
                    v = (f32x4){lo[0], lo[1], hi[0], hi[1]};
                }
                // rows 8-15 keep the two 8-byte halves of every 16-byte chunk swapped: the 16 lanes of one LDS pass
                // (rows 0-15 of one fq) would otherwise meet in pairs on the same banks (row pitch 36 dwords)
                *reinterpret_cast<uint2*>(slab + fr * kEpiPitch + j * 32 + ((fq ^ (fr >> 3)) * 8)) =
                    make_uint2(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]));
            }
            OASR_WAVE_LDS_SYNC();
            if (c_packed) {   // wave-uniform
                u32x4 p0 = *reinterpret_cast<const u32x4*>(slab + prow * kEpiPitch + pchunk * 16);
                u32x4 p1 = *reinterpret_cast<const u32x4*>(slab + prow * kEpiPitch + 64 + pchunk * 16);
                OASR_WAVE_LDS_SYNC();
                if (prow >= 8) {
                    p0 = (u32x4){p0[2], p0[3], p0[0], p0[1]};
                    p1 = (u32x4){p1[2], p1[3], p1[0], p1[1]};
                }
                char* dst = pbase + (int64_t)i * (N >> 5) * 1024;
                store_nt16(dst, p0);
                store_nt16(dst + 1024, p1);
            } else {
                const u32x4 lo = *reinterpret_cast<const u32x4*>(slab + rrow * kEpiPitch + rchunk * 16);
                const u32x4 hs = *reinterpret_cast<const u32x4*>(slab + (rrow + 8) * kEpiPitch + rchunk * 16);
                OASR_WAVE_LDS_SYNC();
                const u32x4 hi = {hs[2], hs[3], hs[0], hs[1]};
                store_nt16(cbase + (int64_t)(i * 16) * N, lo);
                store_nt16(cbase + (int64_t)(i * 16 + 8) * N, hi);
            }
        }
    } else if (EPI == OASR_EPI_GLU_BF16) {
        // interleaved weights: tiles j = 0, 2 hold values, j = 1, 3 their gates; N/2 output columns
        const int rrow = lane >> 2, rchunk = lane & 3;   // 16 rows x 64 B
        uint16_t* cbase = reinterpret_cast<uint16_t*>(C) + (mbase + rrow) * (N >> 1) + (nbase >> 1) + rchunk * 8;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const f32x4 v = acc[i][j], g = acc[i][j + 1];
                const f32x2 lo = (f32x2){v[0], v[1]} * sigmoid2((f32x2){g[0], g[1]});
                const f32x2 hi = (f32x2){v[2], v[3]} * sigmoid2((f32x2){g[2], g[3]});
                *reinterpret_cast<uint2*>(slab + fr * kEpiPitch + (j >> 1) * 32 + fq * 8) =
                    make_uint2(cvt_pk_bf16(lo[0], lo[1]), cvt_pk_bf16(hi[0], hi[1]));
            }
            OASR_WAVE_LDS_SYNC();
            const u32x4 o = *reinterpret_cast<const u32x4*>(slab + rrow * kEpiPitch + rchunk * 16);
            OASR_WAVE_LDS_SYNC();
            store_nt16(cbase + (int64_t)(i * 16) * (N >> 1), o);
        }
    } else {  // float32 outputs: two 32-column halves per 16-row slice
        const int rrow = lane >> 3, rchunk = lane & 7;
        // element index of (slice i, half h, row-pair r2) for this lane's read-back role
        auto index_of = [&](int i, int h, int r2) {
            return (mbase + i * 16 + rrow + 8 * r2) * N + nbase + h * 32 + rchunk * 4;
        };
        // The residual may alias the output (in-place update), so the compiler keeps loads and stores
        // in program order: the residual of slice i+1 is therefore requested before slice i is stored,
        // which keeps one slice of loads in flight instead of paying a memory round trip per store.
        float4 rcur[2][2], rnext[2][2];
        if (EPI == OASR_EPI_RESIDUAL_F32) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) rcur[h][r2] = *reinterpret_cast<const float4*>(resid + index_of(0, h, r2));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (EPI == OASR_EPI_RESIDUAL_F32 && i + 1 < MI) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
                        rnext[h][r2] = *reinterpret_cast<const float4*>(resid + index_of(i + 1, h, r2));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    *reinterpret_cast<f32x4*>(slab + fr * kEpiPitch + jj * 64 + fq * 16) = acc[i][2 * h + jj];
                OASR_WAVE_LDS_SYNC();
                float4 v[2];
                v[0] = *reinterpret_cast<const float4*>(slab + rrow * kEpiPitch + rchunk * 16);
                v[1] = *reinterpret_cast<const float4*>(slab + (rrow + 8) * kEpiPitch + rchunk * 16);
                OASR_WAVE_LDS_SYNC();
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    float4 o = v[r2];
                    if (EPI == OASR_EPI_RESIDUAL_F32) {
                        const float4 r = rcur[h][r2];
                        o = make_float4(r.x + alpha * o.x, r.y + alpha * o.y, r.z + alpha * o.z, r.w + alpha * o.w);
                    }
                    // the residual stream is read back by the next LayerNorm at once: ordinary stores let part
                    // of it stay in the memory-side cache (35.11 -> 35.03 ms per step); other outputs stream
                    if (EPI == OASR_EPI_RESIDUAL_F32)
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + index_of(i, h, r2)) = o;
                    else
                        store_nt16(reinterpret_cast<float*>(C) + index_of(i, h, r2),
                                   (u32x4){__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)});
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
}

// LONGK only names the instantiation (K >= 2048 or not): the code is the same, but profilers then list
// the K = 4096 down-projection and the K = 1024 projections that share an epilogue as separate kernels.
template <int EPI, bool LONGK>
__global__ __launch_bounds__(512) void gemm_bf16_kernel(const uint16_t* __restrict__ A,
                                                        const uint16_t* __restrict__ W,
                                                        const float* __restrict__ bias,
                                                        const float* resid, void* C, int64_t N,
                                                        int64_t K, float alpha, int n_tiles, int flags) {
    const bool w_packed = flags & OASR_GEMM_W_PACKED, a_packed = flags & OASR_GEMM_A_PACKED, c_packed = flags & OASR_GEMM_C_PACKED;
    // ONE object on purpose: with the ring in an array of its own the compiler orders every fragment
    // read after all copies in flight (s_waitcnt vmcnt(0) per slice), which serialises the ring
    __shared__ __attribute__((aligned(1024))) char smem[2 * kRing * kSliceBytes + 8 * kEpiSlab + 8 * 2 * 256];  // A ring | W ring | slabs | bias strips
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

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nph = (int)(K / BK);          // phases (K slices) per output tile
    const int total = my_tiles * nph;
    const int fr = lane & 15, fq = lane >> 4;
    // byte offset of this lane's 16-byte fragment inside a 16-row group of a slice
    const int off = fr * 64 + ((fq ^ ((4 - (fr >> 2)) & 3)) << 4);

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
    // Staging cursor: the next slice to copy is slice st_t of tile st_i; st_a / st_w are the
    // wave-uniform source addresses of this wave's first A and W row at that k offset.
    const uint32_t lane_off =
        (uint32_t)(((lane >> 2) * K + (((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) << 3)) * 2);
    const int64_t piece_stride = 16 * K * 2;
    // packed weights: piece (row group g, slice t) lies at ((g * nph) + t) KiB, lanes in order
    const uint32_t lane_off_w = w_packed ? (uint32_t)lane * 16 : lane_off;
    const int64_t piece_stride_w = w_packed ? (int64_t)nph * 1024 : piece_stride;
    const int w_step = w_packed ? 1024 : BK * 2;
    const uint32_t lane_off_a = a_packed ? (uint32_t)lane * 16 : lane_off;
    const int64_t piece_stride_a = a_packed ? (int64_t)nph * 1024 : piece_stride;
    const int a_step = a_packed ? 1024 : BK * 2;
    int st_i = 0, st_t = 0;
    const char *st_a, *st_w;
    auto stage_seek = [&](int i) {
        int64_t m0, n0;
        tile_origin(i, m0, n0);
        st_a = reinterpret_cast<const char*>(A) + (a_packed ? ((m0 + wave * 32) >> 4) * (int64_t)nph * 1024 : (m0 + wave * 32) * K * 2);
        st_w = reinterpret_cast<const char*>(W) + (w_packed ? ((n0 + wave * 32) >> 4) * (int64_t)nph * 1024 : (n0 + wave * 32) * K * 2);
    };
    auto stage_next = [&](int p) {  // global phase p = st_i * nph + st_t  ->  ring slot p & 3
        stage_slice(st_a, lane_off_a, piece_stride_a, smem + (p & 3) * kSliceBytes, wave);
        stage_slice(st_w, lane_off_w, piece_stride_w, smem + (kRing + (p & 3)) * kSliceBytes, wave);
        st_a += a_step;
        st_w += w_step;
        if (++st_t == nph) {
            st_t = 0;
            if (++st_i < my_tiles) stage_seek(st_i);
        }
    };
    stage_seek(0);

    const bool has_bias = EPI != OASR_EPI_GLU_BF16 && bias != nullptr;   // wave-uniform
    char* bias_strip = smem + 2 * kRing * kSliceBytes + 8 * kEpiSlab + wave * 512;
    if (has_bias) {
        int64_t m0, n0;
        tile_origin(0, m0, n0);
        bias_to_lds(bias, n0 + wn * 64, lane, bias_strip);
    }
    stage_next(0);
    if (total > 1) {
        stage_next(1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    OASR_RAW_BARRIER();
    if (wm == 1) OASR_RAW_BARRIER();  // group B runs one barrier behind

    bf16x8 xf[8], wf[4];
    int p = 0;
    f32x4 bv[4];
    for (int i = 0; i < my_tiles; ++i) {
        int64_t m0, n0;
        tile_origin(i, m0, n0);
        for (int t = 0; t < nph; ++t, ++p) {
            const char* sa = smem + (p & 3) * kSliceBytes + (wm * 128) * 64 + off;
            const char* sb = smem + (kRing + (p & 3)) * kSliceBytes + (wn * 64) * 64 + off;
            // ---- LOAD segment
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * 64);
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) xf[ii] = *reinterpret_cast<const bf16x8*>(sa + ii * 16 * 64);
            const bool more = p + 2 < total;
            // the next tile's bias goes out with the last slice of this one (newest operation: vmcnt(5))
            const bool pre = has_bias && t == nph - 1 && i + 1 < my_tiles;
            if (more) stage_next(p + 2);
            if (pre) {
                int64_t m1, n1;
                tile_origin(i + 1, m1, n1);
                bias_to_lds(bias, n1 + wn * 64, lane, bias_strip + ((i + 1) & 1) * 256);
            }
            if (!more) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (pre) {
                asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            if (t == 0) {   // this tile's bias strip has landed: it is older than the copies just issued
                if (has_bias) {
                    bias_from_lds(bv, bias_strip + (i & 1) * 256, lane);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            OASR_RAW_BARRIER();
            // ---- MFMA segment
            __builtin_amdgcn_s_setprio(1);
            if (t == 0) {   // a tile's first slice accumulates onto the bias: no clearing pass, no bias add later
#pragma unroll
                for (int ii = 0; ii < 8; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[ii], bv[j], 0, 0, 0);
            } else {
#pragma unroll
                for (int ii = 0; ii < 8; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[ii], acc[ii][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            // Group B holds the barrier that ends a tile's last MFMA segment until its epilogue is
            // done, so that the two groups' epilogues overlap instead of running one after the other
            // (group A's epilogue + first LOAD segment already sit between the same two barriers).
            if (!(wm == 1 && t == nph - 1)) OASR_RAW_BARRIER();
        }
        gemm_epilogue<EPI>(acc, resid, C, N, alpha, m0 + wm * 128, n0 + wn * 64, lane,
                           smem + 2 * kRing * kSliceBytes + wave * kEpiSlab, c_packed);
        if (wm == 1) OASR_RAW_BARRIER();
    }
    if (wm == 0) OASR_RAW_BARRIER();  // same barrier count in both groups
}


// Small-problem variant: 128 x 128 output tiles, 4 waves as 2 (m) x 2 (n), a wave owns 64 x 64.  Used
// when the 256 x 256 tiling would leave more than half of the CUs without a tile (a single 30 s
// utterance has 6 m-tiles).
// Same K slices, ring, swizzle, MFMA and epilogues as the kernel above, so every output element is
// computed by the same sequence of operations and the results are bit-identical to it.  One tile per
// work-group and one barrier per K slice; two work-groups fit a CU (73 KiB of LDS each) and cover each
// other's LOAD segments and epilogues.
constexpr int kSmallTile = 128;
constexpr int kSmallSlice = kSmallTile * BK * 2;   // 8 KiB per operand per ring slot

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_small_kernel(const uint16_t* __restrict__ A,
                                                                 const uint16_t* __restrict__ W,
                                                                 const float* __restrict__ bias, const float* resid,
                                                                 void* C, int64_t N, int64_t K, float alpha,
                                                                 int tiles_m, int n_tiles, int flags) {
    const bool w_packed = flags & OASR_GEMM_W_PACKED, a_packed = flags & OASR_GEMM_A_PACKED, c_packed = flags & OASR_GEMM_C_PACKED;
    __shared__ __attribute__((aligned(1024))) char smem[2 * kRing * kSmallSlice + 4 * kEpiSlab];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // contiguous run of tiles per XCD (work-groups with equal blockIdx % 8 share one), m fastest: the
    // work-groups of an XCD share few weight panels
    int tile;
    {
        const int q8 = n_tiles / 8, r8 = n_tiles % 8, xcd = blockIdx.x % 8;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
    }
    const int64_t m0 = (int64_t)(tile % tiles_m) * kSmallTile, n0 = (int64_t)(tile / tiles_m) * kSmallTile;
    const int fr = lane & 15, fq = lane >> 4;
    const int off = fr * 64 + ((fq ^ ((4 - (fr >> 2)) & 3)) << 4);
    const int nph = (int)(K / BK);

    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (EPI != OASR_EPI_GLU_BF16 && bias) {
            const float4 t = *reinterpret_cast<const float4*>(bias + n0 + wn * 64 + j * 16 + fq * 4);
            b = (f32x4){t.x, t.y, t.z, t.w};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = b;   // the accumulators start from the bias
    }

    // staging: a slice is 8 pieces of 16 rows per operand; wave w copies pieces 2 w, 2 w + 1 of both
    const uint32_t lane_off =
        (uint32_t)(((lane >> 2) * K + (((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) << 3)) * 2);
    const int64_t piece_stride = 16 * K * 2;
    const uint32_t lane_off_w = w_packed ? (uint32_t)lane * 16 : lane_off;
    const int64_t piece_stride_w = w_packed ? (int64_t)nph * 1024 : piece_stride;
    const int w_step = w_packed ? 1024 : BK * 2;
    const uint32_t lane_off_a = a_packed ? (uint32_t)lane * 16 : lane_off;
    const int64_t piece_stride_a = a_packed ? (int64_t)nph * 1024 : piece_stride;
    const int a_step = a_packed ? 1024 : BK * 2;
    const char* st_a = reinterpret_cast<const char*>(A) + (a_packed ? ((m0 + wave * 32) >> 4) * (int64_t)nph * 1024 : (m0 + wave * 32) * K * 2);
    const char* st_w = reinterpret_cast<const char*>(W) + (w_packed ? ((n0 + wave * 32) >> 4) * (int64_t)nph * 1024 : (n0 + wave * 32) * K * 2);
    auto stage_next = [&](int p) {
        stage_slice(st_a, lane_off_a, piece_stride_a, smem + (p & 3) * kSmallSlice, wave);
        stage_slice(st_w, lane_off_w, piece_stride_w, smem + (kRing + (p & 3)) * kSmallSlice, wave);
        st_a += a_step;
        st_w += w_step;
    };
    stage_next(0);
    if (nph > 1) {
        stage_next(1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    OASR_RAW_BARRIER();

    bf16x8 xf[4], wf[4];
    for (int p = 0; p < nph; ++p) {
        const char* sa = smem + (p & 3) * kSmallSlice + (wm * 64) * 64 + off;
        const char* sb = smem + (kRing + (p & 3)) * kSmallSlice + (wn * 64) * 64 + off;
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * 64);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) xf[ii] = *reinterpret_cast<const bf16x8*>(sa + ii * 16 * 64);
        // slot (p + 2) & 3 held slice p - 2: every wave finished reading it before the barrier of slice p - 1
        if (p + 2 < nph) {
            stage_next(p + 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // own pieces of slice p + 1 have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        OASR_RAW_BARRIER();                                    // ... and so have everyone else's
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[ii], acc[ii][j], 0, 0, 0);
    }
    gemm_epilogue<EPI, 4>(acc, resid, C, N, alpha, m0 + wm * 64, n0 + wn * 64, lane,
                          smem + 2 * kRing * kSmallSlice + wave * kEpiSlab, c_packed);
}

// Stream variant for small problems (a single utterance: 12 row tiles).  At that size a work-group
// has a CU to itself or shares it with one other, and the kernel above - every wave copying its share of
// slice p + 2 between its fragment reads and its MFMAs - runs at the latency of a global -> LDS copy per
// two slices (0.45 us per slice measured on 1536 x 1024 x 4096: 57 us, 0.09 of the MFMA peak).  Here the
// copying and the arithmetic belong to different waves: waves 4-7 (producers) do nothing but keep a ring
// of eight slices filled, up to five slices ahead of the arithmetic; waves 0-3 (consumers) read the
// fragments of slice p + 1 from LDS while their MFMAs run on slice p.  One s_barrier per slice:
//   barrier_p is passed once every producer has seen its pieces of slices <= p + 1 land;
//   after it the consumers read slice p + 1 and multiply slice p (whose fragments they hold), and the
//   producers refill the ring slot of slice p - 1 (its fragments were in registers before the consumers
//   arrived at barrier_p) with slice p + 7.
// Tile 128 x TN with TN = 128 (consumers 2 x 2, a wave owns 64 x 64) or TN = 64 (consumers 4 x 1, a wave
// owns 32 x 64: twice the work-groups when 128 x 128 tiles would leave most CUs idle).  Same K slices,
// swizzle, MFMA, accumulation order (from the bias, slice by slice) and epilogues as the other kernels:
// every output element is computed by the same sequence of operations, so results stay bit-identical to
// them - which is why K is not split over work-groups here: a split sum is another sum, and an
// utterance's states would depend on the batch it was in.
constexpr int kStreamRing = 8;

template <int N>
__device__ __forceinline__ void wait_vmcnt_slices(int slices) {   // s_waitcnt vmcnt(N * slices), slices in 0 .. 6
    switch (slices) {
        case 0: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * N) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * N) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * N) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * N) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * N) : "memory"); break;
    }
}

// One producer wave of the stream kernel: NA pieces (16 rows x one K slice, 1 KiB) of A from row a_row and NW pieces
// of W from row w_row per slice, to lds_a / lds_w inside ring slot (slice % RING); RING - 1 slices are requested
// before the first barrier, then one per barrier.  barrier_it is passed with this wave's pieces of slices <= it + 1
// in LDS; slice it + RING - 1 then goes into the slot of slice it - 1, whose fragments the consumers had in registers
// before they arrived at barrier_it (the slot of slice `it` itself may still be being read: slice 0 is read after
// barrier_0).
// The weight pieces are requested non-temporal (aux 2): at this size every step streams the whole model from HBM
// (0.58 GB, more than the memory-side cache holds), a piece is used by the few row tiles of one utterance and never
// again, and the hint shortens the copy's issue-to-landed time - which, with a fixed number of bytes in flight, is
// the rate.  Step of one 30 s utterance 2.63 -> 2.57 ms; the same hint on the activation pieces (re-read by every
// column tile) costs 13 %, and on the weights of the kernel above 4 %.
template <int NA, int NW, int RING, int SLOT>
__device__ __forceinline__ void stream_producer(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, int64_t a_row,
                                                int64_t w_row, int64_t K, int nph, int flags, char* lds_a, char* lds_w, int lane) {
    const bool w_packed = flags & OASR_GEMM_W_PACKED, a_packed = flags & OASR_GEMM_A_PACKED;
    const uint32_t lane_off =
        (uint32_t)(((lane >> 2) * K + (((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) << 3)) * 2);
    const int64_t piece_stride = 16 * K * 2;
    const uint32_t lane_off_w = w_packed ? (uint32_t)lane * 16 : lane_off;
    const int64_t piece_stride_w = w_packed ? (int64_t)nph * 1024 : piece_stride;
    const int w_step = w_packed ? 1024 : BK * 2;
    const uint32_t lane_off_a = a_packed ? (uint32_t)lane * 16 : lane_off;
    const int64_t piece_stride_a = a_packed ? (int64_t)nph * 1024 : piece_stride;
    const int a_step = a_packed ? 1024 : BK * 2;
    const char* st_a = reinterpret_cast<const char*>(A) + (a_packed ? (a_row >> 4) * (int64_t)nph * 1024 : a_row * K * 2);
    const char* st_w = reinterpret_cast<const char*>(W) + (w_packed ? (w_row >> 4) * (int64_t)nph * 1024 : w_row * K * 2);
    auto issue = [&](int p) {
        const int slot = (p % RING) * SLOT;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(st_a + i * piece_stride_a + lane_off_a), (lds_void*)(lds_a + slot + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NW; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(st_w + i * piece_stride_w + lane_off_w), (lds_void*)(lds_w + slot + i * 1024), 16, 0, 2);
        st_a += a_step;
        st_w += w_step;
    };
    int issued = 0;
    for (; issued < RING - 1 && issued < nph; ++issued) issue(issued);
    for (int it = 0; it < nph; ++it) {
        const int ahead = issued - it - 2;      // everything issued but the newest `ahead` slices must have landed
        wait_vmcnt_slices<NA + NW>(ahead < 0 ? 0 : ahead);
        OASR_RAW_BARRIER();
        if (issued < nph) issue(issued++);
    }
}

template <int EPI, int TM, int TN>
__global__ __launch_bounds__(512) void gemm_bf16_stream_kernel(const uint16_t* __restrict__ A,
                                                               const uint16_t* __restrict__ W,
                                                               const float* __restrict__ bias, const float* resid,
                                                               void* C, int64_t N, int64_t K, float alpha,
                                                               int tiles_m, int n_tiles, int flags) {
    // Eight waves: consumers CWM x CWN, a wave owning (16 MI) x 64 of the tile, then the producers.
    //   128 x 128: consumers 2 x 2 (64 x 64 each), 4 producers of 4 pieces;  128 x 64: 4 x 1 (32 x 64), 4 producers of 3;
    //   96 x 64: 3 x 1 (32 x 64), 5 producers of 2 - the tile for 1536 rows x 1024 columns (one 30 s utterance):
    //   16 x 16 = 256 equal work-groups, one per CU, where 128 x 64 tiles give 192.
    constexpr int CWN = TN / 64, CWM = TM == 96 ? 3 : 4 / CWN, kConsumers = CWM * CWN, kProducers = 8 - kConsumers;
    constexpr int MI = TM / (16 * CWM);
    constexpr int kRingSlots = kStreamRing;                            // 8 slots of 16, 12 or 10 KiB
    constexpr int kSlotA = TM * BK * 2, kSlot = (TM + TN) * BK * 2;
    static_assert(MI * 16 * CWM == TM && kProducers == (TM == 96 ? 5 : 4), "tile does not divide over the waves");
    const bool c_packed = flags & OASR_GEMM_C_PACKED;
    __shared__ __attribute__((aligned(1024))) char smem[kRingSlots * kSlot + kConsumers * kEpiSlab];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile;
    {   // contiguous run of tiles per XCD (work-groups with equal blockIdx % 8 share one), m fastest; cutting the
        // tile grid into one compact block per XCD instead changed no shape's time
        const int q8 = n_tiles / 8, r8 = n_tiles % 8, xcd = blockIdx.x % 8;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
    }
    const int tile_m = tile % tiles_m, tile_n = tile / tiles_m;
    const int64_t m0 = (int64_t)tile_m * TM, n0 = (int64_t)tile_n * TN;
    const int nph = (int)(K / BK);

    if (wave >= kConsumers) {
        // ---- producers.  Tiles of 128 rows: wave pw copies A pieces 2 pw, 2 pw + 1 and W piece(s) (TN / 64) pw .. of
        // every slice; 96 x 64: waves 0-2 copy two A pieces each, waves 3-4 two W pieces each.  The loop of a
        // producer is straight-line code for its piece counts (a per-copy choice between the operands inside one
        // generic loop ran the single-utterance step 5-11 % slower).
        const int pw = wave - kConsumers;
        if (TM == 96) {
            if (pw < 3)
                stream_producer<2, 0, kRingSlots, kSlot>(A, W, m0 + pw * 32, 0, K, nph, flags, smem + pw * 2048, nullptr, lane);
            else
                stream_producer<0, 2, kRingSlots, kSlot>(A, W, 0, n0 + (pw - 3) * 32, K, nph, flags, nullptr,
                                                         smem + kSlotA + (pw - 3) * 2048, lane);
        } else {
            constexpr int kW = TN / 64;
            stream_producer<2, kW, kRingSlots, kSlot>(A, W, m0 + pw * 32, n0 + pw * (16 * kW), K, nph, flags, smem + pw * 2048,
                                                      smem + kSlotA + pw * kW * 1024, lane);
        }
        return;
    }

    // ---- consumers
    const int wm = wave / CWN, wn = wave % CWN;
    const int fr = lane & 15, fq = lane >> 4;
    const int off = fr * 64 + ((fq ^ ((4 - (fr >> 2)) & 3)) << 4);
    f32x4 acc[MI][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (EPI != OASR_EPI_GLU_BF16 && bias) {
            const float4 t = *reinterpret_cast<const float4*>(bias + n0 + wn * 64 + j * 16 + fq * 4);
            b = (f32x4){t.x, t.y, t.z, t.w};
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = b;   // the accumulators start from the bias
    }
    bf16x8 xf[2][MI], wf[2][4];
    auto read_frags = [&](int p, bf16x8 (&x)[MI], bf16x8 (&w)[4]) {
        const char* slot = smem + (p % kRingSlots) * kSlot;
        const char* sa = slot + (wm * 16 * MI) * 64 + off;
        const char* sb = slot + kSlotA + (wn * 64) * 64 + off;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < MI; ++i) x[i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * 64);
    };
    auto multiply = [&](const bf16x8 (&x)[MI], const bf16x8 (&w)[4]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], x[i], acc[i][j], 0, 0, 0);
    };
    // A consumer passes barrier_p only with the fragments of slice p in its registers (the producers overwrite
    // that slot next): the LDS counter is drained in front of every barrier (the reads were issued a whole
    // MFMA batch earlier, so this does not wait in practice).
#define OASR_CONSUMER_BARRIER()                                \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        OASR_RAW_BARRIER();                                    \
    } while (0)
    OASR_RAW_BARRIER();                              // barrier_0: slices 0 and 1 are in LDS
    read_frags(0, xf[0], wf[0]);
    for (int it = 0; it < nph; it += 2) {
        if (it + 1 < nph) read_frags(it + 1, xf[1], wf[1]);
        multiply(xf[0], wf[0]);
        if (it + 1 >= nph) break;
        OASR_CONSUMER_BARRIER();                     // barrier_(it + 1)
        if (it + 2 < nph) read_frags(it + 2, xf[0], wf[0]);
        multiply(xf[1], wf[1]);
        if (it + 2 < nph) OASR_CONSUMER_BARRIER();   // barrier_(it + 2)
    }
#undef OASR_CONSUMER_BARRIER
    gemm_epilogue<EPI, MI>(acc, resid, C, N, alpha, m0 + wm * 16 * MI, n0 + wn * 64, lane,
                           smem + kRingSlots * kSlot + wave * kEpiSlab, c_packed);
}

// Row-major bf16 weights [n][k] -> the packed form read by the kernels above: one thread per 16-byte chunk;
// chunk `lane` of piece (row group g, K slice t) is row 16 g + (lane >> 2), source chunk
// (lane & 3) ^ (-(row >> 2) & 3) of the slice, as lane_off places it.
__global__ __launch_bounds__(256) void gemm_pack_w_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ out,
                                                          int64_t n, int64_t k) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // 16-byte chunk of the output
    const int64_t nph = k / BK, total = (n / 16) * nph * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int64_t piece = idx >> 6, g = piece / nph, t = piece - g * nph;
    const int row = lane >> 2, chunk = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
    const u32x4 v = *reinterpret_cast<const u32x4*>(w + (g * 16 + row) * k + t * BK + chunk * 8);
    *reinterpret_cast<u32x4*>(out + idx * 8) = v;
}

}  // namespace oasr

using namespace oasr;

#define OASR_TRY_GEMM(call)             \
    do {                                \
        const int rc_ = (call);         \
        if (rc_ != OASR_OK) return rc_; \
    } while (0)

static int gemm_bf16_launch(const uint16_t* a, const uint16_t* w, const float* bias, const float* resid, void* c,
                            int64_t m, int64_t n, int64_t k, float alpha, int32_t epilogue, void* stream,
                            int flags) {
    OASR_REQUIRE(a && w && c, "gemm: null pointer");
    OASR_REQUIRE(m > 0 && n > 0 && k > 0 && m % BM == 0 && n % BN == 0 && k % BK == 0,
                 "gemm: m=%lld n=%lld must be multiples of 256 and k=%lld of 32", (long long)m,
                 (long long)n, (long long)k);
    OASR_REQUIRE((m / BM) * (n / BN) < ((int64_t)1 << 30), "gemm: too many tiles");
    OASR_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)c & 15) == 0 &&
                     ((uintptr_t)bias & 15) == 0 && ((uintptr_t)resid & 15) == 0,
                 "gemm: operands must be 16-byte aligned");
    OASR_REQUIRE(epilogue != OASR_EPI_RESIDUAL_F32 || resid != nullptr, "gemm: residual epilogue without resid");
    OASR_REQUIRE((flags & ~(OASR_GEMM_W_PACKED | OASR_GEMM_A_PACKED | OASR_GEMM_C_PACKED)) == 0, "gemm: unknown flag bits %d", flags);
    OASR_REQUIRE(!(flags & OASR_GEMM_C_PACKED) || epilogue == OASR_EPI_BIAS_BF16 || epilogue == OASR_EPI_BIAS_SWISH_BF16,
                 "gemm: packed output exists for the two bf16 bias epilogues only");
    const int tiles = (int)((m / BM) * (n / BN));
#ifndef OASR_SMALL_STREAM
#define OASR_SMALL_STREAM 1
#endif
#ifndef OASR_STREAM_TN_CUT
#define OASR_STREAM_TN_CUT 128
#endif
    if (OASR_SMALL_STREAM && tiles < kMaxCUs / 2 && (m / 128) * (n / 128) <= kMaxCUs) {
        // Small problems (a single utterance) whose 128 x 128 tiles number at most one per CU: producer /
        // consumer waves on 128 x 128 tiles, or on 128 x 64 tiles when the former would occupy at most half of
        // the CUs; same arithmetic per output element as every other path.  Measured for one 30 s utterance
        // (1536 rows) inside the step, against the kernel below: FFN down-projection and d x d projections
        // 36.2 -> 21.1 us on average, GLU 15.7 -> 13.3.  A work-group takes in 50-60 GB/s whatever its shape -
        // what a CU's global -> LDS path delivered in every variant tried (four or eight producer waves, 128 x 64
        // to 256 x 128 tiles, tiles grouped per XCD or not) - so the time is the bytes of the busiest CU.  With
        // more tiles than CUs the kernel below, two work-groups to a CU, is as fast (FFN up-projection, 384
        // tiles: 21.5 us against 26.8 on 128 x 128 and 20.0 on 256 x 128 tiles here; QKV 16.8 against 23.0 and
        // 18.9) and stays in charge.
        const int64_t t128 = (m / 128) * (n / 128);
        const int tile_n = t128 <= OASR_STREAM_TN_CUT ? 64 : 128;
        // 96-row tiles when they divide m and still number at most one per CU: more, smaller, equal work-groups
        // (1536 x 1024, the shapes of one 30 s utterance: 256 work-groups where 128-row tiles give 192)
#ifndef OASR_STREAM_96
#define OASR_STREAM_96 1
#endif
        const int tile_m = OASR_STREAM_96 && tile_n == 64 && m % 96 == 0 && (m / 96) * (n / 64) <= kMaxCUs ? 96 : 128;
        const int tm = (int)(m / tile_m), tn = (int)(n / tile_n), nt = tm * tn;
        hipStream_t st = (hipStream_t)stream;
#define OASR_LAUNCH_STREAM(E)                                                                                              \
    do {                                                                                                                   \
        if (tile_m == 96)                                                                                                  \
            hipLaunchKernelGGL((gemm_bf16_stream_kernel<E, 96, 64>), dim3((unsigned)nt), dim3(512), 0, st, a, w, bias, resid, c, n, k, alpha, tm, nt, flags);   \
        else if (tile_n == 64)                                                                                             \
            hipLaunchKernelGGL((gemm_bf16_stream_kernel<E, 128, 64>), dim3((unsigned)nt), dim3(512), 0, st, a, w, bias, resid, c, n, k, alpha, tm, nt, flags);  \
        else                                                                                                               \
            hipLaunchKernelGGL((gemm_bf16_stream_kernel<E, 128, 128>), dim3((unsigned)nt), dim3(512), 0, st, a, w, bias, resid, c, n, k, alpha, tm, nt, flags); \
    } while (0)
        switch (epilogue) {
            case OASR_EPI_BIAS_BF16: OASR_LAUNCH_STREAM(OASR_EPI_BIAS_BF16); break;
            case OASR_EPI_BIAS_SWISH_BF16: OASR_LAUNCH_STREAM(OASR_EPI_BIAS_SWISH_BF16); break;
            case OASR_EPI_GLU_BF16: OASR_LAUNCH_STREAM(OASR_EPI_GLU_BF16); break;
            case OASR_EPI_RESIDUAL_F32: OASR_LAUNCH_STREAM(OASR_EPI_RESIDUAL_F32); break;
            case OASR_EPI_BIAS_F32: OASR_LAUNCH_STREAM(OASR_EPI_BIAS_F32); break;
            default:
                set_error("gemm: unknown epilogue %d", epilogue);
                return OASR_EINVAL;
        }
#undef OASR_LAUNCH_STREAM
        OASR_CHECK_LAUNCH("gemm (stream)");
        return OASR_OK;
    }
    if (tiles < kMaxCUs / 2) {   // 256 x 256 tiles for fewer than half of the CUs: 128 x 128 tiles, same arithmetic
        // (measured: faster below ~128 large tiles, slower at 192)
        const int tm = (int)(m / kSmallTile), nt = tm * (int)(n / kSmallTile);
        hipStream_t st = (hipStream_t)stream;
#define OASR_LAUNCH_SMALL(E) \
    hipLaunchKernelGGL((gemm_bf16_small_kernel<E>), dim3((unsigned)nt), dim3(256), 0, st, a, w, bias, resid, c, n, k, alpha, tm, nt, flags)
        switch (epilogue) {
            case OASR_EPI_BIAS_BF16: OASR_LAUNCH_SMALL(OASR_EPI_BIAS_BF16); break;
            case OASR_EPI_BIAS_SWISH_BF16: OASR_LAUNCH_SMALL(OASR_EPI_BIAS_SWISH_BF16); break;
            case OASR_EPI_GLU_BF16: OASR_LAUNCH_SMALL(OASR_EPI_GLU_BF16); break;
            case OASR_EPI_RESIDUAL_F32: OASR_LAUNCH_SMALL(OASR_EPI_RESIDUAL_F32); break;
            case OASR_EPI_BIAS_F32: OASR_LAUNCH_SMALL(OASR_EPI_BIAS_F32); break;
            default:
                set_error("gemm: unknown epilogue %d", epilogue);
                return OASR_EINVAL;
        }
#undef OASR_LAUNCH_SMALL
        OASR_CHECK_LAUNCH("gemm (small tiles)");
        return OASR_OK;
    }
#ifndef OASR_SPLIT_TAIL
#define OASR_SPLIT_TAIL 1
#endif
    if (OASR_SPLIT_TAIL && tiles > kMaxCUs) {
        // 256 work-groups walk the tiles in rounds; a last round with few tiles leaves most CUs idle for a whole
        // tile time (six utterances, 36 row tiles x 16: 2.25 rounds cost 3, and ran slower than eight utterances).
        // Rows are independent, so when whole rounds can be cut off at a row-tile boundary and the rest is less than
        // 5/8 of a round, the rest runs as a problem of its own on the small-tile kernels above (same arithmetic per
        // output element: the rows' results do not change).
        const int64_t tn = n / BN, tm = m / BM;
        int64_t g = tn, h = kMaxCUs;                // row tiles per whole round = 256 / gcd(tn, 256)
        while (h) { const int64_t t = g % h; g = h; h = t; }
        const int64_t per_round = kMaxCUs / g, full = tm / per_round * per_round, rest = tm - full;
        if (full > 0 && rest > 0 && rest * tn * 8 <= kMaxCUs * 5) {
            OASR_TRY_GEMM(gemm_bf16_launch(a, w, bias, resid, c, full * BM, n, k, alpha, epilogue, stream, flags));
            const int64_t row = full * BM;
            const bool f32_out = epilogue == OASR_EPI_RESIDUAL_F32 || epilogue == OASR_EPI_BIAS_F32;
            const int64_t c_cols = epilogue == OASR_EPI_GLU_BF16 ? n / 2 : n;
            // packed matrices keep a 16-row group's pieces together: [row group][K slice] of 1 KiB
            const uint16_t* a2 = a + row * k;       // the same element offset in both layouts (row is a multiple of 16)
            char* c2 = reinterpret_cast<char*>(c) + row * c_cols * (f32_out ? 4 : 2);
            const float* r2 = resid ? resid + row * c_cols : nullptr;
            return gemm_bf16_launch(a2, w, bias, r2, c2, rest * BM, n, k, alpha, epilogue, stream, flags);
        }
    }
    // one persistent work-group per CU; the grid is a multiple of 8 so that every XCD gets the same
    // number of work-groups
    dim3 grid((unsigned)(tiles < kMaxCUs ? (tiles + 7) / 8 * 8 : kMaxCUs)), blk(512);
    hipStream_t st = (hipStream_t)stream;
#define OASR_LAUNCH_GEMM(E)                                                                                          \
    do {                                                                                                             \
        if (k >= 2048)                                                                                               \
            hipLaunchKernelGGL((gemm_bf16_kernel<E, true>), grid, blk, 0, st, a, w, bias, resid, c, n, k, alpha, tiles, flags);  \
        else                                                                                                         \
            hipLaunchKernelGGL((gemm_bf16_kernel<E, false>), grid, blk, 0, st, a, w, bias, resid, c, n, k, alpha, tiles, flags); \
    } while (0)
    switch (epilogue) {
        case OASR_EPI_BIAS_BF16: OASR_LAUNCH_GEMM(OASR_EPI_BIAS_BF16); break;
        case OASR_EPI_BIAS_SWISH_BF16: OASR_LAUNCH_GEMM(OASR_EPI_BIAS_SWISH_BF16); break;
        case OASR_EPI_GLU_BF16: OASR_LAUNCH_GEMM(OASR_EPI_GLU_BF16); break;
        case OASR_EPI_RESIDUAL_F32: OASR_LAUNCH_GEMM(OASR_EPI_RESIDUAL_F32); break;
        case OASR_EPI_BIAS_F32: OASR_LAUNCH_GEMM(OASR_EPI_BIAS_F32); break;
        default:
            set_error("gemm: unknown epilogue %d", epilogue);
            return OASR_EINVAL;
    }
#undef OASR_LAUNCH_GEMM
    OASR_CHECK_LAUNCH("gemm");
    return OASR_OK;
}

extern "C" int oasr_gemm_bf16(const uint16_t* a, const uint16_t* w, const float* bias,
                              const float* resid, void* c, int64_t m, int64_t n, int64_t k,
                              float alpha, int32_t epilogue, void* stream) {
    return gemm_bf16_launch(a, w, bias, resid, c, m, n, k, alpha, epilogue, stream, 0);
}

extern "C" int oasr_gemm_bf16_pw(const uint16_t* a, const uint16_t* w_packed, const float* bias,
                                 const float* resid, void* c, int64_t m, int64_t n, int64_t k,
                                 float alpha, int32_t epilogue, void* stream) {
    return gemm_bf16_launch(a, w_packed, bias, resid, c, m, n, k, alpha, epilogue, stream, OASR_GEMM_W_PACKED);
}

extern "C" int oasr_gemm_bf16_ex(const uint16_t* a, const uint16_t* w, const float* bias, const float* resid, void* c,
                                 int64_t m, int64_t n, int64_t k, float alpha, int32_t epilogue, int32_t flags,
                                 void* stream) {
    return gemm_bf16_launch(a, w, bias, resid, c, m, n, k, alpha, epilogue, stream, flags);
}

extern "C" int oasr_gemm_pack_w_bf16(const uint16_t* w, int64_t n, int64_t k, uint16_t* out, void* stream) {
    OASR_REQUIRE(w && out && w != out, "gemm_pack_w: null pointer or in-place call");
    OASR_REQUIRE(n > 0 && k > 0 && n % 16 == 0 && k % BK == 0, "gemm_pack_w: n=%lld must be a multiple of 16 and k=%lld of 32",
                 (long long)n, (long long)k);
    OASR_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0, "gemm_pack_w: operands must be 16-byte aligned");
    const int64_t chunks = n * k / 8;
    OASR_REQUIRE(chunks / 256 + 1 < ((int64_t)1 << 31), "gemm_pack_w: matrix too large");
    hipLaunchKernelGGL(gemm_pack_w_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, out, n, k);
    OASR_CHECK_LAUNCH("gemm_pack_w");
    return OASR_OK;
}
