// bf16 MFMA GEMM with fused epilogues for gfx950:  C[m][n] = epi( sum_k A[m][k] * W[n][k] ).
//
// Both operands are K-contiguous (activations [m][k], Linear weights [n][k]), so both tiles are
// staged the same way: 256 rows x 64 k (128-byte rows) per K-step, copied HBM -> LDS directly by
// global_load_lds (16 bytes per lane, no VGPR round trip), double buffered.  The LDS image is
// lane-linear per wave instruction, so the bank-conflict swizzle (16-byte chunk index XOR row&7)
// is applied to the per-lane *source* address and again on the fragment read.
//
// 8 wavefronts as 2 (m) x 4 (n); a wave owns 128 m x 64 n as 8 x 4 tiles of
// v_mfma_f32_16x16x32_bf16.  The weight fragment is fed as the MFMA's A operand and the
// activation fragment as B, so a lane ends up holding 4 consecutive n for one m and the epilogue
// stores 8 (bf16) or 16 (f32) contiguous bytes per lane.
#include "oasr_common.h"

namespace oasr {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kTileBytes = BM * BK * 2;  // 32 KiB per operand per buffer

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// Stage one 256 x 64 bf16 tile.  Each wave issues 4 global_load_lds of 1 KiB (8 rows each).
__device__ __forceinline__ void stage_tile(const uint16_t* __restrict__ g, int64_t ld, int64_t row0,
                                           int64_t k0, char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;             // 8 rows per piece
        const int row = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);   // logical 16-byte chunk this lane must fetch
        const uint16_t* src = g + (row0 + row) * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds_tile + piece * 1024), 16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_kernel(const uint16_t* __restrict__ A,
                                                        const uint16_t* __restrict__ W,
                                                        const float* __restrict__ bias,
                                                        const float* resid, void* C, int64_t M,
                                                        int64_t N, int64_t K, float alpha) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * kTileBytes];  // A0 A1 B0 B1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // XCD-aware block order: ids that share an XCD (id % 8) get a contiguous run of tiles, n
    // fastest, so the A rows of an m-tile are fetched into one L2 and reused by all its n-tiles.
    const int tiles_n = (int)(N / BN);
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int64_t m0 = (int64_t)(bid / tiles_n) * BM;
    const int64_t n0 = (int64_t)(bid % tiles_n) * BN;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    // byte offset of this lane's 16-byte fragment inside a 16-row group, for the two k-halves
    const int off0 = fr * 128 + (((0 + fq) ^ (fr & 7)) << 4);
    const int off1 = fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);

    const int nt = (int)(K / BK);
    stage_tile(A, K, m0, 0, smem, wave, lane);
    stage_tile(W, K, n0, 0, smem + 2 * kTileBytes, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            stage_tile(A, K, m0, (int64_t)(t + 1) * BK, smem + (cur ^ 1) * kTileBytes, wave, lane);
            stage_tile(W, K, n0, (int64_t)(t + 1) * BK, smem + (2 + (cur ^ 1)) * kTileBytes, wave, lane);
        }
        const char* sa = smem + cur * kTileBytes + (wm * 128) * 128;
        const char* sb = smem + (2 + cur) * kTileBytes + (wn * 64) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int off = kk ? off1 : off0;
            bf16x8 xf[8], wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * 128 + off);
#pragma unroll
            for (int i = 0; i < 8; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * 128 + off);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n .. n+3], m = .. + (lane & 15), n = .. + 4 * (lane >> 4) ----------
    const int64_t mrow0 = m0 + wm * 128 + fr;
    const int64_t ncol0 = n0 + wn * 64 + fq * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t m = mrow0 + i * 16;
        if (EPI == OASR_EPI_GLU_BF16) {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const f32x4 v = acc[i][j], g = acc[i][j + 1];
                const int64_t oc = ((n0 + wn * 64 + j * 16) >> 1) + fq * 4;
                uint2 pk = make_uint2(pack_bf16x2(v[0] * sigmoidf(g[0]), v[1] * sigmoidf(g[1])),
                                      pack_bf16x2(v[2] * sigmoidf(g[2]), v[3] * sigmoidf(g[3])));
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(C) + m * (N >> 1) + oc) = pk;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = ncol0 + j * 16;
                f32x4 v = acc[i][j];
                if (bias != nullptr) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (EPI == OASR_EPI_BIAS_SWISH_BF16) {
                    v[0] = swishf(v[0]); v[1] = swishf(v[1]); v[2] = swishf(v[2]); v[3] = swishf(v[3]);
                }
                if (EPI == OASR_EPI_BIAS_BF16 || EPI == OASR_EPI_BIAS_SWISH_BF16) {
                    uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(C) + m * N + n) = pk;
                } else if (EPI == OASR_EPI_RESIDUAL_F32) {
                    const float4 r = *reinterpret_cast<const float4*>(resid + m * N + n);
                    float4 o = make_float4(r.x + alpha * v[0], r.y + alpha * v[1], r.z + alpha * v[2],
                                           r.w + alpha * v[3]);
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + m * N + n) = o;
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + m * N + n) =
                        make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

}  // namespace oasr

using namespace oasr;

extern "C" int oasr_gemm_bf16(const uint16_t* a, const uint16_t* w, const float* bias,
                              const float* resid, void* c, int64_t m, int64_t n, int64_t k,
                              float alpha, int32_t epilogue, void* stream) {
    OASR_REQUIRE(a && w && c, "gemm: null pointer");
    OASR_REQUIRE(m > 0 && n > 0 && k > 0 && m % BM == 0 && n % BN == 0 && k % BK == 0,
                 "gemm: m=%lld n=%lld must be multiples of 256 and k=%lld of 64", (long long)m,
                 (long long)n, (long long)k);
    OASR_REQUIRE((m / BM) * (n / BN) < ((int64_t)1 << 31), "gemm: too many tiles");
    OASR_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)c & 15) == 0,
                 "gemm: operands must be 16-byte aligned");
    OASR_REQUIRE(epilogue != OASR_EPI_RESIDUAL_F32 || resid != nullptr, "gemm: residual epilogue without resid");
    dim3 grid((unsigned)((m / BM) * (n / BN))), blk(512);
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case OASR_EPI_BIAS_BF16:
            hipLaunchKernelGGL((gemm_bf16_kernel<OASR_EPI_BIAS_BF16>), grid, blk, 0, st, a, w, bias, resid, c, m, n, k, alpha);
            break;
        case OASR_EPI_BIAS_SWISH_BF16:
            hipLaunchKernelGGL((gemm_bf16_kernel<OASR_EPI_BIAS_SWISH_BF16>), grid, blk, 0, st, a, w, bias, resid, c, m, n, k, alpha);
            break;
        case OASR_EPI_GLU_BF16:
            hipLaunchKernelGGL((gemm_bf16_kernel<OASR_EPI_GLU_BF16>), grid, blk, 0, st, a, w, bias, resid, c, m, n, k, alpha);
            break;
        case OASR_EPI_RESIDUAL_F32:
            hipLaunchKernelGGL((gemm_bf16_kernel<OASR_EPI_RESIDUAL_F32>), grid, blk, 0, st, a, w, bias, resid, c, m, n, k, alpha);
            break;
        case OASR_EPI_BIAS_F32:
            hipLaunchKernelGGL((gemm_bf16_kernel<OASR_EPI_BIAS_F32>), grid, blk, 0, st, a, w, bias, resid, c, m, n, k, alpha);
            break;
        default:
            set_error("gemm: unknown epilogue %d", epilogue);
            return OASR_EINVAL;
    }
    OASR_CHECK_LAUNCH("gemm");
    return OASR_OK;
}
