// Fused multi-head self-attention for gfx950 with clamped relative-key bias and key-length mask:
//   scores[i][j] = (q_i . k_j + q_i . E[clip(j - i, -L, R) + L]) / sqrt(head);  softmax over j < len;
//   ctx_i = sum_j p[i][j] v_j,  head size 64 or 128.
// The caller hands in q already multiplied by log2(e) / sqrt(head) (folded into the q rows of the
// fused projection), so the MFMA results are scores in log2 units and no per-element scaling is left.
// Flash-style: the T x T scores never leave the CU.  Each wavefront owns 32 query rows of one
// (utterance, head) and walks the keys in tiles of 64.  Head size 64: 4 waves (128 queries) per
// work-group, three work-groups per CU.  Head size 128: K and V tiles are twice as large and a wave
// needs twice the output accumulators, so 8 waves (256 queries) share one pair of tiles and one
// work-group fills a CU (two waves per SIMD, as before).
//
// MFMA orientation (v_mfma_f32_32x32x16_bf16): S^T = K . Q^T, so a lane holds one query column and
// 16 keys of it in registers — the row max is in-register plus one exchange with lane^32, and the
// rescale factor is a per-lane scalar.  The S^T accumulator, converted to bf16, is directly the B
// operand of O^T = V^T . P^T (summing over its row index), so P never touches LDS; the row sums are
// float32 additions of the unrounded p on the vector ALU (an all-ones MFMA per k-step was tried first:
// it loads the matrix pipe, the busier one here, with a fifth more work and holds 16 registers).
// Softmax bookkeeping is relative to a lazily raised running max m: the S^T accumulators start from
// (bias - m) instead of zero in tiles where the relative-key term is one value per query, so there
// p = exp2(accumulator) with no arithmetic in between; m moves (and O, the sums and the tile's
// scores are rescaled) only when some score exceeds it by more than a threshold.
// The relative-key term needs only the L+R+1 distinct values q_i . E[c]: they are computed once per
// wave by MFMA into LDS; key tiles entirely left (right) of the clamp window use one register value.
// A wave's copies of the next key tile are issued between the MFMA pairs of S^T (their issue time is then
// covered by the matrix pipe), and its finished output tile crosses LDS once so that it is stored as whole rows.
#include <type_traits>

#include "oasr_common.h"

namespace oasr {

constexpr int kKTile = 64;     // keys per iteration
constexpr int kDistRows = 96;  // rows of the (zero padded) distance-embedding table in global memory
constexpr int kQeRows = 80;    // rows of q.E kept in LDS (>= left_max + right_max + 1)
constexpr float kRescaleThreshold = 6.0f;  // log2 units: the running max is raised only past this

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}

// This file is compiled with -fno-honor-nans (see the Makefile): without it every fmaxf() on an MFMA
// result gets a canonicalising v_max per operand (3 instructions per max).  Scores are never NaN:
// masked keys are -inf and the running max is finite from the first tile on.  (Hand-written
// v_max in inline asm is not an option: hipcc pads no MFMA -> VALU wait states around asm, and the
// max would read accumulators the matrix pipe has not written yet.)
__device__ __forceinline__ float vmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float vmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
// max over the 32 scores a lane holds: four chains of four dependent instructions, not one of sixteen
__device__ __forceinline__ float max32(const f32x16& a, const f32x16& b) {
    float m[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) m[c] = vmax(a[c], b[c]);
#pragma unroll
    for (int e = 4; e < 16; e += 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) m[c] = vmax3(m[c], a[e + c], b[e + c]);
    }
    return vmax(vmax3(m[0], m[1], m[2]), m[3]);
}

// LDS images, double buffered (one barrier per key tile):
//   k : [key][HD] bf16, rows of 2 HD bytes; 16-byte chunk c of key r sits at chunk c ^ kswz(r), which
//       makes the 32-row ds_read_b128 fragment reads conflict-free: kswz(r) = (r >> 1) & 7 for 128-byte
//       rows (two rows span the 64 banks), r & 15 for 256-byte rows (the 16 lanes of a read group
//       hold 16 different values of r & 15)
//   v : [key][HD] bf16 as it lies in memory; chunk c of key r sits at chunk c ^ vswz(r).  The V^T
//       operand of O^T = V^T . P^T is fetched with the transposing read ds_read_b64_tr_b16 (per 16
//       lanes a block of 4 keys x 16 channels, delivered as 4 keys of one channel per lane); vswz
//       sends the four keys of a block to four different bank quarters (((r >> 1) & 1) << 2 for
//       128-byte rows, where odd rows already start half a bank period later; (r & 3) << 2 for
//       256-byte rows), so the reads are conflict-free
template <int HD>
struct AttnSmem {
    static constexpr int kWaves = HD / 16;   // 4 or 8
    __attribute__((aligned(16))) char k[2][kKTile * HD * 2];
    __attribute__((aligned(16))) char v[2][kKTile * HD * 2];
    __attribute__((aligned(16))) _Float16 qe[kWaves][kQeRows * 32];   // per wave: [c][q] = q . E[c] (fp16 keeps the work-group small)
};

template <int HD>
__device__ __forceinline__ int kswz(int r) { return HD == 64 ? (r >> 1) & 7 : r & 15; }
template <int HD>
__device__ __forceinline__ int vswz(int r) { return HD == 64 ? ((r >> 1) & 1) << 2 : (r & 3) << 2; }

template <int HD>
__global__ __launch_bounds__(HD * 4, HD == 64 ? 3 : 1) void attention_relkey_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dist_emb,
    const int32_t* __restrict__ lengths, uint16_t* __restrict__ ctx, int t_pad, int n_heads,
    int q_blocks, int left_max, int right_max) {
    constexpr int kHead = HD;
    constexpr int kRowB = HD * 2;          // bytes per K / V row in LDS
    constexpr int kSteps = HD / 16;        // MFMA k-steps of S^T = K . Q^T
    constexpr int kOut = HD / 32;          // 32-channel output accumulators per wave
    constexpr int kCpr = HD / 8;           // 16-byte chunks per row
    constexpr int kQBlock = 32 * (HD / 16);  // query rows per work-group
    const int d_model = n_heads * kHead;
    __shared__ AttnSmem<HD> sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    // Work item = (utterance, head, query block).  Work-groups with equal blockIdx % 8 share an XCD
    // (observed round-robin placement; locality only): give each XCD a contiguous run of work items,
    // query block fastest, so the q-blocks that re-read one (utterance, head)'s K and V hit one L2.
    int item;
    {
        const int nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
        item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
    }
    const int qb = item % q_blocks;
    const int head = (item / q_blocks) % n_heads, b = item / (q_blocks * n_heads);
    const int len = min(lengths[b], t_pad);   // a length past the row pitch would walk into the next utterance
    const int q_blk0 = qb * kQBlock;
    if (q_blk0 >= len) return;  // whole block is padding (uniform)
    const int ld = 3 * d_model;
    const uint16_t* base = qkv + (int64_t)b * t_pad * ld + head * kHead;
    const int qw0 = q_blk0 + wave * 32;
    const bool wave_active = qw0 < len;
    const int n_dist = left_max + right_max + 1;

    // ---- Q fragments (B operand): lane (r, hh) holds Q[qw0 + r][16 s + 8 hh .. +8], s = 0..3 ---------
    bf16x8 qf[kSteps];
    {
        const int qrow = min(qw0 + r, t_pad - 1);
        const uint16_t* qp = base + qrow * ld + 8 * hh;
#pragma unroll
        for (int s = 0; s < kSteps; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
        // Pin the wait for these loads here.  Left to the compiler it lands at their first use inside the
        // key loop as s_waitcnt vmcnt(3..0), where the counter also sees the copies of the next tile
        // (issued from inline assembly, unknown to the compiler) and would wait for those every tile.
#pragma unroll
        for (int s = 0; s < kSteps; ++s) asm volatile("" : "+v"(qf[s]));
    }
    // ---- QE^T[c][q] = E[c] . Q[q]: 32-row blocks of the distance table ---------------------------------
    _Float16* qe = sm.qe[wave];
    if (wave_active) {
#pragma unroll
        for (int cb = 0; cb < kDistRows / 32; ++cb) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const uint16_t* ep = dist_emb + (cb * 32 + r) * kHead + 8 * hh;
#pragma unroll
            for (int s = 0; s < kSteps; ++s) {
                const bf16x8 ef = *reinterpret_cast<const bf16x8*>(ep + 16 * s);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ef, qf[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (c < kQeRows) qe[c * 32 + r] = (_Float16)acc[e];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const float bias_left = wave_active ? (float)qe[0 * 32 + r] : 0.f;
    const float bias_right = wave_active ? (float)qe[(n_dist - 1) * 32 + r] : 0.f;

    f32x16 o[kOut], offv;
    float lsum = 0.f;   // this lane's share of its query's row sum (keys of its half only)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
        for (int db = 0; db < kOut; ++db) o[db][e] = 0.f;
        offv[e] = 0.f;
    }
    float m_run = 0.f;       // running max (log2 units); set from the first tile's true max
    bool first_tile = true;  // wave-uniform

    // ---- staging: global -> LDS copies without a register round trip ---------------------------------------
    // A tile of K (or V) is 64 rows x kRowB bytes = kPieces pieces of 1 KiB; one global_load_lds moves a
    // piece (16 bytes per lane, placed lane-linearly), so a lane's position in the piece fixes its LDS
    // chunk and the swizzle is applied to the *source* chunk it fetches.  Wave w copies pieces w and
    // w + kWaves of both operands: rows 32 apart, for which both swizzles agree, so the second piece only
    // adds 32 rows to the source pointer.  Only the last tile can reach past row t_pad - 1: there the
    // row is clamped (those keys are masked).  Masked keys get p = 0 exactly and their V rows are
    // multiplied by it, so rows of qkv between an utterance's length and t_pad must be finite.
    constexpr int kWaves = HD / 16;
    constexpr int kRowsPerPiece = 1024 / kRowB;              // 8 or 4
    const int n_tiles = (len + kKTile - 1) / kKTile;
    const int last_row = t_pad - 1;
    const int prow = wave * kRowsPerPiece + lane / kCpr;     // this lane's row in the tile (first piece)
    const int pchunk = lane % kCpr;                          // its chunk position in LDS
    const uint16_t* kp0 = base + d_model + (pchunk ^ kswz<HD>(prow)) * 8 + (int64_t)prow * ld;
    const uint16_t* vp0 = base + 2 * d_model + (pchunk ^ vswz<HD>(prow)) * 8 + (int64_t)prow * ld;
    const int64_t tile_step = (int64_t)kKTile * ld, half_step = (int64_t)32 * ld;
    // The copy is written as inline assembly: through the builtin the compiler orders every later LDS
    // read of the same array after all copies in flight (s_waitcnt vmcnt(0) in front of the first V read
    // of each tile), which would make the prefetch of tile kt+1 wait in the middle of tile kt.  The
    // ordering that matters is explicit here: vmcnt(0) and a barrier at the end of every tile.
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) _Float16 lds_half;
    const int qe_lds = (int)(uint32_t)(uintptr_t)(lds_char*)(char*)qe;   // LDS byte address of this wave's table
    auto dma16 = [&](const uint16_t* src, char* dst) {   // dst: wave-uniform LDS address of a 1 KiB piece
        const uint32_t lds_off = (uint32_t)(uintptr_t)(lds_char*)dst;
        uint32_t saved;                       // M0 is left as it was found: the compiler reserves it
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(saved) : "v"(src), "s"(lds_off) : "memory");
    };
    // Source pointers of this wave's four pieces of tile kt (K rows, K rows + 32, V rows, V rows + 32).
    // Tiles are staged in order: the base pointers advance by one tile per call.  A tile that reaches
    // past row t_pad - 1 (only the last one, or the tile after the last that the final iteration
    // requests and nobody reads) has its rows clamped.
    struct Pieces { const uint16_t* src[4]; };
    auto next_pieces = [&](int kt) {
        Pieces pc;
        int64_t back0 = 0, back1 = 0;
        if ((kt + 1) * kKTile > t_pad) {
            const int k0 = kt * kKTile;
            back0 = (int64_t)(k0 + prow - min(k0 + prow, last_row)) * ld;
            back1 = (int64_t)(k0 + 32 + prow - min(k0 + 32 + prow, last_row)) * ld;
        }
        pc.src[0] = kp0 - back0;
        pc.src[1] = kp0 + half_step - back1;
        pc.src[2] = vp0 - back0;
        pc.src[3] = vp0 + half_step - back1;
        kp0 += tile_step;
        vp0 += tile_step;
        return pc;
    };
    auto stage_piece = [&](int buf, const Pieces& pc, int i) {
        char* dst = (i < 2 ? sm.k[buf] : sm.v[buf]) + wave * 1024 + (i & 1) * (kWaves * 1024);
        dma16(pc.src[i], dst);
    };

    // K fragment addresses: lane (r, hh) reads key r (+32), logical chunk 2 s + hh
    int k_frag[kSteps];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) k_frag[s] = r * kRowB + (((2 * s + hh) ^ kswz<HD>(r)) << 4);
    // kswz(32 + r) = kswz(r), so the second half only adds 32 rows
    // V^T fragments by transposing reads.  In its group of 16 lanes, lane 4 q + p supplies the address
    // of key (block base + q), channels 16 (g & 1) + 4 p .. + 3, and receives the block's 4 keys of
    // channel 16 (g & 1) + (lane & 15) = lane & 31 (g = lane >> 4; hh = g >> 1 picks the keys).
    // Block base of piece w of k-step (u, s2): key 32 u + 16 s2 + 8 w + 4 hh.
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
    int v_frag[kOut];   // byte offset inside the tile for output block db at key block 0 (key = 4 hh + tq)
#pragma unroll
    for (int db = 0; db < kOut; ++db) {
        const int key = 4 * hh + tq;
        const int cbyte = (32 * db + 16 * tg + 4 * tp) * 2;   // byte offset of the channels inside the row
        v_frag[db] = key * kRowB + ((((cbyte >> 4) ^ vswz<HD>(key)) << 4) | (cbyte & 15));
    }

    {
        const Pieces pc = next_pieces(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_piece(0, pc, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // One key tile.  FAR: every (query, key) distance of this wave's rows lies outside the clamp
    // window on one side and no key is masked, so the relative-key term is one value per lane and
    // the accumulators start from offv = (that value - m_run).  NEAR: per-element table gather.
    // NEAR_MASKED: the last tile when it has keys >= len.  The cases are separate loops below (not
    // flags inside one loop) so that the far path carries none of the gather and masking code.
    constexpr int FAR = 0, NEAR = 1, NEAR_MASKED = 2;
    auto process_tile = [&](auto kind_c, int kt) {
        constexpr int KIND = decltype(kind_c)::value;
        const int buf = kt & 1;
        // Tile kt+1 goes into the other buffer (free since the barrier that ended tile kt-1).  Its four
        // copies are issued between the MFMAs of S^T, one per pair: a copy holds the wave's issue for
        // ~100 cycles, which the matrix pipe spends on the pair before it.  The last iteration requests
        // one tile more than exists (rows clamped, nobody reads it) so that the loop has no branch here.
        const Pieces pc = next_pieces(kt + 1);
        if (wave_active) {
            const int k0 = kt * kKTile;
            const char* kbuf = sm.k[buf];
            const char* vbuf = sm.v[buf];
            // ---- S^T = K . Q^T for the two 32-key halves; the accumulators start from offv ------------------
            f32x16 s0 = offv, s1 = offv;   // FAR: bias - m_run; otherwise -m_run
            // K fragments of four k-steps are requested together, so the MFMAs wait for the LDS once per
            // group instead of once per pair
#pragma unroll
            for (int s4 = 0; s4 < kSteps; s4 += 4) {
                bf16x8 ka[4], kb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ka[j] = *reinterpret_cast<const bf16x8*>(kbuf + k_frag[s4 + j]);
                    kb[j] = *reinterpret_cast<const bf16x8*>(kbuf + 32 * kRowB + k_frag[s4 + j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[j], qf[s4 + j], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[j], qf[s4 + j], s1, 0, 0, 0);
                    if (s4 == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        stage_piece(buf ^ 1, pc, j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (s4 != 0) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // the 8 LDS reads first ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // ... then the 8 MFMAs
                }
            }
            if (KIND != FAR) {
                // table entry (c, r) lies at LDS byte qe_lds + 64 c + 2 r with c = clamp(key - q, -L, R) + L:
                // one add and one three-operand median give the address (the accumulators already hold
                // score - m_run, so the entry is only converted and added)
                const int cb64 = qe_lds + (k0 - (qw0 + r) + left_max + 4 * hh) * 64 + 2 * r;
                const int lo = qe_lds + 2 * r, hi = lo + (n_dist - 1) * 64;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int krow = (e & 3) + 8 * (e >> 2);
                    int a0 = cb64 + krow * 64, a1 = cb64 + (krow + 32) * 64;
                    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(a0) : "v"(a0), "v"(lo), "v"(hi));
                    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(a1) : "v"(a1), "v"(lo), "v"(hi));
                    const float t0 = s0[e] + (float)*(const lds_half*)(uintptr_t)(uint32_t)a0;
                    const float t1 = s1[e] + (float)*(const lds_half*)(uintptr_t)(uint32_t)a1;
                    if (KIND == NEAR_MASKED) {
                        const int key = k0 + krow + 4 * hh;
                        s0[e] = key < len ? t0 : -INFINITY;
                        s1[e] = key + 32 < len ? t1 : -INFINITY;
                    } else {
                        s0[e] = t0;
                        s1[e] = t1;
                    }
                }
            }
            // ---- raise the running max if needed (always on the first tile) ---------------------------------
            // A lane sees the 32 keys of its half; the other half of its query's keys lives in lane ^ 32.
            // The common case needs no exchange: no lane of the wave is past the threshold.
            float tmax = max32(s0, s1);
            if (__builtin_amdgcn_ballot_w64(first_tile || tmax > kRescaleThreshold) != 0) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
                tmax = vmax(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                // both lanes of a query (r, r + 32) now hold the same tmax and m_run: same decision
                const bool raise = first_tile || tmax > kRescaleThreshold;
                const float delta = raise ? tmax : 0.f;
                const float alpha = first_tile ? 1.f : __builtin_amdgcn_exp2f(-delta);   // O and sums are 0 at first
                m_run += delta;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
#pragma unroll
                    for (int db = 0; db < kOut; ++db) o[db][e] *= alpha;
                    s0[e] -= delta;
                    s1[e] -= delta;
                    offv[e] -= delta;
                }
                lsum *= alpha;
                first_tile = false;
            }
            // ---- p = exp2(score - m_run) -----------------------------------------------------------------------
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s0[e] = __builtin_amdgcn_exp2f(s0[e]);
                s1[e] = __builtin_amdgcn_exp2f(s1[e]);
            }
            {   // row sum on the vector ALU: four independent chains
                float a0 = s0[0] + s1[0], a1 = s0[1] + s1[1], a2 = s0[2] + s1[2], a3 = s0[3] + s1[3];
#pragma unroll
                for (int e = 4; e < 16; e += 4) {
                    a0 += s0[e] + s1[e];
                    a1 += s0[e + 1] + s1[e + 1];
                    a2 += s0[e + 2] + s1[e + 2];
                    a3 += s0[e + 3] + s1[e + 3];
                }
                lsum += (a0 + a1) + (a2 + a3);
            }
            // ---- O^T += V^T . P^T ------------------------------------------------------------------------------
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
                    // key blocks 32 u + 16 s2 + 8 w (+ 4 hh + tq in v_frag): multiples of 8 leave vswz alone
                    const int kb_off = (32 * u + 16 * s2) * kRowB;
#pragma unroll
                    for (int db = 0; db < kOut; ++db) {
                        typedef __attribute__((ext_vector_type(4))) short s16x4;
                        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                        const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vbuf + kb_off + v_frag[db]));
                        const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vbuf + kb_off + 8 * kRowB + v_frag[db]));
                        const bf16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
                    }
                }
            }
        }
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) stage_piece(buf ^ 1, pc, i);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt+1 have landed
        __syncthreads();  // tile kt+1 is staged; everyone is done reading buffer kt & 1
    };
    // start value of the S^T accumulators: (bias - m_run) in FAR tiles, -m_run in the others
    auto set_far_bias = [&](float bias) {
#pragma unroll
        for (int e = 0; e < 16; ++e) offv[e] = bias - m_run;
    };

    // Tile ranges of this wave (wave-uniform; every wave still runs n_tiles tiles = n_tiles barriers):
    //   [0, n_left)        all distances <= -left_max        -> FAR with the left-edge bias
    //   [n_left, first_r)  inside the clamp window           -> general
    //   [first_r, n_full)  all distances >= right_max        -> FAR with the right-edge bias
    //   [n_full, n_tiles)  the last tile when it has masked keys -> general
    const int n_full = len / kKTile;   // tiles without masked keys
    int n_left = (qw0 - (kKTile - 1) - left_max >= 0) ? (qw0 - (kKTile - 1) - left_max) / kKTile + 1 : 0;
    int first_r = (qw0 + 31 + right_max + kKTile - 1) / kKTile;
    n_left = min(n_left, n_full);
    first_r = min(max(first_r, n_left), n_full);
    int kt = 0;
    set_far_bias(bias_left);
    for (; kt < n_left; ++kt) process_tile(std::integral_constant<int, FAR>{}, kt);
    set_far_bias(0.f);
    for (; kt < first_r; ++kt) process_tile(std::integral_constant<int, NEAR>{}, kt);
    set_far_bias(bias_right);
    for (; kt < n_full; ++kt) process_tile(std::integral_constant<int, FAR>{}, kt);
    set_far_bias(0.f);
    for (; kt < n_tiles; ++kt) process_tile(std::integral_constant<int, NEAR_MASKED>{}, kt);

    if (!wave_active) return;
    // ---- normalise and store: lane holds ctx[q][32 db + 8 g + 4 hh + {0..3}] ---------------------------------
    float inv;
    {   // the other half of this query's keys was summed by lane ^ 32
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
        inv = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
    }
    // The wave's 32 x HD output tile crosses LDS once (the K/V buffers are free: every wave is past the
    // last barrier and no copy is in flight), so that it leaves as whole rows: a lane holds 8-byte pieces
    // of one row, and stored from there each instruction would touch 64 different lines.  Layout of the
    // wave's strip: row q at q * kRowB, 16-byte chunk c at position c ^ (q % kCpr).
    char* strip = sm.k[0] + wave * (32 * kRowB);   // k and v are contiguous: 4 x 4 KiB or 8 x 8 KiB
#pragma unroll
    for (int db = 0; db < kOut; ++db) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 pk = make_uint2(cvt_pk_bf16(o[db][4 * g] * inv, o[db][4 * g + 1] * inv),
                                        cvt_pk_bf16(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
            const int chunk = 4 * db + g;   // channels 32 db + 8 g + 4 hh .. + 3: half hh of chunk 4 db + g
            *reinterpret_cast<uint2*>(strip + r * kRowB + ((chunk ^ (r & (kCpr - 1))) << 4) + 8 * hh) = pk;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int kRowsPerStore = 64 / kCpr;   // 8 or 4 whole rows per store instruction
    const int srow = lane / kCpr, schunk = lane % kCpr;
    uint16_t* op = ctx + ((int64_t)b * t_pad + qw0) * d_model + head * kHead + 8 * schunk;
#pragma unroll
    for (int i = 0; i < 32 / kRowsPerStore; ++i) {
        const int q = i * kRowsPerStore + srow;
        const uint4 v = *reinterpret_cast<const uint4*>(strip + q * kRowB + ((schunk ^ (q & (kCpr - 1))) << 4));
        if (qw0 + q < t_pad) *reinterpret_cast<uint4*>(op + (int64_t)q * d_model) = v;
    }
}

}  // namespace oasr

using namespace oasr;

extern "C" int oasr_attention_relkey(const uint16_t* qkv, const uint16_t* dist_emb,
                                     const int32_t* lengths, uint16_t* ctx, int32_t batch,
                                     int64_t t_pad, int32_t n_heads, int32_t head_dim, int32_t left_max,
                                     int32_t right_max, void* stream) {
    OASR_REQUIRE(qkv && dist_emb && lengths && ctx, "attention: null pointer");
    OASR_REQUIRE(batch > 0 && batch <= 65535 && n_heads > 0 && n_heads <= 65535, "attention: bad batch/heads");
    OASR_REQUIRE(head_dim == 64 || head_dim == 128, "attention: head size %d (64 and 128 are built)", head_dim);
    OASR_REQUIRE(t_pad > 0 && t_pad % 32 == 0, "attention: t_pad must be a positive multiple of 32");
    OASR_REQUIRE(left_max >= 0 && right_max >= 0 && left_max + right_max + 1 <= kQeRows,
                 "attention: left_max + right_max + 1 must be <= %d", kQeRows);
    OASR_REQUIRE((int64_t)batch * t_pad * 3 * n_heads * head_dim < ((int64_t)1 << 31),
                 "attention: qkv matrix too large for 32-bit element offsets");
    const int q_block = 2 * head_dim;   // 32 query rows per wave, head_dim / 16 waves
    const int q_blocks = (int)((t_pad + q_block - 1) / q_block);
    OASR_REQUIRE((int64_t)q_blocks * n_heads * batch < ((int64_t)1 << 31), "attention: too many work items");
    dim3 grid((unsigned)(q_blocks * n_heads * batch));
    if (head_dim == 64)
        hipLaunchKernelGGL(attention_relkey_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, qkv,
                           dist_emb, lengths, ctx, (int)t_pad, n_heads, q_blocks, left_max, right_max);
    else
        hipLaunchKernelGGL(attention_relkey_kernel<128>, grid, dim3(512), 0, (hipStream_t)stream, qkv,
                           dist_emb, lengths, ctx, (int)t_pad, n_heads, q_blocks, left_max, right_max);
    OASR_CHECK_LAUNCH("attention");
    return OASR_OK;
}
