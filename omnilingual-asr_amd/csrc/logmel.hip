// Log-mel feature extraction for gfx950: framing, DC removal, pre-emphasis, Povey window,
// 512-point real FFT, power, mel triangles, log; then per-utterance per-bin normalisation and
// frame-pair stacking.
//
// Work decomposition: one 256-thread work-group produces kFramesPerBlock = 32 consecutive frames of
// one utterance.  The PCM span those frames cover is staged once into LDS with coalesced 16-byte
// loads (frames overlap 240 of 400 samples, so this removes the 2.5x re-read from HBM).
//
// FFT.  The 512 real samples of a frame are 256 complex points z[n], n = 16 n1 + n2.  Sixteen lanes
// own one frame (a wavefront works on four frames at once): lane n2 holds the 16 points z[16 n1 + n2]
// in registers, runs a 16-point FFT over n1 entirely in registers, multiplies by W256^(n2 k1), and the
// 16 x 16 block is transposed across the 16 lanes through a wave-private LDS tile (one write and one
// read of 16 float2 per lane - the only exchange of the transform); lane k1 then runs the second
// 16-point FFT over n2 and holds Z[k1 + 16 k2], k2 = 0..15.  The real-FFT split needs Z[256 - k],
// which sits in lane (16 - k1) mod 16: one lane permutation per register.
//
// Mel filterbank on the matrix pipe.  The power spectrum P[frame][bin] is split into a bf16 high part
// and a bf16 remainder (relative error 2^-17) and written to LDS; the 80 x 256 weight matrix, split
// the same way on the host, waits in registers as MFMA operands.  D[filter][frame] = Wh Ph + Wh Pl +
// Wl Ph with v_mfma_f32_16x16x32_bf16 over 16 frames at a time; only the 12 (filter block, K-step)
// pairs that hold a non-zero weight are multiplied (the triangles are narrow).  A lane ends up with
// four consecutive filters of one frame: the log-mel row leaves as 16-byte stores.
#include "logmel_tables.h"
#include "oasr_common.h"

namespace oasr {

constexpr int kFramesPerBlock = 32;
constexpr float kMelFloor = 1.192092955078125e-07f;
constexpr float kNormEps = 1e-7f;
constexpr float kPcmScale = 32768.0f;          // Kaldi convention: 16-bit integer range
constexpr int kPPitchDw = 128;                 // dwords per frame of the split power spectrum: 8 K-steps x 32 bf16
constexpr int kPPlaneDw = 16 * kPPitchDw;

// Orders this wave's LDS writes before its later LDS reads (DS operations of one wave execute in
// order; this only pins the compiler).
#define OASR_WAVE_SYNC()                                        \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

struct cpx {
    float re, im;
};
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cpx cmul(cpx a, cpx b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
// exp(-2 pi i k / 512) for any k >= 0 from the half-period table
__device__ __forceinline__ cpx tw512(const LogmelBlob* t, int k) {
    k &= 511;
    const float s = (k & 256) ? -1.0f : 1.0f;
    k &= 255;
    return {s * t->tw_re[k], s * t->tw_im[k]};
}

// In-place 4-point DFT: a[q] <- sum_j a[j] (-i)^(j q)
__device__ __forceinline__ void radix4(cpx (&a)[4]) {
    const cpx t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]);
    const cpx t2 = cadd(a[1], a[3]), d = csub(a[1], a[3]);
    const cpx t3 = {d.im, -d.re};
    a[0] = cadd(t0, t2);
    a[1] = cadd(t1, t3);
    a[2] = csub(t0, t2);
    a[3] = csub(t1, t3);
}

// 16-point DFT in registers, natural order in and out: X[k] = sum_n a[n] W16^(n k), as two passes of
// 4-point DFTs (n = 4 na + nb, k = ka + 4 kb) with the W16^(nb ka) twiddles in between.
__device__ __forceinline__ void fft16(const cpx (&a)[16], cpx (&X)[16]) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r = 0.70710678118654752f;
    cpx y[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        cpx v[4] = {a[nb], a[4 + nb], a[8 + nb], a[12 + nb]};
        radix4(v);
#pragma unroll
        for (int ka = 0; ka < 4; ++ka) y[nb][ka] = v[ka];
    }
    auto mul_w2 = [&](cpx v) { return cpx{(v.re + v.im) * r, (v.im - v.re) * r}; };     // W16^2 = (r, -r)
    auto mul_w6 = [&](cpx v) { return cpx{(v.im - v.re) * r, -(v.re + v.im) * r}; };    // W16^6 = (-r, -r)
    y[1][1] = cmul(y[1][1], cpx{c1, -s1});    // W16^1
    y[1][2] = mul_w2(y[1][2]);
    y[1][3] = cmul(y[1][3], cpx{s1, -c1});    // W16^3
    y[2][1] = mul_w2(y[2][1]);
    y[2][2] = cpx{y[2][2].im, -y[2][2].re};   // W16^4 = -i
    y[2][3] = mul_w6(y[2][3]);
    y[3][1] = cmul(y[3][1], cpx{s1, -c1});    // W16^3
    y[3][2] = mul_w6(y[3][2]);
    y[3][3] = cmul(y[3][3], cpx{-c1, s1});    // W16^9
#pragma unroll
    for (int ka = 0; ka < 4; ++ka) {
        cpx v[4] = {y[0][ka], y[1][ka], y[2][ka], y[3][ka]};
        radix4(v);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) X[ka + 4 * kb] = v[kb];
    }
}

__device__ __forceinline__ uint32_t cvt_pk_bf16_rne(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// Sum over the 16 lanes that share a frame, left in all of them: four rotations within the DPP row
// (row_ror:8, 4, 2, 1), on the vector ALU - no trip through the LDS crossbar.
__device__ __forceinline__ float sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// LDS of one work-group (38.6 KB: four work-groups per CU, four waves per SIMD at <= 128 registers):
//   span   the PCM span of the 16 frames of one iteration, scaled
//   tabs   window[400] | W256^(c k) for the twiddle between the FFTs [k][c] | W512^(c + 16 k) of the
//          real-FFT split [k][c] - read where they are used instead of living in registers
//   pbuf   the split power spectrum of 16 frames (MFMA B operand): [plane][frame][32 chunks of 16 bytes],
//          chunk j of frame f stored at chunk j ^ f (conflict-free 16-byte fragment reads without padding);
//          between barriers the same 16 KB serve as the transpose tiles of the four waves (real and
//          imaginary parts in turn): tile[k1][c] stored at column c ^ k1, so that both the row-wise
//          writes and the column-wise reads of 16 lanes touch 16 different banks
//   obuf   the 16 x 80 log-mel values of a unit, so that they leave as whole 128-byte lines (rows of 320
//          bytes written 64 bytes at a time would be fetched and written back twice by the memory side)
constexpr int kBufferWord3 = 0x00020000;   // raw buffer descriptor, 32-bit data format (gfx9 family)
constexpr int kIterFrames = 16;
constexpr int kIterSpan = (kIterFrames - 1) * OASR_FRAME_HOP + OASR_FRAME_LEN;   // 2800 samples
constexpr int kTileDw = 16 * 16;                                                  // dwords per frame tile
constexpr int kOPitch = OASR_NMEL + 4;                                            // floats per obuf row: 16-byte writes of 16 frames spread over the banks

__global__ __launch_bounds__(256, 4) void logmel_kernel(const float* __restrict__ pcm,
                                                        int64_t pcm_stride,
                                                        const int32_t* __restrict__ n_samples,
                                                        int64_t frame_cap,
                                                        const LogmelBlob* __restrict__ tab,
                                                        float* __restrict__ out,
                                                        float* __restrict__ stats,
                                                        int64_t stats_slots, int blocks_per_utt, int n_items) {
    __shared__ __attribute__((aligned(16))) float span[kIterSpan + 24];   // + 24: an absent frame's lanes may read (and discard) past the last frame
    __shared__ __attribute__((aligned(16))) float win[OASR_FRAME_LEN + 16];
    __shared__ __attribute__((aligned(16))) float2 twm[16 * 16], twp[16 * 16];
    __shared__ __attribute__((aligned(16))) uint32_t pbuf[2 * kPPlaneDw];
    __shared__ __attribute__((aligned(16))) float obuf[kIterFrames * kOPitch];
    static_assert(4 * 4 * kTileDw <= 2 * kPPlaneDw, "transpose tiles must fit the power-spectrum buffer");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, c = lane & 15;   // frame slot of the wave; n2 in the first FFT, k1 after the transpose

    // ---- tables to LDS (once: the work-group is persistent) -------------------------------------------
    // the 2^15 scale of the samples (Kaldi's 16-bit range) rides on the window, and so does the 1/2 of the
    // real-FFT split (E = (Z[k] + conj Z[256-k]) / 2): powers of two, so every product, sum and difference
    // downstream is the scaled one bit for bit
    for (int i = tid; i < OASR_FRAME_LEN + 16; i += 256) win[i] = i < OASR_FRAME_LEN ? tab->window[i] * (0.5f * kPcmScale) : 0.0f;
    {
        const int k = tid >> 4, cc = tid & 15;
        const cpx a = tw512(tab, 2 * cc * k), p = tw512(tab, cc + 16 * k);
        twm[k * 16 + cc] = make_float2(a.re, a.im);
        twp[k * 16 + cc] = make_float2(p.re, p.im);
    }
    // mel weights of this wave's tasks (wave 0: filter blocks 0 and 1; waves 1..3: blocks 2..4)
    const int task0 = wave == 0 ? 0 : (wave == 1 ? 3 : (wave == 2 ? 5 : 8));
    const int ntask = wave == 0 ? 3 : (wave == 1 ? 2 : (wave == 2 ? 3 : 4));
    // the frames q = 0, 1 (and 2, 3) of a wave share a tile of 16 rows x 32 dwords, 16 columns each: the two
    // frames that one LDS pass serves together then sit on different banks
    float* tile = reinterpret_cast<float*>(pbuf) + (4 * wave + (q & 2)) * kTileDw + 16 * (q & 1);

    // A unit is one iteration: 16 frames of one (utterance, block of 32 frames) item.  The work-group is
    // persistent and walks units u = 0, 1, ...: item blockIdx.x + (u >> 1) * gridDim.x, half u & 1.
    struct Unit {
        bool item_ok;     // the item exists
        int nfb;          // frames of the item (0: the utterance ends before this block)
        int nfi;          // frames of this unit (0: nothing to do)
        int b, blk;       // utterance and block index of the item
        const float* src; // first sample of the unit's span
    };
    auto unit_of = [&](int u) {
        Unit r;
        const int item = blockIdx.x + (u >> 1) * gridDim.x, it = u & 1;
        r.item_ok = item < n_items;
        r.nfb = r.nfi = 0;
        r.b = r.blk = 0;
        r.src = pcm;
        if (r.item_ok) {
            r.b = item / blocks_per_utt;
            r.blk = item - r.b * blocks_per_utt;
            int ns = n_samples[r.b];
            if (ns > (int)pcm_stride) ns = (int)pcm_stride;       // a row never extends past its pitch
            const int nf = ns < OASR_FRAME_LEN ? 0 : 1 + (ns - OASR_FRAME_LEN) / OASR_FRAME_HOP;
            const int nf_cap = nf < (int)frame_cap ? nf : (int)frame_cap;
            const int f0 = r.blk * kFramesPerBlock;
            r.nfb = max(0, min(kFramesPerBlock, nf_cap - f0));
            r.nfi = max(0, min(kIterFrames, r.nfb - kIterFrames * it));
            r.src = pcm + (int64_t)r.b * pcm_stride + (int64_t)(f0 + kIterFrames * it) * OASR_FRAME_HOP;
        }
        return r;
    };
    // Asynchronous copy of a unit's PCM span into `span` (global -> LDS directly, no registers), 16 bytes
    // per lane when the source is 16-byte aligned, else 4.  Written as inline assembly so that the only
    // waits are the explicit ones: s_waitcnt vmcnt(0) and a barrier before the span is read.
    typedef __attribute__((address_space(3))) char lds_char;
    auto stage_span = [&](const Unit& un) {
        const int count = (un.nfi - 1) * OASR_FRAME_HOP + OASR_FRAME_LEN;      // a multiple of 4
        if ((reinterpret_cast<uintptr_t>(un.src) & 15) == 0) {
            for (int base = wave * 256; base < count; base += 1024) {          // wave-uniform loop
                const uint32_t lds_off = (uint32_t)(uintptr_t)(lds_char*)(span + base);
                const float* p = un.src + base + 4 * lane;
                if (base + 4 * lane < count) {
                    uint32_t saved;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                                 "s_mov_b32 m0, %0"
                                 : "=&s"(saved) : "v"(p), "s"(lds_off) : "memory");
                }
            }
        } else {
            for (int base = wave * 64; base < count; base += 256) {
                const uint32_t lds_off = (uint32_t)(uintptr_t)(lds_char*)(span + base);
                const float* p = un.src + base + lane;
                if (base + lane < count) {
                    uint32_t saved;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
                                 "s_mov_b32 m0, %0"
                                 : "=&s"(saved) : "v"(p), "s"(lds_off) : "memory");
                }
            }
        }
    };

    float* pend_dst = nullptr;   // log-mel rows waiting in obuf: where they go and how many float4 (uniform)
    int pend_vec = 0;
    auto flush = [&]() {         // call after a barrier that follows the writes to obuf
        if (pend_dst) {
            for (int i = tid; i < pend_vec; i += 256) {
                const int row = i / (OASR_NMEL / 4), col = i - row * (OASR_NMEL / 4);
                reinterpret_cast<float4*>(pend_dst)[i] = *reinterpret_cast<const float4*>(obuf + row * kOPitch + 4 * col);
            }
            pend_dst = nullptr;
        }
    };
    Unit cur = unit_of(0);
    if (cur.nfi > 0) stage_span(cur);
    for (int u = 0; cur.item_ok; ++u) {
        const Unit nxt = unit_of(u + 1);
        const int it = u & 1;
        float* my_stats = stats + ((int64_t)cur.b * stats_slots + cur.blk) * (OASR_NMEL * 2);
        if (cur.nfi == 0) {                                   // uniform over the work-group
            if (it == 0 && threadIdx.x < OASR_NMEL * 2) my_stats[threadIdx.x] = 0.0f;   // the utterance ends before this block
            __syncthreads();                                  // obuf is complete; nobody reads the span any more
            flush();
            if (nxt.nfi > 0) stage_span(nxt);
            cur = nxt;
            continue;
        }
        const int nfb = cur.nfb, nfi = cur.nfi;
        f32x4 keep0[2], keep1[2];   // log-mel values of this lane, first and second half of the item: [accumulator]
        keep0[0] = keep0[1] = keep1[0] = keep1[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the span has landed
            __syncthreads();   // span staged; every wave has read the previous power spectrum and written its part of obuf
            flush();           // the previous unit's log-mel rows, as whole lines
            // this lane group's frame within the unit; a group past the unit's last frame repeats that frame
            // (its results are discarded below), so no lane ever reads samples that were not staged
            const int fi = min(4 * wave + q, nfi - 1);
            const float* x = span + fi * OASR_FRAME_HOP;
            // ---- load, DC removal, pre-emphasis, window -------------------------------------------------
            cpx a[16], y1[16];
            {
                float2 v[13];
                float xp[13];
                float part = 0.f;
#pragma unroll
                for (int n1 = 0; n1 < 13; ++n1) {
                    const int s = 32 * n1 + 2 * c;
                    v[n1] = *reinterpret_cast<const float2*>(x + s);
                    xp[n1] = x[s > 0 ? s - 1 : 0];
                    if (n1 == 12 && s >= OASR_FRAME_LEN) {   // past the frame's 400 samples (lanes c >= 8 of the last step)
                        v[n1] = make_float2(0.f, 0.f);
                        xp[n1] = 0.f;
                    }
                    part += v[n1].x + v[n1].y;
                }
                const float mean = sum16(part) * (1.0f / OASR_FRAME_LEN);
#pragma unroll
                for (int n1 = 0; n1 < 13; ++n1) {
                    const int s = 32 * n1 + 2 * c;
                    const float2 w = *reinterpret_cast<const float2*>(win + s);   // zero beyond the frame
                    const float fe = v[n1].x - mean, fo = v[n1].y - mean, fp = xp[n1] - mean;
                    // y[i] = f[i] - 0.97 f[i-1];  y[0] = f[0] - 0.97 f[0]
                    const float ye = (n1 == 0 && c == 0) ? fe - 0.97f * fe : fe - 0.97f * fp;
                    const float yo = fo - 0.97f * fe;
                    a[n1].re = ye * w.x;
                    a[n1].im = yo * w.y;
                }
#pragma unroll
                for (int n1 = 13; n1 < 16; ++n1) a[n1] = cpx{0.f, 0.f};
            }
            // ---- 256-point complex FFT as 16 x 16 ---------------------------------------------------------
            fft16(a, y1);
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) {
                const float2 w = twm[k1 * 16 + c];
                y1[k1] = cmul(y1[k1], cpx{w.x, w.y});
            }
            // transpose across the 16 lanes of the frame, real parts then imaginary parts
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) tile[k1 * 32 + (c ^ k1)] = y1[k1].re;
            OASR_WAVE_SYNC();
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) a[n2].re = tile[c * 32 + (n2 ^ c)];
            OASR_WAVE_SYNC();
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) tile[k1 * 32 + (c ^ k1)] = y1[k1].im;
            OASR_WAVE_SYNC();
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) a[n2].im = tile[c * 32 + (n2 ^ c)];
            fft16(a, y1);                                    // y1[k2] = Z[c + 16 k2]
            // ---- real-FFT split and power.  X[k] = E + T and X[256 - k] = conj(E - T) with E, O from the pair
            //      (Z[k], Z[256 - k]) and T = W512^k O: one pair gives both bins.  256 - (c + 16 k2) = (16 - c) + 16 (15 - k2),
            //      so the lanes of columns c and 16 - c share their pairs: each takes the pairs of its own k2 = 0..7
            //      (partner's k2 = 15..8, eight values fetched instead of sixteen) and ends with the powers of its own
            //      column's k2 = 0..7 and of the partner column's k2 = 8..15.  Columns 0 and 8 are their own partners
            //      (column 0 pairs k2 with 16 - k2: its k2 = 8 is the self-paired bin 128, |X|^2 = 4 |Z|^2 under
            //      the 1/2 that rides on the window). --------------------------------------------------------------
            float pw[16];   // [0..7]: column c, k2 = 0..7; [8..15]: column (16 - c) & 15, k2 = 8..15
            {
                const int partner = (lane & 48) | ((16 - c) & 15);
                cpx far[8];   // the partner's y1[8 .. 15]
#pragma unroll
                for (int j = 0; j < 8; ++j) far[j] = cpx{__shfl(y1[8 + j].re, partner, 64), __shfl(y1[8 + j].im, partner, 64)};
                float pm[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // column 0: partner index 16 - j (its own registers, j = 0: itself); otherwise 15 - j
                    cpx zm = far[7 - j];
                    if (j == 0) {
                        if (c == 0) zm = y1[0];
                    } else if (c == 0) {
                        zm = far[8 - j];
                    }
                    const cpx zk = y1[j];
                    const float2 w = twp[j * 16 + c];
                    const float er = zk.re + zm.re, ei = zk.im - zm.im;        // E (the 1/2 is in the window)
                    const float orr = zk.im + zm.im, oi = zm.re - zk.re;       // O
                    const float tr = w.x * orr - w.y * oi, ti = w.x * oi + w.y * orr;
                    const float ar = er + tr, ai = ei + ti, br = er - tr, bi = ei - ti;
                    pw[j] = ar * ar + ai * ai;
                    pm[j] = br * br + bi * bi;
                }
                const float p128 = 4.0f * (y1[8].re * y1[8].re + y1[8].im * y1[8].im);
#pragma unroll
                for (int t = 0; t < 8; ++t) {   // k2 = 8 + t of the partner column
                    const float zero_col = t == 0 ? p128 : pm[8 - t];
                    pw[8 + t] = c == 0 ? zero_col : pm[7 - t];
                }
            }
            // this wave's mel weights, re-read every unit (L2-resident) and only now, so that they do not
            // occupy registers during the transforms; the zero the compiler cannot see keeps the loads here
            bf16x8 wfh[4], wfl[4];
            {
                // buffer addressing (scalar descriptor + 16 lane bytes): a per-lane 64-bit pointer kept across the
                // unit loop was one of the values the register allocator spilled at 128 registers
                int opaque = 0;
                asm volatile("" : "+s"(opaque));
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<uint32_t*>(&tab->mel_frag[task0 + opaque][0][0][0]), 0, 4 * 2 * 64 * 16, kBufferWord3);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int tt = t < ntask ? t : 0;
                    wfh[t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, tt * 2048, 0));
                    wfl[t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, tt * 2048 + 1024, 0));
                }
            }
            __syncthreads();   // every wave is done with its transpose tiles: the buffer becomes the power spectrum
            // From here on the lane's LDS and buffer offsets are derived from a copy of the lane number that the
            // compiler cannot see through: otherwise it computes all of them once, in front of the unit loop, and at
            // 128 registers keeps eight of them in scratch memory (9 MB of spill stores per launch, every
            // work-group's, counted as HBM writes; recomputing them costs a dozen integer instructions per unit).
            int lane_here = lane;
            asm volatile("" : "+v"(lane_here));
            const int q = lane_here >> 4, c = lane_here & 15;
            // ---- bf16 high part and remainder to LDS: K-step s holds bins 32 s + col (slot 2 col) and
            //      32 s + 16 + col (slot 2 col + 1); this lane has K-steps 0..3 of its own column and 4..7 of its
            //      partner column ---------------------------------------------------------------------------------
            {
                // frame f = 4 wave + q; dword 16 s + col of its row lives in chunk (4 s + (col >> 2)) ^ f
                //   = 4 (s ^ wave) + ((col >> 2) ^ q): the K-step part of the swizzle is wave-uniform
                const int cp = (16 - c) & 15;
                uint32_t* dst = pbuf + (4 * wave + q) * kPPitchDw + 4 * ((c >> 2) ^ q) + (c & 3);
                uint32_t* dstp = pbuf + (4 * wave + q) * kPPitchDw + 4 * ((cp >> 2) ^ q) + (cp & 3);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float pa = pw[2 * s], pb = pw[2 * s + 1];
                    const uint32_t hi = cvt_pk_bf16_rne(pa, pb);
                    const float ra = pa - __uint_as_float(hi << 16), rb = pb - __uint_as_float(hi & 0xFFFF0000u);
                    uint32_t* d = s < 4 ? dst : dstp;
                    d[16 * (s ^ wave)] = hi;
                    d[kPPlaneDw + 16 * (s ^ wave)] = cvt_pk_bf16_rne(ra, rb);
                }
            }
            __syncthreads();   // power spectrum written; every wave has read its frames from the span
            if (nxt.nfi > 0) stage_span(nxt);               // lands while this unit finishes
            // ---- mel filterbank: D[filter][frame] over this wave's (filter block, K-step) tasks -----------
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            // B operand: column = frame c, K slots 8 q .. 8 q + 7 = chunk (4 s + q) ^ c = 4 (s ^ (c >> 2)) + (q ^ (c & 3))
            const uint32_t* frag = pbuf + c * kPPitchDw + 4 * (q ^ (c & 3));
            const int fh = c >> 2;
            auto mm = [&](f32x4 acc, int t, int s) {
                const bf16x8 ph = *reinterpret_cast<const bf16x8*>(frag + 16 * (s ^ fh));
                const bf16x8 pl = *reinterpret_cast<const bf16x8*>(frag + kPPlaneDw + 16 * (s ^ fh));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfh[t], ph, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfh[t], pl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfl[t], ph, acc, 0, 0, 0);
                return acc;
            };
            if (wave == 0) {
                acc0 = mm(acc0, 0, 0);
                acc1 = mm(acc1, 1, 0);
                acc1 = mm(acc1, 2, 1);
            } else if (wave == 1) {
                acc0 = mm(acc0, 0, 1);
                acc0 = mm(acc0, 1, 2);
            } else if (wave == 2) {
                acc0 = mm(acc0, 0, 2);
                acc0 = mm(acc0, 1, 3);
                acc0 = mm(acc0, 2, 4);
            } else {
                acc0 = mm(acc0, 0, 4);
                acc0 = mm(acc0, 1, 5);
                acc0 = mm(acc0, 2, 6);
                acc0 = mm(acc0, 3, 7);
            }
            // ---- floor, log, store: this lane holds filters 16 blk + 4 q .. + 3 of frame 16 it + c ---------
            const int blk0 = wave == 0 ? 0 : wave + 1;
            float* orow = obuf + c * kOPitch + 4 * q;
            // the item's first half still sits in obuf where this lane wrote it (already copied out): the
            // statistics below need it, and it costs no registers in between
            if (it == 1) {
                keep0[0] = *reinterpret_cast<const f32x4*>(orow + 16 * blk0);
                if (wave == 0) keep0[1] = *reinterpret_cast<const f32x4*>(orow + 16);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) keep1[0][r] = __logf(fmaxf(acc0[r], kMelFloor));
            *reinterpret_cast<f32x4*>(orow + 16 * blk0) = keep1[0];
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) keep1[1][r] = __logf(fmaxf(acc1[r], kMelFloor));
                *reinterpret_cast<f32x4*>(orow + 16) = keep1[1];
            }
            // the unit's rows are contiguous in the output: they leave together after the next barrier
            pend_dst = out + ((int64_t)cur.b * frame_cap + (cur.blk * kFramesPerBlock + kIterFrames * it)) * OASR_NMEL;
            pend_vec = nfi * (OASR_NMEL / 4);
        }
        if (!(it == 0 && nfb > kIterFrames)) {     // the item is complete (its second half, or a first half that ends it)
            // ---- per-block (mean, M2) partials over the block's frames, two passes over the kept values --
            // v0: frames 0..15 of the block, v1: frames 16..31 (absent when the block ends in its first half)
            const float inv = 1.0f / (float)nfb;
            int lane_here = lane;                  // as above: offsets recomputed here, not kept across the loop
            asm volatile("" : "+v"(lane_here));
            const int q = lane_here >> 4, c = lane_here & 15;
            const bool ok0 = c < nfb, ok1 = 16 + c < nfb;
            const __amdgpu_buffer_rsrc_t stats_rs = __builtin_amdgcn_make_buffer_rsrc(my_stats, 0, OASR_NMEL * 2 * 4, kBufferWord3);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (a == 1 && wave != 0) break;
                const int blk = a == 1 ? 1 : (wave == 0 ? 0 : wave + 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v0 = it == 0 ? keep1[a][r] : keep0[a][r], v1 = keep1[a][r];
                    const float mean = sum16((ok0 ? v0 : 0.f) + (ok1 ? v1 : 0.f)) * inv;
                    const float d0 = ok0 ? v0 - mean : 0.f, d1 = ok1 ? v1 - mean : 0.f;
                    const float m2 = sum16(d0 * d0 + d1 * d1);
                    if (c == 0) {   // (mean, M2) of filter j = 16 blk + 4 q + r -> my_stats[2 j], [2 j + 1]
                        typedef __attribute__((ext_vector_type(2))) float f32x2s;
                        __builtin_amdgcn_raw_buffer_store_b64((f32x2s){mean, m2}, stats_rs, q * 32, (16 * blk + r) * 8, 0);
                    }
                }
            }
        }
        cur = nxt;
    }   // units
    __syncthreads();
    flush();
}

// One work-group per utterance: merge the per-block partials (Chan's pairwise update, in double)
// into (mean, 1/sqrt(var_ddof1 + eps)) kept in the last stats slot.  Eight lanes share a mel bin and
// each sums every eighth slot, so the dependent chain of strided loads is 8x shorter.
constexpr int kFinalizeLanes = 8;
__global__ __launch_bounds__(OASR_NMEL * kFinalizeLanes) void stats_finalize_kernel(
    float* __restrict__ stats, int64_t stats_slots, const int32_t* __restrict__ n_samples,
    int64_t frame_cap) {
    const int b = blockIdx.x, j = threadIdx.x / kFinalizeLanes, part = threadIdx.x % kFinalizeLanes;
    const int ns = n_samples[b];
    int nf = ns < OASR_FRAME_LEN ? 0 : 1 + (ns - OASR_FRAME_LEN) / OASR_FRAME_HOP;
    if (nf > (int)frame_cap) nf = (int)frame_cap;
    float* base = stats + (int64_t)b * stats_slots * (OASR_NMEL * 2);
    const int nblk = (nf + kFramesPerBlock - 1) / kFramesPerBlock;
    double sum = 0.0;
    for (int s = part; s < nblk; s += kFinalizeLanes) {
        const int cnt = min(kFramesPerBlock, nf - s * kFramesPerBlock);
        sum += (double)cnt * (double)base[(s * OASR_NMEL + j) * 2];
    }
#pragma unroll
    for (int o = 1; o < kFinalizeLanes; o <<= 1) sum += __shfl_xor(sum, o, 64);
    const double mean = nf > 0 ? sum / nf : 0.0;
    double m2 = 0.0;
    for (int s = part; s < nblk; s += kFinalizeLanes) {
        const int cnt = min(kFramesPerBlock, nf - s * kFramesPerBlock);
        const double dm = (double)base[(s * OASR_NMEL + j) * 2] - mean;
        m2 += (double)base[(s * OASR_NMEL + j) * 2 + 1] + cnt * dm * dm;
    }
#pragma unroll
    for (int o = 1; o < kFinalizeLanes; o <<= 1) m2 += __shfl_xor(m2, o, 64);
    if (part == 0) {
        const double var = nf > 1 ? m2 / (nf - 1) : 0.0;
        float* fin = base + (stats_slots - 1) * (OASR_NMEL * 2);
        fin[j * 2 + 0] = (float)mean;
        fin[j * 2 + 1] = (float)(1.0 / sqrt(var + (double)kNormEps));
    }
}

// Normalise and stack: the [frames][80] log-mel rows of an utterance are contiguous, so a stacked
// row of 160 is two consecutive frames; only the per-utterance row pitch changes.
__global__ __launch_bounds__(256) void norm_stack_kernel(const float* __restrict__ logmel,
                                                         const float* __restrict__ stats,
                                                         int64_t stats_slots,
                                                         const int32_t* __restrict__ n_samples,
                                                         int64_t frame_cap, int64_t t_pad,
                                                         float* __restrict__ feats,
                                                         int32_t* __restrict__ lengths) {
    const int b = blockIdx.y;
    const int ns = n_samples[b];
    int nf = ns < OASR_FRAME_LEN ? 0 : 1 + (ns - OASR_FRAME_LEN) / OASR_FRAME_HOP;
    if (nf > (int)frame_cap) nf = (int)frame_cap;
    int len = nf / 2;
    if (len > (int)t_pad) len = (int)t_pad;
    if (blockIdx.x == 0 && threadIdx.x == 0) lengths[b] = len;
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;  // element within the utterance
    if (e >= t_pad * OASR_FEAT_DIM) return;
    const int row = (int)(e / OASR_FEAT_DIM);
    const int bin = (int)(e % OASR_FEAT_DIM) % OASR_NMEL;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < len) {
        const float4 x = *reinterpret_cast<const float4*>(logmel + (int64_t)b * frame_cap * OASR_NMEL + e);
        const float* fin = stats + ((int64_t)b * stats_slots + (stats_slots - 1)) * (OASR_NMEL * 2) + bin * 2;
        v.x = (x.x - fin[0]) * fin[1];
        v.y = (x.y - fin[2]) * fin[3];
        v.z = (x.z - fin[4]) * fin[5];
        v.w = (x.w - fin[6]) * fin[7];
    }
    *reinterpret_cast<float4*>(feats + (int64_t)b * t_pad * OASR_FEAT_DIM + e) = v;
}

}  // namespace oasr

using namespace oasr;

extern "C" {

int64_t oasr_logmel_stats_slots(int64_t frame_cap) {
    return (frame_cap + kFramesPerBlock - 1) / kFramesPerBlock + 1;
}

int oasr_logmel_f32(const float* pcm, int64_t pcm_stride, const int32_t* n_samples, int32_t batch,
                    int64_t frame_cap, const uint32_t* tables, float* logmel, float* stats,
                    void* stream) {
    OASR_REQUIRE(pcm && n_samples && tables && logmel && stats, "logmel: null pointer");
    OASR_REQUIRE(batch > 0 && batch <= 65535, "logmel: batch %d out of range", batch);
    OASR_REQUIRE(frame_cap > 0 && (frame_cap & 1) == 0, "logmel: frame_cap must be even and positive");
    OASR_REQUIRE(pcm_stride >= OASR_FRAME_LEN, "logmel: pcm_stride shorter than one frame");
    const int64_t slots = oasr_logmel_stats_slots(frame_cap);
    const int64_t items = (slots - 1) * (int64_t)batch;
    OASR_REQUIRE(items < ((int64_t)1 << 31), "logmel: too many frame blocks");
    // persistent work-groups, four per CU (what LDS and registers allow)
    dim3 grid((unsigned)(items < 1024 ? items : 1024));
    hipLaunchKernelGGL(logmel_kernel, grid, dim3(256), 0, (hipStream_t)stream, pcm, pcm_stride,
                       n_samples, frame_cap, reinterpret_cast<const LogmelBlob*>(tables), logmel,
                       stats, slots, (int)(slots - 1), (int)items);
    OASR_CHECK_LAUNCH("logmel");
    return OASR_OK;
}

int oasr_feature_norm_stack(const float* logmel, float* stats, const int32_t* n_samples,
                            int32_t batch, int64_t frame_cap, int64_t t_pad, float* feats,
                            int32_t* lengths, void* stream) {
    OASR_REQUIRE(logmel && stats && n_samples && feats && lengths, "norm_stack: null pointer");
    OASR_REQUIRE(batch > 0 && batch <= 65535, "norm_stack: batch %d out of range", batch);
    OASR_REQUIRE(t_pad > 0 && 2 * t_pad <= frame_cap, "norm_stack: need 2*t_pad <= frame_cap");
    const int64_t slots = oasr_logmel_stats_slots(frame_cap);
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((unsigned)batch), dim3(OASR_NMEL * kFinalizeLanes), 0,
                       (hipStream_t)stream, stats, slots, n_samples, frame_cap);
    OASR_CHECK_LAUNCH("stats_finalize");
    const int64_t vec = t_pad * OASR_FEAT_DIM / 4;
    dim3 grid((unsigned)((vec + 255) / 256), (unsigned)batch);
    hipLaunchKernelGGL(norm_stack_kernel, grid, dim3(256), 0, (hipStream_t)stream, logmel, stats,
                       slots, n_samples, frame_cap, t_pad, feats, lengths);
    OASR_CHECK_LAUNCH("norm_stack");
    return OASR_OK;
}

}  // extern "C"
