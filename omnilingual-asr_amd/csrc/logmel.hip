// Log-mel feature extraction for gfx950: framing, DC removal, pre-emphasis, Povey window,
// 512-point real FFT (as a 256-point complex radix-4 FFT held across one wavefront), power,
// sparse mel triangles, log; then per-utterance per-bin normalisation and frame-pair stacking.
//
// Work decomposition: one 256-thread work-group produces kFramesPerBlock consecutive frames of one
// utterance.  The PCM span those frames cover is staged once into LDS with coalesced 16-byte
// loads (frames overlap 240 of 400 samples, so this removes the 2.5x re-read from HBM); each of
// the four wavefronts then transforms frames on its own, four complex points per lane, exchanging
// points between radix-4 stages through a wave-private LDS slab.
#include "logmel_tables.h"
#include "oasr_common.h"

namespace oasr {

constexpr int kFramesPerBlock = 32;
constexpr int kWaves = 4;
constexpr int kSpan = (kFramesPerBlock - 1) * OASR_FRAME_HOP + OASR_FRAME_LEN;  // 5360 samples
constexpr float kMelFloor = 1.192092955078125e-07f;
constexpr float kNormEps = 1e-7f;
constexpr float kPcmScale = 32768.0f;
constexpr int kMelPitch = kMelWidth + 1;       // odd pitch: lanes reading column i of 64 rows hit distinct banks
constexpr int kPPitch = 257 + kMelWidth + 3;   // room for a padded triangle starting at the last bin  // Kaldi convention: 16-bit integer range

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

// In-place radix-4 decimation-in-frequency butterfly: a[q] <- sum_j a[j] (-i)^(j q)
__device__ __forceinline__ void radix4(cpx (&a)[4]) {
    const cpx t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]);
    const cpx t2 = cadd(a[1], a[3]), d = csub(a[1], a[3]);
    const cpx t3 = {d.im, -d.re};
    a[0] = cadd(t0, t2);
    a[1] = cadd(t1, t3);
    a[2] = csub(t0, t2);
    a[3] = csub(t1, t3);
}

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ pcm,
                                                     int64_t pcm_stride,
                                                     const int32_t* __restrict__ n_samples,
                                                     int64_t frame_cap,
                                                     const LogmelBlob* __restrict__ tab,
                                                     float* __restrict__ out,
                                                     float* __restrict__ stats,
                                                     int64_t stats_slots) {
    __shared__ __attribute__((aligned(16))) float span[kSpan + 8];
    __shared__ __attribute__((aligned(16))) float2 zbuf[kWaves][256];
    __shared__ float pbuf[kWaves][kPPitch];            // power spectrum; entries >= 257 stay zero
    __shared__ float melw[81 * kMelPitch];             // padded triangle weights (block copy, odd row pitch)
    __shared__ float wstat[kWaves][OASR_NMEL][2];

    const int b = blockIdx.y;
    const int f0 = blockIdx.x * kFramesPerBlock;
    const int ns = n_samples[b];
    const int nf = ns < OASR_FRAME_LEN ? 0 : 1 + (ns - OASR_FRAME_LEN) / OASR_FRAME_HOP;
    const int nf_cap = nf < (int)frame_cap ? nf : (int)frame_cap;
    float* my_stats = stats + ((int64_t)b * stats_slots + blockIdx.x) * (OASR_NMEL * 2);
    if (f0 >= nf_cap) {
        if (threadIdx.x < OASR_NMEL * 2) my_stats[threadIdx.x] = 0.0f;
        return;
    }
    const int nfb = min(kFramesPerBlock, nf_cap - f0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- stage the PCM span (16-byte coalesced loads, scalar tail) --------------------------------
    {
        const float* src = pcm + (int64_t)b * pcm_stride + (int64_t)f0 * OASR_FRAME_HOP;
        const int count = (nfb - 1) * OASR_FRAME_HOP + OASR_FRAME_LEN;
        const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
        if (aligned) {
            const int n4 = count >> 2;
            for (int i = tid; i < n4; i += 256) {
                float4 v = reinterpret_cast<const float4*>(src)[i];
                v.x *= kPcmScale; v.y *= kPcmScale; v.z *= kPcmScale; v.w *= kPcmScale;
                reinterpret_cast<float4*>(span)[i] = v;
            }
            for (int i = (n4 << 2) + tid; i < count; i += 256) span[i] = src[i] * kPcmScale;
        } else {
            for (int i = tid; i < count; i += 256) span[i] = src[i] * kPcmScale;
        }
    }

    for (int i = tid; i < 81 * kMelWidth; i += 256) melw[(i / kMelWidth) * kMelPitch + i % kMelWidth] = tab->mel_w[i];
    for (int i = tid; i < kWaves * kPPitch; i += 256) (&pbuf[0][0])[i] = 0.0f;

    // ---- per-lane constants ------------------------------------------------------------------------
    // lane holds complex points n = lane + 64 j (j = 0..3), i.e. samples 2n and 2n+1 of the frame
    float w_even[4], w_odd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = 2 * (lane + 64 * j);
        w_even[j] = s < OASR_FRAME_LEN ? tab->window[s] : 0.0f;
        w_odd[j] = s + 1 < OASR_FRAME_LEN ? tab->window[s + 1] : 0.0f;
    }
    cpx tw1[3], tw2[3], tw3[3], twp[4];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        tw1[q - 1] = tw512(tab, 2 * lane * q);          // W256^(n q),  n = lane
        tw2[q - 1] = tw512(tab, 8 * (lane & 15) * q);   // W64^(n' q),  n' = lane & 15
        tw3[q - 1] = tw512(tab, 32 * (lane & 3) * q);   // W16^(n'' q), n'' = lane & 3
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) twp[m] = tw512(tab, lane + 64 * m);

    // mel triangles owned by this lane: A = lane, and B = 79 - lane for lane < 16 (the widest
    // triangles are paired with the narrowest; other lanes point B at the all-zero dummy row 80).
    // Weights are zero padded to kMelWidth bins, so the sum below is a fixed-length, branch-free loop.
    const int mjA = lane, mjB = lane < 16 ? 79 - lane : 80;
    const float* wA = melw + mjA * kMelPitch;
    const float* wB = melw + mjB * kMelPitch;
    const int msA = tab->mel_start[mjA], msB = tab->mel_start[mjB];

    float2* z = zbuf[wave];
    float* p = pbuf[wave];
    const int qa = lane >> 4, qb = (lane >> 2) & 3, qc = lane & 3;

    // running statistics (shifted by the first value this wave sees)
    float c0 = 0.f, c1 = 0.f, s1_0 = 0.f, s2_0 = 0.f, s1_1 = 0.f, s2_1 = 0.f;
    int nw = 0;

    __syncthreads();

    for (int fi = wave; fi < nfb; fi += kWaves) {
        const float* x = span + fi * OASR_FRAME_HOP;
        // ---- load, DC removal, pre-emphasis, window -------------------------------------------------
        float xe[4], xo[4], xp[4];
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = 2 * (lane + 64 * j);
            const bool in = s < OASR_FRAME_LEN;  // frame length is even: s+1 is in range too
            const float2 v = in ? *reinterpret_cast<const float2*>(x + s) : make_float2(0.f, 0.f);
            xe[j] = v.x;
            xo[j] = v.y;
            xp[j] = (in && s > 0) ? x[s - 1] : 0.0f;
            part += v.x + v.y;
        }
        const float mean = wave_sum(part) * (1.0f / OASR_FRAME_LEN);
        cpx a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = 2 * (lane + 64 * j);
            const float fe = xe[j] - mean, fo = xo[j] - mean, fp = xp[j] - mean;
            // y[i] = f[i] - 0.97 f[i-1];  y[0] = f[0] - 0.97 f[0]
            const float ye = (s == 0) ? fe - 0.97f * fe : fe - 0.97f * fp;
            const float yo = fo - 0.97f * fe;
            a[j].re = ye * w_even[j];   // window is zero beyond the frame
            a[j].im = yo * w_odd[j];
        }
        // ---- 256-point complex FFT, radix-4 DIF, three exchanges through LDS -----------------------
        radix4(a);
#pragma unroll
        for (int q = 1; q < 4; ++q) a[q] = cmul(a[q], tw1[q - 1]);
#pragma unroll
        for (int q = 0; q < 4; ++q) z[q * 64 + lane] = make_float2(a[q].re, a[q].im);
        OASR_WAVE_SYNC();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 v = z[qa * 64 + (lane & 15) + 16 * j];
            a[j] = {v.x, v.y};
        }
        OASR_WAVE_SYNC();
        radix4(a);
#pragma unroll
        for (int q = 1; q < 4; ++q) a[q] = cmul(a[q], tw2[q - 1]);
#pragma unroll
        for (int q = 0; q < 4; ++q) z[qa * 64 + q * 16 + (lane & 15)] = make_float2(a[q].re, a[q].im);
        OASR_WAVE_SYNC();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 v = z[qa * 64 + qb * 16 + qc + 4 * j];
            a[j] = {v.x, v.y};
        }
        OASR_WAVE_SYNC();
        radix4(a);
#pragma unroll
        for (int q = 1; q < 4; ++q) a[q] = cmul(a[q], tw3[q - 1]);
#pragma unroll
        for (int q = 0; q < 4; ++q) z[qa * 64 + qb * 16 + q * 4 + qc] = make_float2(a[q].re, a[q].im);
        OASR_WAVE_SYNC();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 v = z[qa * 64 + qb * 16 + qc * 4 + j];
            a[j] = {v.x, v.y};
        }
        OASR_WAVE_SYNC();
        radix4(a);
        // lane (qa,qb,qc), register q4 holds Z[64 q4 + 16 qc + 4 qb + qa]: store in natural order
#pragma unroll
        for (int q = 0; q < 4; ++q) z[64 * q + 16 * qc + 4 * qb + qa] = make_float2(a[q].re, a[q].im);
        OASR_WAVE_SYNC();
        // ---- real-FFT post-processing and power: X[k] = E[k] + W512^k O[k] -------------------------
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int k = lane + 64 * m;
            const float2 zk = z[k], zm = z[(256 - k) & 255];
            const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);
            const float orr = 0.5f * (zk.y + zm.y), oi = -0.5f * (zk.x - zm.x);
            const float xr = er + twp[m].re * orr - twp[m].im * oi;
            const float xi = ei + twp[m].re * oi + twp[m].im * orr;
            p[k] = xr * xr + xi * xi;
        }
        if (lane == 0) {
            const float2 z0 = z[0];
            const float x256 = z0.x - z0.y;
            p[256] = x256 * x256;
        }
        OASR_WAVE_SYNC();
        // ---- mel triangles, floor, log --------------------------------------------------------------
        float m0 = 0.f, m1 = 0.f;
#pragma unroll
        for (int i = 0; i < kMelWidth; ++i) {
            m0 = fmaf(wA[i], p[msA + i], m0);
            m1 = fmaf(wB[i], p[msB + i], m1);
        }
        const float l0 = logf(fmaxf(m0, kMelFloor));
        const float l1 = logf(fmaxf(m1, kMelFloor));
        OASR_WAVE_SYNC();  // p and z are rewritten by the next frame
        float* dst = out + ((int64_t)b * frame_cap + (f0 + fi)) * OASR_NMEL;
        dst[lane] = l0;
        if (lane < 16) dst[79 - lane] = l1;
        if (nw == 0) {
            c0 = l0;
            c1 = l1;
        }
        const float d0 = l0 - c0, d1 = l1 - c1;
        s1_0 += d0;
        s2_0 += d0 * d0;
        s1_1 += d1;
        s2_1 += d1 * d1;
        ++nw;
    }

    // ---- per-block (mean, M2) partials ---------------------------------------------------------------
    {
        const float n = (float)nw;
        const float inv = nw > 0 ? 1.0f / n : 0.0f;
        wstat[wave][lane][0] = c0 + s1_0 * inv;
        wstat[wave][lane][1] = fmaxf(s2_0 - s1_0 * s1_0 * inv, 0.0f);
        if (lane < 16) {
            wstat[wave][79 - lane][0] = c1 + s1_1 * inv;
            wstat[wave][79 - lane][1] = fmaxf(s2_1 - s1_1 * s1_1 * inv, 0.0f);
        }
    }
    __syncthreads();
    if (tid < OASR_NMEL) {
        float cnt[kWaves], tot = 0.f, mean = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            cnt[w] = w < nfb ? (float)((nfb - w + kWaves - 1) / kWaves) : 0.0f;
            tot += cnt[w];
            mean += cnt[w] * wstat[w][tid][0];
        }
        mean /= tot;
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const float dm = wstat[w][tid][0] - mean;
            m2 += wstat[w][tid][1] + cnt[w] * dm * dm;
        }
        my_stats[tid * 2 + 0] = mean;
        my_stats[tid * 2 + 1] = m2;
    }
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
    dim3 grid((unsigned)(slots - 1), (unsigned)batch);
    hipLaunchKernelGGL(logmel_kernel, grid, dim3(256), 0, (hipStream_t)stream, pcm, pcm_stride,
                       n_samples, frame_cap, reinterpret_cast<const LogmelBlob*>(tables), logmel,
                       stats, slots);
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
