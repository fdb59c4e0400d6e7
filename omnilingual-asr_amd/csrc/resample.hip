// Rational polyphase resampler for gfx950:
//   y[n] = sum_{j < taps} hp[phase(n)][j] * x[k_hi(n) - j],
//   pos = n * down + half,  phase = pos % up,  k_hi = pos / up,   x[k] = 0 outside 0 <= k < n_in,
// where hp[p][j] = h[p + j * up] are the polyphase branches of the low-pass h (built by the host in
// float64).  A thread produces one output sample: its `taps` inputs are consecutive (neighbouring
// outputs overlap, so they come from L1/L2) and its coefficients are one row of the branch table
// (tens of kilobytes, cache resident).  Accumulation is float64 so that the float32 result equals the
// host implementation's to the last bit or one unit in the last place.  HBM-bound: n_in * 4 bytes
// in, n_out * 4 bytes out per signal.
#include "oasr_common.h"

namespace oasr {

__global__ __launch_bounds__(256) void resample_poly_kernel(const float* __restrict__ x, int64_t n_in,
                                                            int64_t x_stride, const double* __restrict__ hp,
                                                            int up, int down, int taps, int half,
                                                            float* __restrict__ y, int64_t n_out,
                                                            int64_t y_stride) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_out) return;
    const float* xs = x + (int64_t)blockIdx.y * x_stride;
    const int64_t pos = n * down + half;
    const int phase = (int)(pos % up);
    const int64_t k_hi = pos / up;
    const double* c = hp + (int64_t)phase * taps;
    double acc = 0.0;
    if (k_hi - (taps - 1) >= 0 && k_hi < n_in) {
        for (int j = 0; j < taps; ++j) acc = fma(c[j], (double)xs[k_hi - j], acc);
    } else {
        for (int j = 0; j < taps; ++j) {
            const int64_t k = k_hi - j;
            if (k >= 0 && k < n_in) acc = fma(c[j], (double)xs[k], acc);
        }
    }
    y[(int64_t)blockIdx.y * y_stride + n] = (float)acc;
}

// Interleaved integer PCM as a file holds it -> mono float32 in [-1, 1): a sample is the integer / 2^(bits - 1),
// rounded to float32 (exact up to 24 bits), and a frame is the mean of its channels taken in float64 and rounded
// once - the arithmetic of the host readers (float32 samples, channels averaged in float64).  BYTES is the
// container: 1 = unsigned 8-bit (offset 128), 2 / 4 = little-endian int16 / int32 (the buffer is aligned to the
// container), 3 = packed little-endian 24-bit.  A thread produces one frame; HBM-bound.
template <int BYTES>
__global__ __launch_bounds__(256) void pcm_to_mono_kernel(const uint8_t* __restrict__ pcm, int64_t frames, int channels,
                                                          double inv_full_scale, float* __restrict__ out) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= frames) return;
    const uint8_t* p = pcm + f * channels * BYTES;
    double sum = 0.0;
    for (int c = 0; c < channels; ++c, p += BYTES) {
        int v;
        if (BYTES == 1) v = (int)p[0] - 128;
        else if (BYTES == 2) v = *reinterpret_cast<const int16_t*>(p);
        else if (BYTES == 3) v = (int)p[0] | ((int)p[1] << 8) | ((int)(int8_t)p[2] << 16);
        else v = *reinterpret_cast<const int32_t*>(p);
        sum += (double)(float)((double)v * inv_full_scale);
    }
    out[f] = (float)(sum / channels);
}

// Batch assembly on the device: row i of `out` = pcm[start[i] .. start[i] + len[i]) followed by zeros up to `width`.
// Work-group (x, i) handles 1024 consecutive samples of row i, four per thread; HBM-bound: 4 bytes in, 4 out.
__global__ __launch_bounds__(256) void gather_chunks_kernel(const float* __restrict__ pcm, const int64_t* __restrict__ start,
                                                            const int32_t* __restrict__ len, int64_t width,
                                                            float* __restrict__ out) {
    const int i = blockIdx.y;
    const int64_t s0 = start[i];
    const int n = len[i];
    float* row = out + (int64_t)i * width;
    const int64_t j0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t j = j0 + u;
        if (j < width) row[j] = j < n ? pcm[s0 + j] : 0.0f;
    }
}

}  // namespace oasr

using namespace oasr;

extern "C" int oasr_gather_chunks(const float* pcm, int64_t pcm_len, const int64_t* start, const int32_t* len,
                                  int32_t batch, int64_t width, float* out, void* stream) {
    OASR_REQUIRE(pcm && start && len && out, "gather_chunks: null pointer");
    OASR_REQUIRE(batch > 0 && batch <= 65535 && width > 0 && pcm_len >= 0, "gather_chunks: bad batch, width or signal length");
    const int64_t blocks = (width + 1023) / 1024;
    OASR_REQUIRE(blocks < ((int64_t)1 << 31), "gather_chunks: rows too wide");
    // start / len are device arrays: their range (0 <= start, start + len <= pcm_len, len <= width) is the caller's
    // to guarantee - the host pipeline checks its plan before it uploads them
    hipLaunchKernelGGL(gather_chunks_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, pcm,
                       start, len, width, out);
    OASR_CHECK_LAUNCH("gather_chunks");
    return OASR_OK;
}

extern "C" int oasr_pcm_to_mono_f32(const void* pcm, int64_t frames, int32_t channels, int32_t sample_bytes, int32_t bits,
                                    float* out, void* stream) {
    OASR_REQUIRE(pcm && out, "pcm_to_mono: null pointer");
    OASR_REQUIRE(frames >= 0 && channels >= 1 && channels <= 64, "pcm_to_mono: frames=%lld must be >= 0 and channels=%d in 1..64",
                 (long long)frames, channels);
    OASR_REQUIRE(sample_bytes >= 1 && sample_bytes <= 4 && bits >= 2 && bits <= 8 * sample_bytes,
                 "pcm_to_mono: sample_bytes=%d must be 1..4 and bits=%d in 2..8*sample_bytes", sample_bytes, bits);
    OASR_REQUIRE(((uintptr_t)pcm & (sample_bytes == 3 ? 0 : sample_bytes - 1)) == 0, "pcm_to_mono: buffer not aligned to its samples");
    if (frames == 0) return OASR_OK;
    const int64_t blocks = (frames + 255) / 256;
    OASR_REQUIRE(blocks < ((int64_t)1 << 31), "pcm_to_mono: too many frames per launch");
    const double inv = 1.0 / (double)((int64_t)1 << (bits - 1));
    const uint8_t* src = static_cast<const uint8_t*>(pcm);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)blocks), blk(256);
    switch (sample_bytes) {
        case 1: hipLaunchKernelGGL(pcm_to_mono_kernel<1>, grid, blk, 0, st, src, frames, channels, inv, out); break;
        case 2: hipLaunchKernelGGL(pcm_to_mono_kernel<2>, grid, blk, 0, st, src, frames, channels, inv, out); break;
        case 3: hipLaunchKernelGGL(pcm_to_mono_kernel<3>, grid, blk, 0, st, src, frames, channels, inv, out); break;
        default: hipLaunchKernelGGL(pcm_to_mono_kernel<4>, grid, blk, 0, st, src, frames, channels, inv, out); break;
    }
    OASR_CHECK_LAUNCH("pcm_to_mono");
    return OASR_OK;
}

extern "C" int64_t oasr_resample_length(int64_t n_in, int32_t up, int32_t down) {
    if (n_in < 0 || up <= 0 || down <= 0) return -1;
    return (n_in * up + down - 1) / down;
}

extern "C" int oasr_resample_poly(const float* x, int64_t n_in, int64_t x_stride, int32_t batch,
                                  const double* hp, int32_t up, int32_t down, int32_t taps,
                                  int32_t half, float* y, int64_t n_out, int64_t y_stride,
                                  void* stream) {
    OASR_REQUIRE(x && hp && y, "resample: null pointer");
    OASR_REQUIRE(batch > 0 && batch <= 65535, "resample: bad batch");
    OASR_REQUIRE(up > 0 && down > 0 && taps > 0 && half >= 0, "resample: bad up/down/taps/half");
    OASR_REQUIRE(n_in >= 0 && x_stride >= n_in && y_stride >= n_out, "resample: strides shorter than the signals");
    OASR_REQUIRE(n_out == oasr_resample_length(n_in, up, down), "resample: n_out must be ceil(n_in * up / down) = %lld",
                 (long long)oasr_resample_length(n_in, up, down));
    OASR_REQUIRE(n_in < ((int64_t)1 << 40), "resample: signal too long");
    if (n_out == 0) return OASR_OK;
    const int64_t blocks = (n_out + 255) / 256;
    OASR_REQUIRE(blocks < ((int64_t)1 << 31), "resample: too many output samples per launch");
    hipLaunchKernelGGL(resample_poly_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, x, n_in, x_stride, hp, up, down, taps, half, y, n_out, y_stride);
    OASR_CHECK_LAUNCH("resample");
    return OASR_OK;
}
