// Host-only entry points of liboasr_hip.so: version, error text, chunk plan, shard ranges, frame
// arithmetic and the constant tables of the feature kernel.
#include <math.h>
#include <stdarg.h>

#include <vector>

#include "logmel_tables.h"
#include "oasr_common.h"

namespace oasr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- feature tables (double precision, rounded once to float) -------------------------------

static double mel_of_hz(double f) { return 1127.0 * log(1.0 + f / 700.0); }

void build_window(float* w) {
    for (int n = 0; n < OASR_FRAME_LEN; ++n) {
        double hann = 0.5 - 0.5 * cos(2.0 * M_PI * n / (OASR_FRAME_LEN - 1));
        w[n] = (float)pow(hann, 0.85);
    }
}

void build_mel_dense(double* m /* [257][80] */) {
    const int nb = OASR_NFFT / 2 + 1;
    const double lo = mel_of_hz(20.0), hi = mel_of_hz(OASR_SAMPLE_RATE / 2.0);
    std::vector<double> edge(OASR_NMEL + 2);
    // same spacing rule as numpy.linspace: start + i*step, last point exact
    const double step = (hi - lo) / (OASR_NMEL + 1);
    for (int i = 0; i < OASR_NMEL + 2; ++i) edge[i] = lo + step * i;
    edge[OASR_NMEL + 1] = hi;
    for (int k = 0; k < nb; ++k) {
        const double bm = mel_of_hz(k * ((double)OASR_SAMPLE_RATE / OASR_NFFT));
        for (int j = 0; j < OASR_NMEL; ++j) {
            const double up = (bm - edge[j]) / (edge[j + 1] - edge[j]);
            const double down = (edge[j + 2] - bm) / (edge[j + 2] - edge[j + 1]);
            const double v = fmin(up, down);
            m[k * OASR_NMEL + j] = v > 0.0 ? v : 0.0;
        }
    }
}

int build_logmel_blob(uint32_t* blob) {
    LogmelBlob* b = reinterpret_cast<LogmelBlob*>(blob);
    memset(b, 0, sizeof(LogmelBlob));
    build_window(b->window);
    for (int k = 0; k < 256; ++k) {
        b->tw_re[k] = (float)cos(2.0 * M_PI * k / 512.0);
        b->tw_im[k] = (float)(-sin(2.0 * M_PI * k / 512.0));
    }
    const int nb = OASR_NFFT / 2 + 1;
    std::vector<double> dense((size_t)nb * OASR_NMEL);
    build_mel_dense(dense.data());
    int widest = 0;
    for (int j = 0; j < OASR_NMEL; ++j) {
        int first = -1, last = -1;
        for (int k = 0; k < nb; ++k)
            if (dense[k * OASR_NMEL + j] > 0.0) {
                if (first < 0) first = k;
                last = k;
            }
        if (first < 0) {
            first = 0;
            last = -1;
        }
        const int cnt = last - first + 1;
        if (cnt > kMelWidth) return -1;
        if (cnt > widest) widest = cnt;
        b->mel_start[j] = first;
        for (int i = 0; i < cnt; ++i) b->mel_w[j * kMelWidth + i] = (float)dense[(first + i) * OASR_NMEL + j];
    }
    b->mel_start[OASR_NMEL] = 0;
    b->mel_width_max = widest;
    // MFMA form of the weights; every non-zero weight must fall into one of the listed tasks
    auto to_bf16 = [](float f) {
        uint32_t u;
        memcpy(&u, &f, 4);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    auto from_bf16 = [](uint16_t h) {
        uint32_t u = (uint32_t)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    std::vector<char> covered((size_t)nb * OASR_NMEL, 0);
    for (int t = 0; t < kMelTasks; ++t)
        for (int l = 0; l < 64; ++l)
            for (int d = 0; d < 4; ++d) {
                uint32_t hi = 0, lo = 0;
                for (int h = 0; h < 2; ++h) {
                    const int sigma = 8 * (l >> 4) + 2 * d + h;
                    const int bin = 32 * kMelTaskStep[t] + (sigma >> 1) + 16 * (sigma & 1);
                    const int filt = 16 * kMelTaskBlock[t] + (l & 15);
                    const float w = (float)dense[(size_t)bin * OASR_NMEL + filt];
                    covered[(size_t)bin * OASR_NMEL + filt] = 1;
                    const uint16_t wh = to_bf16(w), wl = to_bf16(w - from_bf16(wh));
                    hi |= (uint32_t)wh << (16 * h);
                    lo |= (uint32_t)wl << (16 * h);
                }
                b->mel_frag[t][0][l][d] = hi;
                b->mel_frag[t][1][l][d] = lo;
            }
    for (int k = 0; k < nb; ++k)
        for (int j = 0; j < OASR_NMEL; ++j)
            if (dense[(size_t)k * OASR_NMEL + j] > 0.0 && !covered[(size_t)k * OASR_NMEL + j]) return -1;
    return 0;
}

}  // namespace oasr

using namespace oasr;

extern "C" {

int oasr_version(void) { return OASR_VERSION; }

const char* oasr_last_error(void) { return g_err; }

int oasr_plan_chunks(int64_t total_samples, int32_t sample_rate, double chunk_seconds,
                     double min_chunking_seconds, int64_t* out_bounds, int64_t capacity_chunks,
                     int64_t* n_chunks) {
    OASR_REQUIRE(n_chunks != nullptr, "plan_chunks: n_chunks is null");
    OASR_REQUIRE(sample_rate > 0, "plan_chunks: sample_rate must be positive");
    OASR_REQUIRE(out_bounds != nullptr || capacity_chunks == 0, "plan_chunks: out_bounds is null");
    const int64_t total = total_samples > 0 ? total_samples : 0;
    const double duration = (double)total / (double)sample_rate;
    int64_t step = total;  // one chunk
    if (total > 0 && duration > min_chunking_seconds) {
        step = (int64_t)nearbyint(chunk_seconds * (double)sample_rate);   // ties to even, like the oracle
        OASR_REQUIRE(step > 0, "plan_chunks: chunk_seconds too small");
    }
    const int64_t n = total == 0 ? 1 : (total + step - 1) / step;
    *n_chunks = n;
    if (n > capacity_chunks) {
        set_error("plan_chunks: need room for %lld chunks, have %lld", (long long)n,
                  (long long)capacity_chunks);
        return OASR_ECAPACITY;
    }
    for (int64_t i = 0; i < n; ++i) {
        const int64_t s = i * step;
        const int64_t e = s + step < total ? s + step : total;
        out_bounds[2 * i] = s;
        out_bounds[2 * i + 1] = e;
    }
    return OASR_OK;
}

int oasr_shard_range(int64_t n_items, int32_t world, int32_t rank, int64_t* lo, int64_t* hi) {
    OASR_REQUIRE(lo && hi, "shard_range: null output");
    OASR_REQUIRE(world > 0 && rank >= 0 && rank < world && n_items >= 0,
                 "shard_range: bad world/rank/count");
    const int64_t q = n_items / world, r = n_items % world;
    *lo = rank * q + (rank < r ? rank : r);
    *hi = *lo + q + (rank < r ? 1 : 0);
    return OASR_OK;
}

int oasr_adpcm_ima_decode(const uint8_t* data, int64_t size, int32_t channels, int32_t block_align, int16_t* out,
                          int64_t capacity_per_channel, int64_t* written_per_channel) {
    OASR_REQUIRE(data && out && written_per_channel && size >= 0, "adpcm_ima_decode: null pointer");
    OASR_REQUIRE(channels >= 1 && channels <= 8 && block_align >= 4 * channels && block_align % (4 * channels) == 0,
                 "adpcm_ima_decode: a block is a 4-byte header and whole 4-byte groups per channel");
    static const int16_t steps[89] = {
        7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45, 50, 55, 60, 66, 73, 80, 88, 97, 107, 118,
        130, 143, 157, 173, 190, 209, 230, 253, 279, 307, 337, 371, 408, 449, 494, 544, 598, 658, 724, 796, 876, 963, 1060,
        1166, 1282, 1411, 1552, 1707, 1878, 2066, 2272, 2499, 2749, 3024, 3327, 3660, 4026, 4428, 4871, 5358, 5894, 6484,
        7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899, 15289, 16818, 18500, 20350, 22385, 24623, 27086, 29794, 32767};
    static const int8_t moves[8] = {-1, -1, -1, -1, 2, 4, 6, 8};
    const int64_t per_block = 1 + (int64_t)(block_align / channels - 4) * 2;   // samples per channel in a block
    int64_t done = 0;
    for (int64_t at = 0; at + block_align <= size; at += block_align) {
        if (done + per_block > capacity_per_channel) {
            set_error("adpcm_ima_decode: output holds %lld samples per channel, the data has more", (long long)capacity_per_channel);
            return OASR_ECAPACITY;
        }
        const uint8_t* blk = data + at;
        for (int c = 0; c < channels; ++c) {
            int pred = (int16_t)(blk[4 * c] | (blk[4 * c + 1] << 8));
            int idx = blk[4 * c + 2];
            if (idx > 88) idx = 88;
            int16_t* o = out + done * channels + c;
            *o = (int16_t)pred;
            int64_t n = 1;
            // after the headers: groups of 4 bytes (8 samples, low nibble first) per channel, channels in turn
            for (int g = 0; 4 * channels + (g * channels + c) * 4 + 4 <= block_align; ++g) {
                const uint8_t* q = blk + 4 * channels + (g * channels + c) * 4;
                for (int k = 0; k < 8; ++k) {
                    const int code = (q[k >> 1] >> ((k & 1) * 4)) & 15;
                    const int step = steps[idx];
                    int diff = step >> 3;
                    if (code & 4) diff += step;
                    if (code & 2) diff += step >> 1;
                    if (code & 1) diff += step >> 2;
                    pred += (code & 8) ? -diff : diff;
                    pred = pred > 32767 ? 32767 : pred < -32768 ? -32768 : pred;
                    idx += moves[code & 7];
                    idx = idx < 0 ? 0 : idx > 88 ? 88 : idx;
                    o[n * channels] = (int16_t)pred;
                    ++n;
                }
            }
        }
        done += per_block;
    }
    *written_per_channel = done;
    return OASR_OK;
}

int64_t oasr_num_frames(int64_t n) {
    return n < OASR_FRAME_LEN ? 0 : 1 + (n - OASR_FRAME_LEN) / OASR_FRAME_HOP;
}
int64_t oasr_num_stacked(int64_t n) { return oasr_num_frames(n) / 2; }

int oasr_logmel_tables(float* window, float* mel_weights) {
    OASR_REQUIRE(window && mel_weights, "logmel_tables: null output");
    build_window(window);
    const int nb = OASR_NFFT / 2 + 1;
    std::vector<double> dense((size_t)nb * OASR_NMEL);
    build_mel_dense(dense.data());
    for (size_t i = 0; i < dense.size(); ++i) mel_weights[i] = (float)dense[i];
    return OASR_OK;
}

int64_t oasr_logmel_table_words(void) { return (int64_t)(sizeof(LogmelBlob) / 4); }

int oasr_logmel_table_blob(uint32_t* blob, int64_t capacity_words) {
    OASR_REQUIRE(blob != nullptr, "logmel_table_blob: null output");
    if (capacity_words < oasr_logmel_table_words()) {
        set_error("logmel_table_blob: need %lld words", (long long)oasr_logmel_table_words());
        return OASR_ECAPACITY;
    }
    if (build_logmel_blob(blob) != 0) {
        set_error("logmel_table_blob: mel filter table overflow");
        return OASR_EINVAL;
    }
    return OASR_OK;
}

}  // extern "C"
