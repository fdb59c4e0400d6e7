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
