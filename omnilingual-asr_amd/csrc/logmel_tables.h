// Constant tables of the log-mel kernel, built on the host in double precision and handed to the
// kernel as one device blob (the library keeps no device-side global state).
#pragma once
#include <stdint.h>

namespace oasr {

constexpr int kMelNnzCap = 640;  // non-zero triangle weights over all 80 filters (actual: ~510)

struct LogmelBlob {
    float window[400];       // Povey window
    float tw_re[256];        // cos(2 pi k / 512)
    float tw_im[256];        // -sin(2 pi k / 512)
    int32_t mel_start[80];   // first FFT bin with non-zero weight
    int32_t mel_count[80];   // number of consecutive bins
    int32_t mel_offset[80];  // offset of the filter's weights in mel_w
    float mel_w[kMelNnzCap];
};

int build_logmel_blob(uint32_t* blob);

}  // namespace oasr
