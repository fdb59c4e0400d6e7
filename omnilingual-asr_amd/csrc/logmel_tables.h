// Constant tables of the log-mel kernel, built on the host in double precision and handed to the
// kernel as one device blob (the library keeps no device-side global state).
#pragma once
#include <stdint.h>

namespace oasr {

constexpr int kMelWidth = 20;    // padded number of FFT bins per mel triangle (widest real one: 18)

struct LogmelBlob {
    float window[400];       // Povey window
    float tw_re[256];        // cos(2 pi k / 512)
    float tw_im[256];        // -sin(2 pi k / 512)
    int32_t mel_start[81];            // first FFT bin of each triangle; entry 80 is a dummy (bin 0)
    int32_t mel_width_max;            // widest triangle, in bins (checked against kMelWidth)
    int32_t pad_[2];
    float mel_w[81 * kMelWidth];      // [filter][kMelWidth] weights, zero padded; row 80 is all zeros
};

int build_logmel_blob(uint32_t* blob);

}  // namespace oasr
