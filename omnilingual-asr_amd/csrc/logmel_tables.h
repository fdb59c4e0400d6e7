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
    // The same weights as MFMA operands (v_mfma_f32_16x16x32_bf16, weights = A operand): for each of the
    // kMelTasks (filter block of 16, K-step of 32 bins) pairs that hold a non-zero weight, the bf16 high
    // part and the bf16 remainder of the float weight, laid out per lane: lane l holds filter
    // 16 * block + (l & 15), K slots 8 * (l >> 4) .. + 7, two slots per dword (low half first); slot sigma
    // of K-step s is FFT bin 32 s + (sigma >> 1) + 16 * (sigma & 1).
    uint32_t mel_frag[12][2][64][4];
};

constexpr int kMelTasks = 12;
// (filter block, K-step) of each task; tasks of one block are consecutive
constexpr int kMelTaskBlock[kMelTasks] = {0, 1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 4};
constexpr int kMelTaskStep[kMelTasks] = {0, 0, 1, 1, 2, 2, 3, 4, 4, 5, 6, 7};

int build_logmel_blob(uint32_t* blob);

}  // namespace oasr
