// MPEG audio Layer III ("MP3") decoder (host code): the lossy format of both demo recordings the reference ships
// and the first of the upload types it accepts (ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:106-118,
// ref:workflows/wav2elan_web/app.py:76-77; the reference forwards the file bytes to its remote model, here they are
// decoded locally).  Implements the published format (ISO/IEC 11172-3 Layer III, the lower sampling frequencies of
// ISO/IEC 13818-3, and the "2.5" extension to 8 / 11.025 / 12 kHz): frame sync with ID3v2 skipping, header CRC-16,
// side information, the bit reservoir, scale factors (with scfsi; the second standard's partition tables),
// Huffman-coded spectra (pair books with linbits, both quadruple books), requantisation, mid/side and intensity
// stereo, short-block reordering, alias reduction, the 36- and 12-point inverse MDCTs with the four window shapes,
// frequency inversion and the 32-band polyphase synthesis filter.  An Info / Xing frame is not audio and is skipped;
// the encoder delay and padding it states (plus the decoder's own 529 samples) are trimmed, as ffmpeg — the tool the
// reference slices audio with (ref gemini_pipeline.py:273-285) — does.  Output: interleaved float32, nominally [-1, 1).
// Free-format streams and Layers I / II are not decoded.
//
// Long files are decoded by several host threads, each on a range of frames: a thread starts early enough (as many
// frames as hold the largest bit reservoir, and two more: `warm_start`) and discards what those produce, so that
// every piece of state a frame inherits (bit reservoir <= 511 bytes, one granule of overlap, 16 time slots of
// synthesis history) equals that of an in-order decode and the samples are identical for any thread count.
// OASR_MP3_THREADS in the environment overrides the count.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <thread>
#include <vector>

#include "oasr_common.h"

namespace oasr {
namespace mp3 {

#include "mp3_tables.inc"

// ---- constant tables ---------------------------------------------------------------------------------------------
struct Book {
    const int16_t (*nodes)[2];
    int linbits;
    int id;   // which of the 16 distinct trees (index of its first-level table)
};
static const Book kBooks[32] = {
    {nullptr, 0, 0},   {kMp3Book1, 0, 1},  {kMp3Book2, 0, 2},  {kMp3Book3, 0, 3},   {nullptr, 0, 0},    {kMp3Book5, 0, 4},  {kMp3Book6, 0, 5},
    {kMp3Book7, 0, 6}, {kMp3Book8, 0, 7},  {kMp3Book9, 0, 8},  {kMp3Book10, 0, 9},  {kMp3Book11, 0, 10}, {kMp3Book12, 0, 11}, {kMp3Book13, 0, 12},
    {nullptr, 0, 0},   {kMp3Book15, 0, 13}, {kMp3Book16, 1, 14}, {kMp3Book16, 2, 14},  {kMp3Book16, 3, 14}, {kMp3Book16, 4, 14}, {kMp3Book16, 6, 14},
    {kMp3Book16, 8, 14}, {kMp3Book16, 10, 14}, {kMp3Book16, 13, 14}, {kMp3Book24, 4, 15}, {kMp3Book24, 5, 15}, {kMp3Book24, 6, 15}, {kMp3Book24, 7, 15},
    {kMp3Book24, 8, 15}, {kMp3Book24, 9, 15}, {kMp3Book24, 11, 15}, {kMp3Book24, 13, 15}};

// scale factor band edges; row = sampling frequency index: 44.1, 48, 32 | 22.05, 24, 16 | 11.025, 12, 8 kHz
static const int16_t kLongEdges[9][23] = {
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 52, 62, 74, 90, 110, 134, 162, 196, 238, 288, 342, 418, 576},
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 42, 50, 60, 72, 88, 106, 128, 156, 190, 230, 276, 330, 384, 576},
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 54, 66, 82, 102, 126, 156, 194, 240, 296, 364, 448, 550, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 114, 136, 162, 194, 232, 278, 332, 394, 464, 540, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
    {0, 12, 24, 36, 48, 60, 72, 88, 108, 132, 160, 192, 232, 280, 336, 400, 476, 566, 568, 570, 572, 574, 576}};
static const int16_t kShortEdges[9][14] = {
    {0, 4, 8, 12, 16, 22, 30, 40, 52, 66, 84, 106, 136, 192}, {0, 4, 8, 12, 16, 22, 28, 38, 50, 64, 80, 100, 126, 192},
    {0, 4, 8, 12, 16, 22, 30, 42, 58, 78, 104, 138, 180, 192}, {0, 4, 8, 12, 18, 24, 32, 42, 56, 74, 100, 132, 174, 192},
    {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 136, 180, 192}, {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192},
    {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192}, {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192},
    {0, 8, 16, 24, 36, 52, 72, 96, 124, 160, 162, 164, 166, 192}};
static const uint8_t kPretab[22] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2, 0};
static const uint8_t kSlen1[16] = {0, 0, 0, 0, 3, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4};
static const uint8_t kSlen2[16] = {0, 1, 2, 3, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 3};
// second standard: scale factors per partition, [rule][long | short | mixed][partition]
static const uint8_t kLsfPartitions[6][3][4] = {
    {{6, 5, 5, 5}, {9, 9, 9, 9}, {6, 9, 9, 9}},   {{6, 5, 7, 3}, {9, 9, 12, 6}, {6, 9, 12, 6}},
    {{11, 10, 0, 0}, {18, 18, 0, 0}, {15, 18, 0, 0}}, {{7, 7, 7, 0}, {12, 12, 12, 0}, {6, 15, 12, 0}},
    {{6, 6, 6, 3}, {12, 9, 9, 6}, {6, 12, 9, 6}}, {{8, 8, 5, 0}, {15, 12, 9, 0}, {6, 18, 9, 0}}};
// synthesis window: the prototype low-pass h[n] * 65536 for n = 0 .. 256 (h[512 - n] = h[n]); the standard's D[n] is
// h[n] with the sign of every second group of 64 flipped
static const int32_t kWindowBase[257] = {
    0, -1, -1, -1, -1, -1, -1, -2, -2, -2, -2, -3, -3, -4, -4, -5, -5, -6, -7, -7, -8, -9, -10, -11, -13, -14, -16, -17,
    -19, -21, -24, -26, -29, -31, -35, -38, -41, -45, -49, -53, -58, -63, -68, -73, -79, -85, -91, -97, -104, -111,
    -117, -125, -132, -139, -147, -154, -161, -169, -176, -183, -190, -196, -202, -208, -213, -218, -222, -225, -227,
    -228, -228, -227, -224, -221, -215, -208, -200, -189, -177, -163, -146, -127, -106, -83, -57, -29, 2, 36, 72, 111,
    153, 197, 244, 294, 347, 401, 459, 519, 581, 645, 711, 779, 848, 919, 991, 1064, 1137, 1210, 1283, 1356, 1428,
    1498, 1567, 1634, 1698, 1759, 1817, 1870, 1919, 1962, 2001, 2032, 2057, 2075, 2085, 2087, 2080, 2063, 2037, 2000,
    1952, 1893, 1822, 1739, 1644, 1535, 1414, 1280, 1131, 970, 794, 605, 402, 185, -45, -288, -545, -814, -1095, -1388,
    -1692, -2006, -2330, -2663, -3004, -3351, -3705, -4063, -4425, -4788, -5153, -5517, -5879, -6237, -6589, -6935,
    -7271, -7597, -7910, -8209, -8491, -8755, -8998, -9219, -9416, -9585, -9727, -9838, -9916, -9959, -9966, -9935,
    -9863, -9750, -9592, -9389, -9139, -8840, -8492, -8092, -7640, -7134, -6574, -5959, -5288, -4561, -3776, -2935,
    -2037, -1082, -70, 998, 2122, 3300, 4533, 5818, 7154, 8540, 9975, 11455, 12980, 14548, 16155, 17799, 19478, 21189,
    22929, 24694, 26482, 28289, 30112, 31947, 33791, 35640, 37489, 39336, 41176, 43006, 44821, 46617, 48390, 50137,
    51853, 53534, 55178, 56778, 58333, 59838, 61289, 62684, 64019, 65290, 66494, 67629, 68692, 69679, 70590, 71420,
    72169, 72835, 73415, 73908, 74313, 74630, 74856, 74992, 75038};

struct Tables {
    float pow43[8207];        // n^(4/3)
    // [coefficient][output], outputs padded to whole vectors: columns 0..8 are outputs 0..8, columns 9..17 outputs
    // 18..26 (the other 18 follow by symmetry).  Coefficient-major so that the inner loop runs over outputs: a sum
    // over coefficients per output is a reduction the compiler must not reorder, a loop over outputs it vectorises.
    float imdct_cos[18][20];
    float long_window[4][36];
    float imdct_short[6][12];      // [coefficient][output], window folded in
    float lee[32];                 // fast DCT: 1 / (2 cos((2k + 1) pi / 2N)) at [N/2 + k]
    float gain[512];               // 2^(e / 4) at [e + 400], e = -400 .. 111
    // first level of every code tree: the next 8 bits -> (leaf << 4 | code length) for codes of at most 8 bits,
    // else -(node reached after 8 bits) - 1; [16] is the quadruple book
    int16_t first[17][256];
    float window[512];        // the standard's D[n]
    float cs[8], ca[8];
    float pan[7][2];          // first-standard intensity ratios (left, right)
    Tables() {
        const double pi = 3.14159265358979323846;
        for (int i = 0; i < 8207; ++i) pow43[i] = (float)pow((double)i, 4.0 / 3.0);
        for (int bt = 0; bt < 4; ++bt)
            for (int i = 0; i < 36; ++i) {
                double w = sin(pi / 36 * (i + 0.5));
                if (bt == 1) w = i < 18 ? w : i < 24 ? 1.0 : i < 30 ? sin(pi / 12 * (i - 18 + 0.5)) : 0.0;
                if (bt == 3) w = i < 6 ? 0.0 : i < 12 ? sin(pi / 12 * (i - 6 + 0.5)) : i < 18 ? 1.0 : w;
                long_window[bt][i] = (float)w;
            }
        memset(imdct_cos, 0, sizeof(imdct_cos));
        for (int r = 0; r < 18; ++r)
            for (int k = 0; k < 18; ++k) {
                const int i = r < 9 ? r : r + 9;
                imdct_cos[k][r] = (float)cos(pi / 72 * (2 * i + 19) * (2 * k + 1));
            }
        for (int n = 2; n <= 32; n *= 2)
            for (int k = 0; k < n / 2; ++k) lee[n / 2 + k] = (float)(1.0 / (2.0 * cos((2 * k + 1) * pi / (2 * n))));
        for (int e = -400; e < 112; ++e) gain[e + 400] = (float)exp2(0.25 * e);
        memset(first, 0, sizeof(first));
        for (int t = 0; t <= 32; ++t) {
            const int16_t(*nodes)[2] = t == 32 ? kMp3Book32 : kBooks[t].nodes;
            if (!nodes) continue;
            int16_t* f = first[t == 32 ? 16 : kBooks[t].id];
            for (int w = 0; w < 256; ++w) {
                int at = 0, used = 0;
                while (at >= 0 && used < 8) at = nodes[at][(w >> (7 - used)) & 1], ++used;
                f[w] = at < 0 ? (int16_t)(((-at - 1) << 4) | used) : (int16_t)(-at - 1);
            }
        }
        for (int i = 0; i < 12; ++i)
            for (int k = 0; k < 6; ++k)
                imdct_short[k][i] = (float)(sin(pi / 12 * (i + 0.5)) * cos(pi / 24 * (2 * i + 7) * (2 * k + 1)));
        for (int n = 0; n < 512; ++n) {
            const double h = kWindowBase[n <= 256 ? n : 512 - n] / 65536.0;
            window[n] = (float)(((n >> 6) & 1) ? -h : h);
        }
        static const double c[8] = {-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037};
        for (int i = 0; i < 8; ++i) {
            cs[i] = (float)(1.0 / sqrt(1.0 + c[i] * c[i]));
            ca[i] = (float)(c[i] / sqrt(1.0 + c[i] * c[i]));
        }
        for (int p = 0; p < 7; ++p) {
            if (p == 6) {
                pan[p][0] = 1.0f, pan[p][1] = 0.0f;
            } else {
                const double r = tan(p * pi / 12);
                pan[p][0] = (float)(r / (1 + r)), pan[p][1] = (float)(1 / (1 + r));
            }
        }
    }
};
static const Tables& tables() {
    static const Tables t;
    return t;
}

// X[m] = sum_k x[k] cos((2k + 1) m pi / 2N), in place (Lee's recursion: two half-size transforms of the sums and of
// the scaled differences), for kLanes independent inputs at once: the 18 time slots of a granule side by side, so that
// every step is a loop over lanes the compiler turns into vector instructions.  The scale factors reach 10 and multiply
// along the recursion: relative error about 1e-6, i.e. 120 dB below the signal.
static const int kLanes = 20;   // 18 slots, padded to whole vectors
template <int N>
static inline void dct2(float (*x)[kLanes], const float* lee) {
    float a[N / 2][kLanes], b[N / 2][kLanes];
    for (int k = 0; k < N / 2; ++k) {
        const float f = lee[N / 2 + k];
        for (int l = 0; l < kLanes; ++l) {
            a[k][l] = x[k][l] + x[N - 1 - k][l];
            b[k][l] = (x[k][l] - x[N - 1 - k][l]) * f;
        }
    }
    dct2<N / 2>(a, lee);
    dct2<N / 2>(b, lee);
    for (int m = 0; m < N / 2 - 1; ++m)
        for (int l = 0; l < kLanes; ++l) {
            x[2 * m][l] = a[m][l];
            x[2 * m + 1][l] = b[m][l] + b[m + 1][l];
        }
    for (int l = 0; l < kLanes; ++l) {
        x[N - 2][l] = a[N / 2 - 1][l];
        x[N - 1][l] = b[N / 2 - 1][l];
    }
}
template <>
inline void dct2<1>(float (*)[kLanes], const float*) {}

// ---- frame header -----------------------------------------------------------------------------------------------
struct Header {
    bool lsf;        // second standard or its 2.5 extension: one granule per frame
    int sr_idx;      // row of the band tables
    int rate, kbps, channels, mode, mode_ext, frame_bytes, side_bytes;
    bool crc;
    int granules() const { return lsf ? 1 : 2; }
    int samples() const { return lsf ? 576 : 1152; }
    bool same_stream(const Header& o) const { return sr_idx == o.sr_idx && lsf == o.lsf; }
};

static bool parse_header(const uint8_t* p, Header& h) {
    if (p[0] != 0xFF || (p[1] & 0xE0) != 0xE0) return false;
    const int ver = (p[1] >> 3) & 3, layer = (p[1] >> 1) & 3;
    if (ver == 1 || layer != 1) return false;
    const int bri = p[2] >> 4, sri = (p[2] >> 2) & 3;
    if (bri == 0 || bri == 15 || sri == 3) return false;
    static const int r1[3] = {44100, 48000, 32000};
    static const int b1[15] = {0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320};
    static const int b2[15] = {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160};
    h.lsf = ver != 3;
    h.sr_idx = (ver == 3 ? 0 : ver == 2 ? 3 : 6) + sri;
    h.rate = r1[sri] >> (ver == 3 ? 0 : ver == 2 ? 1 : 2);
    h.kbps = h.lsf ? b2[bri] : b1[bri];
    h.crc = !(p[1] & 1);
    h.mode = p[3] >> 6;
    h.mode_ext = (p[3] >> 4) & 3;
    h.channels = h.mode == 3 ? 1 : 2;
    h.frame_bytes = (h.lsf ? 72000 : 144000) * h.kbps / h.rate + ((p[2] >> 1) & 1);
    h.side_bytes = h.lsf ? (h.channels == 1 ? 9 : 17) : (h.channels == 1 ? 17 : 32);
    return h.frame_bytes >= 4 + (h.crc ? 2 : 0) + h.side_bytes;
}

struct FrameRef {
    int64_t at;
    Header h;
};

// All frames of the stream, in order.  A frame counts when its header parses, it lies inside the data, it belongs to
// the same stream as the first one, and either it starts where an accepted frame ended or the position after it holds
// another such header (or the data ends there); anything else is stepped over byte by byte.
static void scan_frames(const uint8_t* d, int64_t size, std::vector<FrameRef>& out, int64_t& skipped) {
    int64_t pos = 0;
    skipped = 0;
    while (pos + 10 <= size && d[pos] == 'I' && d[pos + 1] == 'D' && d[pos + 2] == '3') {
        const int64_t n = ((int64_t)(d[pos + 6] & 0x7F) << 21) | ((d[pos + 7] & 0x7F) << 14) | ((d[pos + 8] & 0x7F) << 7) |
                          (d[pos + 9] & 0x7F);
        pos += 10 + n + ((d[pos + 5] & 0x10) ? 10 : 0);
    }
    if (pos > size) pos = size;
    Header h, hn;
    while (pos + 4 <= size) {
        bool ok = parse_header(d + pos, h) && pos + h.frame_bytes <= size;
        if (ok && !out.empty() && !h.same_stream(out.front().h)) ok = false;
        const bool chained = !out.empty() && pos == out.back().at + out.back().h.frame_bytes;
        if (ok && !chained && pos + h.frame_bytes + 4 <= size)
            ok = parse_header(d + pos + h.frame_bytes, hn) && hn.same_stream(h);
        if (ok) {
            out.push_back({pos, h});
            pos += h.frame_bytes;
        } else {
            ++pos;
            ++skipped;
        }
    }
}

// Info / Xing frame: where the identifier sits, -1 when the frame is ordinary audio.
static int info_tag_offset(const uint8_t* f, const Header& h) {
    const int at = 4 + (h.crc ? 2 : 0) + h.side_bytes;   // where the main data of an audio frame would begin
    if (at + 8 > h.frame_bytes) return -1;
    if (memcmp(f + at, "Info", 4) == 0 || memcmp(f + at, "Xing", 4) == 0) return at;
    return -1;
}

struct StreamPlan {
    std::vector<FrameRef> frames;   // audio frames only
    int64_t skipped = 0;
    int64_t lead = 0;               // decoded samples per channel dropped at the start
    int64_t keep = 0;               // samples per channel delivered
};

static int plan_stream(const uint8_t* d, int64_t size, StreamPlan& sp) {
    scan_frames(d, size, sp.frames, sp.skipped);
    if (sp.frames.empty()) {
        set_error("mp3: no Layer III frame found");
        return OASR_EINVAL;
    }
    int64_t delay = -1, padding = 0;
    const FrameRef f0 = sp.frames.front();
    const int tag = info_tag_offset(d + f0.at, f0.h);
    if (tag >= 0) {
        const uint8_t* t = d + f0.at + tag;
        const int avail = f0.h.frame_bytes - tag;
        const uint32_t flags = ((uint32_t)t[4] << 24) | (t[5] << 16) | (t[6] << 8) | t[7];
        const int ext = 8 + ((flags & 1) ? 4 : 0) + ((flags & 2) ? 4 : 0) + ((flags & 4) ? 100 : 0) + ((flags & 8) ? 4 : 0);
        if (ext + 24 <= avail && (memcmp(t + ext, "LAME", 4) == 0 || memcmp(t + ext, "Lavc", 4) == 0 ||
                                  memcmp(t + ext, "Lavf", 4) == 0)) {
            delay = (t[ext + 21] << 4) | (t[ext + 22] >> 4);
            padding = ((t[ext + 22] & 15) << 8) | t[ext + 23];
        }
        sp.frames.erase(sp.frames.begin());
        if (sp.frames.empty()) {
            set_error("mp3: the stream holds no audio frame");
            return OASR_EINVAL;
        }
    }
    const int64_t total = (int64_t)sp.frames.size() * sp.frames.front().h.samples();
    if (delay >= 0) {   // 529 = the decoder's own delay (528) + 1, the convention of the encoder that writes the tag
        sp.lead = delay + 529;
        sp.keep = total - delay - padding;
        if (sp.lead > total) sp.lead = total;
        if (sp.keep > total - sp.lead) sp.keep = total - sp.lead;
        if (sp.keep < 0) sp.keep = 0;
    } else {
        sp.lead = 0;
        sp.keep = total;
    }
    return OASR_OK;
}

// ---- bit reader; the buffer holds 8 readable bytes past `cap` bits, and reads beyond `cap` return zeros --------------
struct Bits {
    const uint8_t* p;
    int64_t pos;   // in bits
    int64_t cap;
    uint32_t peek32() const {
        if (pos > cap) return 0;
        uint64_t v;
        memcpy(&v, p + (pos >> 3), 8);
        return (uint32_t)((__builtin_bswap64(v) << (pos & 7)) >> 32);
    }
    uint32_t get(int n) {   // 0 <= n <= 24
        if (n == 0) return 0;
        const uint32_t v = peek32() >> (32 - n);
        pos += n;
        return v;
    }
    int walk(const int16_t (*nodes)[2], const int16_t* first) {
        uint32_t w = peek32();
        const int e = first[w >> 24];
        if (e >= 0) {   // a code of at most 8 bits
            pos += e & 15;
            return e >> 4;
        }
        int at = -e - 1, used = 8;
        w <<= 8;
        do {
            at = nodes[at][w >> 31];
            w <<= 1;
            ++used;
        } while (at >= 0);
        pos += used;
        return -at - 1;
    }
};

struct Granule {
    int part23, big_values, global_gain, scf_compress, block_type, table[3], sub_gain[3], region0, region1;
    bool switching, mixed, low_normal, preflag, scf_scale, count1_b;   // mixed: short granule with long low bands
    // flat band list in transmission order: long bands, then short bands (each band once per window)
    int n_long, n_bands;
    int16_t width[40];
    bool supported;
};

static void build_bands(Granule& g, const Header& h) {
    const int16_t* le = kLongEdges[h.sr_idx];
    const int16_t* se = kShortEdges[h.sr_idx];
    int n = 0;
    g.supported = true;
    if (g.block_type != 2) {
        for (int b = 0; b < 22; ++b) g.width[n++] = le[b + 1] - le[b];
        g.n_long = 22;
    } else {
        int first_short = 0;
        g.n_long = 0;
        if (g.mixed) {   // long bands over the lowest 36 lines, short bands from the one that starts at line 12 of a window
            while (le[g.n_long] < 36) g.width[n++] = le[g.n_long + 1] - le[g.n_long], ++g.n_long;
            while (first_short < 13 && se[first_short] * 3 < 36) ++first_short;
            if (le[g.n_long] != 36 || se[first_short] * 3 != 36) g.supported = false;
        }
        for (int b = first_short; b < 13; ++b)
            for (int w = 0; w < 3; ++w) g.width[n++] = se[b + 1] - se[b];
    }
    g.n_bands = n;
}

struct SideInfo {
    int main_data_begin;
    int scfsi[2][4];
    Granule gr[2][2];   // [granule][channel]
};

static bool parse_side_info(const uint8_t* p, const Header& h, SideInfo& si) {
    uint8_t buf[48] = {0};
    memcpy(buf, p, h.side_bytes);
    Bits b{buf, 0, 40 * 8};
    if (h.lsf) {
        si.main_data_begin = b.get(8);
        b.get(h.channels == 1 ? 1 : 2);
    } else {
        si.main_data_begin = b.get(9);
        b.get(h.channels == 1 ? 5 : 3);
        for (int ch = 0; ch < h.channels; ++ch)
            for (int k = 0; k < 4; ++k) si.scfsi[ch][k] = b.get(1);
    }
    for (int gi = 0; gi < h.granules(); ++gi)
        for (int ch = 0; ch < h.channels; ++ch) {
            Granule& g = si.gr[gi][ch];
            g.part23 = b.get(12);
            g.big_values = b.get(9);
            g.global_gain = b.get(8);
            g.scf_compress = b.get(h.lsf ? 9 : 4);
            g.block_type = 0;
            g.mixed = g.low_normal = false;
            g.sub_gain[0] = g.sub_gain[1] = g.sub_gain[2] = 0;
            g.switching = b.get(1);
            if (g.switching) {
                g.block_type = b.get(2);
                g.mixed = b.get(1);
                g.table[0] = b.get(5), g.table[1] = b.get(5), g.table[2] = 0;
                for (int w = 0; w < 3; ++w) g.sub_gain[w] = b.get(3);
                if (g.block_type == 0) return false;   // window switching must name a window
                g.region0 = (g.block_type == 2 && !g.mixed) ? 8 : 7;
                g.region1 = 20 - g.region0;
            } else {
                for (int r = 0; r < 3; ++r) g.table[r] = b.get(5);
                g.region0 = b.get(4);
                g.region1 = b.get(3);
            }
            g.preflag = h.lsf ? false : (bool)b.get(1);
            g.scf_scale = b.get(1);
            g.count1_b = b.get(1);
            if (g.big_values > 288) return false;
            // the flag of a window-switching granule: its two lowest subbands keep the normal window whatever the block
            // type; only a short granule also changes its band layout
            g.low_normal = g.mixed;
            if (g.block_type != 2) g.mixed = false;
            build_bands(g, h);
        }
    return true;
}

// per-channel state carried from frame to frame
struct Channel {
    float overlap[32][18];
    float v[2048];    // synthesis history, every vector stored twice (at o and o + 1024) so that 1024 in a row are readable
    int voff;
    int live;         // subbands whose overlap may be non-zero
    uint8_t scf_prev[40];   // granule 0's scale factors (for scfsi)
};

struct Decoder {
    const Tables& T = tables();
    Channel ch[2];
    std::vector<uint8_t> main;   // main data seen so far (tail kept), then 8 zero bytes while in use
    int64_t diag[5] = {0, 0, 0, 0, 0};
    bool counting = true;
    void note(int k, int64_t n = 1) {
        if (counting) diag[k] += n;
    }
    Decoder() { reset(); }
    void reset() {
        memset(ch, 0, sizeof(ch));
        main.clear();
    }

    // ---- scale factors -> flat list `scf` (one per band of the granule's band list); returns false on overrun ----
    void read_scalefactors(Bits& b, const Header& h, const SideInfo& si, int gi, int c, Granule& g, uint8_t* scf) {
        memset(scf, 0, 40);
        if (!h.lsf) {
            const int s1 = kSlen1[g.scf_compress], s2 = kSlen2[g.scf_compress];
            if (g.block_type == 2) {
                const int n1 = g.mixed ? g.n_long + 9 : 18;   // long bands and short bands 3..5 | short bands 0..5
                int i = 0;
                for (; i < n1; ++i) scf[i] = b.get(s1);
                for (int k = 0; k < 18; ++k, ++i) scf[i] = b.get(s2);
            } else {
                static const int edge[5] = {0, 6, 11, 16, 21};
                for (int k = 0; k < 4; ++k)
                    for (int i = edge[k]; i < edge[k + 1]; ++i)
                        scf[i] = (gi == 1 && si.scfsi[c][k]) ? ch[c].scf_prev[i] : b.get(k < 2 ? s1 : s2);
            }
            if (gi == 0) memcpy(ch[c].scf_prev, scf, 40);
            return;
        }
        int sfc = g.scf_compress, slen[4], rule;
        const bool right_intensity = h.mode == 1 && (h.mode_ext & 1) && c == 1;
        g.preflag = false;
        if (!right_intensity) {
            if (sfc < 400) {
                slen[0] = (sfc >> 4) / 5, slen[1] = (sfc >> 4) % 5, slen[2] = (sfc & 15) >> 2, slen[3] = sfc & 3, rule = 0;
            } else if (sfc < 500) {
                sfc -= 400;
                slen[0] = (sfc >> 2) / 5, slen[1] = (sfc >> 2) % 5, slen[2] = sfc & 3, slen[3] = 0, rule = 1;
            } else {
                sfc -= 500;
                slen[0] = sfc / 3, slen[1] = sfc % 3, slen[2] = slen[3] = 0, rule = 2;
                g.preflag = true;
            }
        } else {
            sfc >>= 1;
            if (sfc < 180) {
                slen[0] = sfc / 36, slen[1] = (sfc % 36) / 6, slen[2] = (sfc % 36) % 6, slen[3] = 0, rule = 3;
            } else if (sfc < 244) {
                sfc -= 180;
                slen[0] = (sfc & 63) >> 4, slen[1] = (sfc & 15) >> 2, slen[2] = sfc & 3, slen[3] = 0, rule = 4;
            } else {
                sfc -= 244;
                slen[0] = sfc / 3, slen[1] = sfc % 3, slen[2] = slen[3] = 0, rule = 5;
            }
        }
        const uint8_t* parts = kLsfPartitions[rule][g.block_type != 2 ? 0 : g.mixed ? 2 : 1];
        int i = 0;
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < parts[k] && i < 39; ++j, ++i) {
                const int v = b.get(slen[k]);
                // an intensity position equal to the largest value of its field means "not intensity coded"
                scf[i] = (right_intensity && v == (1 << slen[k]) - 1) ? 255 : v;
            }
    }

    // ---- Huffman-coded spectrum -> is[576]; returns the index past the last decoded line ----
    int read_spectrum(Bits& b, int64_t end, const Header& h, const Granule& g, int* is) {
        memset(is, 0, 576 * sizeof(int));
        int edge[3];
        if (g.switching) {   // two regions
            edge[0] = g.block_type == 2 ? (h.sr_idx == 8 && !g.mixed ? 72 : 36) : kLongEdges[h.sr_idx][8];
            edge[1] = 576;
        } else {
            edge[0] = kLongEdges[h.sr_idx][g.region0 + 1];
            int r = g.region0 + g.region1 + 2;
            edge[1] = kLongEdges[h.sr_idx][r > 22 ? 22 : r];
        }
        edge[2] = 576;
        const int big_end = g.big_values * 2;
        int i = 0;
        bool over = false;
        for (int r = 0; r < 3 && i < big_end && !over; ++r) {
            const int stop = edge[r] < big_end ? edge[r] : big_end;
            const Book& bk = kBooks[g.table[r]];
            const int16_t* first = T.first[bk.id];
            if (!bk.nodes) {
                if (i < stop) i = stop;
                continue;
            }
            while (i < stop) {
                if (b.pos >= end) {
                    over = true;
                    break;
                }
                int x, y;
                if (!bk.linbits) {   // code (at most 19 bits) and the two sign bits out of one 32-bit look
                    uint32_t w = b.peek32();
                    const int e = first[w >> 24];
                    int used, v;
                    if (e >= 0) {
                        used = e & 15, v = e >> 4;
                    } else {
                        int at = -e - 1;
                        used = 8;
                        do at = bk.nodes[at][(w << used) >> 31], ++used;
                        while (at >= 0);
                        v = -at - 1;
                    }
                    x = v >> 4, y = v & 15;
                    if (x) {
                        if ((w << used) >> 31) x = -x;
                        ++used;
                    }
                    if (y) {
                        if ((w << used) >> 31) y = -y;
                        ++used;
                    }
                    b.pos += used;
                } else {
                    const int v = b.walk(bk.nodes, first);
                    x = v >> 4, y = v & 15;
                    if (x == 15) x += b.get(bk.linbits);
                    if (x && b.get(1)) x = -x;
                    if (y == 15) y += b.get(bk.linbits);
                    if (y && b.get(1)) y = -y;
                }
                is[i] = x, is[i + 1] = y;
                i += 2;
            }
        }
        if (b.pos > end) over = true;
        while (!over && i <= 572 && b.pos < end) {
            const uint32_t w = b.peek32();   // code (at most 6 bits) and up to four sign bits
            int used, v;
            if (g.count1_b) {
                v = 15 - (int)(w >> 28), used = 4;
            } else {
                const int e = T.first[16][w >> 24];
                v = e >> 4, used = e & 15;
            }
            int q[4];
            for (int k = 0; k < 4; ++k) {
                q[k] = (v >> (3 - k)) & 1;
                if (q[k]) {
                    if ((w << used) >> 31) q[k] = -1;
                    ++used;
                }
            }
            b.pos += used;
            if (b.pos > end) {   // the quadruple ran past the granule's data: it is stuffing, not spectrum
                over = true;
                break;
            }
            for (int k = 0; k < 4; ++k) is[i + k] = q[k];
            i += 4;
        }
        if (over) {   // the pair that crossed the end of the granule's data is not spectrum either
            note(2);
            if (i <= big_end && i >= 2 && b.pos > end) is[i - 2] = is[i - 1] = 0, i -= 2;
        } else if (b.pos < end) {
            note(4, end - b.pos);
        }
        b.pos = end;
        return i;
    }

    void requantize(const Granule& g, const uint8_t* scf, const int* is, int n_used, float* xr) {
        int o = 0;
        const int mult = g.scf_scale ? 4 : 2;
        for (int bnd = 0; bnd < g.n_bands; ++bnd) {
            const int w = g.width[bnd];
            if (o >= n_used) {
                memset(xr + o, 0, (576 - o) * sizeof(float));
                return;
            }
            int e = g.global_gain - 210;
            const int s = scf[bnd] == 255 ? 0 : scf[bnd];
            if (bnd < g.n_long)
                e -= mult * (s + ((g.preflag && bnd < 22) ? kPretab[bnd] : 0));
            else
                e -= 8 * g.sub_gain[(bnd - g.n_long) % 3] + mult * s;
            const float gain = T.gain[(e < -400 ? -400 : e) + 400];
            for (int j = 0; j < w; ++j) {
                const int v = is[o + j];
                const int a = v < 0 ? -v : v;
                const float m = T.pow43[a < 8207 ? a : 8206] * gain;
                xr[o + j] = v < 0 ? -m : m;
            }
            o += w;
        }
    }

    // ---- joint stereo on the flat band order of the right channel (both channels share one layout) ----
    void stereo(float* l, float* r, const Header& h, const Granule& g, uint8_t* pos) {
        const bool ms = h.mode_ext & 2, intensity = h.mode_ext & 1;
        const float rs = 0.70710678118654752f;
        if (!intensity) {
            if (ms)
                for (int i = 0; i < 576; ++i) {
                    const float a = l[i], b = r[i];
                    l[i] = (a + b) * rs, r[i] = (a - b) * rs;
                }
            return;
        }
        int top[3] = {-1, -1, -1};
        int o = 0;
        for (int bnd = 0; bnd < g.n_bands; ++bnd) {
            for (int j = 0; j < g.width[bnd]; ++j)
                if (r[o + j] != 0.0f) {
                    top[bnd % 3] = bnd;
                    break;
                }
            o += g.width[bnd];
        }
        const int blocks = g.n_bands > g.n_long ? 3 : 1;
        if (g.n_long) {
            const int m = top[0] > top[1] ? (top[0] > top[2] ? top[0] : top[2]) : (top[1] > top[2] ? top[1] : top[2]);
            top[0] = top[1] = top[2] = m;
        }
        // the last band (one per window) carries no position of its own: it follows the band below it
        for (int i = 0; i < blocks; ++i) {
            const int last = g.n_bands - blocks + i, below = last - blocks;
            pos[last] = top[last % 3] >= below ? (h.lsf ? 0 : 3) : pos[below];
        }
        const unsigned limit = h.lsf ? 64 : 7;
        const int scale_sel = g.scf_compress & 1;
        o = 0;
        for (int bnd = 0; bnd < g.n_bands; ++bnd) {
            const int w = g.width[bnd];
            const unsigned p = pos[bnd];
            if (bnd > top[bnd % 3] && p < limit) {
                float kl, kr;
                if (!h.lsf) {
                    kl = T.pan[p][0], kr = T.pan[p][1];
                } else {
                    const float k = (float)exp2(-0.25 * (double)(((p + 1) >> 1) << scale_sel));
                    kl = (p & 1) ? k : 1.0f;
                    kr = (p & 1) ? 1.0f : k;
                }
                for (int j = 0; j < w; ++j) {
                    const float a = l[o + j];
                    l[o + j] = a * kl, r[o + j] = a * kr;
                }
            } else if (ms) {
                for (int j = 0; j < w; ++j) {
                    const float a = l[o + j], b = r[o + j];
                    l[o + j] = (a + b) * rs, r[o + j] = (a - b) * rs;
                }
            }
            o += w;
        }
    }

    // ---- one channel of one granule: reorder, alias reduction, inverse MDCT, synthesis -> pcm[18 * 32] ----
    void hybrid_and_synthesis(const Granule& g, float* xr, Channel& c, float* pcm, int stride) {
        if (g.block_type == 2) {   // short bands: [window][line] -> [line][window]
            float tmp[576];
            int o = 0;
            for (int bnd = 0; bnd < g.n_long; ++bnd) o += g.width[bnd];
            for (int bnd = g.n_long; bnd < g.n_bands; bnd += 3) {
                const int w = g.width[bnd];
                for (int j = 0; j < w; ++j)
                    for (int k = 0; k < 3; ++k) tmp[o + 3 * j + k] = xr[o + k * w + j];
                o += 3 * w;
            }
            int lo = 0;
            for (int bnd = 0; bnd < g.n_long; ++bnd) lo += g.width[bnd];
            memcpy(xr + lo, tmp + lo, (576 - lo) * sizeof(float));
        }
        const int long_subbands = g.block_type != 2 ? 32 : g.mixed ? 2 : 0;
        for (int sb = 1; sb < long_subbands; ++sb)
            for (int i = 0; i < 8; ++i) {
                const float lo = xr[18 * sb - 1 - i], hi = xr[18 * sb + i];
                xr[18 * sb - 1 - i] = lo * T.cs[i] - hi * T.ca[i];
                xr[18 * sb + i] = hi * T.cs[i] + lo * T.ca[i];
            }
        int top = 576;   // lines at and above `top` are zero
        while (top > 0 && xr[top - 1] == 0.0f) --top;
        const int coded = (top + 17) / 18;                      // subbands holding spectrum
        const int busy = coded > c.live ? coded : c.live;       // ... or overlap from the granule before
        float slots[32][kLanes];   // [subband][time slot]
        memset(slots, 0, sizeof(slots));
        for (int sb = 0; sb < busy; ++sb) {
            float y[36];
            const float* x = xr + 18 * sb;
            if (sb >= coded) {
                memset(y, 0, sizeof(y));
            } else if (sb < long_subbands) {
                float raw[20];
                for (int r = 0; r < 20; ++r) raw[r] = 0.0f;
                for (int k = 0; k < 18; ++k) {
                    const float xk = x[k];
                    const float* cr = T.imdct_cos[k];
                    for (int r = 0; r < 20; ++r) raw[r] += cr[r] * xk;
                }
                const float* w = T.long_window[(g.block_type == 2 || (g.low_normal && sb < 2)) ? 0 : g.block_type];
                for (int i = 0; i < 9; ++i) {   // outputs 17 - i and 53 - i mirror outputs i and 18 + i
                    y[i] = w[i] * raw[i];
                    y[17 - i] = -w[17 - i] * raw[i];
                    y[18 + i] = w[18 + i] * raw[9 + i];
                    y[35 - i] = w[35 - i] * raw[9 + i];
                }
            } else {
                memset(y, 0, sizeof(y));
                for (int w = 0; w < 3; ++w) {
                    float acc[12];
                    for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
                    for (int k = 0; k < 6; ++k) {
                        const float xk = x[3 * k + w];
                        for (int i = 0; i < 12; ++i) acc[i] += T.imdct_short[k][i] * xk;
                    }
                    for (int i = 0; i < 12; ++i) y[6 + 6 * w + i] += acc[i];
                }
            }
            for (int i = 0; i < 18; ++i) {
                float s = y[i] + c.overlap[sb][i];
                c.overlap[sb][i] = y[18 + i];
                if ((sb & 1) && (i & 1)) s = -s;   // frequency inversion
                slots[sb][i] = s;
            }
        }
        c.live = coded;
        // matrixing: v[i] = sum_k cos((16 + i)(2k + 1) pi / 64) s[k] from one 32-point DCT B[m] per time slot
        dct2<32>(slots, T.lee);
        for (int t = 0; t < 18; ++t) {
            c.voff = (c.voff - 64) & 1023;
            float* v = c.v + c.voff;
            for (int i = 0; i < 16; ++i) v[i] = slots[16 + i][t];
            v[16] = 0.0f;
            for (int i = 17; i < 48; ++i) v[i] = -slots[48 - i][t];
            for (int i = 48; i < 64; ++i) v[i] = -slots[i - 48][t];
            memcpy(v + 1024, v, 64 * sizeof(float));
            float acc[32];
            for (int j = 0; j < 32; ++j) acc[j] = 0.0f;
            for (int i = 0; i < 8; ++i) {
                const float* w = T.window + 64 * i;
                const float* va = v + 128 * i;
                for (int j = 0; j < 32; ++j) acc[j] += w[j] * va[j] + w[32 + j] * va[96 + j];
            }
            for (int j = 0; j < 32; ++j) pcm[(size_t)(32 * t + j) * stride] = acc[j];
        }
    }

    // ---- one frame -> pcm (interleaved, h.samples() per channel); silence when its main data is not all there ----
    void decode_frame(const uint8_t* f, const Header& h, float* pcm) {
        const int nch = h.channels, spf = h.samples();
        memset(pcm, 0, sizeof(float) * spf * nch);
        note(0);
        const uint8_t* side = f + 4 + (h.crc ? 2 : 0);
        SideInfo si;
        bool ok = parse_side_info(side, h, si);
        if (ok && h.crc) {   // CRC-16 (x^16 + x^15 + x^2 + 1, start 0xFFFF) over header bytes 2..3 and the side information
            uint32_t crc = 0xFFFF;
            auto feed = [&](uint8_t byte) {
                crc ^= (uint32_t)byte << 8;
                for (int k = 0; k < 8; ++k) crc = (crc & 0x8000) ? ((crc << 1) ^ 0x8005) : (crc << 1);
                crc &= 0xFFFF;
            };
            feed(f[2]), feed(f[3]);
            for (int i = 0; i < h.side_bytes; ++i) feed(side[i]);
            ok = crc == (uint32_t)((f[4] << 8) | f[5]);
        }
        const uint8_t* md = side + h.side_bytes;
        const int md_bytes = h.frame_bytes - (int)(md - f);
        // main data of this frame starts main_data_begin bytes before the end of what earlier frames brought
        const int64_t have = (int64_t)main.size();
        if (have > 4096) {
            main.erase(main.begin(), main.end() - 1024);
        }
        const int64_t before = (int64_t)main.size();
        main.insert(main.end(), md, md + md_bytes);
        if (!ok) {
            note(3);
            for (int c = 0; c < nch; ++c) memset(ch[c].overlap, 0, sizeof(ch[c].overlap));
            return;
        }
        const int64_t start = before - si.main_data_begin;
        if (start < 0) {
            note(3);
            for (int c = 0; c < nch; ++c) memset(ch[c].overlap, 0, sizeof(ch[c].overlap));
            return;
        }
        main.insert(main.end(), 8, 0);
        const int64_t limit = ((int64_t)main.size() - 8) * 8;
        Bits b{main.data(), start * 8, limit};
        float xr[2][576];
        uint8_t scf[2][40];
        for (int gi = 0; gi < h.granules(); ++gi) {
            bool bad = false;
            for (int c = 0; c < nch; ++c) {
                Granule& g = si.gr[gi][c];
                const int64_t end = b.pos + g.part23;
                if (end > limit || !g.supported) {
                    bad = true;
                    b.pos = end > limit ? limit : end;
                    memset(xr[c], 0, sizeof(xr[c]));
                    memset(scf[c], 0, 40);
                    continue;
                }
                read_scalefactors(b, h, si, gi, c, g, scf[c]);
                int is[576 + 4];
                int used = 0;
                if (b.pos <= end)
                    used = read_spectrum(b, end, h, g, is);
                else
                    memset(is, 0, sizeof(is)), b.pos = end, note(2);
                requantize(g, scf[c], is, used, xr[c]);
            }
            if (bad) note(3);
            if (nch == 2 && h.mode == 1 && h.mode_ext && !bad) stereo(xr[0], xr[1], h, si.gr[gi][1], scf[1]);
            for (int c = 0; c < nch; ++c)
                hybrid_and_synthesis(si.gr[gi][c], xr[c], ch[c], pcm + (size_t)gi * 576 * nch + c, nch);
        }
        main.resize(main.size() - 8);
    }
};

// How far before frame `first` a decoder must start so that frame `first` comes out as in an in-order decode.
// Frame `first` needs (a) the main data that main_data_begin can reach back to, (b) the overlap of the granule
// before it and (c) 16 time slots of synthesis history - one granule, which in the second standard is a whole
// frame whose own samples depend on the overlap of the frame before that.  So the two frames before `first`
// must decode correctly, which takes the reservoir the earlier of them can point into: the walk goes back from
// frame first - 3 until the main-data bytes gathered reach the largest main_data_begin (511 bytes; 255 in the
// second standard).  At 128 kbit/s that is five frames in all; at 8 kbit/s and 22.05 kHz (13 bytes of main data
// per frame) it is twenty-three - a fixed count of frames is wrong for one or wasteful for the other.
static int64_t warm_start(const StreamPlan& sp, int64_t first) {
    if (first <= 0) return 0;
    const Header& h0 = sp.frames.front().h;
    const int reach = h0.samples() == 576 ? 255 : 511;
    int64_t f = first - 2 > 0 ? first - 2 : 0;
    int gathered = 0;
    for (int walked = 0; f > 0 && gathered < reach && walked < 512; ++walked) {
        --f;
        const Header& h = sp.frames[f].h;
        const int md = h.frame_bytes - 4 - (h.crc ? 2 : 0) - h.side_bytes;
        if (md > 0) gathered += md;
    }
    return f;
}

// Frames [first, last) of the plan; of what they produce, the output samples [lo, hi) of the stream go to
// out[(s - lo) * channels ...].
static void decode_range(const uint8_t* d, const StreamPlan& sp, int64_t first, int64_t last, float* out,
                         int64_t lo, int64_t hi, int64_t* diag, bool mono) {
    Decoder dec;
    const int nch = sp.frames.front().h.channels;
    const int nout = mono ? 1 : nch;
    const int spf = sp.frames.front().h.samples();
    std::vector<float> pcm((size_t)spf * 2);
    const int64_t warm = warm_start(sp, first);
    for (int64_t f = warm; f < last; ++f) {
        Header h = sp.frames[f].h;
        if (h.channels != nch) {   // a channel-count change inside a stream: deliver silence of the stream's shape
            h = sp.frames.front().h;
            std::fill(pcm.begin(), pcm.end(), 0.0f);
            dec.reset();
        } else {
            dec.counting = f >= first;   // warm-up frames are another thread's to report
            dec.decode_frame(d + sp.frames[f].at, h, pcm.data());
        }
        if (f < first) continue;
        if (mono && nch == 2)   // the mean of the two channels, formed in double and rounded once
            for (int i = 0; i < spf; ++i) pcm[i] = (float)(((double)pcm[2 * i] + (double)pcm[2 * i + 1]) * 0.5);
        int64_t s0 = f * spf - sp.lead, n = spf, skip = 0;   // this frame covers output samples [s0, s0 + spf)
        if (s0 < lo) skip = lo - s0;
        if (s0 + n > sp.keep) n = sp.keep - s0;
        if (s0 + n > hi) n = hi - s0;
        if (n > skip)
            memcpy(out + (size_t)(s0 + skip - lo) * nout, pcm.data() + (size_t)skip * nout, sizeof(float) * (size_t)(n - skip) * nout);
    }
    for (int k = 0; k < 5; ++k) diag[k] = dec.diag[k];
}

}  // namespace mp3
}  // namespace oasr

using namespace oasr;

extern "C" {

int oasr_mp3_info(const uint8_t* data, int64_t size, int32_t* sample_rate, int32_t* channels,
                  int64_t* samples_per_channel) {
    OASR_REQUIRE(data && size > 0 && sample_rate && channels && samples_per_channel, "mp3_info: null pointer");
    mp3::StreamPlan sp;
    const int rc = mp3::plan_stream(data, size, sp);
    if (rc != OASR_OK) return rc;
    *sample_rate = sp.frames.front().h.rate;
    *channels = sp.frames.front().h.channels;
    *samples_per_channel = sp.keep;
    return OASR_OK;
}

}  // extern "C"

// Output samples [lo, hi) of a planned stream -> out, by one or several threads on frame ranges.
static void decode_window(const uint8_t* data, const mp3::StreamPlan& sp, int64_t lo, int64_t hi, float* out, bool mono,
                          int64_t* diagnostics) {
    const int spf = sp.frames.front().h.samples();
    const int64_t all = (int64_t)sp.frames.size();
    int64_t f0 = (lo + sp.lead) / spf, f1 = (hi + sp.lead + spf - 1) / spf;   // frames that hold the window
    if (f0 > all) f0 = all;
    if (f1 > all) f1 = all;
    const int64_t nf = f1 > f0 ? f1 - f0 : 0;
    int threads = 1;
    if (const char* e = getenv("OASR_MP3_THREADS")) {
        threads = atoi(e);
    } else if (nf >= 4096) {
        const unsigned hc = std::thread::hardware_concurrency();
        threads = (int)(hc ? hc : 1);
        if (threads > 16) threads = 16;
        if (threads > nf / 1024) threads = (int)(nf / 1024);
    }
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (threads > nf) threads = nf > 0 ? (int)nf : 1;
    std::vector<int64_t> diag((size_t)threads * 5, 0);
    if (threads == 1) {
        mp3::decode_range(data, sp, f0, f1, out, lo, hi, diag.data(), mono);
    } else {
        std::vector<std::thread> pool;
        bool spawn_failed = false;
        for (int t = 0; t < threads; ++t) {
            const int64_t a = f0 + nf * t / threads, b = f0 + nf * (t + 1) / threads;
            try {
                pool.emplace_back(mp3::decode_range, data, std::cref(sp), a, b, out, lo, hi, diag.data() + (size_t)t * 5,
                                  mono);
            } catch (...) {
                spawn_failed = true;
                break;
            }
        }
        for (auto& th : pool) th.join();
        if (spawn_failed) {   // a thread could not be started: decode everything in order instead
            std::fill(diag.begin(), diag.end(), 0);
            threads = 1;
            mp3::decode_range(data, sp, f0, f1, out, lo, hi, diag.data(), mono);
        }
    }
    if (diagnostics) {
        for (int k = 0; k < 5; ++k) diagnostics[k] = 0;
        for (int t = 0; t < threads; ++t)
            for (int k = 0; k < 5; ++k) diagnostics[k] += diag[(size_t)t * 5 + k];
        diagnostics[0] = nf;
        diagnostics[1] = sp.skipped;
    }
}

static int mp3_decode_to(const uint8_t* data, int64_t size, float* out, int64_t capacity_per_channel,
                         int64_t* written_per_channel, int64_t* diagnostics, bool mono) {
    OASR_REQUIRE(data && size > 0 && out && written_per_channel, "mp3_decode: null pointer");
    OASR_REQUIRE(capacity_per_channel >= 0, "mp3_decode: negative capacity");
    mp3::StreamPlan sp;
    const int rc = mp3::plan_stream(data, size, sp);
    if (rc != OASR_OK) return rc;
    if (sp.keep > capacity_per_channel) {
        set_error("mp3_decode: output holds %lld samples per channel, the stream has %lld", (long long)capacity_per_channel,
                  (long long)sp.keep);
        return OASR_ECAPACITY;
    }
    decode_window(data, sp, 0, sp.keep, out, mono, diagnostics);
    *written_per_channel = sp.keep;
    return OASR_OK;
}

struct oasr_mp3_stream {
    const uint8_t* data;
    mp3::StreamPlan plan;
};

extern "C" {

int oasr_mp3_decode(const uint8_t* data, int64_t size, float* out, int64_t capacity_per_channel,
                    int64_t* written_per_channel, int64_t* diagnostics) {
    return mp3_decode_to(data, size, out, capacity_per_channel, written_per_channel, diagnostics, false);
}

int oasr_mp3_decode_mono(const uint8_t* data, int64_t size, float* out, int64_t capacity, int64_t* written,
                         int64_t* diagnostics) {
    return mp3_decode_to(data, size, out, capacity, written, diagnostics, true);
}

int oasr_mp3_open(const uint8_t* data, int64_t size, oasr_mp3_stream** stream, int32_t* sample_rate, int32_t* channels,
                  int64_t* samples_per_channel) {
    OASR_REQUIRE(data && size > 0 && stream && sample_rate && channels && samples_per_channel, "mp3_open: null pointer");
    oasr_mp3_stream* s = new (std::nothrow) oasr_mp3_stream;
    if (!s) {
        set_error("mp3_open: out of host memory");
        return OASR_ENOMEM;
    }
    s->data = data;
    const int rc = mp3::plan_stream(data, size, s->plan);
    if (rc != OASR_OK) {
        delete s;
        return rc;
    }
    *sample_rate = s->plan.frames.front().h.rate;
    *channels = s->plan.frames.front().h.channels;
    *samples_per_channel = s->plan.keep;
    *stream = s;
    return OASR_OK;
}

int oasr_mp3_decode_mono_range(const oasr_mp3_stream* stream, int64_t first, int64_t count, float* out) {
    OASR_REQUIRE(stream && out, "mp3_decode_mono_range: null pointer");
    OASR_REQUIRE(first >= 0 && count >= 0 && first + count <= stream->plan.keep,
                 "mp3_decode_mono_range: samples %lld .. %lld lie outside the stream's %lld", (long long)first,
                 (long long)(first + count), (long long)stream->plan.keep);
    if (count > 0) decode_window(stream->data, stream->plan, first, first + count, out, true, nullptr);
    return OASR_OK;
}

void oasr_mp3_close(oasr_mp3_stream* stream) { delete stream; }

}  // extern "C"
