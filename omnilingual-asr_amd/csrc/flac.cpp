// FLAC stream decoder (host code): the lossless format among the upload types the reference accepts
// (ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:106-118, ref:workflows/wav2elan_web/app.py:76-77;
// the reference forwards the file bytes to its remote model, here they are decoded locally).
// Implements the FLAC format as published (RFC 9639): STREAMINFO, frame headers with CRC-8, the
// CONSTANT / VERBATIM / FIXED / LPC subframes, both Rice residual codings with escape partitions, wasted
// bits, the three stereo decorrelations, frame CRC-16.  Output: interleaved int32 samples.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "oasr_common.h"

namespace oasr {
namespace flac {

struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;   // byte position of the next byte to load
    uint64_t acc = 0;
    int bits = 0;        // valid bits in acc (MSB-aligned use: top `bits` of the low part)
    bool bad = false;
    BitReader(const uint8_t* d, size_t len) : p(d), n(len) {}
    void fill() {
        if (bits <= 32 && pos + 4 <= n) {   // four bytes at once while they last
            uint32_t w;
            memcpy(&w, p + pos, 4);
            acc = (acc << 32) | __builtin_bswap32(w);
            pos += 4;
            bits += 32;
        }
        while (bits <= 56 && pos < n) {
            acc = (acc << 8) | p[pos++];
            bits += 8;
        }
    }
    uint32_t get(int k) {   // 0 <= k <= 32
        if (k == 0) return 0;
        fill();
        if (bits < k) {
            bad = true;
            return 0;
        }
        const uint32_t v = (uint32_t)((acc >> (bits - k)) & ((k == 32) ? 0xFFFFFFFFull : ((1ull << k) - 1)));
        bits -= k;
        return v;
    }
    int64_t get_signed(int k) {   // 1 <= k <= 33
        if (k <= 32) {
            const uint32_t v = get(k);
            return (int64_t)((uint64_t)v << (64 - k)) >> (64 - k);
        }
        const uint64_t hi = get(k - 32);
        const uint64_t v = (hi << 32) | get(32);
        return (int64_t)(v << (64 - k)) >> (64 - k);
    }
    uint32_t unary() {   // number of 0 bits before the next 1 bit
        uint32_t q = 0;
        for (;;) {
            fill();
            if (bits == 0) {
                bad = true;
                return 0;
            }
            const uint64_t window = acc << (64 - bits);   // valid bits at the top
            if (window == 0) {
                q += bits;
                bits = 0;
                continue;
            }
            const int z = __builtin_clzll(window);
            q += z;
            bits -= z + 1;
            return q;
        }
    }
    // One Rice-coded value: unary quotient, a 1 bit, `param` (< 32) low bits; the zig-zag fold is the caller's.
    // Fast path: the whole code lies in the bits already loaded (the common case: the refill leaves >= 57).
    uint64_t rice(int param) {
        fill();
        if (bits > 0) {
            const uint64_t window = acc << (64 - bits);   // valid bits at the top
            if (window != 0) {
                const int z = __builtin_clzll(window);
                const int used = z + 1 + param;
                if (used <= bits) {
                    const uint64_t low = param ? ((window << (z + 1)) >> (64 - param)) : 0;
                    bits -= used;
                    return ((uint64_t)z << param) | low;
                }
            }
        }
        const uint64_t q = unary();
        return (q << param) | get(param);
    }
    void align() { bits -= bits & 7; }
    size_t byte_pos() const { return pos - bits / 8; }   // only meaningful when aligned
};

// CRC-8 (polynomial 0x07) and CRC-16 (0x8005), MSB first, one table look-up per byte.
struct CrcTables {
    uint8_t t8[256];
    uint16_t t16[256];
    CrcTables() {
        for (int v = 0; v < 256; ++v) {
            uint8_t c = (uint8_t)v;
            uint16_t w = (uint16_t)(v << 8);
            for (int b = 0; b < 8; ++b) {
                c = (c & 0x80) ? (uint8_t)((c << 1) ^ 0x07) : (uint8_t)(c << 1);
                w = (w & 0x8000) ? (uint16_t)((w << 1) ^ 0x8005) : (uint16_t)(w << 1);
            }
            t8[v] = c;
            t16[v] = w;
        }
    }
};
static const CrcTables kCrc;

static uint8_t crc8(const uint8_t* d, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) c = kCrc.t8[c ^ d[i]];
    return c;
}

static uint16_t crc16(const uint8_t* d, size_t n) {
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ kCrc.t16[(c >> 8) ^ d[i]]);
    return c;
}

struct StreamInfo {
    int rate = 0, channels = 0, bits = 0, min_block = 0, max_block = 0;
    int64_t total = 0;
    uint8_t md5[16] = {0};
    size_t audio_start = 0;
};

static int parse_header(const uint8_t* d, size_t n, StreamInfo& si) {
    size_t pos = 0;
    if (n >= 10 && !memcmp(d, "ID3", 3))   // a tag some writers put in front
        pos = 10 + (((size_t)d[6] & 0x7F) << 21 | ((size_t)d[7] & 0x7F) << 14 | ((size_t)d[8] & 0x7F) << 7 | ((size_t)d[9] & 0x7F));
    if (pos + 4 > n || memcmp(d + pos, "fLaC", 4)) {
        set_error("flac: no fLaC stream marker");
        return OASR_EINVAL;
    }
    pos += 4;
    bool seen = false;
    for (;;) {
        if (pos + 4 > n) {
            set_error("flac: truncated metadata");
            return OASR_EINVAL;
        }
        const int last = d[pos] >> 7, type = d[pos] & 0x7F;
        const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
        pos += 4;
        if (pos + len > n) {
            set_error("flac: truncated metadata block");
            return OASR_EINVAL;
        }
        if (type == 0) {
            if (len < 34) {
                set_error("flac: short STREAMINFO");
                return OASR_EINVAL;
            }
            const uint8_t* b = d + pos;
            si.min_block = (b[0] << 8) | b[1];
            si.max_block = (b[2] << 8) | b[3];
            uint64_t x = 0;
            for (int i = 10; i < 18; ++i) x = (x << 8) | b[i];
            si.rate = (int)(x >> 44);
            si.channels = (int)((x >> 41) & 7) + 1;
            si.bits = (int)((x >> 36) & 31) + 1;
            si.total = (int64_t)(x & ((1ull << 36) - 1));
            memcpy(si.md5, b + 18, 16);
            seen = true;
        }
        pos += len;
        if (last) break;
    }
    if (!seen || si.rate <= 0 || si.bits < 4 || si.bits > 32) {
        set_error("flac: missing or invalid STREAMINFO");
        return OASR_EINVAL;
    }
    si.audio_start = pos;
    return OASR_OK;
}

static bool read_residual(BitReader& br, int64_t* out, int blocksize, int pred_order) {
    const int method = (int)br.get(2);
    if (method > 1) return false;
    const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int porder = (int)br.get(4);
    const int parts = 1 << porder;
    if ((blocksize >> porder) << porder != blocksize && porder > 0) return false;
    int idx = 0;
    for (int pt = 0; pt < parts; ++pt) {
        int count = (blocksize >> porder) - (pt == 0 ? pred_order : 0);
        if (porder == 0) count = blocksize - pred_order;
        if (count < 0) return false;
        const int param = (int)br.get(pbits);
        if (param == esc) {
            const int nb = (int)br.get(5);
            for (int i = 0; i < count; ++i) out[idx++] = nb ? br.get_signed(nb) : 0;
        } else {
            for (int i = 0; i < count; ++i) {
                const uint64_t u = br.rice(param);
                out[idx++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
            }
        }
        if (br.bad) return false;
    }
    return idx == blocksize - pred_order;
}

// s[i] += (sum_j coef[j] * s[i - 1 - j]) >> shift for i >= ORDER: the order is a compile-time constant for the
// orders encoders use, so the inner sum is straight-line code.
// (Sums and products wrap modulo 2^64 - unsigned arithmetic, reinterpreted: a valid stream stays far inside the range,
// a damaged one may not, and its garbage must not be undefined behaviour.)
template <int ORDER>
static void lpc_restore_n(int64_t* s, int blocksize, const int64_t* coef, int shift) {
    for (int i = ORDER; i < blocksize; ++i) {
        uint64_t acc = 0;
#pragma GCC unroll 32
        for (int j = 0; j < ORDER; ++j) acc += (uint64_t)coef[j] * (uint64_t)s[i - 1 - j];
        s[i] = (int64_t)((uint64_t)s[i] + (uint64_t)((int64_t)acc >> shift));
    }
}
static void lpc_restore(int64_t* s, int blocksize, int order, const int64_t* coef, int shift) {
    switch (order) {
#define OASR_LPC_CASE(N) case N: lpc_restore_n<N>(s, blocksize, coef, shift); return;
        OASR_LPC_CASE(1) OASR_LPC_CASE(2) OASR_LPC_CASE(3) OASR_LPC_CASE(4) OASR_LPC_CASE(5) OASR_LPC_CASE(6)
        OASR_LPC_CASE(7) OASR_LPC_CASE(8) OASR_LPC_CASE(9) OASR_LPC_CASE(10) OASR_LPC_CASE(11) OASR_LPC_CASE(12)
#undef OASR_LPC_CASE
        default: break;
    }
    for (int i = order; i < blocksize; ++i) {
        uint64_t acc = 0;
        for (int j = 0; j < order; ++j) acc += (uint64_t)coef[j] * (uint64_t)s[i - 1 - j];
        s[i] = (int64_t)((uint64_t)s[i] + (uint64_t)((int64_t)acc >> shift));
    }
}

static bool read_subframe(BitReader& br, int64_t* s, int blocksize, int bps) {
    if (br.get(1)) return false;
    const int type = (int)br.get(6);
    int wasted = 0;
    if (br.get(1)) wasted = (int)br.unary() + 1;
    bps -= wasted;
    if (bps < 1 || br.bad) return false;
    if (type == 0) {
        const int64_t v = br.get_signed(bps);
        for (int i = 0; i < blocksize; ++i) s[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < blocksize; ++i) s[i] = br.get_signed(bps);
    } else if (type >= 8 && type <= 12) {
        const int order = type - 8;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) s[i] = br.get_signed(bps);
        if (!read_residual(br, s + order, blocksize, order)) return false;
        for (int i = order; i < blocksize; ++i) {
            const uint64_t a = (uint64_t)s[i - 1], b = order > 1 ? (uint64_t)s[i - 2] : 0, c = order > 2 ? (uint64_t)s[i - 3] : 0,
                           d = order > 3 ? (uint64_t)s[i - 4] : 0;
            uint64_t pred = 0;   // modulo 2^64, as above
            switch (order) {
                case 1: pred = a; break;
                case 2: pred = 2 * a - b; break;
                case 3: pred = 3 * a - 3 * b + c; break;
                case 4: pred = 4 * a - 6 * b + 4 * c - d; break;
                default: break;
            }
            s[i] = (int64_t)((uint64_t)s[i] + pred);
        }
    } else if (type >= 32) {
        const int order = (type & 31) + 1;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) s[i] = br.get_signed(bps);
        const int prec = (int)br.get(4) + 1;
        if (prec == 16) return false;
        const int shift = (int)br.get_signed(5);
        if (shift < 0) return false;
        int64_t coef[32];
        for (int j = 0; j < order; ++j) coef[j] = br.get_signed(prec);
        if (!read_residual(br, s + order, blocksize, order)) return false;
        lpc_restore(s, blocksize, order, coef, shift);
    } else {
        return false;   // reserved subframe type
    }
    if (wasted)
        for (int i = 0; i < blocksize; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
    return !br.bad;
}

// One frame at d[pos..]: header (sync, CRC-8), subframes, CRC-16, stereo reconstruction.  On success returns
// the frame's length in bytes and fills `blocksize`, `first_sample` (from the coded frame / sample number) and
// ch[c][0..blocksize).  On failure returns 0 and, if `why` is given, a short reason (no error state is set here:
// this also runs in worker threads and on candidate positions that turn out not to be frames).
struct FrameScratch {
    std::vector<int64_t> ch[8];
};
static size_t decode_frame(const uint8_t* d, size_t n, size_t pos, const StreamInfo& si, FrameScratch& fs,
                           int& blocksize, int64_t& first_sample, int& nch_out, const char** why) {
    static const int kBlock[16] = {0, 192, 576, 1152, 2304, 4608, -8, -16, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
    static const int kBits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    auto fail = [&](const char* msg) {
        if (why) *why = msg;
        return (size_t)0;
    };
    if (pos + 6 > n || !(d[pos] == 0xFF && (d[pos + 1] & 0xFE) == 0xF8)) return fail("lost frame sync");
    const bool variable = d[pos + 1] & 1;   // blocking strategy: the coded number counts samples, not frames
    BitReader br(d + pos, n - pos);
    br.get(16);
    const int bs_code = (int)br.get(4), sr_code = (int)br.get(4);
    const int ch_code = (int)br.get(4), bits_code = (int)br.get(3);
    if (br.get(1) || bs_code == 0 || sr_code == 15 || ch_code > 10 || kBits[bits_code] < 0)
        return fail("reserved field in a frame header");
    uint64_t number;
    {   // frame or sample number, "UTF-8" coded, up to 36 bits
        const uint32_t b0 = br.get(8);
        int extra = 0;
        if (b0 >= 0xFE) extra = 6;
        else if (b0 >= 0xFC) extra = 5;
        else if (b0 >= 0xF8) extra = 4;
        else if (b0 >= 0xF0) extra = 3;
        else if (b0 >= 0xE0) extra = 2;
        else if (b0 >= 0xC0) extra = 1;
        else if (b0 >= 0x80) return fail("malformed frame number");
        number = extra ? (b0 & (0x3Fu >> extra)) : b0;
        for (int i = 0; i < extra; ++i) {
            const uint32_t bx = br.get(8);
            if ((bx & 0xC0) != 0x80) return fail("malformed frame number");
            number = (number << 6) | (bx & 0x3F);
        }
    }
    blocksize = kBlock[bs_code];
    if (blocksize == -8) blocksize = (int)br.get(8) + 1;
    else if (blocksize == -16) blocksize = (int)br.get(16) + 1;
    if (sr_code == 12) br.get(8);
    else if (sr_code == 13 || sr_code == 14) br.get(16);
    const size_t hdr_len = br.byte_pos();
    const uint32_t want8 = br.get(8);
    if (br.bad || crc8(d + pos, hdr_len) != want8) return fail("frame header CRC mismatch");
    const int bps = kBits[bits_code] ? kBits[bits_code] : si.bits;
    const int nch = ch_code < 8 ? ch_code + 1 : 2;
    if (nch != si.channels) return fail("channel count changes inside the stream");
    first_sample = variable ? (int64_t)number : (int64_t)number * si.max_block;
    nch_out = nch;
    for (int c = 0; c < nch; ++c) {
        fs.ch[c].resize(blocksize);
        const int side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
        if (!read_subframe(br, fs.ch[c].data(), blocksize, bps + side)) return fail("malformed subframe");
    }
    br.align();
    const size_t body_len = br.byte_pos();
    const uint32_t want16 = br.get(16);
    if (br.bad || crc16(d + pos, body_len) != want16) return fail("frame CRC mismatch");
    int64_t* c0 = fs.ch[0].data();
    int64_t* c1 = fs.ch[1].data();
    // (modulo 2^64 like the predictors: a damaged frame's values may be anything)
    if (ch_code == 8) {
        for (int i = 0; i < blocksize; ++i) c1[i] = (int64_t)((uint64_t)c0[i] - (uint64_t)c1[i]);
    } else if (ch_code == 9) {
        for (int i = 0; i < blocksize; ++i) c0[i] = (int64_t)((uint64_t)c1[i] + (uint64_t)c0[i]);
    } else if (ch_code == 10) {
        for (int i = 0; i < blocksize; ++i) {
            const uint64_t side = (uint64_t)c1[i], mid = (uint64_t)c0[i] * 2 + (side & 1);
            c0[i] = (int64_t)(mid + side) >> 1;
            c1[i] = (int64_t)(mid - side) >> 1;
        }
    }
    return br.byte_pos();
}

// Where decoded samples go: interleaved int32 at the stream's bit depth, or interleaved int16 scaled to 16 bits
// (streams of at most 16 bits per sample; value << (16 - bits), so that sample / 32768 is the same number).
struct Sink {
    int32_t* o32 = nullptr;
    int16_t* o16 = nullptr;
    int shift = 0;
    explicit operator bool() const { return o32 || o16; }
};
// samples [skip, skip + take) of the frame go to sample positions at .. at + take of the output
static void emit(const FrameScratch& fs, int nch, int64_t take, const Sink& sk, int64_t at, int64_t skip = 0) {
    if (sk.o16) {
        int16_t* dst = sk.o16 + at * nch;
        for (int c = 0; c < nch; ++c) {
            const int64_t* src = fs.ch[c].data() + skip;
            for (int64_t i = 0; i < take; ++i) dst[i * nch + c] = (int16_t)((uint64_t)src[i] << sk.shift);
        }
    } else {
        int32_t* dst = sk.o32 + at * nch;
        for (int c = 0; c < nch; ++c) {
            const int64_t* src = fs.ch[c].data() + skip;
            for (int64_t i = 0; i < take; ++i) dst[i * nch + c] = (int32_t)src[i];
        }
    }
}

// Decodes every frame in order; returns samples per channel written (<= capacity), or a negative status.
static int64_t decode(const uint8_t* d, size_t n, const StreamInfo& si, const Sink& out, int64_t capacity) {
    size_t pos = si.audio_start;
    int64_t done = 0;
    FrameScratch fs;
    while (pos + 2 <= n) {
        int blocksize = 0, nch = 0;
        int64_t first = 0;
        const char* why = "";
        const size_t len = decode_frame(d, n, pos, si, fs, blocksize, first, nch, &why);
        if (len == 0) {
            if (si.total && done >= si.total) break;   // trailing bytes after the last frame
            set_error("flac: %s at byte %zu", why, pos);
            return OASR_EINVAL;
        }
        int64_t take = blocksize;
        if (si.total && done + take > si.total) take = si.total - done;
        if (out) {
            if (done + take > capacity) {
                set_error("flac: output holds %lld samples per channel, stream has more", (long long)capacity);
                return OASR_ECAPACITY;
            }
            emit(fs, nch, take, out, done);
        }
        done += take;
        pos += len;
    }
    return done;
}

// The same result from several threads.  The bytes after the metadata are cut into equal ranges; a worker
// finds the first position in its range where a whole frame decodes (sync code, header CRC-8, subframes,
// CRC-16), decodes on from there until a frame starts at or past the end of its range, and writes each frame
// at the sample position its header states.  Afterwards the ranges must chain: every worker stopped exactly
// where the next one started, the first started at the first frame, the last reached the end of the data, and
// the sample positions are contiguous.  Anything else (a false start inside a frame, a damaged file, a stream
// without a fixed block size or total) returns -1 and the caller decodes sequentially.
static int64_t decode_parallel(const uint8_t* d, size_t n, const StreamInfo& si, const Sink& out, int64_t capacity,
                               int threads) {
    if (!out || si.total <= 0 || si.total > capacity || si.max_block <= 0 || si.min_block != si.max_block) return -1;
    const size_t begin = si.audio_start, span = n - begin;
    struct Part {
        size_t first = 0, end = 0;          // byte position of the first frame decoded / after the last one
        int64_t s_first = -1, s_end = -1;   // sample range covered
        bool ok = false;
    };
    std::vector<Part> parts(threads);
    auto work = [&](int k) {
        Part& pt = parts[k];
        const size_t lo = begin + span * k / threads, hi = begin + span * (k + 1) / threads;
        FrameScratch fs;
        size_t pos = lo;
        int blocksize = 0, nch = 0;
        int64_t first = 0;
        size_t len = 0;
        for (; pos < hi; ++pos) {   // first whole frame at or after lo
            if (d[pos] != 0xFF || pos + 1 >= n || (d[pos + 1] & 0xFE) != 0xF8) continue;
            len = decode_frame(d, n, pos, si, fs, blocksize, first, nch, nullptr);
            if (len) break;
        }
        if (!len) {                 // no frame starts in this range: an empty part that chains through
            pt.first = pt.end = hi;
            pt.ok = true;
            return;
        }
        pt.first = pos;
        pt.s_first = first;
        for (;;) {
            if (first < 0 || first >= si.total) return;                      // not ok
            const int64_t take = first + blocksize > si.total ? si.total - first : blocksize;
            emit(fs, nch, take, out, first);
            pt.s_end = first + take;
            pos += len;
            if (pos >= hi || pos + 2 > n) break;
            len = decode_frame(d, n, pos, si, fs, blocksize, first, nch, nullptr);
            if (!len) {
                if (pt.s_end >= si.total) break;                             // trailing bytes after the last frame
                return;                                                      // not ok
            }
            if (first != pt.s_end) return;                                   // not ok
        }
        pt.end = pos;
        pt.ok = true;
    };
    std::vector<std::thread> pool;
    bool started = true;
    try {
        for (int k = 1; k < threads; ++k) pool.emplace_back(work, k);
    } catch (...) {   // no more threads to be had: finish what started, then decode in order
        started = false;
    }
    if (started) work(0);
    for (auto& t : pool) t.join();
    if (!started) return -1;
    // chain check
    size_t at = begin;
    int64_t s_at = 0;
    for (int k = 0; k < threads; ++k) {
        const Part& pt = parts[k];
        if (!pt.ok) return -1;
        if (pt.s_first < 0) {            // empty part
            continue;
        }
        if (pt.first != at || pt.s_first != s_at) return -1;
        at = pt.end;
        s_at = pt.s_end;
    }
    if (s_at != si.total) return -1;
    return si.total;
}

// Samples [first, first + count) of the stream into out[0 .. count): frames are self-contained, so only the
// frames that cover the range are decoded.  The frame that holds `first` is found from a byte position guessed
// by proportion (the coded sample number in every frame header says where a candidate lies): a candidate is a
// position where a whole frame decodes (sync code, header CRC-8, subframes, CRC-16) and the next frame follows it
// sample for sample; if the first candidate found already lies past `first`, the search restarts further back.
// Returns OASR_OK or a negative status (error text set).
static int decode_range_one(const uint8_t* d, size_t n, const StreamInfo& si, const Sink& out, int64_t first, int64_t count) {
    if (count <= 0) return OASR_OK;
    const size_t begin = si.audio_start;
    const int64_t last = first + count;
    FrameScratch fs;
    size_t guess = begin + (size_t)((double)(n - begin) * ((double)first / (double)si.total));
    size_t back = 1 << 16;
    for (;;) {
        size_t pos = guess > begin + back ? guess - back : begin;
        const bool from_start = pos == begin;
        int blocksize = 0, nch = 0;
        int64_t f0 = 0;
        size_t len = 0;
        for (; pos + 2 <= n; ++pos) {
            if (d[pos] != 0xFF || (d[pos + 1] & 0xFE) != 0xF8) continue;
            len = decode_frame(d, n, pos, si, fs, blocksize, f0, nch, nullptr);
            if (!len) continue;
            if (f0 + blocksize >= si.total) break;          // the stream's last frame: nothing to chain to
            // a chance match inside a frame's data would not be followed by a frame that continues it
            FrameScratch probe;
            int bs2 = 0, nch2 = 0;
            int64_t f2 = 0;
            if (decode_frame(d, n, pos + len, si, probe, bs2, f2, nch2, nullptr) && f2 == f0 + blocksize) break;
            len = 0;
        }
        if (!len && !from_start) {       // the guess lay inside the stream's last frame(s): look further back
            back *= 4;
            continue;
        }
        if (!len) {
            set_error("flac: no frame found for sample %lld", (long long)first);
            return OASR_EINVAL;
        }
        if (f0 > first) {
            if (from_start) {
                set_error("flac: the stream starts at sample %lld, after the requested %lld", (long long)f0, (long long)first);
                return OASR_EINVAL;
            }
            back *= 4;
            continue;
        }
        // decode forward from here
        int64_t at = f0;
        for (;;) {
            const int64_t f_end = f0 + blocksize;
            const int64_t lo = f0 > first ? f0 : first, hi = f_end < last ? f_end : last;
            if (hi > lo) emit(fs, nch, hi - lo, out, lo - first, lo - f0);
            at = f_end;
            if (at >= last) return OASR_OK;
            pos += len;
            const char* why = "";
            len = decode_frame(d, n, pos, si, fs, blocksize, f0, nch, &why);
            if (!len) {
                set_error("flac: %s at byte %zu", why, pos);
                return OASR_EINVAL;
            }
            if (f0 != at) {
                set_error("flac: frame at byte %zu starts at sample %lld, expected %lld", pos, (long long)f0, (long long)at);
                return OASR_EINVAL;
            }
        }
    }
}

// The range cut into equal parts for several threads (each finds its own first frame).
static int decode_range(const uint8_t* d, size_t n, const StreamInfo& si, const Sink& out, int64_t first, int64_t count, int threads) {
    const int64_t per_thread_min = 1 << 21;
    if (threads > count / per_thread_min) threads = (int)(count / per_thread_min);
    if (threads < 2) return decode_range_one(d, n, si, out, first, count);
    std::vector<int> rc(threads, OASR_OK);
    std::vector<std::string> msg(threads);     // the error text is per thread: carry it to the caller's
    auto work = [&](int k) {
        const int64_t lo = count * k / threads, hi = count * (k + 1) / threads;
        Sink part = out;
        if (part.o16) part.o16 += lo * si.channels;
        if (part.o32) part.o32 += lo * si.channels;
        rc[k] = decode_range_one(d, n, si, part, first + lo, hi - lo);
        if (rc[k] != OASR_OK) msg[k] = oasr_last_error();
    };
    std::vector<std::thread> pool;
    int started = 1;
    try {
        for (int k = 1; k < threads; ++k, ++started) pool.emplace_back(work, k);
    } catch (...) {
    }
    work(0);
    for (auto& t : pool) t.join();
    for (int k = started; k < threads; ++k) work(k);     // parts whose thread could not be started
    for (int k = 0; k < threads; ++k)
        if (rc[k] != OASR_OK) {
            set_error("%s", msg[k].c_str());
            return rc[k];
        }
    return OASR_OK;
}

}  // namespace flac
}  // namespace oasr

using namespace oasr;

extern "C" {

int oasr_flac_info(const uint8_t* data, int64_t size, int32_t* sample_rate, int32_t* channels, int32_t* bits,
                   int64_t* samples_per_channel, uint8_t* md5_16) {
    OASR_REQUIRE(data && size > 0 && sample_rate && channels && bits && samples_per_channel, "flac_info: null pointer");
    flac::StreamInfo si;
    const int rc = flac::parse_header(data, (size_t)size, si);
    if (rc != OASR_OK) return rc;
    if (si.total == 0) {   // unknown in the header (streamed encoders): count by decoding
        const int64_t n = flac::decode(data, (size_t)size, si, flac::Sink{}, 0);
        if (n < 0) return (int)n;
        si.total = n;
    }
    *sample_rate = si.rate;
    *channels = si.channels;
    *bits = si.bits;
    *samples_per_channel = si.total;
    if (md5_16) memcpy(md5_16, si.md5, 16);
    return OASR_OK;
}

static int flac_decode_to(const uint8_t* data, int64_t size, const flac::StreamInfo& si, const flac::Sink& sink,
                          int64_t capacity_per_channel, int64_t* written_per_channel) {
    // large files: several threads, each on a byte range (at least 1 MiB apiece, at most 16)
    int threads = (int)std::thread::hardware_concurrency();
    if (const char* env = getenv("OASR_FLAC_THREADS")) threads = atoi(env);
    if (threads > 16) threads = 16;
    const int64_t by_size = (size - (int64_t)si.audio_start) >> 20;
    if (threads > by_size) threads = (int)by_size;
    int64_t n = -1;
    if (threads >= 2) n = flac::decode_parallel(data, (size_t)size, si, sink, capacity_per_channel, threads);
    if (n < 0) n = flac::decode(data, (size_t)size, si, sink, capacity_per_channel);
    if (n < 0) return (int)n;
    *written_per_channel = n;
    return OASR_OK;
}

int oasr_flac_decode(const uint8_t* data, int64_t size, int32_t* out, int64_t capacity_per_channel,
                     int64_t* written_per_channel) {
    OASR_REQUIRE(data && size > 0 && out && written_per_channel, "flac_decode: null pointer");
    flac::StreamInfo si;
    const int rc = flac::parse_header(data, (size_t)size, si);
    if (rc != OASR_OK) return rc;
    flac::Sink sink;
    sink.o32 = out;
    return flac_decode_to(data, size, si, sink, capacity_per_channel, written_per_channel);
}

int oasr_flac_decode16(const uint8_t* data, int64_t size, int16_t* out, int64_t capacity_per_channel,
                       int64_t* written_per_channel) {
    OASR_REQUIRE(data && size > 0 && out && written_per_channel, "flac_decode16: null pointer");
    flac::StreamInfo si;
    const int rc = flac::parse_header(data, (size_t)size, si);
    if (rc != OASR_OK) return rc;
    OASR_REQUIRE(si.bits <= 16, "flac_decode16: the stream has %d bits per sample (at most 16 fit)", si.bits);
    flac::Sink sink;
    sink.o16 = out;
    sink.shift = 16 - si.bits;
    return flac_decode_to(data, size, si, sink, capacity_per_channel, written_per_channel);
}

int oasr_flac_decode_range(const uint8_t* data, int64_t size, int64_t first, int64_t count, void* out, int32_t out_bits) {
    OASR_REQUIRE(data && size > 0 && out, "flac_decode_range: null pointer");
    OASR_REQUIRE(out_bits == 16 || out_bits == 32, "flac_decode_range: out_bits must be 16 or 32");
    flac::StreamInfo si;
    const int rc = flac::parse_header(data, (size_t)size, si);
    if (rc != OASR_OK) return rc;
    OASR_REQUIRE(si.total > 0, "flac_decode_range: the header states no sample count (decode the whole stream instead)");
    OASR_REQUIRE(first >= 0 && count >= 0 && first + count <= si.total, "flac_decode_range: samples [%lld, %lld) lie outside the stream's %lld",
                 (long long)first, (long long)(first + count), (long long)si.total);
    OASR_REQUIRE(out_bits == 32 || si.bits <= 16, "flac_decode_range: the stream has %d bits per sample (at most 16 fit int16)", si.bits);
    // a frame header of a fixed-block-size stream holds a frame number: its sample position is that number times
    // the block size, which STREAMINFO must then state unambiguously (variable-block-size frames hold sample numbers)
    const bool variable = (size_t)size > si.audio_start + 1 && (data[si.audio_start + 1] & 1);
    OASR_REQUIRE(variable || (si.max_block > 0 && si.min_block == si.max_block),
                 "flac_decode_range: STREAMINFO states block sizes %d..%d, so frame numbers do not give sample positions",
                 si.min_block, si.max_block);
    flac::Sink sink;
    if (out_bits == 16) {
        sink.o16 = static_cast<int16_t*>(out);
        sink.shift = 16 - si.bits;
    } else {
        sink.o32 = static_cast<int32_t*>(out);
    }
    int threads = (int)std::thread::hardware_concurrency();
    if (const char* env = getenv("OASR_FLAC_THREADS")) threads = atoi(env);
    if (threads > 16) threads = 16;
    return flac::decode_range((const uint8_t*)data, (size_t)size, si, sink, first, count, threads);
}

}  // extern "C"
