// Ogg Vorbis decoder (host code): `.ogg` is one of the upload types the reference accepts
// (ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:106-118, ref:workflows/wav2elan_web/app.py:76-77;
// the reference forwards the file bytes to its remote model, here they are decoded locally).
// Implements the Ogg container (RFC 3533: pages, lacing, CRC-32) and the Vorbis I format as published by Xiph.Org:
// identification / comment / setup headers, codebooks (Huffman lengths -> codewords, lookup types 1 and 2), floor 1,
// residues 0 / 1 / 2, channel coupling, the two-size MDCT with the Vorbis window, overlap-add, granule positions.
// Floor 0 (written by no encoder since 2001) is refused.  A packet depends on nothing but the block before it, so a long
// stream is decoded by several threads on packet ranges, each starting one packet early; the samples are the same for
// any number of threads.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "oasr_common.h"

namespace oasr {
namespace vorbis {

// ---- bit reader: Vorbis packs least-significant bit first ----
struct Bits {
    const uint8_t* p;
    size_t n, pos = 0;
    uint64_t acc = 0;
    int bits = 0;
    bool eop = false;   // a read ran past the end of the packet: every later read returns 0
    Bits(const uint8_t* d, size_t len) : p(d), n(len) {}
    inline void fill() {
        while (bits <= 56 && pos < n) {
            acc |= (uint64_t)p[pos++] << bits;
            bits += 8;
        }
    }
    inline uint32_t get(int k) {   // 0 <= k <= 32
        if (k == 0) return 0;
        if (bits < k) {
            fill();
            if (bits < k) {
                eop = true;
                bits = 0;
                acc = 0;
                pos = n;
                return 0;
            }
        }
        const uint32_t v = (uint32_t)(acc & (k == 32 ? 0xFFFFFFFFull : ((1ull << k) - 1)));
        acc >>= k;
        bits -= k;
        return v;
    }
    int64_t bits_left() const { return (int64_t)(n - pos) * 8 + bits; }
};

static inline int ilog(uint32_t v) {
    int r = 0;
    while (v) {
        ++r;
        v >>= 1;
    }
    return r;
}

static float float32_unpack(uint32_t x) {
    const double mant = (double)(x & 0x1fffff);
    const int expo = (int)((x & 0x7fe00000u) >> 21);
    return (float)ldexp((x & 0x80000000u) ? -mant : mant, expo - 788);
}

// greatest r with r^dims <= entries
static int lookup1_values(int entries, int dims) {
    int r = (int)floor(exp(log((double)entries) / dims));
    auto pw = [&](int b) {
        double a = 1;
        for (int i = 0; i < dims; ++i) a *= b;
        return a;
    };
    while (pw(r + 1) <= entries) ++r;
    while (r > 0 && pw(r) > entries) --r;
    return r;
}

// ---- Ogg pages -> packets of the first Vorbis stream ----
// slice-by-8 tables of the Ogg CRC-32 (polynomial 0x04c11db7, no reflection); built once, by whichever thread comes first
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t r = i << 24;
            for (int b = 0; b < 8; ++b) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
            t[0][i] = r;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] << 8) ^ t[0][t[k - 1][i] >> 24];
    }
};
static const uint32_t (&crc_tables())[8][256] {
    static const CrcTables tables;   // thread-safe initialisation
    return tables.t;
}
static inline uint32_t crc_run(uint32_t crc, const uint8_t* p, size_t n) {
    const uint32_t (&g)[8][256] = crc_tables();
    while (n >= 8) {
        const uint32_t a = crc ^ ((uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]);
        crc = g[7][a >> 24] ^ g[6][(a >> 16) & 0xff] ^ g[5][(a >> 8) & 0xff] ^ g[4][a & 0xff] ^ g[3][p[4]] ^ g[2][p[5]] ^
              g[1][p[6]] ^ g[0][p[7]];
        p += 8;
        n -= 8;
    }
    for (size_t i = 0; i < n; ++i) crc = (crc << 8) ^ g[0][((crc >> 24) & 0xff) ^ p[i]];
    return crc;
}

struct Packet {
    size_t off;
    uint32_t len;
    int64_t granule;   // the page's granule position if this is the last packet completed on its page, else -1
    bool eos_page;
};

struct Stream {
    std::vector<uint8_t> body;   // packet bytes, contiguous per packet
    std::vector<Packet> packets;
    bool truncated = false;      // the file ends inside a page (the pages before it are used)
};

static inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

// `keep_bytes` false: only the packets' first byte and lengths are wanted (the sample count), so page bodies are neither
// checked nor copied.  `flac`: the stream looked for is the FLAC mapping's (first packet 0x7f "FLAC") instead of Vorbis.
// `start` / `end_pos`: where the walk begins and where it ended (behind the stream's last page: the next link of a chained
// file starts there); `none`: set when the walk failed only because no stream of the kind begins at or after `start`.
static bool demux(const uint8_t* d, size_t n, Stream& s, std::string& err, bool keep_bytes = true, bool flac = false,
                  size_t start = 0, size_t* end_pos = nullptr, bool* none = nullptr) {
    size_t pos = start;
    if (none) *none = false;
    if (end_pos) *end_pos = n;
    bool have = false, in_packet = false, skipping = false, have_seq = false;
    uint32_t serial = 0, next_seq = 0;
    size_t cur_off = 0;
    std::string other;
    s.body.reserve(keep_bytes ? n : std::min<size_t>(n, 1 << 20));
    while (pos + 27 <= n) {
        if (memcmp(d + pos, "OggS", 4) != 0) {
            if (have) {
                err = "ogg: page synchronisation lost";
                return false;
            }
            ++pos;   // bytes in front of the first page (a tag block) are skipped
            continue;
        }
        const int version = d[pos + 4], htype = d[pos + 5], nseg = d[pos + 26];
        if (version != 0) {
            err = "ogg: unknown stream structure version";
            return false;
        }
        if (pos + 27 + nseg > n) {
            s.truncated = true;
            break;
        }
        size_t body_len = 0;
        for (int i = 0; i < nseg; ++i) body_len += d[pos + 27 + i];
        const size_t page_len = 27 + nseg + body_len;
        if (pos + page_len > n) {
            s.truncated = true;
            break;
        }
        uint64_t granule = 0;
        for (int i = 7; i >= 0; --i) granule = (granule << 8) | d[pos + 6 + i];
        const uint32_t ser = le32(d + pos + 14), seq = le32(d + pos + 18), crc = le32(d + pos + 22);
        if (keep_bytes) {
            static const uint8_t zero4[4] = {0, 0, 0, 0};
            uint32_t c = crc_run(0, d + pos, 22);
            c = crc_run(c, zero4, 4);
            c = crc_run(c, d + pos + 26, page_len - 26);
            if (c != crc) {
                err = "ogg: page checksum mismatch";
                return false;
            }
        }
        const uint8_t* seg = d + pos + 27;
        const uint8_t* bp = seg + nseg;
        const size_t page_end = pos + page_len;
        if (!have) {
            const bool is_vorbis = (htype & 2) && body_len >= 7 && memcmp(bp, "\x01vorbis", 7) == 0;
            const bool is_flac = (htype & 2) && body_len >= 5 && memcmp(bp, "\x7f" "FLAC", 5) == 0;
            if (flac ? is_flac : is_vorbis) {
                serial = ser;
                have = true;
            } else {
                if ((htype & 2) && other.empty() && body_len >= 8) {
                    if (memcmp(bp, "OpusHead", 8) == 0) other = "Opus";
                    else if (memcmp(bp, "Speex   ", 8) == 0) other = "Speex";
                    else if (is_flac) other = "FLAC (to be unwrapped with oasr_ogg_flac_unwrap)";
                    else if (is_vorbis) other = "Vorbis";
                    else other = "another codec";
                }
                pos = page_end;
                continue;
            }
        } else if (ser != serial) {
            // streams multiplexed with this one all begin before its first data page; a stream of the same kind that
            // begins later is the next link of a chain whose last page was not marked
            const bool again = (htype & 2) && s.packets.size() > 3 &&
                               (flac ? body_len >= 5 && memcmp(bp, "\x7f" "FLAC", 5) == 0
                                     : body_len >= 7 && memcmp(bp, "\x01vorbis", 7) == 0);
            if (again) break;
            pos = page_end;   // a multiplexed stream of another kind
            continue;
        } else if (htype & 2) {
            break;   // the next link of a chain under the same serial number
        }
        if (have_seq && seq != next_seq) {
            err = "ogg: a page of the stream is missing";
            return false;
        }
        have_seq = true;
        next_seq = seq + 1;
        const bool continued = (htype & 1) != 0;
        if (continued && !in_packet) skipping = true;   // the tail of a packet whose start is not there
        if (!continued && in_packet) {                  // a packet that was never finished
            s.body.resize(cur_off);
            in_packet = false;
        }
        long last_done = -1;
        for (int i = 0; i < nseg; ++i) {
            const int l = seg[i];
            if (skipping) {
                bp += l;
                if (l < 255) skipping = false;
                continue;
            }
            if (!in_packet) {
                cur_off = s.body.size();
                in_packet = true;
            }
            size_t take = (size_t)l;
            if (!keep_bytes && !flac && s.packets.size() >= 3) {   // an audio packet's mode is in its first byte
                const size_t have = s.body.size() - cur_off;
                take = have >= 2 ? 0 : std::min<size_t>(take, 2 - have);
            }
            s.body.insert(s.body.end(), bp, bp + take);
            bp += l;
            if (l < 255) {
                s.packets.push_back(Packet{cur_off, (uint32_t)(s.body.size() - cur_off), -1, false});
                last_done = (long)s.packets.size() - 1;
                in_packet = false;
            }
        }
        if (last_done >= 0) {
            s.packets[last_done].granule = (int64_t)granule;
            s.packets[last_done].eos_page = (htype & 4) != 0;
        }
        pos = page_end;
        if (htype & 4) break;
    }
    if (end_pos) *end_pos = pos;
    if (!have) {
        if (none) *none = true;
        const std::string want = flac ? "FLAC" : "Vorbis";
        err = other.empty() ? "ogg: no " + want + " stream found" : "ogg: the file holds " + other + ", not " + want;
        return false;
    }
    if (in_packet) s.body.resize(cur_off);
    return true;
}

// ---- setup ----
struct Codebook {
    int dims = 0, entries = 0;
    std::vector<uint8_t> len;
    int lookup = 0;
    std::vector<float> vq;       // [entries][dims], lookup types 1 and 2
    static const int kFast = 10;
    std::vector<int32_t> fast_val;   // entry for codes of at most kFast bits
    std::vector<uint8_t> fast_len;   // their length; 0 = longer code (walk the tree), 255 = no such code
    std::vector<int32_t> child;      // binary tree, two slots per node: 0 absent, < 0 leaf (~entry), > 0 node
    int single = -1, single_len = 0; // a book with one used entry reads its length and returns that entry
    int used = 0;

    inline int decode(Bits& b) const {
        if (single >= 0) {
            b.get(single_len);
            return b.eop ? -1 : single;
        }
        if (b.bits < kFast) b.fill();
        const uint32_t v = (uint32_t)b.acc & ((1u << kFast) - 1);
        const int l = fast_len[v];
        if (l != 0 && l != 255) {
            if (l > b.bits) {
                b.get(32);
                b.get(32);   // runs the reader into its end-of-packet state
                return -1;
            }
            b.acc >>= l;
            b.bits -= l;
            return fast_val[v];
        }
        if (l == 255 && b.bits >= kFast) return -2;   // a bit pattern no codeword starts with
        int node = 0;
        for (;;) {
            const uint32_t bit = b.get(1);
            if (b.eop) return -1;
            const int32_t c = child[2 * node + bit];
            if (c < 0) return ~c;
            if (c == 0) return -2;
            node = c;
        }
    }
};

struct Floor1 {
    int partitions = 0;
    int part_class[32];
    int class_dim[16], class_sub[16], class_master[16], sub_book[16][8];
    int multiplier = 1, rangebits = 0, values = 0;
    int x[65], low[65], high[65], order[65];
};

struct Residue {
    int type = 0;
    uint32_t begin = 0, end = 0, psize = 1;
    int classifications = 1, classbook = 0;
    int books[64][8];
};

struct Mapping {
    int submaps = 1, steps = 0;
    int mag[256], ang[256];
    std::vector<int> mux;
    int floor[16], residue[16];
};

struct Mode {
    int blockflag = 0, mapping = 0;
};

struct Setup {
    int channels = 0, rate = 0, bs[2] = {0, 0};
    std::vector<Codebook> books;
    std::vector<Floor1> floors;
    std::vector<Residue> residues;
    std::vector<Mapping> mappings;
    std::vector<Mode> modes;
    std::vector<float> win[2];                    // rising slope of bs[i] / 2 samples
    std::vector<float> tw_pre[2], tw_post[2];     // IMDCT twiddles (cos, sin pairs), M / 2 each
    std::vector<float> fft_tw[2];                 // FFT twiddles of size M / 2
    std::vector<uint16_t> rev[2];                 // bit reversal of the FFT
    float inv_db[256];
};

static bool build_codebook(Codebook& cb, std::string& err) {
    const int n = cb.entries;
    cb.used = 0;
    int last_used = -1;
    for (int i = 0; i < n; ++i)
        if (cb.len[i]) {
            ++cb.used;
            last_used = i;
        }
    cb.fast_val.assign(1u << Codebook::kFast, 0);
    cb.fast_len.assign(1u << Codebook::kFast, 255);
    cb.child.assign(2, 0);
    if (cb.used == 0) return true;   // legal as long as no packet uses it: every decode reports an invalid code
    if (cb.used == 1) {
        cb.single = last_used;
        cb.single_len = cb.len[last_used];
        return true;
    }
    uint32_t avail[33];
    memset(avail, 0, sizeof(avail));
    bool first = true;
    for (int e = 0; e < n; ++e) {
        const int L = cb.len[e];
        if (!L) continue;
        uint32_t code;
        if (first) {
            code = 0;
            for (int i = 1; i <= L; ++i) avail[i] = 1u << (32 - i);
            first = false;
        } else {
            int z = L;
            while (z > 0 && !avail[z]) --z;
            if (z == 0) {
                err = "vorbis: a codebook states more codewords than its lengths leave room for";
                return false;
            }
            code = avail[z];
            avail[z] = 0;
            for (int y = L; y > z; --y) avail[y] = code + (1u << (32 - y));
        }
        // insert: the first bit read is the codeword's most significant bit
        int node = 0;
        for (int b = 0; b < L; ++b) {
            const int bit = (code >> (31 - b)) & 1;
            if (b == L - 1) {
                cb.child[2 * node + bit] = ~e;
            } else {
                int32_t c = cb.child[2 * node + bit];
                if (c == 0) {
                    // a complete prefix code of n words has n - 1 inner nodes, and filling the tree from the left
                    // never holds more on the way: more means the lengths cannot come out even
                    if (cb.child.size() / 2 > (size_t)n) {
                        err = "vorbis: a codebook's lengths leave codewords unassigned";
                        return false;
                    }
                    c = (int32_t)(cb.child.size() / 2);
                    cb.child.push_back(0);
                    cb.child.push_back(0);
                    cb.child[2 * node + bit] = c;
                }
                node = c;
            }
        }
    }
    for (int i = 1; i <= 32; ++i)
        if (avail[i]) {
            err = "vorbis: a codebook's lengths leave codewords unassigned";
            return false;
        }
    for (uint32_t v = 0; v < (1u << Codebook::kFast); ++v) {
        int node = 0;
        for (int b = 0; b < Codebook::kFast; ++b) {
            const int32_t c = cb.child[2 * node + ((v >> b) & 1)];
            if (c < 0) {
                cb.fast_val[v] = ~c;
                cb.fast_len[v] = (uint8_t)(b + 1);
                break;
            }
            if (c == 0) break;   // stays 255
            node = c;
            if (b == Codebook::kFast - 1) cb.fast_len[v] = 0;
        }
    }
    return true;
}

#define VB_FAIL(msg)      \
    do {                  \
        err = msg;        \
        return false;     \
    } while (0)

static bool parse_codebook(Bits& b, Codebook& cb, std::string& err) {
    if (b.get(24) != 0x564342) VB_FAIL("vorbis: codebook synchronisation pattern not found");
    cb.dims = (int)b.get(16);
    cb.entries = (int)b.get(24);
    if (b.eop || cb.dims < 1 || cb.entries < 1) VB_FAIL("vorbis: codebook of no dimensions or entries");
    // (the format allows 2^24 entries; the largest books encoders write have a few thousand, and a hostile header must
    // not be able to ask for gigabytes of tables)
    if (cb.entries > (1 << 16)) VB_FAIL("vorbis: codebook beyond 65536 entries");
    cb.len.assign(cb.entries, 0);
    const int ordered = (int)b.get(1);
    if (!ordered) {
        const int sparse = (int)b.get(1);
        for (int i = 0; i < cb.entries; ++i) {
            if (sparse && !b.get(1)) continue;
            cb.len[i] = (uint8_t)(b.get(5) + 1);
        }
    } else {
        int cur = 0, L = (int)b.get(5) + 1;
        while (cur < cb.entries) {
            const int num = (int)b.get(ilog((uint32_t)(cb.entries - cur)));
            if (b.eop || cur + num > cb.entries || L > 32) VB_FAIL("vorbis: ordered codebook lengths overrun");
            for (int i = 0; i < num; ++i) cb.len[cur + i] = (uint8_t)L;
            cur += num;
            ++L;
        }
    }
    if (b.eop) VB_FAIL("vorbis: setup header ends inside a codebook");
    cb.lookup = (int)b.get(4);
    if (cb.lookup > 2) VB_FAIL("vorbis: unknown codebook lookup type");
    if (cb.lookup) {
        const float minv = float32_unpack(b.get(32)), delta = float32_unpack(b.get(32));
        const int value_bits = (int)b.get(4) + 1, seq = (int)b.get(1);
        int64_t lv = cb.lookup == 1 ? lookup1_values(cb.entries, cb.dims) : (int64_t)cb.entries * cb.dims;
        if ((int64_t)cb.entries * cb.dims > (1 << 22)) VB_FAIL("vorbis: codebook vector table beyond 2^22 values");
        std::vector<uint32_t> mult((size_t)lv);
        for (int64_t i = 0; i < lv; ++i) mult[i] = b.get(value_bits);
        if (b.eop) VB_FAIL("vorbis: setup header ends inside a codebook");
        cb.vq.assign((size_t)cb.entries * cb.dims, 0.f);
        for (int e = 0; e < cb.entries; ++e) {
            if (!cb.len[e]) continue;
            float last = 0.f;
            int64_t div = 1;
            for (int i = 0; i < cb.dims; ++i) {
                const int64_t off = cb.lookup == 1 ? (e / div) % lv : (int64_t)e * cb.dims + i;
                const float v = (float)mult[off] * delta + minv + last;
                cb.vq[(size_t)e * cb.dims + i] = v;
                if (seq) last = v;
                if (cb.lookup == 1) div *= lv;
            }
        }
    }
    return build_codebook(cb, err);
}

static bool parse_floor1(Bits& b, Floor1& f, int n_books, std::string& err) {
    f.partitions = (int)b.get(5);
    int max_class = -1;
    for (int i = 0; i < f.partitions; ++i) {
        f.part_class[i] = (int)b.get(4);
        max_class = std::max(max_class, f.part_class[i]);
    }
    for (int c = 0; c <= max_class; ++c) {
        f.class_dim[c] = (int)b.get(3) + 1;
        f.class_sub[c] = (int)b.get(2);
        f.class_master[c] = 0;
        if (f.class_sub[c]) {
            f.class_master[c] = (int)b.get(8);
            if (f.class_master[c] >= n_books) VB_FAIL("vorbis: floor names a codebook that does not exist");
        }
        for (int j = 0; j < (1 << f.class_sub[c]); ++j) {
            f.sub_book[c][j] = (int)b.get(8) - 1;
            if (f.sub_book[c][j] >= n_books) VB_FAIL("vorbis: floor names a codebook that does not exist");
        }
    }
    f.multiplier = (int)b.get(2) + 1;
    f.rangebits = (int)b.get(4);
    f.x[0] = 0;
    f.x[1] = 1 << f.rangebits;
    f.values = 2;
    for (int i = 0; i < f.partitions; ++i) {
        const int c = f.part_class[i];
        for (int j = 0; j < f.class_dim[c]; ++j) {
            if (f.values >= 65) VB_FAIL("vorbis: floor with more than 65 posts");
            f.x[f.values++] = (int)b.get(f.rangebits);
        }
    }
    if (b.eop) VB_FAIL("vorbis: setup header ends inside a floor");
    for (int i = 0; i < f.values; ++i) f.order[i] = i;
    std::sort(f.order, f.order + f.values, [&](int a, int c) { return f.x[a] < f.x[c]; });
    for (int i = 1; i < f.values; ++i)
        if (f.x[f.order[i]] == f.x[f.order[i - 1]]) VB_FAIL("vorbis: floor posts at equal positions");
    for (int i = 2; i < f.values; ++i) {
        int lo = 0, hi = 1;   // position of the nearest lower / higher post among those before i
        for (int j = 0; j < i; ++j) {
            if (f.x[j] < f.x[i] && f.x[j] > f.x[lo]) lo = j;
            if (f.x[j] > f.x[i] && f.x[j] < f.x[hi]) hi = j;
        }
        f.low[i] = lo;
        f.high[i] = hi;
    }
    return true;
}

static bool parse_residue(Bits& b, Residue& r, int type, const std::vector<Codebook>& books, std::string& err) {
    r.type = type;
    r.begin = b.get(24);
    r.end = b.get(24);
    r.psize = b.get(24) + 1;
    r.classifications = (int)b.get(6) + 1;
    r.classbook = (int)b.get(8);
    if (r.classbook >= (int)books.size()) VB_FAIL("vorbis: residue names a codebook that does not exist");
    int cascade[64];
    for (int i = 0; i < r.classifications; ++i) {
        int high = 0;
        const int low = (int)b.get(3);
        if (b.get(1)) high = (int)b.get(5);
        cascade[i] = high * 8 + low;
    }
    for (int i = 0; i < r.classifications; ++i)
        for (int j = 0; j < 8; ++j) {
            r.books[i][j] = -1;
            if (cascade[i] & (1 << j)) {
                r.books[i][j] = (int)b.get(8);
                if (r.books[i][j] >= (int)books.size()) VB_FAIL("vorbis: residue names a codebook that does not exist");
                if (!books[r.books[i][j]].lookup) VB_FAIL("vorbis: residue codebook without a value table");
            }
        }
    if (b.eop) VB_FAIL("vorbis: setup header ends inside a residue");
    if (r.end < r.begin) VB_FAIL("vorbis: residue ends before it begins");
    return true;
}

static void make_tables(Setup& S) {
    for (int w = 0; w < 2; ++w) {
        const int n = S.bs[w], M = n / 2, Q = M / 2;
        S.win[w].resize(M);
        for (int i = 0; i < M; ++i) {
            const double s = sin((i + 0.5) / M * (M_PI / 2));
            S.win[w][i] = (float)sin(M_PI / 2 * s * s);
        }
        S.tw_pre[w].resize(2 * Q);
        S.tw_post[w].resize(2 * Q);
        for (int j = 0; j < Q; ++j) {
            S.tw_pre[w][2 * j] = (float)cos(-M_PI * (j + 0.25) / M);
            S.tw_pre[w][2 * j + 1] = (float)sin(-M_PI * (j + 0.25) / M);
            S.tw_post[w][2 * j] = (float)cos(-M_PI * j / M);
            S.tw_post[w][2 * j + 1] = (float)sin(-M_PI * j / M);
        }
        S.fft_tw[w].resize(Q);   // (cos, sin) of -2 pi j / Q for j < Q / 2
        for (int j = 0; j < Q / 2; ++j) {
            S.fft_tw[w][2 * j] = (float)cos(-2 * M_PI * j / Q);
            S.fft_tw[w][2 * j + 1] = (float)sin(-2 * M_PI * j / Q);
        }
        S.rev[w].resize(Q);
        int lg = 0;
        while ((1 << lg) < Q) ++lg;
        for (int i = 0; i < Q; ++i) {
            int r = 0;
            for (int k = 0; k < lg; ++k)
                if (i & (1 << k)) r |= 1 << (lg - 1 - k);
            S.rev[w][i] = (uint16_t)r;
        }
    }
    // the format's table of 256 floor amplitudes is a geometric series: entry i is (i - 255) * 35 / 64 dB
    for (int i = 0; i < 256; ++i) S.inv_db[i] = (float)pow(10.0, (i - 255) * 0.02734375);
}

static bool parse_headers(const Stream& st, Setup& S, std::string& err) {
    if (st.packets.size() < 3) VB_FAIL("vorbis: the three header packets are not all there");
    {
        const Packet& p = st.packets[0];
        if (p.len < 30 || memcmp(st.body.data() + p.off, "\x01vorbis", 7) != 0) VB_FAIL("vorbis: identification header not found");
        Bits b(st.body.data() + p.off + 7, p.len - 7);
        if (b.get(32) != 0) VB_FAIL("vorbis: unknown format version");
        S.channels = (int)b.get(8);
        S.rate = (int)b.get(32);
        b.get(32);
        b.get(32);
        b.get(32);
        const int e0 = (int)b.get(4), e1 = (int)b.get(4);
        const int framing = (int)b.get(1);
        if (b.eop || !framing) VB_FAIL("vorbis: identification header cut short");
        if (S.channels < 1 || S.rate < 1) VB_FAIL("vorbis: no channels or no sampling rate");
        if (S.channels > 32) VB_FAIL("vorbis: more than 32 channels");
        if (e0 < 6 || e1 > 13 || e0 > e1) VB_FAIL("vorbis: block sizes outside 64 ... 8192 or out of order");
        S.bs[0] = 1 << e0;
        S.bs[1] = 1 << e1;
    }
    {
        const Packet& p = st.packets[1];
        if (p.len < 7 || memcmp(st.body.data() + p.off, "\x03vorbis", 7) != 0) VB_FAIL("vorbis: comment header not found");
    }
    const Packet& p = st.packets[2];
    if (p.len < 7 || memcmp(st.body.data() + p.off, "\x05vorbis", 7) != 0) VB_FAIL("vorbis: setup header not found");
    Bits b(st.body.data() + p.off + 7, p.len - 7);
    const int n_books = (int)b.get(8) + 1;
    S.books.resize(n_books);
    int64_t table_values = 0;
    for (int i = 0; i < n_books; ++i) {
        if (!parse_codebook(b, S.books[i], err)) return false;
        table_values += (int64_t)S.books[i].vq.size();
        if (table_values > (1 << 24)) VB_FAIL("vorbis: the codebooks' vector tables hold more than 2^24 values");
    }
    const int n_time = (int)b.get(6) + 1;
    for (int i = 0; i < n_time; ++i)
        if (b.get(16) != 0) VB_FAIL("vorbis: time-domain transform entry not zero");
    const int n_floors = (int)b.get(6) + 1;
    S.floors.resize(n_floors);
    for (int i = 0; i < n_floors; ++i) {
        const int type = (int)b.get(16);
        if (b.eop) VB_FAIL("vorbis: setup header cut short");
        if (type == 0) VB_FAIL("vorbis: floor type 0 (streams of encoders before 2001) is not supported");
        if (type != 1) VB_FAIL("vorbis: unknown floor type");
        if (!parse_floor1(b, S.floors[i], n_books, err)) return false;
    }
    const int n_res = (int)b.get(6) + 1;
    S.residues.resize(n_res);
    for (int i = 0; i < n_res; ++i) {
        const int type = (int)b.get(16);
        if (b.eop || type > 2) VB_FAIL("vorbis: unknown residue type");
        if (!parse_residue(b, S.residues[i], type, S.books, err)) return false;
    }
    const int n_map = (int)b.get(6) + 1;
    S.mappings.resize(n_map);
    for (int i = 0; i < n_map; ++i) {
        Mapping& m = S.mappings[i];
        if (b.get(16) != 0) VB_FAIL("vorbis: unknown mapping type");
        m.submaps = b.get(1) ? (int)b.get(4) + 1 : 1;
        m.steps = 0;
        if (b.get(1)) {
            m.steps = (int)b.get(8) + 1;
            const int fb = ilog((uint32_t)(S.channels - 1));
            for (int j = 0; j < m.steps; ++j) {
                m.mag[j] = (int)b.get(fb);
                m.ang[j] = (int)b.get(fb);
                if (m.mag[j] == m.ang[j] || m.mag[j] >= S.channels || m.ang[j] >= S.channels)
                    VB_FAIL("vorbis: channel coupling names a channel that does not exist");
            }
        }
        if (b.get(2) != 0) VB_FAIL("vorbis: reserved mapping field not zero");
        m.mux.assign(S.channels, 0);
        if (m.submaps > 1)
            for (int c = 0; c < S.channels; ++c) {
                m.mux[c] = (int)b.get(4);
                if (m.mux[c] >= m.submaps) VB_FAIL("vorbis: channel assigned to a submap that does not exist");
            }
        for (int j = 0; j < m.submaps; ++j) {
            b.get(8);
            m.floor[j] = (int)b.get(8);
            m.residue[j] = (int)b.get(8);
            if (m.floor[j] >= n_floors || m.residue[j] >= n_res) VB_FAIL("vorbis: mapping names a floor or residue that does not exist");
        }
    }
    const int n_modes = (int)b.get(6) + 1;
    S.modes.resize(n_modes);
    for (int i = 0; i < n_modes; ++i) {
        S.modes[i].blockflag = (int)b.get(1);
        const int wt = (int)b.get(16), tt = (int)b.get(16);
        S.modes[i].mapping = (int)b.get(8);
        if (wt != 0 || tt != 0) VB_FAIL("vorbis: unknown window or transform type");
        if (S.modes[i].mapping >= n_map) VB_FAIL("vorbis: mode names a mapping that does not exist");
    }
    if (!b.get(1) || b.eop) VB_FAIL("vorbis: setup header cut short");
    make_tables(S);
    return true;
}

// ---- one audio packet -> one windowed block ----
struct Work {
    std::vector<std::vector<float>> spec;    // per channel, bs1 / 2
    std::vector<std::vector<float>> block;   // per channel, bs1: windowed IMDCT output
    std::vector<float> r2;                   // residue type 2: the interleaved vector
    std::vector<float> u, fre, fim;
    std::vector<int> classif;
    std::vector<int> final_y;                // [channels][65]
    std::vector<uint8_t> step2, unused;
    void init(const Setup& S) {
        spec.assign(S.channels, std::vector<float>(S.bs[1] / 2));
        block.assign(S.channels, std::vector<float>(S.bs[1]));
        r2.resize((size_t)S.channels * S.bs[1] / 2);
        u.resize(S.bs[1] / 2);
        fre.resize(S.bs[1] / 4);
        fim.resize(S.bs[1] / 4);
        final_y.resize((size_t)S.channels * 65);
        step2.resize((size_t)S.channels * 65);
        unused.resize(S.channels);
    }
};

static inline int render_point(int x0, int y0, int x1, int y1, int X) {
    const int dy = y1 - y0, adx = x1 - x0, ady = dy < 0 ? -dy : dy;
    const int off = ady * (X - x0) / adx;
    return dy < 0 ? y0 - off : y0 + off;
}

static inline void render_line(int x0, int y0, int x1, int y1, float* v, int n, const float* tab) {
    const int dy = y1 - y0, adx = x1 - x0;
    int ady = dy < 0 ? -dy : dy;
    const int base = dy / adx, sy = dy < 0 ? base - 1 : base + 1;
    int x = x0, y = y0, e = 0;
    ady -= (base < 0 ? -base : base) * adx;
    const int lim = x1 < n ? x1 : n;
    if (x < lim) v[x] *= tab[y & 255];
    for (x = x0 + 1; x < lim; ++x) {
        e += ady;
        if (e >= adx) {
            e -= adx;
            y += sy;
        } else {
            y += base;
        }
        v[x] *= tab[y & 255];
    }
}

// floor 1: the packet part (posts) -> final_y / step2; false = channel unused in this block
static bool floor1_posts(const Setup& S, const Floor1& f, Bits& b, int* final_y, uint8_t* step2) {
    if (!b.get(1)) return false;
    static const int ranges[4] = {256, 128, 86, 64};
    const int range = ranges[f.multiplier - 1], yb = ilog((uint32_t)(range - 1));
    int Y[65];
    Y[0] = (int)b.get(yb);
    Y[1] = (int)b.get(yb);
    int off = 2;
    for (int i = 0; i < f.partitions; ++i) {
        const int c = f.part_class[i], cdim = f.class_dim[c], cbits = f.class_sub[c], csub = (1 << cbits) - 1;
        int cval = 0;
        if (cbits) {
            cval = S.books[f.class_master[c]].decode(b);
            if (cval < 0) return false;
        }
        for (int j = 0; j < cdim; ++j) {
            const int book = f.sub_book[c][cval & csub];
            cval >>= cbits;
            int y = 0;
            if (book >= 0) {
                y = S.books[book].decode(b);
                if (y < 0) return false;
            }
            Y[off + j] = y;
        }
        off += cdim;
    }
    if (b.eop) return false;
    step2[0] = step2[1] = 1;
    final_y[0] = Y[0];
    final_y[1] = Y[1];
    for (int i = 2; i < f.values; ++i) {
        const int lo = f.low[i], hi = f.high[i];
        const int pred = render_point(f.x[lo], final_y[lo], f.x[hi], final_y[hi], f.x[i]);
        const int val = Y[i], highroom = range - pred, lowroom = pred;
        const int room = (highroom < lowroom ? highroom : lowroom) * 2;
        if (val) {
            step2[lo] = step2[hi] = step2[i] = 1;
            if (val >= room)
                final_y[i] = highroom > lowroom ? val - lowroom + pred : pred - val + highroom - 1;
            else
                final_y[i] = (val & 1) ? pred - (val + 1) / 2 : pred + val / 2;
        } else {
            step2[i] = 0;
            final_y[i] = pred;
        }
        final_y[i] = final_y[i] < 0 ? 0 : (final_y[i] > range - 1 ? range - 1 : final_y[i]);
    }
    return true;
}

static void floor1_apply(const Setup& S, const Floor1& f, const int* final_y, const uint8_t* step2, float* v, int n) {
    int lx = 0, ly = final_y[f.order[0]] * f.multiplier, hx = 0, hy = ly;
    for (int k = 1; k < f.values; ++k) {
        const int i = f.order[k];
        if (!step2[i]) continue;
        hy = final_y[i] * f.multiplier;
        hx = f.x[i];
        render_line(lx, ly, hx, hy, v, n, S.inv_db);
        lx = hx;
        ly = hy;
        if (lx >= n) break;
    }
    if (lx < n) {
        const float a = S.inv_db[ly & 255];
        for (int x = lx; x < n; ++x) v[x] *= a;
    }
}

// residue of `nch` vectors of `n` values each (type 2: called with the one interleaved vector)
static void residue_decode(const Setup& S, const Residue& r, Bits& b, float** vec, const uint8_t* skip, int nch, int n,
                           int format, std::vector<int>& classif) {
    bool any = false;
    for (int j = 0; j < nch; ++j) any = any || !skip[j];
    if (!any) return;   // nothing is coded for these channels: no loop over partitions either
    const uint32_t lb = std::min<uint32_t>(r.begin, (uint32_t)n), le = std::min<uint32_t>(r.end, (uint32_t)n);
    const Codebook& cbk = S.books[r.classbook];
    const int cw = cbk.dims;
    const int to_read = (int)(le - lb), parts = to_read / (int)r.psize;
    if (parts <= 0) return;
    const int stride = parts + cw;
    classif.assign((size_t)nch * stride, 0);
    for (int pass = 0; pass < 8; ++pass) {
        int pc = 0;
        while (pc < parts) {
            if (pass == 0) {
                for (int j = 0; j < nch; ++j) {
                    if (skip[j]) continue;
                    int t = cbk.decode(b);
                    if (t < 0) return;
                    for (int i = cw - 1; i >= 0; --i) {
                        classif[(size_t)j * stride + pc + i] = t % r.classifications;
                        t /= r.classifications;
                    }
                }
            }
            for (int i = 0; i < cw && pc < parts; ++i, ++pc) {
                for (int j = 0; j < nch; ++j) {
                    if (skip[j]) continue;
                    const int book = r.books[classif[(size_t)j * stride + pc]][pass];
                    if (book < 0) continue;
                    const Codebook& vb = S.books[book];
                    float* out = vec[j] + lb + (size_t)pc * r.psize;
                    const int dims = vb.dims, ps = (int)r.psize;
                    if (format == 0) {
                        const int step = ps / dims;
                        for (int k = 0; k < step; ++k) {
                            const int e = vb.decode(b);
                            if (e < 0) return;
                            const float* q = vb.vq.data() + (size_t)e * dims;
                            for (int d = 0; d < dims; ++d) out[k + d * step] += q[d];
                        }
                    } else {
                        int k = 0;
                        while (k < ps) {
                            const int e = vb.decode(b);
                            if (e < 0) return;
                            const float* q = vb.vq.data() + (size_t)e * dims;
                            for (int d = 0; d < dims && k < ps; ++d) out[k++] += q[d];
                        }
                    }
                }
            }
        }
    }
}

// y[0 .. 2M) = sum_k X[k] cos(pi / M (n + 1/2 + M/2)(k + 1/2)): a DCT-IV of size M (one complex FFT of size M / 2
// between two twiddle passes), unfolded by the transform's symmetries
static void imdct(const Setup& S, int w, const float* X, float* y, Work& W) {
    const int M = S.bs[w] / 2, Q = M / 2;
    float* re = W.fre.data();
    float* im = W.fim.data();
    const float* pre = S.tw_pre[w].data();
    const uint16_t* rev = S.rev[w].data();
    for (int j = 0; j < Q; ++j) {
        const float a = X[2 * j], c = X[M - 1 - 2 * j], wr = pre[2 * j], wi = pre[2 * j + 1];
        const int t = rev[j];
        re[t] = a * wr - c * wi;
        im[t] = a * wi + c * wr;
    }
    const float* tw = S.fft_tw[w].data();
    for (int len = 2; len <= Q; len <<= 1) {
        const int half = len >> 1, step = Q / len;
        for (int s = 0; s < Q; s += len)
            for (int k = 0; k < half; ++k) {
                const float wr = tw[2 * k * step], wi = tw[2 * k * step + 1];
                const int i0 = s + k, i1 = i0 + half;
                const float xr = re[i1] * wr - im[i1] * wi, xi = re[i1] * wi + im[i1] * wr;
                re[i1] = re[i0] - xr;
                im[i1] = im[i0] - xi;
                re[i0] += xr;
                im[i0] += xi;
            }
    }
    float* u = W.u.data();
    const float* post = S.tw_post[w].data();
    for (int k = 0; k < Q; ++k) {
        const float wr = post[2 * k], wi = post[2 * k + 1];
        u[2 * k] = re[k] * wr - im[k] * wi;
        u[M - 1 - 2 * k] = -(re[k] * wi + im[k] * wr);
    }
    for (int i = 0; i < M / 2; ++i) y[i] = u[M / 2 + i];
    for (int i = 0; i < M; ++i) y[M / 2 + i] = -u[M - 1 - i];
    for (int i = 0; i < M / 2; ++i) y[3 * M / 2 + i] = -u[i];
}

// the mode of an audio packet, or -1 if the packet is not audio / names no mode
static int packet_mode(const Setup& S, const uint8_t* p, size_t len) {
    Bits b(p, len);
    if (b.get(1) != 0 || b.eop) return -1;
    const int mode = (int)b.get(ilog((uint32_t)(S.modes.size() - 1)));
    if (b.eop || mode >= (int)S.modes.size()) return -1;
    return mode;
}

// decodes one audio packet into W.block (windowed, n samples per channel); a damaged packet gives a block of silence
static void decode_packet(const Setup& S, const uint8_t* p, size_t len, Work& W, int64_t* tail_bits) {
    Bits b(p, len);
    b.get(1);
    const int mode_i = (int)b.get(ilog((uint32_t)(S.modes.size() - 1)));
    const Mode& mode = S.modes[mode_i];
    const Mapping& map = S.mappings[mode.mapping];
    const int w = mode.blockflag, n = S.bs[w], M = n / 2, ch = S.channels;
    int prev_long = 1, next_long = 1;
    if (w) {
        prev_long = (int)b.get(1);
        next_long = (int)b.get(1);
    }
    // floors (posts)
    for (int c = 0; c < ch; ++c) {
        const Floor1& f = S.floors[map.floor[map.mux[c]]];
        W.unused[c] = floor1_posts(S, f, b, &W.final_y[(size_t)c * 65], &W.step2[(size_t)c * 65]) ? 0 : 1;
    }
    uint8_t no_res[256];
    for (int c = 0; c < ch; ++c) no_res[c] = W.unused[c];
    for (int j = 0; j < map.steps; ++j)
        if (!no_res[map.mag[j]] || !no_res[map.ang[j]]) no_res[map.mag[j]] = no_res[map.ang[j]] = 0;
    // residues per submap
    for (int c = 0; c < ch; ++c) memset(W.spec[c].data(), 0, sizeof(float) * M);
    for (int sm = 0; sm < map.submaps; ++sm) {
        float* vec[256];
        uint8_t skip[256];
        int nch = 0;
        for (int c = 0; c < ch; ++c)
            if (map.mux[c] == sm) {
                vec[nch] = W.spec[c].data();
                skip[nch] = no_res[c];
                ++nch;
            }
        if (!nch) continue;
        const Residue& r = S.residues[map.residue[sm]];
        if (r.type == 2) {
            bool any = false;
            for (int j = 0; j < nch; ++j) any = any || !skip[j];
            if (!any) continue;
            float* big = W.r2.data();
            memset(big, 0, sizeof(float) * (size_t)nch * M);
            uint8_t s0 = 0;
            residue_decode(S, r, b, &big, &s0, 1, nch * M, 1, W.classif);
            for (int j = 0; j < nch; ++j) {
                float* o = vec[j];
                for (int i = 0; i < M; ++i) o[i] = big[(size_t)i * nch + j];
            }
        } else {
            residue_decode(S, r, b, vec, skip, nch, M, r.type, W.classif);
        }
    }
    if (tail_bits) *tail_bits = b.eop ? -1 : b.bits_left();
    // inverse coupling, last step first
    for (int j = map.steps - 1; j >= 0; --j) {
        float* m = W.spec[map.mag[j]].data();
        float* a = W.spec[map.ang[j]].data();
        for (int i = 0; i < M; ++i) {
            const float mv = m[i], av = a[i];
            float nm, na;
            if (mv > 0) {
                if (av > 0) { nm = mv; na = mv - av; } else { na = mv; nm = mv + av; }
            } else {
                if (av > 0) { nm = mv; na = mv + av; } else { na = mv; nm = mv - av; }
            }
            m[i] = nm;
            a[i] = na;
        }
    }
    // window geometry
    const int b0 = S.bs[0];
    int ls = 0, ln = M, rs = M, rn = M, lw = w, rw = w;
    if (w && !prev_long) { ls = n / 4 - b0 / 4; ln = b0 / 2; lw = 0; }
    if (w && !next_long) { rs = n * 3 / 4 - b0 / 4; rn = b0 / 2; rw = 0; }
    for (int c = 0; c < ch; ++c) {
        float* out = W.block[c].data();
        if (W.unused[c]) {
            memset(out, 0, sizeof(float) * n);
            continue;
        }
        const Floor1& f = S.floors[map.floor[map.mux[c]]];
        floor1_apply(S, f, &W.final_y[(size_t)c * 65], &W.step2[(size_t)c * 65], W.spec[c].data(), M);
        imdct(S, w, W.spec[c].data(), out, W);
        const float* wl = S.win[lw].data();
        const float* wr = S.win[rw].data();
        for (int i = 0; i < ls; ++i) out[i] = 0.f;
        for (int i = 0; i < ln; ++i) out[ls + i] *= wl[i];
        for (int i = 0; i < rn; ++i) out[rs + i] *= wr[rn - 1 - i];
        for (int i = rs + rn; i < n; ++i) out[i] = 0.f;
    }
}

// ---- a stream opened: headers parsed, audio packets placed on the sample axis ----
struct Plan {
    Stream st;
    Setup S;
    std::vector<int> audio;        // indices into st.packets
    std::vector<int64_t> centre;   // centre of block i on the axis whose 0 is the start of block 0
    int64_t skip = 0, total = 0;   // samples dropped at the start, samples delivered
};

static bool open_stream(const uint8_t* d, size_t n, Plan& P, std::string& err, bool for_info = false, size_t start = 0,
                        size_t* end_pos = nullptr, bool* none = nullptr) {
    if (!demux(d, n, P.st, err, !for_info, false, start, end_pos, none)) return false;
    if (!parse_headers(P.st, P.S, err)) return false;
    const Setup& S = P.S;
    int64_t c = 0;
    int prev_n = 0;
    long first_granule_packet = -1;
    int64_t last_granule = -1;
    for (size_t i = 3; i < P.st.packets.size(); ++i) {
        const Packet& pk = P.st.packets[i];
        const int mode = packet_mode(S, P.st.body.data() + pk.off, pk.len);
        if (mode >= 0) {
            const int bn = S.bs[S.modes[mode].blockflag];
            c = P.audio.empty() ? bn / 2 : c + prev_n / 4 + bn / 4;
            prev_n = bn;
            P.audio.push_back((int)i);
            P.centre.push_back(c);
        }
        if (pk.granule >= 0 && !P.audio.empty()) {
            if (first_granule_packet < 0) first_granule_packet = (long)P.audio.size() - 1;
            last_granule = pk.granule;
        }
    }
    if (P.audio.empty()) {
        P.total = 0;
        return true;
    }
    const int64_t c0 = P.centre[0], decoded = P.centre.back() - c0;
    int64_t base = 0;   // position on the stream's own sample axis of the first sample the blocks produce
    if (first_granule_packet >= 0) {
        const Packet& fp = P.st.packets[P.audio[first_granule_packet]];
        // the first page's granule position may state fewer samples than its packets produce: the excess is cut from
        // the start - unless that page is also the last one (marked as such, or simply the last there is), where the
        // cut is at the end
        const bool last_page = fp.eos_page || (size_t)P.audio[first_granule_packet] + 1 == P.st.packets.size();
        if (!last_page && fp.granule >= 0) base = fp.granule - (P.centre[first_granule_packet] - c0);
    }
    P.skip = base < 0 ? -base : 0;
    if (P.skip > decoded) P.skip = decoded;
    int64_t end = decoded;
    if (last_granule >= 0) end = std::min<int64_t>(decoded, last_granule - base);
    if (end < P.skip) end = P.skip;
    P.total = end - P.skip;
    return true;
}

// packets [a, b) of P.audio (a >= 1) -> their output samples, written where they belong in `out`
// (only the samples lo <= t < hi of the delivered signal, out[0] being sample lo)
template <bool MONO>
static void decode_range(const Plan& P, size_t a, size_t b, float* out, int64_t lo, int64_t hi) {
    const Setup& S = P.S;
    const int ch = S.channels;
    Work W;
    W.init(S);
    std::vector<std::vector<float>> prev(ch, std::vector<float>(S.bs[1] / 2, 0.f));
    int prev_n = 0;
    const int64_t c0 = P.centre[0];
    for (size_t i = a - 1; i < b; ++i) {
        const Packet& pk = P.st.packets[P.audio[i]];
        const int mode = packet_mode(S, P.st.body.data() + pk.off, pk.len);
        const int n = S.bs[S.modes[mode].blockflag];
        decode_packet(S, P.st.body.data() + pk.off, pk.len, W, nullptr);
        if (i >= a) {
            const int L = prev_n / 4 + n / 4;
            const int64_t t0 = P.centre[i - 1] - c0 - P.skip;   // output index of the first finished sample
            const int shift = n / 2 - L;
            for (int s = 0; s < L; ++s) {
                int64_t t = t0 + s;
                if (t < lo || t >= hi) continue;
                t -= lo;
                const int ci = s + shift;
                if (MONO) {
                    double acc = 0;
                    for (int c = 0; c < ch; ++c) {
                        float v = s < prev_n / 2 ? prev[c][s] : 0.f;
                        if (ci >= 0) v += W.block[c][ci];
                        acc += v;
                    }
                    out[t] = (float)(acc / ch);
                } else {
                    for (int c = 0; c < ch; ++c) {
                        float v = s < prev_n / 2 ? prev[c][s] : 0.f;
                        if (ci >= 0) v += W.block[c][ci];
                        out[t * ch + c] = v;
                    }
                }
            }
        }
        for (int c = 0; c < ch; ++c) memcpy(prev[c].data(), W.block[c].data() + n / 2, sizeof(float) * (n / 2));
        prev_n = n;
    }
}

// the packets whose finished samples cover [first, first + count) of the delivered signal: [a, b), a >= 1
static void packets_for(const Plan& P, int64_t first, int64_t count, size_t& a, size_t& b) {
    const int64_t c0 = P.centre[0] + P.skip;   // packet i finishes the samples [centre[i-1] - c0, centre[i] - c0)
    const auto beg = P.centre.begin();
    a = (size_t)(std::upper_bound(beg, P.centre.end(), first + c0) - beg);                 // first centre[i] - c0 > first
    b = (size_t)(std::lower_bound(beg, P.centre.end(), first + count + c0) - beg) + 1;     // through centre[i] - c0 >= end
    a = std::max<size_t>(a, 1);
    b = std::min(b, P.centre.size());
}

// packets [a, b) on up to `threads` host threads (0: as many as the machine has, at most 16; a thread per 512 packets at
// most), each range starting one packet early
template <bool MONO>
static void decode_span(const Plan& P, size_t a, size_t b, float* out, int64_t lo, int64_t hi, int threads) {
    if (a >= b) return;
    size_t T = 1;
    if (threads <= 0) {
        const unsigned hc = std::thread::hardware_concurrency();
        T = std::min<size_t>(16, hc ? hc : 1);
    } else {
        T = (size_t)threads;
    }
    T = std::max<size_t>(1, std::min<size_t>(T, (b - a) / 512 + 1));
    if (T == 1) {
        decode_range<MONO>(P, a, b, out, lo, hi);
        return;
    }
    std::vector<std::thread> pool;
    for (size_t t = 0; t < T; ++t) {
        const size_t a1 = a + (b - a) * t / T, b1 = a + (b - a) * (t + 1) / T;
        if (a1 < b1) pool.emplace_back([&P, a1, b1, out, lo, hi] { decode_range<MONO>(P, a1, b1, out, lo, hi); });
    }
    for (auto& th : pool) th.join();
}

template <bool MONO>
static void decode_all(const Plan& P, float* out, int threads) {
    const size_t K = P.audio.size();
    if (K < 2 || P.total == 0) return;
    size_t a, b;
    packets_for(P, 0, P.total, a, b);   // packets outside what the granule positions keep are not decoded at all
    decode_span<MONO>(P, a, b, out, 0, P.total, threads);
}

// ---- a chained file: logical streams one behind the other (RFC 3533 section 4), delivered as one signal ----
struct Chain {
    std::vector<std::unique_ptr<Plan>> links;
    std::vector<int64_t> first;   // position of link i's first sample in the delivered signal
    int64_t total = 0;
    int rate = 0, channels = 0;
};

static bool open_chain(const uint8_t* d, size_t n, Chain& C, std::string& err, bool for_info = false) {
    size_t pos = 0;
    while (pos < n) {
        std::unique_ptr<Plan> P(new Plan);
        size_t end = n;
        bool none = false;
        if (!open_stream(d, n, *P, err, for_info, pos, &end, &none)) {
            if (none && !C.links.empty()) break;   // whatever follows the last link is not a stream of this kind
            return false;
        }
        if (C.links.empty()) {
            C.rate = P->S.rate;
            C.channels = P->S.channels;
        } else if (P->S.rate != C.rate || P->S.channels != C.channels) {
            char msg[160];
            snprintf(msg, sizeof msg, "ogg: link %zu of the chained file is %d Hz with %d channel(s), the first one %d Hz with %d",
                     C.links.size() + 1, P->S.rate, P->S.channels, C.rate, C.channels);
            err = msg;
            return false;
        }
        C.first.push_back(C.total);
        C.total += P->total;
        C.links.push_back(std::move(P));
        if (end <= pos) break;
        pos = end;
    }
    return true;
}

// the samples [first, first + count) of the delivered signal, from every link that holds some of them
template <bool MONO>
static void decode_chain(const Chain& C, int64_t first, int64_t count, float* out, int threads) {
    const size_t width = MONO ? 1 : (size_t)C.channels;
    for (size_t i = 0; i < C.links.size(); ++i) {
        const Plan& P = *C.links[i];
        const int64_t lo = std::max<int64_t>(first, C.first[i]) - C.first[i];
        const int64_t hi = std::min<int64_t>(first + count, C.first[i] + P.total) - C.first[i];
        if (lo >= hi || P.audio.size() < 2) continue;
        size_t a, b;
        packets_for(P, lo, hi - lo, a, b);
        decode_span<MONO>(P, a, b, out + (size_t)(C.first[i] + lo - first) * width, lo, hi, threads);
    }
}

}  // namespace vorbis
}  // namespace oasr

using namespace oasr;

extern "C" {

int oasr_vorbis_info(const uint8_t* data, int64_t size, int32_t* sample_rate, int32_t* channels,
                     int64_t* samples_per_channel) {
    OASR_REQUIRE(data && size > 0 && sample_rate && channels && samples_per_channel, "vorbis_info: null pointer");
    vorbis::Chain C;
    std::string err;
    if (!vorbis::open_chain(data, (size_t)size, C, err, true)) {
        set_error("%s", err.c_str());
        return OASR_EINVAL;
    }
    *sample_rate = C.rate;
    *channels = C.channels;
    *samples_per_channel = C.total;
    return OASR_OK;
}

static int vorbis_decode_any(const uint8_t* data, int64_t size, float* out, int64_t capacity, int64_t* written,
                             int32_t threads, bool mono) {
    OASR_REQUIRE(data && size > 0 && out && written, "vorbis_decode: null pointer");
    OASR_REQUIRE(capacity >= 0, "vorbis_decode: negative capacity");
    vorbis::Chain C;
    std::string err;
    if (!vorbis::open_chain(data, (size_t)size, C, err)) {
        set_error("%s", err.c_str());
        return OASR_EINVAL;
    }
    if (C.total > capacity) {
        set_error("vorbis_decode: output holds %lld samples per channel, the stream has %lld", (long long)capacity,
                  (long long)C.total);
        return OASR_EINVAL;
    }
    if (mono)
        vorbis::decode_chain<true>(C, 0, C.total, out, threads);
    else
        vorbis::decode_chain<false>(C, 0, C.total, out, threads);
    *written = C.total;
    return OASR_OK;
}

int oasr_vorbis_decode(const uint8_t* data, int64_t size, float* out, int64_t capacity_per_channel,
                       int64_t* written_per_channel, int32_t threads) {
    return vorbis_decode_any(data, size, out, capacity_per_channel, written_per_channel, threads, false);
}

int oasr_vorbis_decode_mono(const uint8_t* data, int64_t size, float* out, int64_t capacity, int64_t* written,
                            int32_t threads) {
    return vorbis_decode_any(data, size, out, capacity, written, threads, true);
}

}  // extern "C"

struct oasr_vorbis_stream {
    vorbis::Chain chain;
};

extern "C" {

int oasr_vorbis_open(const uint8_t* data, int64_t size, oasr_vorbis_stream** stream, int32_t* sample_rate,
                     int32_t* channels, int64_t* samples_per_channel) {
    OASR_REQUIRE(data && size > 0 && stream && sample_rate && channels && samples_per_channel, "vorbis_open: null pointer");
    oasr_vorbis_stream* s = new (std::nothrow) oasr_vorbis_stream;
    if (!s) {
        set_error("vorbis_open: out of host memory");
        return OASR_EINVAL;
    }
    std::string err;
    if (!vorbis::open_chain(data, (size_t)size, s->chain, err)) {
        delete s;
        set_error("%s", err.c_str());
        return OASR_EINVAL;
    }
    *sample_rate = s->chain.rate;
    *channels = s->chain.channels;
    *samples_per_channel = s->chain.total;
    *stream = s;
    return OASR_OK;
}

int oasr_vorbis_decode_mono_range(const oasr_vorbis_stream* stream, int64_t first, int64_t count, float* out) {
    OASR_REQUIRE(stream && out, "vorbis_decode_mono_range: null pointer");
    const vorbis::Chain& C = stream->chain;
    OASR_REQUIRE(first >= 0 && count >= 0 && first + count <= C.total,
                 "vorbis_decode_mono_range: samples [%lld, %lld) of a stream of %lld", (long long)first,
                 (long long)(first + count), (long long)C.total);
    if (count == 0) return OASR_OK;
    vorbis::decode_chain<true>(C, first, count, out, 0);
    return OASR_OK;
}

void oasr_vorbis_close(oasr_vorbis_stream* stream) { delete stream; }

// FLAC in an Ogg container (the mapping of the FLAC format's Ogg annex): the first packet is 0x7f "FLAC", a mapping
// version (major 1), the number of header packets that follow, then the native signature "fLaC" and the STREAMINFO
// block; every further header packet is one metadata block; every audio packet is one FLAC frame.  The native stream is
// therefore the packets' bytes in order, the first one from its ninth byte on - which is what this writes into `out`
// (never more than `size` bytes), for oasr_flac_info / oasr_flac_decode* to read.
int oasr_ogg_flac_unwrap(const uint8_t* data, int64_t size, uint8_t* out, int64_t capacity, int64_t* written) {
    OASR_REQUIRE(data && size > 0 && out && written, "ogg_flac_unwrap: null pointer");
    vorbis::Stream st;
    std::string err;
    if (!vorbis::demux(data, (size_t)size, st, err, true, true)) {
        set_error("%s", err.c_str());
        return OASR_EINVAL;
    }
    OASR_REQUIRE(!st.packets.empty() && st.packets[0].len >= 13 + 4 + 34, "ogg_flac_unwrap: the first packet is too short");
    const uint8_t* p0 = st.body.data() + st.packets[0].off;
    OASR_REQUIRE(p0[5] == 1, "ogg_flac_unwrap: mapping version %d.%d is not known", (int)p0[5], (int)p0[6]);
    OASR_REQUIRE(memcmp(p0 + 9, "fLaC", 4) == 0 && (p0[13] & 0x7f) == 0, "ogg_flac_unwrap: no STREAMINFO in the first packet");
    int64_t need = -9;
    for (const vorbis::Packet& pk : st.packets) need += pk.len;
    OASR_REQUIRE(need <= capacity, "ogg_flac_unwrap: %lld bytes, room for %lld", (long long)need, (long long)capacity);
    // metadata blocks run until one carries the last-block flag; a packet after that which does not start with the
    // frame synchronisation code is not part of the stream
    int64_t at = 0;
    bool in_headers = true;
    for (size_t i = 0; i < st.packets.size(); ++i) {
        const uint8_t* p = st.body.data() + st.packets[i].off;
        size_t len = st.packets[i].len;
        if (i == 0) {
            p += 9;
            len -= 9;
            in_headers = (p[4] & 0x80) == 0;
        } else if (in_headers) {
            if (len < 4) {
                set_error("ogg_flac_unwrap: a header packet of %zu bytes", len);
                return OASR_EINVAL;
            }
            const size_t body = ((size_t)p[1] << 16) | ((size_t)p[2] << 8) | p[3];
            if (body + 4 != len) {
                set_error("ogg_flac_unwrap: a metadata block of %zu bytes in a packet of %zu", body + 4, len);
                return OASR_EINVAL;
            }
            in_headers = (p[0] & 0x80) == 0;
        } else if (len < 2 || p[0] != 0xff || (p[1] & 0xfc) != 0xf8) {
            set_error("ogg_flac_unwrap: audio packet %zu does not start with a frame header", i);
            return OASR_EINVAL;
        }
        memcpy(out + at, p, len);
        at += (int64_t)len;
    }
    *written = at;
    return OASR_OK;
}

// Diagnostic for tests: for every audio packet, the number of bits left unread when its residue decode finished
// (-1: the decode ran into the end of the packet).  An encoder writes no more than it needs, so on an undamaged
// stream decoded with the right code books the figure stays below 8.
int oasr_vorbis_packet_tails(const uint8_t* data, int64_t size, int32_t* tails, int64_t capacity, int64_t* packets) {
    OASR_REQUIRE(data && size > 0 && tails && packets, "vorbis_packet_tails: null pointer");
    vorbis::Chain C;
    std::string err;
    if (!vorbis::open_chain(data, (size_t)size, C, err)) {
        set_error("%s", err.c_str());
        return OASR_EINVAL;
    }
    int64_t n = 0;
    for (const auto& link : C.links) n += (int64_t)link->audio.size();
    OASR_REQUIRE(n <= capacity, "vorbis_packet_tails: %lld packets, room for %lld", (long long)n, (long long)capacity);
    n = 0;
    for (const auto& link : C.links) {
        const vorbis::Plan& P = *link;
        vorbis::Work W;
        W.init(P.S);
        for (size_t i = 0; i < P.audio.size(); ++i) {
            const vorbis::Packet& pk = P.st.packets[P.audio[i]];
            int64_t t = 0;
            vorbis::decode_packet(P.S, P.st.body.data() + pk.off, pk.len, W, &t);
            tails[n++] = (int32_t)t;
        }
    }
    *packets = n;
    return OASR_OK;
}

}  // extern "C"
