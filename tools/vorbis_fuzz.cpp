// Host-side fuzz harness for the Ogg Vorbis decoder (csrc/vorbis.cpp) under AddressSanitizer.  A valid stream is
// damaged at random and decoded with 1, 2 or 5 threads into an exact-size output, whole and by random sample ranges of
// the opened stream; any read or write outside a buffer
// aborts the run.  Page checksums would stop almost every damaged byte at the container, so most kinds of damage are
// applied to page bodies - the setup header's codebooks, floors, residues and mappings, and the audio packets - and
// the checksum of every touched page is recomputed; the rest (truncation, deleted and inserted runs, header fields
// of pages) exercises the container code itself.
// Build and run (host code only; no GPU is touched):
//   hipcc -O1 -g -std=c++17 -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
//         -Xarch_host -fno-sanitize-recover=undefined -Iinclude -Iomnilingual-asr_amd/csrc -x hip \
//         --offload-arch=gfx950 -c omnilingual-asr_amd/csrc/vorbis.cpp -o /tmp/vorbis_asan.o
//   /opt/rocm/lib/llvm/bin/clang++ -O1 -g -fsanitize=address,undefined tools/vorbis_fuzz.cpp /tmp/vorbis_asan.o \
//         -o /tmp/vorbis_fuzz -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread
//   ASAN_OPTIONS=detect_leaks=0 /tmp/vorbis_fuzz tests/golden/mathjax_invalid_keypress.ogg 2000
// (a FLAC-in-Ogg seed - tests/flac_tools.py: wrap_in_ogg - exercises oasr_ogg_flac_unwrap the same way)
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <random>
#include <vector>
namespace oasr { void set_error(const char*, ...) {} }   // the library's error text is not needed here
extern "C" int oasr_vorbis_info(const uint8_t*, int64_t, int32_t*, int32_t*, int64_t*);
extern "C" int oasr_vorbis_decode(const uint8_t*, int64_t, float*, int64_t, int64_t*, int32_t);
extern "C" int oasr_vorbis_decode_mono(const uint8_t*, int64_t, float*, int64_t, int64_t*, int32_t);
struct oasr_vorbis_stream;
extern "C" int oasr_vorbis_open(const uint8_t*, int64_t, oasr_vorbis_stream**, int32_t*, int32_t*, int64_t*);
extern "C" int oasr_vorbis_decode_mono_range(const oasr_vorbis_stream*, int64_t, int64_t, float*);
extern "C" void oasr_vorbis_close(oasr_vorbis_stream*);
extern "C" int oasr_ogg_flac_unwrap(const uint8_t*, int64_t, uint8_t*, int64_t, int64_t*);

static uint32_t crc_tab[256];
static uint32_t page_crc(const uint8_t* p, size_t n) {
    uint32_t c = 0;
    for (size_t i = 0; i < n; ++i) c = (c << 8) ^ crc_tab[((c >> 24) & 0xff) ^ (i >= 22 && i < 26 ? 0 : p[i])];
    return c;
}
struct Page { size_t at, head, len; };
static std::vector<Page> pages_of(const std::vector<uint8_t>& b) {
    std::vector<Page> v;
    size_t pos = 0;
    while (pos + 27 <= b.size() && memcmp(&b[pos], "OggS", 4) == 0) {
        const int ns = b[pos + 26];
        if (pos + 27 + ns > b.size()) break;
        size_t body = 0;
        for (int i = 0; i < ns; ++i) body += b[pos + 27 + i];
        if (pos + 27 + ns + body > b.size()) break;
        v.push_back(Page{pos, (size_t)27 + ns, 27 + ns + body});
        pos += 27 + ns + body;
    }
    return v;
}
static void fix_crc(std::vector<uint8_t>& b, const Page& p) {
    const uint32_t c = page_crc(&b[p.at], p.len);
    b[p.at + 22] = c & 0xff; b[p.at + 23] = (c >> 8) & 0xff; b[p.at + 24] = (c >> 16) & 0xff; b[p.at + 25] = (c >> 24) & 0xff;
}

int main(int argc, char** argv) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t r = i << 24;
        for (int k = 0; k < 8; ++k) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
        crc_tab[i] = r;
    }
    FILE* f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> d(n);
    if (fread(d.data(), 1, n, f) != (size_t)n) return 2;
    const int iters = atoi(argv[2]);
    std::mt19937_64 rng(argc > 3 ? atol(argv[3]) : 5);
    int ok = 0, err = 0, unwrapped = 0;
    long nonfinite = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> b(d);
        std::vector<Page> pg = pages_of(b);
        const int kind = rng() % 8;
        if (kind <= 2 && pg.size() >= 3) {            // the setup header (third page onward until audio): a few bits or bytes
            const Page& p = pg[2];
            const int k = 1 + rng() % (kind == 0 ? 2 : 10);
            for (int i = 0; i < k; i++) {
                const size_t at = p.at + p.head + rng() % (p.len - p.head);
                if (kind == 2) b[at] = (uint8_t)rng(); else b[at] ^= (uint8_t)(1u << (rng() % 8));
            }
            fix_crc(b, p);
        } else if (kind <= 4 && pg.size() >= 4) {     // audio packets
            for (size_t q = 3; q < pg.size(); ++q) {
                const Page& p = pg[q];
                const int k = 1 + rng() % 20;
                for (int i = 0; i < k; i++) {
                    const size_t at = p.at + p.head + rng() % (p.len - p.head);
                    if (kind == 3) b[at] ^= (uint8_t)(1u << (rng() % 8)); else b[at] = (uint8_t)rng();
                }
                fix_crc(b, p);
            }
        } else if (kind == 5 && !pg.empty()) {        // page header fields and lacing values, checksum made right again
            const Page& p = pg[rng() % pg.size()];
            const size_t at = p.at + 4 + rng() % (p.head - 4);
            if (at < p.at + 22 || at >= p.at + 27) b[at] ^= (uint8_t)(1u << (rng() % 8));
            std::vector<Page> again = pages_of(b);
            for (const Page& q : again) fix_crc(b, q);
        } else if (kind == 6) {
            b.resize(1 + rng() % b.size());
        } else {
            size_t i = rng() % (b.size() - 10), m = 1 + rng() % 600;
            if (i + m > b.size()) m = b.size() - i;
            if (rng() & 1) b.erase(b.begin() + i, b.begin() + i + m);
            else {
                std::vector<uint8_t> ins(m);
                for (auto& x : ins) x = (uint8_t)rng();
                b.insert(b.begin() + i, ins.begin(), ins.end());
            }
        }
        uint8_t* heap = (uint8_t*)malloc(b.size());     // exact size: an overread is an error
        memcpy(heap, b.data(), b.size());
        int32_t rate = 0, ch = 0;
        int64_t total = 0, written = 0;
        {   // the FLAC mapping's unwrap on the same bytes (a seed file of that kind gets past the first packet)
            uint8_t* plain = (uint8_t*)malloc(b.size());
            int64_t w = 0;
            if (oasr_ogg_flac_unwrap(heap, (int64_t)b.size(), plain, (int64_t)b.size(), &w) == 0) {
                ++unwrapped;
                if (w < 4 + 4 + 34 || w > (int64_t)b.size() || memcmp(plain, "fLaC", 4) != 0) abort();
            }
            free(plain);
        }
        if (oasr_vorbis_info(heap, (int64_t)b.size(), &rate, &ch, &total) != 0 || total > (1 << 24)) {
            ++err;
            free(heap);
            continue;
        }
        const int threads = (int)(1 + rng() % 3) * 2 - 1;
        const bool mono = rng() & 1;
        float* out = (float*)malloc(sizeof(float) * (size_t)(total * (mono ? 1 : ch)) + 4);
        const int rc = mono ? oasr_vorbis_decode_mono(heap, (int64_t)b.size(), out, total, &written, threads)
                            : oasr_vorbis_decode(heap, (int64_t)b.size(), out, total, &written, threads);
        if (rc == 0) {
            ++ok;
            for (int64_t i = 0; i < written * (mono ? 1 : ch); ++i)
                if (!std::isfinite(out[i])) ++nonfinite;
        } else {
            ++err;
        }
        free(out);
        // the opened-stream form: random sample ranges into exact-size outputs
        oasr_vorbis_stream* st = nullptr;
        int64_t t2 = 0;
        if (oasr_vorbis_open(heap, (int64_t)b.size(), &st, &rate, &ch, &t2) == 0) {
            for (int q = 0; q < 4 && t2 > 0 && t2 < (1 << 24); q++) {
                const int64_t first = (int64_t)(rng() % (uint64_t)t2);
                int64_t count = 1 + (int64_t)(rng() % (uint64_t)(t2 - first));
                if (rng() % 2) count = count < 3000 ? count : 1 + (int64_t)(rng() % 3000);
                float* o = (float*)malloc(sizeof(float) * (size_t)count);
                oasr_vorbis_decode_mono_range(st, first, count, o);
                for (int64_t i = 0; i < count; ++i)
                    if (!std::isfinite(o[i])) ++nonfinite;
                free(o);
            }
            oasr_vorbis_close(st);
        }
        free(heap);
    }
    printf("%d streams: %d decoded, %d refused, %ld non-finite samples; %d unwrapped as FLAC in Ogg\n", iters, ok, err,
           nonfinite, unwrapped);
    return 0;
}
