// Host-side fuzz harness for the MP3 decoder (csrc/mp3.cpp) under AddressSanitizer: a valid stream is damaged at
// random (byte flips anywhere, bit flips aimed at side information, truncation, deleted and inserted runs, header
// damage), copied into an exact-size heap block and decoded with 1, 2 or 5 threads into an exact-size output, whole and
// by random sample ranges of the opened stream.  Any
// read or write outside a buffer aborts the run.
// Build and run (host code only; no GPU is touched):
//   hipcc -O1 -g -std=c++17 -Xarch_host -fsanitize=address -Iinclude -Iomnilingual-asr_amd/csrc -x hip \
//         --offload-arch=gfx950 -c omnilingual-asr_amd/csrc/mp3.cpp -o /tmp/mp3_asan.o
//   g++ -O1 -g -fsanitize=address -no-pie tools/mp3_fuzz.cpp /tmp/mp3_asan.o -o /tmp/mp3_fuzz \
//         -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread
//   ASAN_OPTIONS=detect_leaks=0 /tmp/mp3_fuzz tests/golden/reference_demo_40s.mp3 600
// With UndefinedBehaviorSanitizer as well: add `-Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined`
// to the hipcc line and link with /opt/rocm/lib/llvm/bin/clang++ -fsanitize=address,undefined (its runtime, not gcc's).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <vector>
namespace oasr { void set_error(const char*, ...) {} }   // the library's error text is not needed here
extern "C" int oasr_mp3_info(const uint8_t*, int64_t, int32_t*, int32_t*, int64_t*);
extern "C" int oasr_mp3_decode(const uint8_t*, int64_t, float*, int64_t, int64_t*, int64_t*);
struct oasr_mp3_stream;
extern "C" int oasr_mp3_open(const uint8_t*, int64_t, oasr_mp3_stream**, int32_t*, int32_t*, int64_t*);
extern "C" int oasr_mp3_decode_mono_range(const oasr_mp3_stream*, int64_t, int64_t, float*);
extern "C" void oasr_mp3_close(oasr_mp3_stream*);
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> d(n);
    if (fread(d.data(), 1, n, f) != (size_t)n) return 2;
    const int iters = atoi(argv[2]);
    const size_t keep = argc > 3 ? (size_t)atol(argv[3]) : d.size();   // work on a prefix to keep iterations short
    if (keep < d.size()) d.resize(keep);
    std::mt19937_64 rng(11);
    int ok = 0, err = 0;
    long silent = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> b(d);
        const int kind = rng() % 6;
        if (kind == 0) {
            int k = 1 + rng() % 12;
            for (int i = 0; i < k; i++) b[rng() % b.size()] ^= (uint8_t)(1 + rng() % 255);
        } else if (kind == 1) {
            b.resize(1 + rng() % b.size());
        } else if (kind == 2) {
            size_t i = rng() % (b.size() - 10), m = 1 + rng() % 4000;
            if (i + m > b.size()) m = b.size() - i;
            b.erase(b.begin() + i, b.begin() + i + m);
        } else if (kind == 3) {
            size_t i = rng() % b.size(), m = 1 + rng() % 3000;
            std::vector<uint8_t> ins(m);
            for (auto& x : ins) x = (uint8_t)rng();
            b.insert(b.begin() + i, ins.begin(), ins.end());
        } else if (kind == 4) {   // side information of many frames: the fields that steer every loop of the decoder
            for (size_t p = 0; p + 40 < b.size(); ++p)
                if (b[p] == 0xFF && (b[p + 1] & 0xE0) == 0xE0 && rng() % 3 == 0) {
                    int k = 1 + rng() % 6;
                    for (int i = 0; i < k; i++) b[p + 4 + rng() % 34] ^= (uint8_t)(1u << (rng() % 8));
                    p += 100;
                }
        } else {   // frame headers: version, bit rate, sampling frequency, mode
            for (size_t p = 0; p + 4 < b.size(); ++p)
                if (b[p] == 0xFF && (b[p + 1] & 0xE0) == 0xE0 && rng() % 8 == 0) {
                    b[p + 1 + rng() % 3] ^= (uint8_t)(1u << (rng() % 8));
                    p += 100;
                }
        }
        uint8_t* p = (uint8_t*)malloc(b.size());   // exact-size heap copy so that ASan sees any read past the end
        memcpy(p, b.data(), b.size());
        int32_t sr, ch;
        int64_t total = 0, w = 0, diag[5];
        int rc = oasr_mp3_info(p, b.size(), &sr, &ch, &total);
        if (rc == 0 && total >= 0 && total < (1 << 26)) {
            const char* th[] = {"1", "2", "5"};
            setenv("OASR_MP3_THREADS", th[rng() % 3], 1);
            float* out = (float*)malloc(sizeof(float) * (size_t)(total ? total : 1) * ch);
            rc = oasr_mp3_decode(p, b.size(), out, total, &w, diag);
            if (rc == 0) {
                ok++;
                silent += diag[3];
                for (int64_t i = 0; i < w * ch; i++)
                    if (!(out[i] == out[i])) {
                        printf("NaN in the output\n");
                        return 1;
                    }
            } else {
                err++;
            }
            free(out);
            // the opened-stream form: random sample ranges into exact-size outputs (each range warms up on the frames
            // in front of it)
            oasr_mp3_stream* st = nullptr;
            int64_t t2 = 0;
            if (oasr_mp3_open(p, b.size(), &st, &sr, &ch, &t2) == 0) {
                for (int q = 0; q < 4 && t2 > 0; q++) {
                    const int64_t first = (int64_t)(rng() % (uint64_t)t2);
                    int64_t count = 1 + (int64_t)(rng() % (uint64_t)(t2 - first));
                    if (rng() % 2) count = count < 3000 ? count : 1 + (int64_t)(rng() % 3000);
                    float* o = (float*)malloc(sizeof(float) * (size_t)count);
                    oasr_mp3_decode_mono_range(st, first, count, o);
                    for (int64_t i = 0; i < count; i++)
                        if (!(o[i] == o[i])) {
                            printf("NaN in a range\n");
                            return 1;
                        }
                    free(o);
                }
                oasr_mp3_close(st);
            }
        } else {
            err++;
        }
        free(p);
    }
    printf("asan fuzz: decoded %d, refused %d, frames or granules delivered as silence %ld\n", ok, err, silent);
}
