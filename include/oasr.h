/*
 * oasr.h — C ABI of liboasr_hip.so, the MI355X (gfx950) compute library behind
 * omnilingual_asr's Pipeline.transcribe().
 *
 * The reference has no FFI for this path: its compute is one HTTPS call
 * (ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:512-530) and its only local
 * stages are the duration gate and the fixed-length chunk plan (:217-219, :243-310, :713-714).
 * Each entry point below names the reference behaviour it stands in for; stages the
 * reference does not contain (log-mel features, conformer encoder) say so.
 *
 * Conventions
 *   - every function returns 0 (OASR_OK) or a negative oasr_status; oasr_last_error() returns a
 *     thread-local message for the last failure on the calling thread
 *   - plain pointers and sizes only; device pointers are HIP device addresses owned by the
 *     caller; `stream` is a hipStream_t passed as void* (NULL = the default stream)
 *   - no function allocates or frees device memory, synchronises the device or blocks on the
 *     stream, except oasr_encoder_create/destroy (host memory only) — so every launch function
 *     may be captured into a hipGraph by the caller
 *   - thread-compatible: no mutable global state; an encoder handle is immutable after create.
 *     One exception, documented at its declaration: oasr_ctc_greedy_static works in a per-GPU array
 *     the library owns
 */
#ifndef OASR_H
#define OASR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OASR_VERSION 100 /* 0.1.0 */

typedef enum oasr_status {
    OASR_OK = 0,
    OASR_EINVAL = -1,   /* bad argument (null pointer, shape the kernels do not support) */
    OASR_ECAPACITY = -2,/* caller-provided buffer too small */
    OASR_EHIP = -3,     /* a HIP runtime call or kernel launch failed */
    OASR_ENOMEM = -4    /* host allocation failed */
} oasr_status;

/* Feature geometry (fixed by the kernels). */
#define OASR_SAMPLE_RATE 16000
#define OASR_FRAME_LEN 400
#define OASR_FRAME_HOP 160
#define OASR_NFFT 512
#define OASR_NMEL 80
#define OASR_FEAT_DIM 160 /* two stacked frames */

int oasr_version(void);
const char* oasr_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Chunk plan (host, integer).  Replaces the plan computed inside split_audio_into_chunks
 * (ref gemini_pipeline.py:256-310: `start=0; while start<total: emit; start+=chunk`, last chunk
 * short, no overlap) together with the strict duration gate `duration > MIN` (ref :713-714).
 * Writes (start_sample, end_sample) pairs into out_bounds[2*i], out_bounds[2*i+1].
 * total_samples <= 0 yields the single entry (0, 0), as the reference returns the whole file
 * when it cannot determine a duration (ref :257-259).
 * Returns OASR_OK and stores the chunk count in *n_chunks; OASR_ECAPACITY (with *n_chunks set to
 * the required count) when capacity_chunks is too small. */
int oasr_plan_chunks(int64_t total_samples, int32_t sample_rate, double chunk_seconds,
                     double min_chunking_seconds, int64_t* out_bounds, int64_t capacity_chunks,
                     int64_t* n_chunks);

/* Contiguous shard [lo, hi) of n_items owned by `rank` of `world` (first n%world ranks own one
 * more).  Stands in for the reference's order-independent fan-out over chunks
 * (ref gemini_pipeline.py:623-641), whose results are re-ordered only at merge (:647-654). */
int oasr_shard_range(int64_t n_items, int32_t world, int32_t rank, int64_t* lo, int64_t* hi);

/* IMA / DVI ADPCM as WAV files hold it (format tag 0x11; dictaphones, field recorders - a WAV the reference's ffmpeg
 * path takes, ref gemini_pipeline.py:273-285): `data` = the data chunk, blocks of `block_align` bytes, each a 4-byte
 * header per channel (first sample, step index) followed by 4-byte groups of 8 codes per channel in turn, low nibble
 * first.  Writes interleaved int16; a trailing partial block is ignored. */
int oasr_adpcm_ima_decode(const uint8_t* data, int64_t size, int32_t channels, int32_t block_align, int16_t* out,
                          int64_t capacity_per_channel, int64_t* written_per_channel);

/* Number of 25 ms frames (hop 10 ms, no centring) in n_samples, and of stacked frame pairs. */
int64_t oasr_num_frames(int64_t n_samples);
int64_t oasr_num_stacked(int64_t n_samples);

/* ---------------------------------------------------------------------------------------------
 * Log-mel features.  NOT IN THE REFERENCE (it uploads the file bytes instead,
 * ref gemini_pipeline.py:340-362); geometry declared by this library: x*2^15, frame 400 / hop
 * 160, per-frame DC removal, pre-emphasis 0.97, Povey window, 512-point power spectrum, 80
 * Kaldi-mel triangles 20..8000 Hz, ln(max(.,2^-23)).
 *
 * Host copies of the constant tables the kernel uses (for checking): window[400],
 * mel_weights[257*80] row-major [bin][filter]. */
int oasr_logmel_tables(float* window, float* mel_weights);

/* The same tables (plus FFT twiddles and the mel triangles as bf16 high part + remainder in the
 * operand layout of the matrix instructions that apply them) packed as the device blob the kernel reads.  The library keeps no device-side state: the caller copies the
 * blob to each device once and passes its device address to oasr_logmel_f32. */
int64_t oasr_logmel_table_words(void);
int oasr_logmel_table_blob(uint32_t* blob, int64_t capacity_words);

/* pcm        device, float32, B rows of `pcm_stride` samples; row b holds n_samples[b] valid
 *            samples in [-1, 1]
 * n_samples  device, int32[B]
 * logmel     device, float32 [B][frame_cap][80]; rows >= num_frames(n_samples[b]) are left
 *            untouched
 * tables     device copy of oasr_logmel_table_blob()
 * stats      device, float32 [B][stats_slots][80][2]: per work-group (mean, M2) partials of the
 *            frames it produced, consumed by oasr_feature_norm_stack;
 *            stats_slots = oasr_logmel_stats_slots(frame_cap); the last slot is reserved for the
 *            merged (mean, 1/std) that oasr_feature_norm_stack writes
 * Requires frame_cap even and pcm_stride >= the longest row: n_samples[b] <= pcm_stride for every row
 * (oasr_logmel_f32 clamps a larger value to the pitch, so nothing outside the row is read, but the
 * statistics of oasr_feature_norm_stack are only defined when the caller keeps this bound). */
int64_t oasr_logmel_stats_slots(int64_t frame_cap);
int oasr_logmel_f32(const float* pcm, int64_t pcm_stride, const int32_t* n_samples, int32_t batch,
                    int64_t frame_cap, const uint32_t* tables, float* logmel, float* stats,
                    void* stream);

/* Per-utterance, per-mel-bin normalisation over all frames ((v-mean)/sqrt(var_ddof1+1e-7)) and
 * stacking of frame pairs.  feats: device float32 [B][t_pad][160]; rows >= num_stacked are
 * zero-filled; lengths: device int32[B] receives num_stacked(n_samples[b]).
 * Requires 2*t_pad <= frame_cap. */
int oasr_feature_norm_stack(const float* logmel, float* stats, const int32_t* n_samples,
                            int32_t batch, int64_t frame_cap, int64_t t_pad, float* feats,
                            int32_t* lengths, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Conformer encoder.  NOT IN THE REFERENCE (the remote model call of
 * ref gemini_pipeline.py:512-530 occupies this slot).  Weights are caller-owned device buffers
 * that must outlive the handle. */
typedef struct oasr_encoder_config {
    int32_t feat_dim;     /* 160 */
    int32_t d_model;      /* 512, 1024, 2048 or 4096 */
    int32_t ffn_dim;      /* multiple of 256 */
    int32_t n_heads;      /* d_model / n_heads must be 64 or 128 */
    int32_t n_layers;
    int32_t conv_kernel;  /* 31 */
    int32_t left_max;     /* 64 */
    int32_t right_max;    /* 8;  left_max + right_max + 1 <= 80 */
    float ln_eps;
} oasr_encoder_config;

/* bf16 matrices are row-major [out][in] (the layout of a Linear weight); vectors are float32. */
typedef struct oasr_layer_weights {
    const float *ffn1_ln_w, *ffn1_ln_b;
    const uint16_t* ffn1_w1; const float* ffn1_b1;   /* [ffn][d] */
    const uint16_t* ffn1_w2; const float* ffn1_b2;   /* [d][ffn] */
    const float *attn_ln_w, *attn_ln_b;
    const uint16_t* wqkv; const float* bqkv;         /* [3d][d]: q rows, then k rows, then v rows;
                                                        q rows and q bias pre-multiplied by log2(e)/sqrt(head) */
    const uint16_t* wo; const float* bo;             /* [d][d] */
    const uint16_t* dist_emb;                        /* [96][64] bf16, rows >= left+right+1 zero */
    const float *conv_ln_w, *conv_ln_b;
    const uint16_t* pw1;   /* [2d][d], GLU-interleaved: row 32*g+i (i<16) = value row 16*g+i,
                              row 32*g+16+i = gate row d+16*g+i */
    const float* dw;       /* [conv_kernel][d] float32 (tap-major) */
    const float *dw_ln_w, *dw_ln_b;
    const uint16_t* pw2;   /* [d][d] */
    const float *ffn2_ln_w, *ffn2_ln_b;
    const uint16_t* ffn2_w1; const float* ffn2_b1;
    const uint16_t* ffn2_w2; const float* ffn2_b2;
    const float *final_ln_w, *final_ln_b;
} oasr_layer_weights;

typedef struct oasr_top_weights {
    const float *fp_ln_w, *fp_ln_b;   /* [160] */
    const uint16_t* fp_w;             /* [d][192] bf16, columns >= 160 zero */
    const float* fp_b;                /* [d] */
} oasr_top_weights;

typedef struct oasr_encoder oasr_encoder;

int oasr_encoder_create(const oasr_encoder_config* cfg, const oasr_top_weights* top,
                        const oasr_layer_weights* layers, oasr_encoder** out);
void oasr_encoder_destroy(oasr_encoder* enc);

/* Optional fp8 (OCP e4m3) weights for the eight large projections of every block: q = e4m3 bytes in
 * the layout of the bf16 matrix ([out][in]; pw1 GLU-interleaved, q rows of wqkv pre-scaled), s =
 * float32 scale per output row (weight = q * s).  After this call the forward quantises each
 * projection's bf16 input per row (oasr_quant_fp8_rows) and runs oasr_gemm_fp8; the input
 * projection and everything else stay as they are.  Call before oasr_encoder_workspace_bytes (the
 * workspace grows by the quantised operand).  Passing NULL returns to bf16. */
typedef struct oasr_layer_fp8_weights {
    const uint8_t* ffn1_w1_q; const float* ffn1_w1_s;
    const uint8_t* ffn1_w2_q; const float* ffn1_w2_s;
    const uint8_t* wqkv_q;    const float* wqkv_s;
    const uint8_t* wo_q;      const float* wo_s;
    const uint8_t* pw1_q;     const float* pw1_s;
    const uint8_t* pw2_q;     const float* pw2_s;
    const uint8_t* ffn2_w1_q; const float* ffn2_w1_s;
    const uint8_t* ffn2_w2_q; const float* ffn2_w2_s;
} oasr_layer_fp8_weights;
int oasr_encoder_set_fp8(oasr_encoder* enc, const oasr_layer_fp8_weights* layers);
/* Declares that every bf16 projection matrix handed to oasr_encoder_create (fp_w and the eight per layer:
 * ffn1_w1, ffn1_w2, wqkv, wo, pw1, pw2, ffn2_w1, ffn2_w2) is in the packed layout of oasr_gemm_pack_w_bf16
 * (packed != 0) or row-major (0, the default); with fp8 weights set, the same holds for their e4m3 matrices
 * (oasr_gemm_pack_w_fp8).  Same results either way; packed copies faster. */
int oasr_encoder_set_packed_weights(oasr_encoder* enc, int32_t packed);

/* Rows of the activation matrices for a batch: round_up(batch * t_pad, 256). */
int64_t oasr_encoder_rows(int32_t batch, int64_t t_pad);
/* Bytes of caller-provided device workspace for oasr_encoder_forward. */
int64_t oasr_encoder_workspace_bytes(const oasr_encoder* enc, int32_t batch, int64_t t_pad);

/* feats    device float32 [B][t_pad][160]  (t_pad multiple of 32)
 * lengths  device int32[B], valid stacked frames per utterance (keys >= length are masked)
 * out      device float32 [rows][d_model], rows = oasr_encoder_rows(); row b*t_pad+t is frame t
 *          of utterance b; rows with t >= lengths[b] are unspecified but finite
 * workspace device, >= oasr_encoder_workspace_bytes(), 256-byte aligned; must be zero-filled once
 *          before its first use (later calls may reuse it as is)
 * n_layers_run: run only the first n blocks (<= cfg.n_layers; 0 = input projection only);
 *          pass -1 for all. */
int oasr_encoder_forward(const oasr_encoder* enc, const float* feats, const int32_t* lengths,
                         int32_t batch, int64_t t_pad, float* out, void* workspace,
                         int64_t workspace_bytes, int32_t n_layers_run, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Individual operators of the encoder, exported so each can be checked on its own.  None has a
 * counterpart in the reference. */

/* y[rows][d] = LayerNorm(x[rows][d]) * w + b, optionally followed by Swish.  x float32;
 * y bf16 when out_bf16 != 0 else float32.  d in {160 (y row stride y_stride), multiples of 256
 * up to 4096}.  Columns d..y_stride-1 of y are zero-filled. */
int oasr_layernorm(const float* x, const float* w, const float* b, void* y, int64_t rows,
                   int32_t d, int64_t y_stride, float eps, int32_t out_bf16, int32_t swish,
                   void* stream);

/* Fused double LayerNorm at a block boundary: x <- LN(x; w1,b1) in place (float32) and
 * y(bf16) = LN(x_new; w2,b2). */
int oasr_layernorm2(float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                    uint16_t* y, int64_t rows, int32_t d, float eps, void* stream);

typedef enum oasr_epilogue {
    OASR_EPI_BIAS_BF16 = 0,        /* c_bf16 = acc + bias */
    OASR_EPI_BIAS_SWISH_BF16 = 1,  /* c_bf16 = swish(acc + bias) */
    OASR_EPI_GLU_BF16 = 2,         /* c_bf16[:, n/2] = value * sigmoid(gate), interleaved weights */
    OASR_EPI_RESIDUAL_F32 = 3,     /* c_f32 = resid_f32 + alpha * (acc + bias) (may alias resid) */
    OASR_EPI_BIAS_F32 = 4          /* c_f32 = acc + bias */
} oasr_epilogue;

/* C[m][n] = epilogue(sum_k A[m][k] * W[n][k]).  A bf16 [m][k], W bf16 [n][k]; m, n multiples of
 * 256, k a multiple of 32; bias float32[n] or NULL. */
int oasr_gemm_bf16(const uint16_t* a, const uint16_t* w, const float* bias, const float* resid,
                   void* c, int64_t m, int64_t n, int64_t k, float alpha, int32_t epilogue,
                   void* stream);

/* The same product with the weights in the packed layout the kernel copies fastest: W cut into pieces of
 * 16 rows x 32 k (1 KiB, stored as the kernel's LDS image of that piece), ordered [row group][k slice].
 * oasr_gemm_pack_w_bf16 makes that layout from row-major W[n][k] (n a multiple of 16, k of 32; out must
 * not alias w); results are bit-identical to oasr_gemm_bf16 on the row-major matrix.  Part of the
 * encoder (no counterpart in the reference): see oasr_encoder_set_packed_weights. */
int oasr_gemm_pack_w_bf16(const uint16_t* w, int64_t n, int64_t k, uint16_t* out, void* stream);
int oasr_gemm_bf16_pw(const uint16_t* a, const uint16_t* w_packed, const float* bias, const float* resid,
                      void* c, int64_t m, int64_t n, int64_t k, float alpha, int32_t epilogue,
                      void* stream);
/* The general form: any of the three matrices in the packed layout.  A packed: a[m][k] as made by
 * oasr_gemm_pack_w_bf16(a, m, k) or by a product with OASR_GEMM_C_PACKED; C packed (epilogues
 * OASR_EPI_BIAS_BF16 and OASR_EPI_BIAS_SWISH_BF16 only): c[m][n] is written in that layout, ready to be the
 * next product's packed A.  Packed A pays when n is small (few column tiles share each copy of A):
 * 48128 x 1024 x 4096 runs 7 % faster, 48128 x 4096 x 1024 no faster. */
#define OASR_GEMM_W_PACKED 1
#define OASR_GEMM_A_PACKED 2
#define OASR_GEMM_C_PACKED 4
int oasr_gemm_bf16_ex(const uint16_t* a, const uint16_t* w, const float* bias, const float* resid, void* c,
                      int64_t m, int64_t n, int64_t k, float alpha, int32_t epilogue, int32_t flags,
                      void* stream);

/* FLAC decode on the host (the reference accepts .flac uploads and forwards the bytes to its remote
 * model: ref gemini_pipeline.py:106-118, workflows/wav2elan_web/app.py:76-77).  oasr_flac_info reads
 * STREAMINFO (md5_16 may be NULL; a sample count missing from the header is found by decoding);
 * oasr_flac_decode writes interleaved samples, sign-extended to int32 at the stream's bit depth;
 * oasr_flac_decode16 (streams of at most 16 bits per sample) writes them as int16 scaled to 16 bits, i.e.
 * sample / 32768 is the value in [-1, 1) - a form oasr_pcm_to_mono_f32 takes (sample_bytes 2, bits 16).
 * Frame header CRC-8 and frame CRC-16 are verified.  Streams of a fixed block size with a known sample
 * count are decoded by up to 16 host threads, each on a byte range (a thread starts at the first position
 * of its range where a whole frame decodes and checks, and the ranges must chain byte for byte and sample for
 * sample, else the stream is decoded again in order); OASR_FLAC_THREADS in the environment overrides the count. */
int oasr_flac_info(const uint8_t* data, int64_t size, int32_t* sample_rate, int32_t* channels,
                   int32_t* bits, int64_t* samples_per_channel, uint8_t* md5_16);
int oasr_flac_decode(const uint8_t* data, int64_t size, int32_t* out, int64_t capacity_per_channel,
                     int64_t* written_per_channel);
int oasr_flac_decode16(const uint8_t* data, int64_t size, int16_t* out, int64_t capacity_per_channel,
                       int64_t* written_per_channel);
/* Samples [first, first + count) per channel of a stream whose header states its length, interleaved into
 * out[0 .. count * channels): int16 scaled to 16 bits (out_bits 16, streams of at most 16 bits) or int32 at the
 * stream's depth (out_bits 32) - the values oasr_flac_decode16 / oasr_flac_decode write at those positions.  Only
 * the frames that cover the range are decoded (found from a byte position guessed by proportion and confirmed by
 * both CRCs and the next frame's sample number); long ranges are split over host threads.  This is what lets a
 * rank of a multi-GPU job decode its own shard of chunks and nothing else (the reference cuts its chunks out of
 * the file with ffmpeg -ss/-t, ref gemini_pipeline.py:273-285). */
int oasr_flac_decode_range(const uint8_t* data, int64_t size, int64_t first, int64_t count, void* out,
                           int32_t out_bits);

/* MPEG audio Layer III (MP3) decode on the host: the format of both demo recordings the reference ships and the first
 * upload type it lists (ref gemini_pipeline.py:106-118 MIME table, workflows/wav2elan_web/app.py:76-77; the reference
 * forwards the bytes to its remote model and uses ffmpeg only to cut them, ref :273-285).  First and second standard
 * and the 2.5 extension (8 - 48 kHz), mono / stereo / joint stereo, every block type; ID3v2 tags are stepped over, an
 * Info / Xing frame is not audio, and the encoder delay and padding a LAME-style tag states (plus the decoder's 529
 * samples) are trimmed as ffmpeg does.  oasr_mp3_info scans the frame headers only.  oasr_mp3_decode writes
 * interleaved float32 samples, nominally in [-1, 1) (not clipped); `diagnostics` may be NULL, else 5 words:
 * audio frames, bytes stepped over while searching for frames, granules whose spectrum ran past its stated length,
 * frames or granules delivered as silence (damaged, or their bit reservoir missing), stuffing bits.  Streams of 4096
 * frames or more are decoded by up to 16 host threads on frame ranges (each starts early enough that the frames it
 * discards hold the whole bit reservoir its first kept frame reaches back into: the samples equal an in-order decode); OASR_MP3_THREADS in the environment overrides the count.
 * Free-format streams and Layers I / II return OASR_EINVAL. */
int oasr_mp3_info(const uint8_t* data, int64_t size, int32_t* sample_rate, int32_t* channels,
                  int64_t* samples_per_channel);
int oasr_mp3_decode(const uint8_t* data, int64_t size, float* out, int64_t capacity_per_channel,
                    int64_t* written_per_channel, int64_t* diagnostics);
/* The same decode delivered as one channel: a stereo stream's two channels are averaged (the sum formed in double
 * and rounded to float32 once: what averaging the two-channel output in float64 gives); `out` holds `capacity` samples. */
int oasr_mp3_decode_mono(const uint8_t* data, int64_t size, float* out, int64_t capacity, int64_t* written,
                         int64_t* diagnostics);
/* A stream opened once (its frames found and counted) and decoded by sample ranges, so that a caller can work on
 * one part of a long file while the next is being decoded: oasr_mp3_decode_mono_range writes the samples
 * [first, first + count) of what oasr_mp3_decode_mono would deliver, bit for bit (each range starts early enough to
 * rebuild the bit reservoir and filter history, and discards those frames).  `data` must stay readable until oasr_mp3_close; ranges may be decoded from several threads at once. */
typedef struct oasr_mp3_stream oasr_mp3_stream;
int oasr_mp3_open(const uint8_t* data, int64_t size, oasr_mp3_stream** stream, int32_t* sample_rate, int32_t* channels,
                  int64_t* samples_per_channel);
int oasr_mp3_decode_mono_range(const oasr_mp3_stream* stream, int64_t first, int64_t count, float* out);
void oasr_mp3_close(oasr_mp3_stream* stream);

/* Ogg Vorbis decode on the host: `.ogg` is among the upload types the reference lists (ref gemini_pipeline.py:106-118
 * MIME table, workflows/wav2elan_web/app.py:76-77; the reference forwards the bytes to its remote model).  The first
 * Vorbis stream of an Ogg file (page checksums verified; other multiplexed streams stepped over; Opus, Speex or FLAC in
 * Ogg named in the error): Vorbis I with floor 1, residues 0 / 1 / 2, channel coupling, both block sizes; floor 0
 * returns OASR_EINVAL.  Chained files (logical streams one behind the other) come out as one signal, every link's
 * samples in order; links that differ in sample rate or channel count are refused.  A file that ends inside a page is
 * decoded up to the last whole page.  The sample count follows
 * the granule positions (start and end trimming).  oasr_vorbis_info parses the headers and walks the packets' first
 * bytes only.  oasr_vorbis_decode writes interleaved float32 samples (not clipped), oasr_vorbis_decode_mono the channel
 * mean (formed in double, rounded once).  `threads`: 0 = up to 16 host threads on packet ranges for long streams (a
 * packet depends on the one before it only, so each range starts one packet early; the samples are the same for any
 * count), n > 0 = at most n. */
int oasr_vorbis_info(const uint8_t* data, int64_t size, int32_t* sample_rate, int32_t* channels,
                     int64_t* samples_per_channel);
int oasr_vorbis_decode(const uint8_t* data, int64_t size, float* out, int64_t capacity_per_channel,
                       int64_t* written_per_channel, int32_t threads);
int oasr_vorbis_decode_mono(const uint8_t* data, int64_t size, float* out, int64_t capacity, int64_t* written,
                            int32_t threads);
/* A stream opened once and decoded by sample ranges - what a rank of a process group uses to read only its own shard
 * of chunks, and what lets a caller work on one part of a long file while the next is decoded:
 * oasr_vorbis_decode_mono_range writes the samples [first, first + count) of what oasr_vorbis_decode_mono delivers, bit
 * for bit (a long range on several host threads, as the whole decode).  The stream keeps its own copy of the packets:
 * `data` may be released after oasr_vorbis_open.  Ranges may be decoded from several threads at once. */
typedef struct oasr_vorbis_stream oasr_vorbis_stream;
int oasr_vorbis_open(const uint8_t* data, int64_t size, oasr_vorbis_stream** stream, int32_t* sample_rate,
                     int32_t* channels, int64_t* samples_per_channel);
int oasr_vorbis_decode_mono_range(const oasr_vorbis_stream* stream, int64_t first, int64_t count, float* out);
void oasr_vorbis_close(oasr_vorbis_stream* stream);
/* Test diagnostic: per audio packet, the bits left unread when its residue decode ended (-1: it ran into the end of
 * the packet).  An encoder writes no more than it needs, so with the right code books the figure stays below 8. */
int oasr_vorbis_packet_tails(const uint8_t* data, int64_t size, int32_t* tails, int64_t capacity, int64_t* packets);

/* FLAC in an Ogg container (another of the contents an `.ogg` upload can have, ref gemini_pipeline.py:114): writes the
 * native FLAC stream the file wraps - signature, metadata blocks, frames - into `out` (never more than `size` bytes),
 * for oasr_flac_info / oasr_flac_decode / oasr_flac_decode16 / oasr_flac_decode_range to read.  Page checksums and page
 * sequence are verified as for Vorbis; a file that ends inside a page yields the frames of the whole pages. */
int oasr_ogg_flac_unwrap(const uint8_t* data, int64_t size, uint8_t* out, int64_t capacity, int64_t* written);

/* LayerNorm (optionally followed by Swish) straight to e4m3 rows + per-row scale: byte-identical to
 * oasr_layernorm(out_bf16=1) followed by oasr_quant_fp8_rows, without the intermediate bf16 matrix.
 * oasr_layernorm2_fp8 is the same for oasr_layernorm2 (x is overwritten with its first LayerNorm). */
int oasr_layernorm_fp8(const float* x, const float* w, const float* b, uint8_t* y, float* scale,
                       int64_t rows, int32_t d, float eps, int32_t swish, void* stream);
int oasr_layernorm2_fp8(float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                        uint8_t* y, float* scale, int64_t rows, int32_t d, float eps, void* stream);

/* fp8 path (OCP e4m3).  y[r][:] = e4m3(x[r][:] / scale[r]), scale[r] = max|x[r]| / 448 (1 for an
 * all-zero row); x bf16 [rows][k], k a multiple of 8. */
int oasr_quant_fp8_rows(const uint16_t* x, int64_t rows, int64_t k, uint8_t* y, float* scale,
                        void* stream);

/* C[m][n] = epilogue(scale_a[m] * scale_w[n] * sum_k A8[m][k] * W8[n][k]): the fp8 counterpart of
 * oasr_gemm_bf16 (same epilogues and shape rules); A8, W8 e4m3 bytes, scales float32. */
int oasr_gemm_fp8(const uint8_t* a, const uint8_t* w, const float* scale_a, const float* scale_w,
                  const float* bias, const float* resid, void* c, int64_t m, int64_t n, int64_t k,
                  float alpha, int32_t epilogue, void* stream);
/* The fp8 product with W in the packed layout (as oasr_gemm_bf16_pw): 1 KiB pieces of 16 rows x 64 k made by
 * oasr_gemm_pack_w_fp8 (n a multiple of 16, k of 64; out must not alias w).  Bit-identical results. */
int oasr_gemm_pack_w_fp8(const uint8_t* w, int64_t n, int64_t k, uint8_t* out, void* stream);
int oasr_gemm_fp8_pw(const uint8_t* a, const uint8_t* w_packed, const float* scale_a, const float* scale_w,
                     const float* bias, const float* resid, void* c, int64_t m, int64_t n, int64_t k,
                     float alpha, int32_t epilogue, void* stream);

/* Multi-head self-attention with clamped relative-key bias and key-length masking.
 * qkv bf16 [rows][3*d] (q | k | v), ctx bf16 [rows][d], d = n_heads * head_dim; head_dim 64
 * or 128; dist_emb bf16 [96][head_dim] (rows past left_max + right_max zero).  The kernel works in
 * log2 units: q must arrive multiplied by log2(e)/sqrt(head_dim) (the encoder folds that factor into
 * the q rows of wqkv/bqkv), k and v unscaled.  Rows of an utterance between its length and t_pad are
 * masked, not skipped: they must hold finite values (the encoder keeps them finite). */
int oasr_attention_relkey(const uint16_t* qkv, const uint16_t* dist_emb, const int32_t* lengths,
                          uint16_t* ctx, int32_t batch, int64_t t_pad, int32_t n_heads,
                          int32_t head_dim, int32_t left_max, int32_t right_max, void* stream);

/* Causal depthwise convolution over time within each utterance (k-1 zeros on the left):
 * y[b,t,c] = sum_j w[j][c] * x[b, t-(k-1)+j, c].  x bf16 [rows][d], y float32 [rows][d]. */
int oasr_dwconv_causal(const uint16_t* x, const float* w, float* y, int32_t batch, int64_t t_pad,
                       int32_t d, int32_t kernel, void* stream);

/* The same convolution followed by LayerNorm over d and Swish, in one pass:
 * y bf16 [rows][d] = Swish(LayerNorm(conv(x)) * ln_w + ln_b).  d in {512, 1024, 2048}, kernel 31,
 * t_pad a multiple of 16.  Only rows < batch * t_pad are written. */
int oasr_dwconv_ln_swish(const uint16_t* x, const float* w, const float* ln_w, const float* ln_b,
                         uint16_t* y, int32_t batch, int64_t t_pad, int32_t d, int32_t kernel,
                         float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CTC output stage (the reference returns text from its remote model,
 * ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:512-530; here the text comes from a
 * vocabulary projection of the encoder states, decoded greedily).  The projection itself is
 * oasr_gemm_bf16 with OASR_EPI_BIAS_F32 on the bf16 copy made by oasr_cast_bf16. */

/* y[i] = bf16(x[i]), round to nearest even; n a multiple of 4. */
int oasr_cast_bf16(const float* x, uint16_t* y, int64_t n, void* stream);

/* Greedy CTC decode.  logits float32 [batch * t_pad][ld], columns 0..vocab-1 meaningful.  Per
 * utterance b and frame t < lengths[b]: id = argmax (lowest index on ties); a frame is kept when
 * id != blank and id differs from the previous frame's id.  tokens / frames int32 [batch][t_pad]
 * receive the kept ids and their frame indices, -1 past counts[b].  Any t_pad (a 300 s chunk of
 * the reference's plan, ref gemini_pipeline.py:217, is 15 008 frames).  workspace: device,
 * oasr_ctc_workspace_bytes(batch, t_pad) bytes, 4-byte aligned — receives the per-frame winners
 * between the two launches; contents on entry do not matter. */
int64_t oasr_ctc_workspace_bytes(int32_t batch, int64_t t_pad);
int oasr_ctc_greedy(const float* logits, int64_t ld, int32_t vocab, const int32_t* lengths,
                    int32_t batch, int64_t t_pad, int32_t blank, int32_t* tokens, int32_t* frames,
                    int32_t* counts, void* workspace, void* stream);

/* The same decode for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Takes no
 * workspace: the per-frame winners go to an array the library holds on each GPU it runs on (4 MiB,
 * until the process ends), so batch * t_pad must not exceed oasr_ctc_static_max_frames() (1 048 576;
 * OASR_ECAPACITY otherwise) — a batch of 32 x 30 s is 48 128 frames.  Results are identical to
 * oasr_ctc_greedy's.  Calls no hipMalloc / hipFree.  Every call on a GPU shares that GPU's array: calls
 * that overlap on one GPU (two streams, two threads) race; serialise them.  The first call on a device
 * resolves the array's address (hipGetSymbolAddress) — make it outside stream capture. */
int64_t oasr_ctc_static_max_frames(void);
int oasr_ctc_greedy_static(const float* logits, int64_t ld, int32_t vocab, const int32_t* lengths,
                           int32_t batch, int64_t t_pad, int32_t blank, int32_t* tokens,
                           int32_t* frames, int32_t* counts, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sample-rate conversion to 16 kHz on the device.  The reference sends the file as it is to its
 * remote model (ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:339-362) and its sample
 * file is 22.05 kHz; the local front end needs 16 kHz.
 *
 * y[n] = sum_{j<taps} hp[(n*down+half) % up][j] * x[(n*down+half)/up - j], x = 0 outside [0, n_in).
 * hp float64 [up][taps] = polyphase branches of the low-pass filter (host-built), x float32
 * [batch][x_stride], y float32 [batch][y_stride], n_out = oasr_resample_length(n_in, up, down). */
int64_t oasr_resample_length(int64_t n_in, int32_t up, int32_t down);
int oasr_resample_poly(const float* x, int64_t n_in, int64_t x_stride, int32_t batch,
                       const double* hp, int32_t up, int32_t down, int32_t taps, int32_t half,
                       float* y, int64_t n_out, int64_t y_stride, void* stream);

/* Interleaved integer PCM as a WAV file holds it, or as the FLAC decoder writes it (the formats the reference
 * accepts are listed at ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:106-118), -> mono float32 in
 * [-1, 1) on the device.  sample_bytes is the container: 1 = unsigned 8-bit, 2 / 4 = little-endian int16 / int32
 * (buffer aligned to the container), 3 = packed little-endian 24-bit; a sample's value is the integer /
 * 2^(bits - 1), rounded to float32, and out[f] is the mean of frame f's channels taken in float64 and rounded
 * once - what the host readers compute.  pcm [frames * channels] samples, out float32 [frames]; channels 1..64. */
int oasr_pcm_to_mono_f32(const void* pcm, int64_t frames, int32_t channels, int32_t sample_bytes, int32_t bits,
                         float* out, void* stream);

/* Batch assembly on the device, for a signal that is already there (a file converted by oasr_pcm_to_mono_f32 and
 * oasr_resample_poly): out[i][0 .. width) = pcm[start[i] .. start[i] + len[i]) followed by zeros - what the
 * reference's chunk loop does with file slices (ref gemini_pipeline.py:243-310), without the samples visiting the
 * host.  start int64 [batch] and len int32 [batch] are device arrays; the caller guarantees 0 <= start[i],
 * start[i] + len[i] <= pcm_len and len[i] <= width. */
int oasr_gather_chunks(const float* pcm, int64_t pcm_len, const int64_t* start, const int32_t* len, int32_t batch,
                       int64_t width, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OASR_H */
