// Conformer encoder forward: the launch sequence over the operators of this library.
// Host code only — every launch goes to the caller's stream, nothing is allocated or synchronised,
// so the whole forward can be captured into a hipGraph by the caller.
#include <new>
#include <vector>

#include "oasr_common.h"

struct oasr_encoder {
    oasr_encoder_config cfg;
    oasr_top_weights top;
    std::vector<oasr_layer_weights> layers;
    std::vector<oasr_layer_fp8_weights> fp8;   // empty: bf16 projections
    bool packed_w = false;                      // bf16 projection matrices in the layout of oasr_gemm_pack_w_bf16
};

namespace oasr {

constexpr int kFeatK = 192;  // 160 feature columns padded to a multiple of the GEMM K-step

struct Workspace {
    uint16_t* xn;   // [rows][d]      normalised activations (GEMM A operand)
    uint16_t* big;  // [rows][max(ffn, 3d)]  FFN hidden / fused qkv / padded features
    uint16_t* ctx;  // [rows][d]      attention context / GLU output
    float* tmp;     // [rows][d]      depthwise-conv output
    uint8_t* a8;    // [rows][max(ffn, d)]  e4m3 copy of the projection input (fp8 mode only)
    float* sa;      // [rows]               its row scales
    int64_t bytes;
};

static Workspace carve(const oasr_encoder_config& c, int64_t rows, char* base, bool fp8) {
    Workspace w;
    const int64_t wide = c.ffn_dim > 3 * c.d_model ? c.ffn_dim : 3 * c.d_model;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        char* p = base ? base + off : nullptr;
        off += round_up(bytes, 256);
        return p;
    };
    w.xn = reinterpret_cast<uint16_t*>(take(rows * c.d_model * 2));
    w.big = reinterpret_cast<uint16_t*>(take(rows * wide * 2));
    w.ctx = reinterpret_cast<uint16_t*>(take(rows * c.d_model * 2));
    w.tmp = reinterpret_cast<float*>(take(rows * c.d_model * 4));
    w.a8 = nullptr;
    w.sa = nullptr;
    if (fp8) {
        w.a8 = reinterpret_cast<uint8_t*>(take(rows * (c.ffn_dim > c.d_model ? c.ffn_dim : c.d_model)));
        w.sa = reinterpret_cast<float*>(take(rows * 4));
    }
    w.bytes = off;
    return w;
}

}  // namespace oasr

using namespace oasr;

#define OASR_TRY(call)            \
    do {                          \
        int rc_ = (call);         \
        if (rc_ != OASR_OK) return rc_; \
    } while (0)

extern "C" {

int oasr_encoder_create(const oasr_encoder_config* cfg, const oasr_top_weights* top,
                        const oasr_layer_weights* layers, oasr_encoder** out) {
    OASR_REQUIRE(cfg && top && out, "encoder_create: null pointer");
    OASR_REQUIRE(cfg->n_layers >= 0 && (cfg->n_layers == 0 || layers), "encoder_create: layers missing");
    OASR_REQUIRE(cfg->feat_dim == OASR_FEAT_DIM, "encoder_create: feat_dim must be %d", OASR_FEAT_DIM);
    OASR_REQUIRE(cfg->d_model == 512 || cfg->d_model == 1024 || cfg->d_model == 2048 || cfg->d_model == 4096,
                 "encoder_create: d_model must be 512, 1024, 2048 or 4096 (the LayerNorm kernels' row widths)");
    OASR_REQUIRE(cfg->ffn_dim > 0 && cfg->ffn_dim % 256 == 0, "encoder_create: ffn_dim must be a multiple of 256");
    OASR_REQUIRE(cfg->n_heads > 0 && (cfg->d_model == cfg->n_heads * 64 || cfg->d_model == cfg->n_heads * 128),
                 "encoder_create: head size (d_model / n_heads) must be 64 or 128");
    OASR_REQUIRE(cfg->conv_kernel == 31 || cfg->conv_kernel == 15 || cfg->conv_kernel == 7 || cfg->conv_kernel == 3,
                 "encoder_create: conv_kernel must be 3, 7, 15 or 31");
    OASR_REQUIRE(cfg->left_max >= 0 && cfg->right_max >= 0 && cfg->left_max + cfg->right_max + 1 <= 80,
                 "encoder_create: relative-position window wider than the 80 distances the attention kernel keeps");
    oasr_encoder* e = new (std::nothrow) oasr_encoder;
    if (!e) {
        set_error("encoder_create: out of host memory");
        return OASR_ENOMEM;
    }
    e->cfg = *cfg;
    e->top = *top;
    e->layers.assign(layers, layers + cfg->n_layers);
    *out = e;
    return OASR_OK;
}

int oasr_encoder_set_fp8(oasr_encoder* enc, const oasr_layer_fp8_weights* layers) {
    OASR_REQUIRE(enc, "encoder_set_fp8: null handle");
    if (!layers) {
        enc->fp8.clear();
        return OASR_OK;
    }
    OASR_REQUIRE(enc->cfg.d_model % 64 == 0 && enc->cfg.ffn_dim % 64 == 0, "encoder_set_fp8: dimensions must be multiples of 64");
    for (int l = 0; l < enc->cfg.n_layers; ++l) {
        const oasr_layer_fp8_weights& w = layers[l];
        OASR_REQUIRE(w.ffn1_w1_q && w.ffn1_w1_s && w.ffn1_w2_q && w.ffn1_w2_s && w.wqkv_q && w.wqkv_s && w.wo_q &&
                         w.wo_s && w.pw1_q && w.pw1_s && w.pw2_q && w.pw2_s && w.ffn2_w1_q && w.ffn2_w1_s &&
                         w.ffn2_w2_q && w.ffn2_w2_s,
                     "encoder_set_fp8: layer %d has a null pointer", l);
    }
    enc->fp8.assign(layers, layers + enc->cfg.n_layers);
    return OASR_OK;
}

int oasr_encoder_set_packed_weights(oasr_encoder* enc, int32_t packed) {
    OASR_REQUIRE(enc, "encoder_set_packed_weights: null handle");
    enc->packed_w = packed != 0;
    return OASR_OK;
}

void oasr_encoder_destroy(oasr_encoder* enc) { delete enc; }

int64_t oasr_encoder_rows(int32_t batch, int64_t t_pad) { return round_up((int64_t)batch * t_pad, 256); }

int64_t oasr_encoder_workspace_bytes(const oasr_encoder* enc, int32_t batch, int64_t t_pad) {
    if (!enc || batch <= 0 || t_pad <= 0) return 0;
    return carve(enc->cfg, oasr_encoder_rows(batch, t_pad), nullptr, !enc->fp8.empty()).bytes;
}

int oasr_encoder_forward(const oasr_encoder* enc, const float* feats, const int32_t* lengths,
                         int32_t batch, int64_t t_pad, float* out, void* workspace,
                         int64_t workspace_bytes, int32_t n_layers_run, void* stream) {
    OASR_REQUIRE(enc && feats && lengths && out && workspace, "encoder_forward: null pointer");
    OASR_REQUIRE(batch > 0 && t_pad > 0 && t_pad % 32 == 0, "encoder_forward: t_pad must be a multiple of 32");
    OASR_REQUIRE(((uintptr_t)workspace & 255) == 0, "encoder_forward: workspace must be 256-byte aligned");
    const oasr_encoder_config& c = enc->cfg;
    const int64_t rows = oasr_encoder_rows(batch, t_pad);
    const int64_t live = (int64_t)batch * t_pad;
    const bool fp8 = !enc->fp8.empty();
    Workspace ws = carve(c, rows, reinterpret_cast<char*>(workspace), fp8);
    if (ws.bytes > workspace_bytes) {
        set_error("encoder_forward: workspace needs %lld bytes, got %lld", (long long)ws.bytes,
                  (long long)workspace_bytes);
        return OASR_ECAPACITY;
    }
    int nl = n_layers_run < 0 ? c.n_layers : n_layers_run;
    OASR_REQUIRE(nl <= c.n_layers, "encoder_forward: n_layers_run exceeds the model depth");
    const int d = c.d_model, f = c.ffn_dim;
    float* x = out;  // the residual stream lives in the output buffer

    // input projection: LayerNorm(160) -> bf16 [rows][192] -> Linear -> x
    OASR_TRY(oasr_layernorm(feats, enc->top.fp_ln_w, enc->top.fp_ln_b, ws.big, live, OASR_FEAT_DIM, kFeatK,
                            c.ln_eps, 1, 0, stream));
    const int wflag = enc->packed_w ? OASR_GEMM_W_PACKED : 0;   // projection matrices packed or row-major
    OASR_TRY(oasr_gemm_bf16_ex(ws.big, enc->top.fp_w, enc->top.fp_b, nullptr, x, rows, d, kFeatK, 1.0f, OASR_EPI_BIAS_F32,
                               wflag, stream));
    // The FFN hidden activations go from the up-projection to the down-projection in the packed layout as
    // well (bf16 path): the down-projection has only d / 256 column tiles to share each copy of its A
    // operand, and a packed A makes those copies whole cache lines.
    const int hid_out = fp8 ? 0 : OASR_GEMM_C_PACKED, hid_in = fp8 ? 0 : OASR_GEMM_A_PACKED;
    // one large projection: bf16 GEMM, or row quantisation + fp8 GEMM when fp8 weights are set
    // `a` = nullptr: the operand already sits in ws.a8 / ws.sa (written by a LayerNorm of the fp8 path)
    auto project = [&](const uint16_t* a, int64_t k, const uint16_t* wb, const uint8_t* wq, const float* wscale,
                       const float* bias, const float* resid, void* dst, int64_t n, float alpha, int epi,
                       int layout = 0) -> int {
        if (!fp8) return oasr_gemm_bf16_ex(a, wb, bias, resid, dst, rows, n, k, alpha, epi, wflag | layout, stream);
        if (a) OASR_TRY(oasr_quant_fp8_rows(a, rows, k, ws.a8, ws.sa, stream));
        return (enc->packed_w ? oasr_gemm_fp8_pw : oasr_gemm_fp8)(ws.a8, wq, ws.sa, wscale, bias, resid, dst, rows, n, k, alpha, epi, stream);
    };
    // LayerNorm feeding a projection: bf16 rows in ws.xn, or (fp8 path) e4m3 rows + scales in ws.a8 / ws.sa
    auto norm = [&](const float* src, const float* lw, const float* lb, int64_t nrows, int swish) -> int {
        if (fp8) return oasr_layernorm_fp8(src, lw, lb, ws.a8, ws.sa, nrows, d, c.ln_eps, swish, stream);
        return oasr_layernorm(src, lw, lb, ws.xn, nrows, d, d, c.ln_eps, 1, swish, stream);
    };
    const uint16_t* xn = fp8 ? nullptr : ws.xn;
    static const oasr_layer_fp8_weights kNoFp8 = {};
    for (int l = 0; l < nl; ++l) {
        const oasr_layer_weights& w = enc->layers[l];
        const oasr_layer_fp8_weights& q = fp8 ? enc->fp8[l] : kNoFp8;
        if (l == 0) OASR_TRY(norm(x, w.ffn1_ln_w, w.ffn1_ln_b, rows, 0));
        // half-step feed-forward 1
        OASR_TRY(project(xn, d, w.ffn1_w1, q.ffn1_w1_q, q.ffn1_w1_s, w.ffn1_b1, nullptr, ws.big, f, 1.0f,
                         OASR_EPI_BIAS_SWISH_BF16, hid_out));
        OASR_TRY(project(ws.big, f, w.ffn1_w2, q.ffn1_w2_q, q.ffn1_w2_s, w.ffn1_b2, x, x, d, 0.5f,
                         OASR_EPI_RESIDUAL_F32, hid_in));
        // self-attention
        OASR_TRY(norm(x, w.attn_ln_w, w.attn_ln_b, rows, 0));
        OASR_TRY(project(xn, d, w.wqkv, q.wqkv_q, q.wqkv_s, w.bqkv, nullptr, ws.big, 3 * (int64_t)d, 1.0f,
                         OASR_EPI_BIAS_BF16));
        OASR_TRY(oasr_attention_relkey(ws.big, w.dist_emb, lengths, ws.ctx, batch, t_pad, c.n_heads,
                                       c.d_model / c.n_heads, c.left_max, c.right_max, stream));
        OASR_TRY(project(ws.ctx, d, w.wo, q.wo_q, q.wo_s, w.bo, x, x, d, 1.0f, OASR_EPI_RESIDUAL_F32));
        // convolution module
        OASR_TRY(norm(x, w.conv_ln_w, w.conv_ln_b, rows, 0));
        OASR_TRY(project(xn, d, w.pw1, q.pw1_q, q.pw1_s, nullptr, nullptr, ws.ctx, 2 * (int64_t)d, 1.0f,
                         OASR_EPI_GLU_BF16));
        if (c.conv_kernel == 31 && (d == 512 || d == 1024 || d == 2048) && t_pad % 16 == 0) {
            OASR_TRY(oasr_dwconv_ln_swish(ws.ctx, w.dw, w.dw_ln_w, w.dw_ln_b, ws.xn, batch, t_pad, d,
                                          c.conv_kernel, c.ln_eps, stream));
            OASR_TRY(project(ws.xn, d, w.pw2, q.pw2_q, q.pw2_s, nullptr, x, x, d, 1.0f, OASR_EPI_RESIDUAL_F32));
        } else {
            // rows past batch * t_pad keep the bytes of an earlier operand (finite; their products are
            // never read)
            OASR_TRY(oasr_dwconv_causal(ws.ctx, w.dw, ws.tmp, batch, t_pad, d, c.conv_kernel, stream));
            OASR_TRY(norm(ws.tmp, w.dw_ln_w, w.dw_ln_b, live, 1));
            OASR_TRY(project(xn, d, w.pw2, q.pw2_q, q.pw2_s, nullptr, x, x, d, 1.0f, OASR_EPI_RESIDUAL_F32));
        }
        // half-step feed-forward 2
        OASR_TRY(norm(x, w.ffn2_ln_w, w.ffn2_ln_b, rows, 0));
        OASR_TRY(project(xn, d, w.ffn2_w1, q.ffn2_w1_q, q.ffn2_w1_s, w.ffn2_b1, nullptr, ws.big, f, 1.0f,
                         OASR_EPI_BIAS_SWISH_BF16, hid_out));
        OASR_TRY(project(ws.big, f, w.ffn2_w2, q.ffn2_w2_q, q.ffn2_w2_s, w.ffn2_b2, x, x, d, 0.5f,
                         OASR_EPI_RESIDUAL_F32, hid_in));
        // block output LayerNorm, fused with the next block's first LayerNorm when there is one
        if (l + 1 < nl) {
            const oasr_layer_weights& nx = enc->layers[l + 1];
            if (fp8)
                OASR_TRY(oasr_layernorm2_fp8(x, w.final_ln_w, w.final_ln_b, nx.ffn1_ln_w, nx.ffn1_ln_b, ws.a8, ws.sa,
                                             rows, d, c.ln_eps, stream));
            else
                OASR_TRY(oasr_layernorm2(x, w.final_ln_w, w.final_ln_b, nx.ffn1_ln_w, nx.ffn1_ln_b, ws.xn, rows, d,
                                         c.ln_eps, stream));
        } else {
            OASR_TRY(oasr_layernorm(x, w.final_ln_w, w.final_ln_b, x, rows, d, d, c.ln_eps, 0, 0, stream));
        }
    }
    return OASR_OK;
}

}  // extern "C"
