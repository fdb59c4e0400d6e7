"""Encoder geometries, seeded random weights and their device layout.

The reference ships no weights (recognition is remote,
ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:512-530), so geometries are declared
here and weights are random-initialised from a seed.  A weight set is a dict
`{'fp_ln_w', 'fp_ln_b', 'fp_w', 'fp_b', 'layers': [ {...} ], 'ctc_w', 'ctc_b'}` of float32 CPU tensors in the
natural (Linear `[out, in]`) layout; `pack_for_device` converts it to what liboasr_hip.so reads.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class Geometry:
    name: str = "G300"
    feat_dim: int = 160
    d_model: int = 1024
    ffn_dim: int = 4096
    n_heads: int = 16
    n_layers: int = 12
    conv_kernel: int = 31
    left_max: int = 64
    right_max: int = 8
    ln_eps: float = 1e-5
    vocab_size: int = 32          # CTC head outputs (tokenizer.VOCAB_SIZE)

    @property
    def head_dim(self) -> int:
        return self.d_model // self.n_heads

    @property
    def n_dist(self) -> int:
        return self.left_max + self.right_max + 1

    def n_params(self) -> int:
        d, f = self.d_model, self.ffn_dim
        per = (2 * (2 * d + d * f + f + f * d + d)        # two FFNs with their LayerNorms
               + 2 * d + 4 * (d * d + d) + self.n_dist * self.head_dim   # attention
               + 2 * d + 2 * d * d + d * self.conv_kernel + 2 * d + d * d   # conv module
               + 2 * d)                                    # final LayerNorm
        return per * self.n_layers + 2 * self.feat_dim + self.feat_dim * d + d


GEOMETRIES = {
    "G300": Geometry(),
    "G7B": Geometry(name="G7B", d_model=4096, ffn_dim=16384, n_heads=32, n_layers=18),
    # small shapes for tests
    "GTINY": Geometry(name="GTINY", d_model=512, ffn_dim=1024, n_heads=8, n_layers=2),
}

LAYER_KEYS = (
    "ffn1_ln_w", "ffn1_ln_b", "ffn1_w1", "ffn1_b1", "ffn1_w2", "ffn1_b2",
    "attn_ln_w", "attn_ln_b", "wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "dist_emb",
    "conv_ln_w", "conv_ln_b", "pw1", "dw", "dw_ln_w", "dw_ln_b", "pw2",
    "ffn2_ln_w", "ffn2_ln_b", "ffn2_w1", "ffn2_b1", "ffn2_w2", "ffn2_b2",
    "final_ln_w", "final_ln_b",
)


def resolve_geometry(model) -> Geometry:
    """Map the pipeline's `model` argument to a geometry.  Names the reference used for its remote
    model ("gemini-...", ref diarization/pipeline.py:50) select the default geometry."""
    if isinstance(model, Geometry):
        return model
    if isinstance(model, str) and model in GEOMETRIES:
        return GEOMETRIES[model]
    return GEOMETRIES["G300"]


def _uniform(gen, shape, bound):
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * bound


def init_random_weights(g: Geometry, seed: int = 0, ln_jitter: float = 0.1) -> dict:
    """Seeded random weights: Linear/conv U(-1/sqrt(fan_in), 1/sqrt(fan_in)), LayerNorm gains near 1
    (jittered so that a missed gain or bias shows up in parity tests), distance embedding N(0, 1)."""
    gen = torch.Generator().manual_seed(seed)
    d, f, dh = g.d_model, g.ffn_dim, g.head_dim

    def lin(o, i):
        b = 1.0 / math.sqrt(i)
        return _uniform(gen, (o, i), b), _uniform(gen, (o,), b)

    def ln(n):
        return 1.0 + _uniform(gen, (n,), ln_jitter), _uniform(gen, (n,), ln_jitter)

    w = {}
    w["fp_ln_w"], w["fp_ln_b"] = ln(g.feat_dim)
    w["fp_w"], w["fp_b"] = lin(d, g.feat_dim)
    layers = []
    for _ in range(g.n_layers):
        L = {}
        for pfx in ("ffn1", "ffn2"):
            L[pfx + "_ln_w"], L[pfx + "_ln_b"] = ln(d)
            L[pfx + "_w1"], L[pfx + "_b1"] = lin(f, d)
            L[pfx + "_w2"], L[pfx + "_b2"] = lin(d, f)
        L["attn_ln_w"], L["attn_ln_b"] = ln(d)
        for n in ("q", "k", "v", "o"):
            L["w" + n], L["b" + n] = lin(d, d)
        L["dist_emb"] = torch.randn((g.n_dist, dh), generator=gen, dtype=torch.float32)
        L["conv_ln_w"], L["conv_ln_b"] = ln(d)
        L["pw1"] = _uniform(gen, (2 * d, d), 1.0 / math.sqrt(d))
        L["dw"] = _uniform(gen, (d, g.conv_kernel), 1.0 / math.sqrt(g.conv_kernel))
        L["dw_ln_w"], L["dw_ln_b"] = ln(d)
        L["pw2"] = _uniform(gen, (d, d), 1.0 / math.sqrt(d))
        L["final_ln_w"], L["final_ln_b"] = ln(d)
        layers.append({k: L[k] for k in LAYER_KEYS})
    w["layers"] = layers
    # CTC head, drawn last so that the encoder weights of a seed do not depend on it
    w["ctc_w"], w["ctc_b"] = lin(g.vocab_size, d)
    return w


def glu_interleave(pw1: torch.Tensor) -> torch.Tensor:
    """Reorder the rows of the [2d, d] GLU projection so that each group of 32 rows holds 16 value
    rows followed by their 16 gate rows (what the GEMM's GLU epilogue expects)."""
    d = pw1.shape[0] // 2
    val = pw1[:d].reshape(d // 16, 16, -1)
    gate = pw1[d:].reshape(d // 16, 16, -1)
    return torch.cat([val, gate], dim=1).reshape(2 * d, -1).contiguous()


CTC_LOGIT_PITCH = 256   # columns of the logits matrix (one GEMM tile)

# log2(e): `oasr_attention_relkey` works in log2 units and expects q * log2(e) / sqrt(head_dim)
ATTENTION_Q_SCALE = 1.4426950408889634


def quantize_rows_e4m3(w: torch.Tensor):
    """Per-row OCP e4m3 quantisation: (codes uint8 [out, in], scale float32 [out]) with
    w ~= codes * scale, scale = max|row| / 448 (1 for an all-zero row)."""
    w = w.to(torch.float32)
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    codes = (w * (1.0 / scale)[:, None]).to(torch.float8_e4m3fn)
    return codes.view(torch.uint8), scale


def pack_for_device(weights: dict, g: Geometry, device, fp8: bool = False) -> dict:
    """Device tensors in the layout of `oasr_layer_weights` / `oasr_top_weights` (include/oasr.h);
    with `fp8`, also the e4m3 codes and row scales of `oasr_layer_fp8_weights`."""
    bf = lambda t: t.to(device=device, dtype=torch.bfloat16).contiguous()
    f32 = lambda t: t.to(device=device, dtype=torch.float32).contiguous()
    fp_w = torch.zeros((g.d_model, 192), dtype=torch.float32)
    fp_w[:, : g.feat_dim] = weights["fp_w"]
    top = {"fp_ln_w": f32(weights["fp_ln_w"]), "fp_ln_b": f32(weights["fp_ln_b"]),
           "fp_w": bf(fp_w), "fp_b": f32(weights["fp_b"])}
    layers = []
    q_scale = ATTENTION_Q_SCALE / math.sqrt(g.head_dim)
    for L in weights["layers"]:
        dist = torch.zeros((96, g.head_dim), dtype=torch.float32)
        dist[: g.n_dist] = L["dist_emb"]
        P = {
            # the attention kernel takes q in log2 units: log2(e) / sqrt(head) folded into the q rows
            "wqkv": bf(torch.cat([L["wq"] * q_scale, L["wk"], L["wv"]], dim=0)),
            "bqkv": f32(torch.cat([L["bq"] * q_scale, L["bk"], L["bv"]], dim=0)),
            "wo": bf(L["wo"]), "bo": f32(L["bo"]),
            "dist_emb": bf(dist),
            "pw1": bf(glu_interleave(L["pw1"])),
            "dw": f32(L["dw"].t()),          # tap-major [k, d]
            "pw2": bf(L["pw2"]),
        }
        for pfx in ("ffn1", "ffn2"):
            P[pfx + "_w1"], P[pfx + "_b1"] = bf(L[pfx + "_w1"]), f32(L[pfx + "_b1"])
            P[pfx + "_w2"], P[pfx + "_b2"] = bf(L[pfx + "_w2"]), f32(L[pfx + "_b2"])
        for n in ("ffn1_ln", "attn_ln", "conv_ln", "dw_ln", "ffn2_ln", "final_ln"):
            P[n + "_w"], P[n + "_b"] = f32(L[n + "_w"]), f32(L[n + "_b"])
        if fp8:
            mats = {"ffn1_w1": L["ffn1_w1"], "ffn1_w2": L["ffn1_w2"], "ffn2_w1": L["ffn2_w1"], "ffn2_w2": L["ffn2_w2"],
                    "wqkv": torch.cat([L["wq"] * q_scale, L["wk"], L["wv"]], dim=0), "wo": L["wo"],
                    "pw1": glu_interleave(L["pw1"]), "pw2": L["pw2"]}
            for name, m in mats.items():
                codes, scale = quantize_rows_e4m3(m)
                P[name + "_q"], P[name + "_s"] = codes.to(device).contiguous(), f32(scale)
        layers.append(P)
    packed = {"top": top, "layers": layers}
    if "ctc_w" in weights:
        # vocabulary projection padded to one 256-column GEMM tile; padded outputs can never win the argmax
        v = weights["ctc_w"].shape[0]
        cw = torch.zeros((CTC_LOGIT_PITCH, g.d_model), dtype=torch.float32)
        cw[:v] = weights["ctc_w"]
        cb = torch.full((CTC_LOGIT_PITCH,), -1e30, dtype=torch.float32)
        cb[:v] = weights["ctc_b"]
        packed["ctc_w"], packed["ctc_b"] = bf(cw), f32(cb)
    return packed
