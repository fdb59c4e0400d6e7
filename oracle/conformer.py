"""Oracle for the conformer encoder forward.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED against the reference: ref:src/omnilingual_asr contains no encoder and no
weights.  The block structure below is declared by this build; it mirrors the third-party
stand-in tf:models/wav2vec2_bert/modeling_wav2vec2_bert.py:119-131 (feature projection),
:134-154 (feed-forward), :157-226 (convolution module), :263-337 (relative-key attention),
:398-461 (block) with that model's default hyper-parameters, against which
`tests/golden/standin_encoder.npz` cross-checks it.

Plain PyTorch on the CPU in float32 (or float64 when asked); eval mode, no dropout.

Encoder input  : feats [B, T, 160], lengths [B] (valid stacked frames per utterance)
  h = Linear_160->d(LayerNorm_160(feats))
  per block:
    h = h + 0.5 * FFN1(LN(h))                 FFN = Linear d->f, Swish, Linear f->d
    h = h + MHSA(LN(h))                       H heads of d/H, scores (q.k + q.E[clip(j-i,-L,R)+L]) / sqrt(dh),
                                              keys j >= length masked, softmax, P.V, Linear d->d
    h = h + Conv(LN(h))                       pointwise d->2d (no bias), GLU, causal depthwise k=31
                                              (30 zeros on the left), LN, Swish, pointwise d->d (no bias)
    h = h + 0.5 * FFN2(LN(h))
    h = LN_final(h)
Rows t >= lengths[b] of the output are unspecified (valid rows never depend on them: the
convolution is causal and masked keys get zero weight).

`forward(..., device_rounding=True)` keeps the same mathematics but rounds every tensor the device
path stores in 16 bits where the device rounds it (LayerNorm outputs, the FFN hidden after Swish,
q / k / v, the attention weights that enter P.V, the context, the GLU output, the convolution
branch after LayerNorm + Swish: bfloat16; the q.E table: float16; projection matrices: bfloat16,
the q rows with log2(e) / sqrt(head) folded in before rounding) and, with `fp8=True`, quantises to
e4m3 exactly those rounded rows.  What is left between this emulation and the device is summation
order and the transcendental approximations, so a test against it can use a tolerance several times
tighter than one against the plain float32 forward.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


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

    @property
    def head_dim(self) -> int:
        return self.d_model // self.n_heads

    @property
    def n_dist(self) -> int:
        return self.left_max + self.right_max + 1


G300 = Geometry()
G7B = Geometry(name="G7B", d_model=4096, ffn_dim=16384, n_heads=32, n_layers=18)

LAYER_KEYS = (
    "ffn1_ln_w", "ffn1_ln_b", "ffn1_w1", "ffn1_b1", "ffn1_w2", "ffn1_b2",
    "attn_ln_w", "attn_ln_b", "wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "dist_emb",
    "conv_ln_w", "conv_ln_b", "pw1", "dw", "dw_ln_w", "dw_ln_b", "pw2",
    "ffn2_ln_w", "ffn2_ln_b", "ffn2_w1", "ffn2_b1", "ffn2_w2", "ffn2_b2",
    "final_ln_w", "final_ln_b",
)


# When set, the eight large projections of a block emulate the device's fp8 mode: the input is
# quantised per row and the weight per output row to OCP e4m3 (scale = max|row| / 448), the product
# is taken on the de-quantised values.  Test infrastructure: set/reset by `forward(..., fp8=True)`.
_FP8 = False
# When set, tensors are rounded where the device stores them in 16 bits (`forward(..., device_rounding=True)`).
_DEV = False

LOG2E = 1.4426950408889634
ATTN_KEY_TILE = 64              # keys per step of the attention kernel
ATTN_RESCALE_THRESHOLD = 6.0    # its running maximum moves only past this (log2 units)


def _r(t):
    """A tensor the device holds as bfloat16."""
    return t.to(torch.bfloat16).to(t.dtype) if _DEV else t


def _r16(t):
    """A tensor the device holds as float16 (the attention kernel's q.E table)."""
    return t.to(torch.float16).to(t.dtype)


def _fake_quant_rows(t):
    amax = t.abs().amax(dim=-1, keepdim=True)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    if _DEV:    # the device multiplies by the float32 reciprocal of the scale
        t32, s32 = t.to(torch.float32), scale.to(torch.float32)
        return ((t32 * (1.0 / s32)).to(torch.float8_e4m3fn).to(torch.float32) * s32).to(t.dtype)
    return (t / scale).to(torch.float32).to(torch.float8_e4m3fn).to(t.dtype) * scale


def _proj(x, w, b=None):
    if _FP8:
        return F.linear(_fake_quant_rows(x), _fake_quant_rows(w.to(x.dtype)), b)
    return F.linear(x, _r(w), b)


def _swish(x):
    return x * torch.sigmoid(x)


def _ffn(x, w, pfx):
    return _proj(_r(_swish(_proj(x, w[pfx + "_w1"], w[pfx + "_b1"]))), w[pfx + "_w2"], w[pfx + "_b2"])


def _ln(x, w, b, g):
    """LayerNorm whose output feeds a projection (bfloat16 rows on the device)."""
    return _r(F.layer_norm(x, (x.shape[-1],), w, b, g.ln_eps))


def attention(x, lengths, w, g: Geometry):
    if _DEV:
        return _attention_device_rounding(x, lengths, w, g)
    b, t, d = x.shape
    h, dh = g.n_heads, g.head_dim
    q = _proj(x, w["wq"], w["bq"]).view(b, t, h, dh).transpose(1, 2)
    k = _proj(x, w["wk"], w["bk"]).view(b, t, h, dh).transpose(1, 2)
    v = _proj(x, w["wv"], w["bv"]).view(b, t, h, dh).transpose(1, 2)
    scale = 1.0 / math.sqrt(dh)
    scores = torch.matmul(q, k.transpose(-1, -2)) * scale
    pos = torch.arange(t)
    dist = (pos[None, :] - pos[:, None]).clamp(-g.left_max, g.right_max) + g.left_max   # [i, j]
    qe = torch.matmul(q, w["dist_emb"].t())                   # [b, h, t, n_dist]
    bias = torch.gather(qe, 3, dist[None, None].expand(b, h, t, t))
    scores = scores + bias * scale
    key_ok = pos[None, :] < lengths[:, None].to(torch.long)   # [b, j]
    scores = scores.masked_fill(~key_ok[:, None, None, :], float("-inf"))
    p = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(p, v).transpose(1, 2).reshape(b, t, d)
    return _proj(ctx, w["wo"], w["bo"])


def _attention_device_rounding(x, lengths, w, g: Geometry):
    """The attention branch with the device's roundings: q (in log2 units), k, v and the context are
    bfloat16, the q.E table float16, the weights enter P.V as bfloat16 while their sum stays float32."""
    b, t, d = x.shape
    h, dh = g.n_heads, g.head_dim
    qs = LOG2E / math.sqrt(dh)
    q = _r(_proj(x, w["wq"] * qs, w["bq"] * qs)).view(b, t, h, dh).transpose(1, 2)
    k = _r(_proj(x, w["wk"], w["bk"])).view(b, t, h, dh).transpose(1, 2)
    v = _r(_proj(x, w["wv"], w["bv"])).view(b, t, h, dh).transpose(1, 2)
    pos = torch.arange(t)
    dist = (pos[None, :] - pos[:, None]).clamp(-g.left_max, g.right_max) + g.left_max   # [i, j]
    qe = _r16(torch.matmul(q, _r(w["dist_emb"]).t()))           # [b, h, t, n_dist]
    scores = torch.matmul(q, k.transpose(-1, -2)) + torch.gather(qe, 3, dist[None, None].expand(b, h, t, t))
    key_ok = pos[None, :] < lengths[:, None].to(torch.long)
    scores = scores.masked_fill(~key_ok[:, None, None, :], float("-inf"))
    # The kernel walks the keys in tiles of 64 and keeps, per query, a running maximum m that it sets to the
    # first tile's maximum and afterwards raises to a tile's maximum only when that exceeds m by more than
    # 6 (log2 units).  A tile's weights are p = 2^(score - m) with the m of that moment, and it is this p
    # that is rounded to bfloat16 for P.V (earlier tiles are rescaled in float32 when m moves).  Rounding
    # relative to another maximum gives other bfloat16 values, so the walk is restated here.
    tiles = [slice(k0, min(k0 + ATTN_KEY_TILE, t)) for k0 in range(0, t, ATTN_KEY_TILE)]
    m, m_at = None, []
    for sl in tiles:
        tm = scores[..., sl].amax(dim=-1, keepdim=True)
        m = tm if m is None else torch.where(tm - m > ATTN_RESCALE_THRESHOLD, tm, m)
        m_at.append(m)
    num = torch.zeros((b, h, t, dh), dtype=x.dtype)
    den = torch.zeros((b, h, t, 1), dtype=x.dtype)
    for sl, mk in zip(tiles, m_at):
        p = torch.exp2(scores[..., sl] - mk)
        f = torch.exp2(mk - m)
        num += torch.matmul(_r(p), v[:, :, sl]) * f
        den += p.sum(dim=-1, keepdim=True) * f
    ctx = torch.nan_to_num(num / den)                          # an utterance without keys has no valid row
    return _proj(_r(ctx.transpose(1, 2).reshape(b, t, d)), w["wo"], w["bo"])


def conv_module(x, w, g: Geometry):
    d = g.d_model
    y = _ln(x, w["conv_ln_w"], w["conv_ln_b"], g)
    y = _proj(y, w["pw1"])
    y = _r(y[..., :d] * torch.sigmoid(y[..., d:]))           # GLU over the channel dim
    yt = F.pad(y.transpose(1, 2), (g.conv_kernel - 1, 0))
    yt = F.conv1d(yt, w["dw"][:, None, :], groups=d)
    y = F.layer_norm(yt.transpose(1, 2), (d,), w["dw_ln_w"], w["dw_ln_b"], g.ln_eps)
    return _proj(_r(_swish(y)), w["pw2"])


def block(x, lengths, w, g: Geometry, taps=None):
    d = g.d_model
    x = x + 0.5 * _ffn(_ln(x, w["ffn1_ln_w"], w["ffn1_ln_b"], g), w, "ffn1")
    if taps is not None:
        taps["ffn1"] = x
    x = x + attention(_ln(x, w["attn_ln_w"], w["attn_ln_b"], g), lengths, w, g)
    if taps is not None:
        taps["attn"] = x
    x = x + conv_module(x, w, g)
    if taps is not None:
        taps["conv"] = x
    x = x + 0.5 * _ffn(_ln(x, w["ffn2_ln_w"], w["ffn2_ln_b"], g), w, "ffn2")
    return F.layer_norm(x, (d,), w["final_ln_w"], w["final_ln_b"], g.ln_eps)


def input_projection(feats, w, g: Geometry):
    y = _ln(feats, w["fp_ln_w"], w["fp_ln_b"], g)
    return F.linear(y, _r(w["fp_w"]), w["fp_b"])


@torch.no_grad()
def forward(feats, lengths, weights: dict, g: Geometry = G300, dtype=torch.float32, per_layer=False, fp8=False,
            device_rounding=False):
    """weights: {'fp_ln_w','fp_ln_b','fp_w','fp_b','layers': [ {LAYER_KEYS...} ]}.
    Returns hidden [B, T, d] (and the list of per-layer outputs when per_layer).  `fp8`: the eight large
    projections of a block on e4m3 operands; `device_rounding`: 16-bit storage roundings as on the device."""
    cast = lambda t: t.to(dtype)
    if feats.shape[1] == 0:     # audio shorter than two frames: no encoder rows at all
        x = torch.zeros((feats.shape[0], 0, g.d_model), dtype=dtype)
        return (x, [x] * (len(weights["layers"]) + 1)) if per_layer else x
    top = {k: cast(v) for k, v in weights.items() if k != "layers"}
    global _FP8, _DEV
    _DEV = bool(device_rounding)
    try:
        x = input_projection(cast(feats), top, g)
        outs = [x]
        _FP8 = bool(fp8)
        for lw in weights["layers"]:
            x = block(x, lengths, {k: cast(v) for k, v in lw.items()}, g)
            outs.append(x)
    finally:
        _FP8 = False
        _DEV = False
    return (x, outs) if per_layer else x
