"""Per-block distance of the device encoder from the CPU oracle, through the whole depth of a geometry.

    python tools/depth_parity.py [G7B|G300|GTINY] [--fp8] [--samples 33000,21000] [--layers N]

For every k = 0 .. n_layers the first k blocks run on the device (`oasr_encoder_forward` with
`n_layers_run = k`) and are compared, per utterance, with the oracle's state after k blocks: the
float32 forward (never quantised), and the forward that rounds where the device rounds
(`device_rounding=True`; with --fp8 it also quantises the eight projections to e4m3).  Prints relative L2 and max abs per block.
Weights are drawn on the device (a 7 B-parameter set takes ~40 s from the CPU generator) and copied
to the host for the oracle.
"""

from __future__ import annotations

import argparse
import dataclasses
import math
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "omnilingual-asr_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def device_drawn_weights(g, seed: int) -> dict:
    """A weight set with the distributions of `weights.init_random_weights`, drawn by the device's generator."""
    from omnilingual_asr.weights import LAYER_KEYS
    gen = torch.Generator(device="cuda").manual_seed(seed)
    d, f, dh = g.d_model, g.ffn_dim, g.head_dim

    def uni(shape, bound):
        return ((torch.rand(shape, generator=gen, device="cuda") * 2.0 - 1.0) * bound).cpu()

    def lin(o, i):
        b = 1.0 / math.sqrt(i)
        return uni((o, i), b), uni((o,), b)

    def ln(n):
        return 1.0 + uni((n,), 0.1), uni((n,), 0.1)

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
        L["dist_emb"] = torch.randn((g.n_dist, dh), generator=gen, device="cuda").cpu()
        L["conv_ln_w"], L["conv_ln_b"] = ln(d)
        L["pw1"] = uni((2 * d, d), 1.0 / math.sqrt(d))
        L["dw"] = uni((d, g.conv_kernel), 1.0 / math.sqrt(g.conv_kernel))
        L["dw_ln_w"], L["dw_ln_b"] = ln(d)
        L["pw2"] = uni((d, d), 1.0 / math.sqrt(d))
        L["final_ln_w"], L["final_ln_b"] = ln(d)
        layers.append({k: L[k] for k in LAYER_KEYS})
    w["layers"] = layers
    return w


def depth_curves(g, n_list, fp8: bool, seed: int = 31, weights: dict | None = None):
    """[(k, rel_f32, max_f32, rel_emu, max_emu)] per block k and the worst utterance, device vs the two oracles."""
    from omnilingual_asr.engine_hip import HipEngine
    from oracle import conformer as OC
    from oracle import logmel as OL
    from tests.test_gpu_ops import synth_pcm

    w = weights if weights is not None else device_drawn_weights(g, seed)
    n_max = max(n_list)
    src = synth_pcm(len(n_list), n_max, seed=6)
    pcm = torch.zeros((len(n_list), (n_max + 3) // 4 * 4))
    for i, n in enumerate(n_list):
        pcm[i, :n] = src[i, :n]
    feats, lens = OL.features_batch(pcm.numpy(), n_list)
    og = OC.Geometry(name=g.name, d_model=g.d_model, ffn_dim=g.ffn_dim, n_heads=g.n_heads, n_layers=g.n_layers)
    ft, lt = torch.from_numpy(feats), torch.from_numpy(lens)
    _, ref = OC.forward(ft, lt, w, og, per_layer=True)
    _, emu = OC.forward(ft, lt, w, og, per_layer=True, fp8=fp8, device_rounding=True)
    eng = HipEngine(g, weights=w, fp8=fp8)
    dev = torch.device("cuda")
    nv = torch.tensor(n_list, dtype=torch.int32, device=dev)
    f_dev, l_dev = eng.features(pcm.to(dev), nv)
    assert l_dev.cpu().tolist() == list(lens)
    rows = []
    for k in range(g.n_layers + 1):
        got = eng.encode_features(f_dev, l_dev, n_layers=k).cpu()
        assert torch.isfinite(got).all()
        worst = [0.0, 0.0, 0.0, 0.0]
        for i, n in enumerate(lens):
            for j, r in ((0, ref[k]), (2, emu[k])):
                diff = got[i, :n] - r[i, :n]
                worst[j] = max(worst[j], float(diff.norm() / r[i, :n].norm()))
                worst[j + 1] = max(worst[j + 1], float(diff.abs().max()))
        rows.append((k, *worst))
    eng.close()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="G7B")
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--samples", default="33000,21000")
    ap.add_argument("--layers", type=int, default=-1)
    a = ap.parse_args()
    from omnilingual_asr.weights import GEOMETRIES
    g = GEOMETRIES[a.model]
    if a.layers >= 0:
        g = dataclasses.replace(g, n_layers=a.layers)
    n_list = [int(s) for s in a.samples.split(",")]
    t0 = time.time()
    rows = depth_curves(g, n_list, a.fp8)
    print("%s %s, samples %s, %d blocks (%.0f s)" % (g.name, "fp8" if a.fp8 else "bf16", n_list, g.n_layers, time.time() - t0))
    print("blocks  rel-L2 vs float32  max abs   rel-L2 vs device-rounding oracle  max abs")
    for k, rf, mf, re, me in rows:
        print("%4d    %.3e         %.4f    %.3e                        %.4f" % (k, rf, mf, re, me))


if __name__ == "__main__":
    main()
