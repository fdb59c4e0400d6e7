"""The oracle's log-mel and encoder restatements against the committed vectors of an independent
third-party implementation (transformers' SeamlessM4T feature extractor / Wav2Vec2-BERT encoder with
seeded random weights; see oracle/gen_golden.py part B).  The reference itself contains neither
stage, so these are cross-checks of a declared geometry, not reference parity."""

import numpy as np
import pytest
import torch

from oracle import conformer as OC
from oracle import logmel as OL


@pytest.fixture(scope="module")
def lm(golden_dir):
    return np.load(golden_dir / "standin_logmel.npz")


@pytest.mark.parametrize("name", ["noise_1s", "tone_odd", "speech_6s"])
def test_logmel_matches_standin(lm, name):
    pcm = lm[name + "_pcm"]
    raw, feats, mask = lm[name + "_logmel"], lm[name + "_features"], lm[name + "_mask"]
    assert OL.num_frames(len(pcm)) == raw.shape[0]
    n = int(mask.sum())
    assert OL.num_stacked(len(pcm)) == n                 # odd trailing frame: dropped == masked out
    # the stand-in rounds the spectrum to complex64 before |.|^2; float64 here
    np.testing.assert_allclose(OL.logmel(pcm), raw, atol=2e-5)
    np.testing.assert_allclose(OL.features(pcm)[:n], feats[:n], atol=1e-3)


def test_window_and_mel_tables_shape():
    w, m = OL.povey_window(), OL.mel_matrix()
    assert w.shape == (400,) and m.shape == (257, 80)
    assert w[0] == 0.0 and abs(w[199] - w[200]) < 1e-12 and np.allclose(w, w[::-1])
    nz = (m > 0).sum(axis=1)
    assert nz.max() <= 2 and (m >= 0).all() and m.max() <= 1.0
    assert (m > 0).sum(axis=0).min() >= 1               # every filter has at least one bin


def _load_weights(z):
    hidden, inter, heads, layers = [int(v) for v in z["geometry"]]
    w = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.") and not k.startswith("w.layers.")}
    w["layers"] = []
    for i in range(layers):
        pre = f"w.layers.{i}."
        w["layers"].append({k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)})
    return w, OC.Geometry(name="standin", d_model=hidden, ffn_dim=inter, n_heads=heads, n_layers=layers)


def test_encoder_matches_standin_per_layer(golden_dir):
    z = np.load(golden_dir / "standin_encoder.npz")
    w, g = _load_weights(z)
    feats, lens = torch.from_numpy(z["feats"]), torch.from_numpy(z["lengths"])
    _, per = OC.forward(feats, lens, w, g, per_layer=True)
    for k in range(g.n_layers + 1):
        want = torch.from_numpy(z[f"hidden_{k}"])
        for b, n in enumerate(lens.tolist()):
            torch.testing.assert_close(per[k][b, :n], want[b, :n], atol=2e-5, rtol=1e-5)


def test_encoder_valid_rows_do_not_depend_on_padding(golden_dir):
    """Causal convolution + masked keys: changing padded inputs must not change valid outputs."""
    z = np.load(golden_dir / "standin_encoder.npz")
    w, g = _load_weights(z)
    feats, lens = torch.from_numpy(z["feats"]).clone(), torch.from_numpy(z["lengths"])
    a = OC.forward(feats, lens, w, g)
    n = int(lens[1])
    feats[1, n:] = 1e3 * torch.randn_like(feats[1, n:])
    b = OC.forward(feats, lens, w, g)
    torch.testing.assert_close(a[1, :n], b[1, :n], atol=1e-5, rtol=1e-5)


def test_float64_and_float32_oracles_agree(golden_dir):
    z = np.load(golden_dir / "standin_encoder.npz")
    w, g = _load_weights(z)
    feats, lens = torch.from_numpy(z["feats"]), torch.from_numpy(z["lengths"])
    a = OC.forward(feats, lens, w, g, dtype=torch.float32)
    b = OC.forward(feats, lens, w, g, dtype=torch.float64).float()
    torch.testing.assert_close(a[0], b[0], atol=5e-5, rtol=1e-4)


def test_encoder_matches_standin_at_the_device_width(golden_dir):
    """The fixture the device test uses (d = 512, matrices bf16-exact): the float32 oracle agrees with the
    stand-in's hidden states per layer."""
    from tests.test_gpu_standin import load_standin_weights

    z = np.load(golden_dir / "standin_encoder_d512.npz")
    hidden, inter, heads, layers = [int(v) for v in z["geometry"]]
    w = load_standin_weights(z)
    g = OC.Geometry(name="standin", d_model=hidden, ffn_dim=inter, n_heads=heads, n_layers=layers)
    feats, lens = torch.from_numpy(z["feats"]), torch.from_numpy(z["lengths"])
    _, per = OC.forward(feats, lens, w, g, per_layer=True)
    for k in range(layers + 1):
        want = torch.from_numpy(z[f"hidden_{k}"])
        for b, n in enumerate(lens.tolist()):
            torch.testing.assert_close(per[k][b, :n], want[b, :n], atol=2e-5, rtol=1e-5)


def test_device_rounding_mode_of_the_encoder_oracle_stays_next_to_the_plain_forward():
    """`forward(device_rounding=True)` restates where the device rounds to 16 bits (and, with fp8, which rows it
    quantises): it must stay within bfloat16 / e4m3 noise of the plain float32 forward, and its attention walk
    over key tiles with a lazily raised maximum must give the plain softmax when nothing is rounded."""
    import torch
    from omnilingual_asr.weights import GEOMETRIES, init_random_weights
    from oracle import conformer as OC

    g = GEOMETRIES["GTINY"]
    og = OC.Geometry(name=g.name, d_model=g.d_model, ffn_dim=g.ffn_dim, n_heads=g.n_heads, n_layers=g.n_layers)
    w = {k: v for k, v in init_random_weights(g, 21).items() if not k.startswith("ctc_")}
    gen = torch.Generator().manual_seed(3)
    feats = torch.randn((2, 150, 160), generator=gen)
    lens = torch.tensor([150, 77])
    plain = OC.forward(feats, lens, w, og)
    emu = OC.forward(feats, lens, w, og, device_rounding=True)
    emu8 = OC.forward(feats, lens, w, og, device_rounding=True, fp8=True)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    for i, n in enumerate(lens):
        assert 5e-4 < rel(emu[i, :n], plain[i, :n]) < 8e-3
        assert 1e-2 < rel(emu8[i, :n], plain[i, :n]) < 6e-2
    # the tiled softmax alone (no rounding: float64, identity rounding) equals the plain one
    keep, keep16 = OC._r, OC._r16
    OC._r = OC._r16 = lambda t: t
    try:
        OC._DEV = True
        lw = {k: v.double() for k, v in w["layers"][0].items()}
        x = torch.randn((2, 150, g.d_model), generator=gen, dtype=torch.float64)
        tiled = OC.attention(x, lens, lw, og)
        OC._DEV = False
        direct = OC.attention(x, lens, lw, og)
    finally:
        OC._DEV = False
        OC._r, OC._r16 = keep, keep16
    for i, n in enumerate(lens):
        torch.testing.assert_close(tiled[i, :n], direct[i, :n], atol=1e-9, rtol=1e-9)
