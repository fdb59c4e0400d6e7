"""Prints the measured parity figures of the benchmark geometry at full depth against the CPU oracle
(run by hand on the GPU box: python3 tests/parity_numbers.py; not collected by pytest)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import torch, numpy as np
from omnilingual_asr.engine_hip import HipEngine
from omnilingual_asr.weights import GEOMETRIES, init_random_weights
from oracle import conformer as OC, logmel as OL
from tests.test_gpu_ops import synth_pcm
g = GEOMETRIES["G300"]
w = init_random_weights(g, 1)
eng = HipEngine(g, weights=w)
n_list = [480000, 123456]
src = synth_pcm(2, 480000, seed=100)
pcm = torch.zeros((2, 480000))
for i, n in enumerate(n_list): pcm[i, :n] = src[i, :n]
hidden, lengths = eng.encode_batch(pcm, n_list)
feats, lens = OL.features_batch(pcm.numpy(), n_list)
dev = torch.device("cuda")
f_dev, l_dev = eng.features(pcm.to(dev), torch.tensor(n_list, dtype=torch.int32, device=dev))
print("features max abs diff:", float((f_dev.cpu()[0, :lens[0]] - torch.from_numpy(feats)[0, :lens[0]]).abs().max()))
og = OC.Geometry(name=g.name, d_model=g.d_model, ffn_dim=g.ffn_dim, n_heads=g.n_heads, n_layers=g.n_layers)
ref, outs = OC.forward(torch.from_numpy(feats), torch.from_numpy(lens), w, og, per_layer=True)
got = hidden.cpu()
for i, n in enumerate(lens):
    d = got[i, :n] - ref[i, :n]
    print(f"utt {i} frames {n}: rel-L2 {float(d.norm() / ref[i, :n].norm()):.2e}  max abs {float(d.abs().max()):.3f}  (|ref| max {float(ref[i,:n].abs().max()):.2f})")
for k in (1, 4, 8, 12):
    h = eng.encode_features(f_dev, l_dev, n_layers=k).cpu()
    d = h[0, :lens[0]] - outs[k][0, :lens[0]]
    print(f"after {k:2d} blocks: rel-L2 {float(d.norm() / outs[k][0, :lens[0]].norm()):.2e}")
