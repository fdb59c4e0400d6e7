#!/usr/bin/env python3
"""Run-to-run determinism of the device path at small ragged shapes (debugging aid): the same batch is
pushed through the step repeatedly, with unrelated work in between, and every result is compared bit by
bit with the first."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd")); sys.path.insert(0, str(ROOT))
import numpy as np, torch
from omnilingual_asr.engine_hip import HipEngine
eng = HipEngine("GTINY", seed=2)
dev = eng.device
rng = np.random.default_rng(3)
for shape, nvalid in (((1, 62000), [62000]), ((2, 96000), [58000, 96000]), ((3, 48000), [48000, 100, 31999])):
    pcm = torch.from_numpy((0.1 * rng.standard_normal(shape)).astype(np.float32)).to(dev)
    for i, n in enumerate(nvalid):
        pcm[i, n:] = 0
    nv = torch.tensor(nvalid, dtype=torch.int32, device=dev)
    other = torch.from_numpy((0.1 * rng.standard_normal((2, 80000))).astype(np.float32)).to(dev)
    onv = torch.tensor([80000, 70000], dtype=torch.int32, device=dev)
    ref = None
    bad = {}
    for it in range(40):
        if it % 3 == 1:
            eng.step(other, onv)                      # dirty other buffer sets and the caches
        for nl in (0, 1, 2):
            feats, lens = eng.features(pcm, nv)
            h = eng.encode_features(feats, lens, n_layers=nl).clone()
            torch.cuda.synchronize()
            key = nl
            if ref is None or key not in ref:
                ref = ref or {}
                ref[key] = (feats.clone(), h)
            else:
                lens_h = lens.cpu().tolist()
                for b, n in enumerate(lens_h):
                    if not torch.equal(feats[b, :n], ref[key][0][b, :n]):
                        bad[("feats", b)] = bad.get(("feats", b), 0) + 1
                    if not torch.equal(h[b, :n], ref[key][1][b, :n]):
                        bad[(nl, b)] = bad.get((nl, b), 0) + 1
    print(shape, "mismatching runs per (layers, row):", bad or "none")
