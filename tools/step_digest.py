#!/usr/bin/env python3
"""SHA-256 of the encoder states of a seeded batch (G300, 8 x 30 s with ragged lengths, all blocks): two library
builds that claim bit-identical results must print the same digest.  The library is chosen with OASR_HIP_LIBRARY."""
import hashlib, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd")); sys.path.insert(0, str(ROOT))
import torch
from omnilingual_asr.engine_hip import HipEngine
from omnilingual_asr.weights import init_random_weights, resolve_geometry
from bench import synth_batch

eng = HipEngine("G300", weights=init_random_weights(resolve_geometry("G300"), 0))
n = 480000
pcm = synth_batch(8, n, 99).cuda()
nv = torch.tensor([n, n - 12345, 200000, n, 64000, n - 1, 333333, n], dtype=torch.int32, device="cuda")
hidden, lengths = eng.step(pcm, nv)
torch.cuda.synchronize()
h = hashlib.sha256()
for i, t in enumerate(lengths.cpu().tolist()):
    h.update(hidden[i, :t].contiguous().cpu().numpy().tobytes())
print("digest", h.hexdigest()[:32], "frames", lengths.cpu().tolist())
