#!/usr/bin/env python3
"""Debugging aid: the two-deep chunk loop against one synchronous call per batch, all differences printed."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd")); sys.path.insert(0, str(ROOT))
import numpy as np, torch
from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline, _mean_abs
rng = np.random.default_rng(3)
pcm = (0.1 * rng.standard_normal(16000 * 40)).astype(np.float32)
cuts = [0, 96000, 192000, 250000, 346000, 442000, 538000, 600000]
bounds = np.array(list(zip(cuts[:-1], cuts[1:])), dtype=np.int64)
pipe = GeminiASRPipeline(model="GTINY", seed=2, batch_size=2)
first = pipe._encode_chunks(pcm, bounds)          # slots and graphs created while earlier batches are in flight
sync = {}
for lo in range(0, 7, 2):
    part = bounds[lo: lo + 2]
    width = int(max(e - s for s, e in part))
    batch = np.zeros((len(part), width), dtype=np.float32)
    for i, (s, e) in enumerate(part):
        batch[i, : e - s] = pcm[s:e]
    for rep in range(2):
        hidden, lengths, toks = pipe.engine.transcribe_batch(batch, [int(e - s) for s, e in part])
        acts = _mean_abs(hidden, lengths)
        for i in range(len(part)):
            cur = (int(lengths[i]), acts[i], toks[i][0].tobytes())
            if rep and sync[lo + i] != cur:
                print("SYNC PATH NOT REPEATABLE", lo + i)
            sync[lo + i] = cur
print("sync:", [(v[0], round(v[1], 6)) for v in sync.values()])
for p in range(6):
    recs = first if p == 0 else pipe._encode_chunks(pcm, bounds)
    diff = [(r.chunk_idx, r.n_frames, round(r.activity, 6)) for r in recs
            if (r.n_frames, r.activity, r.tokens.tobytes()) != sync[r.chunk_idx]]
    print("pass", p, "differences:", diff or "none")
