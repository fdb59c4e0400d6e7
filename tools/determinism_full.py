#!/usr/bin/env python3
"""Run-to-run determinism of the full-size step (batch 32 x 30 s, G300, two shortened rows): eager and graph
replays alternate and every result is compared bit by bit with the first.  Run from the repository root."""
import sys
sys.path.insert(0, "omnilingual-asr_amd"); sys.path.insert(0, ".")
import torch
from bench import synth_batch
from omnilingual_asr.engine_hip import HipEngine
eng = HipEngine("G300", seed=0)
pcm = synth_batch(32, 480000, 1234).to(eng.device)
nv = torch.full((32,), 480000, dtype=torch.int32, device=eng.device)
nv[5] = 300001; nv[17] = 123457
ref = None
for i in range(12):
    h, l = (eng.step_graph if i % 2 else eng.step)(pcm, nv)
    torch.cuda.synchronize()
    hc = h.clone()
    lens = l.cpu().tolist()
    if ref is None:
        ref = hc
    else:   # rows past an utterance's length are scratch: only the valid frames are compared
        bad = [b for b in range(32) if not torch.equal(ref[b, :lens[b]], hc[b, :lens[b]])]
        print(i, "equal" if not bad else "DIFFERS in utterances %s" % bad)
        assert not bad
print("deterministic over 12 full-size steps (eager and graph)")
