#!/usr/bin/env python3
"""Time the log-mel kernel at batch 32 x 30 s (tuning / profiling target)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd")); sys.path.insert(0, str(ROOT))
import torch
from omnilingual_asr.engine_hip import HipEngine
from omnilingual_asr import _hip
from bench import synth_batch, time_events
eng = HipEngine("GTINY")
B, n = 32, 480000
pcm = synth_batch(B, n, 1).cuda()
nv = torch.full((B,), n, dtype=torch.int32, device="cuda")
buf = eng._get_buffers(B, n)
L = _hip.lib(); st = _hip.current_stream_ptr()
def f():
    _hip.check(L.oasr_logmel_f32(pcm.data_ptr(), n, nv.data_ptr(), B, buf["frame_cap"], eng._tables.data_ptr(), buf["logmel"].data_ptr(), buf["stats"].data_ptr(), st))
def g():
    _hip.check(L.oasr_feature_norm_stack(buf["logmel"].data_ptr(), buf["stats"].data_ptr(), nv.data_ptr(), B, buf["frame_cap"], buf["t_pad"], buf["feats"].data_ptr(), buf["lengths"].data_ptr(), st))
ms = time_events(f, int(sys.argv[1]) if len(sys.argv) > 1 else 20)
ms2 = time_events(g, 20)
print(f"logmel {ms*1e3:.1f} us -> {B*(n*4+2998*80*4)/ms/1e6:.0f} GB/s ; norm_stack {ms2*1e3:.1f} us")
