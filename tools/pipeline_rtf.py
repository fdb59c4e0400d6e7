#!/usr/bin/env python3
"""End-to-end throughput of the host pipeline (chunk plan -> batches -> H2D -> device path -> CTC ->
records) on synthetic long audio, beside the resident-batch bench figure (tuning aid)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd")); sys.path.insert(0, str(ROOT))
import numpy as np, torch
from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline, plan_chunks
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 256 * 30 / 3600.0   # 256 chunks: whole batches only
n = int(hours * 3600 * 16000)
rng = np.random.default_rng(0)
pcm = (0.1 * rng.standard_normal(n)).astype(np.float32)
pipe = GeminiASRPipeline(model="G300")
bounds = plan_chunks(n, 30.0, 30.0)
pipe._encode_chunks(pcm[: 64 * 480000], bounds[:64])        # warm-up: both slots (buffers, graphs, clocks)
torch.cuda.synchronize()
t0 = time.perf_counter()
recs = pipe._encode_chunks(pcm, bounds)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{hours:.2f} h, {len(bounds)} chunks: {dt:.2f} s wall -> RTF {hours*3600/dt:.0f}x  ({dt/ (len(bounds)/32)*1e3:.1f} ms per batch of 32)")
print("first text:", recs[0].to_result().segments[0].text[:60])
