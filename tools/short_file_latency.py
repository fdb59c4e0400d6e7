#!/usr/bin/env python3
"""Latency of the public call on a short file (the reference's 17.6 s sample, 22.05 kHz mono WAV), steady state,
with the share of audio loading (memory map, upload, device conversion and resampling, copy back)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import numpy as np
from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline

path = ROOT / "tests" / "golden" / "gettysburg.wav"
pipe = GeminiASRPipeline(model="G300")
load_s = [0.0]
orig = pipe._load_audio


def timed_load(p):
    t0 = time.perf_counter()
    out = orig(p)
    load_s[0] += time.perf_counter() - t0
    return out


pipe._load_audio = timed_load
for _ in range(3):
    pipe.transcribe_with_retry(path)
ts, ls = [], []
for _ in range(20):
    load_s[0] = 0.0
    t0 = time.perf_counter()
    res = pipe.transcribe_with_retry(path)
    ts.append(time.perf_counter() - t0)
    ls.append(load_s[0])
print("17.6 s file: median %.2f ms per call (audio loading %.2f ms), min %.2f ms; %d segment(s)" %
      (1e3 * sorted(ts)[10], 1e3 * sorted(ls)[10], 1e3 * min(ts), len(res.segments)))
