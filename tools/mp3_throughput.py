#!/usr/bin/env python3
"""Throughput of the host MP3 decoder by thread count (best of several decodes in one process).
Usage: mp3_throughput.py [file.mp3] [repeats]   (default: the 40 s cut of the reference's demo recording, 8 times over)"""
import ctypes as C, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
from omnilingual_asr import _hip
path = sys.argv[1] if len(sys.argv) > 1 else str(ROOT / "tests" / "golden" / "reference_demo_40s.mp3")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = np.fromfile(path, dtype=np.uint8)
if len(sys.argv) <= 1:      # a longer stream: the audio frames repeated (the bit reservoir makes the joins damaged frames)
    body = d[133 + 208:]
    d = np.concatenate([d] + [body] * (reps - 1))
L = _hip.lib()
r, c, n = C.c_int32(), C.c_int32(), C.c_int64()
_hip.check(L.oasr_mp3_info(d.ctypes.data, d.size, C.byref(r), C.byref(c), C.byref(n)), "info")
out = np.zeros((n.value, c.value), dtype=np.float32)
w = C.c_int64()
secs = n.value / r.value
print("%.1f s of audio, %d channel(s) at %d Hz" % (secs, c.value, r.value))
for th in (1, 2, 4, 8, 16):
    os.environ["OASR_MP3_THREADS"] = str(th)
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        _hip.check(L.oasr_mp3_decode(d.ctypes.data, d.size, out.ctypes.data, n.value, C.byref(w), None), "decode")
        best = min(best, time.perf_counter() - t)
    print("threads %2d: %.3f s = %8.0f x real time" % (th, best, secs / best))
