#!/usr/bin/env python3
"""Throughput of the host FLAC decoder on a long stream (the frames of the reference's own .flac upload,
repeated with fresh frame numbers: tests/flac_tools.py), by thread count.  Usage: flac_throughput.py [repeats]"""
import ctypes as C, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from omnilingual_asr import _hip, audio
import flac_tools as ft

src = ROOT / "tests" / "golden" / "reference_upload.flac"
x, rate = audio.read_flac(src)
ints = np.round(x * 32768).astype(np.int32)
stream, want = ft.repeat_stream(src.read_bytes(), int(sys.argv[1]) if len(sys.argv) > 1 else 400, ints)
seconds = want.shape[0] / rate
print("%.1f MB, %.2f h of audio" % (len(stream) / 1e6, seconds / 3600))
L = _hip.lib()
buf = np.frombuffer(stream, dtype=np.uint8)
out = np.zeros((want.shape[0], 1), dtype=np.int32)
w = C.c_int64()
for th in (1, 2, 4, 8, 16):
    os.environ["OASR_FLAC_THREADS"] = str(th)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        rc = L.oasr_flac_decode(buf.ctypes.data, buf.size, out.ctypes.data, out.shape[0], C.byref(w))
        best = min(best, time.perf_counter() - t0)
    assert rc == 0 and np.array_equal(out, want)
    print("%2d thread(s): %7.1f ms = %6.0fx real time" % (th, best * 1e3, seconds / best), flush=True)
