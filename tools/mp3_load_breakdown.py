#!/usr/bin/env python3
"""Where the audio-loading time of a long MP3 goes: file read, decode (mono, host threads), upload, device resampler.
Usage: mp3_load_breakdown.py [repeats of the 40 s fixture]"""
import ctypes as C, os, sys, tempfile, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
from omnilingual_asr import _hip, audio
from omnilingual_asr.engine_hip import HipEngine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 180
d = (ROOT / "tests" / "golden" / "reference_demo_40s.mp3").read_bytes()
head = 10 + ((d[6] & 127) << 21 | (d[7] & 127) << 14 | (d[8] & 127) << 7 | (d[9] & 127))
path = os.path.join(tempfile.gettempdir(), "oasr_bd.mp3")
with open(path, "wb") as f:
    f.write(d)
    for _ in range(reps - 1):
        f.write(d[head + 208:])
eng = HipEngine("GTINY")
L = _hip.lib()
for rep in range(3):
    t0 = time.perf_counter()
    buf = np.fromfile(path, dtype=np.uint8)
    t1 = time.perf_counter()
    r, c, n = C.c_int32(), C.c_int32(), C.c_int64()
    _hip.check(L.oasr_mp3_info(buf.ctypes.data, buf.size, C.byref(r), C.byref(c), C.byref(n)), "info")
    t2 = time.perf_counter()
    out = np.zeros((n.value,), dtype=np.float32)
    t3 = time.perf_counter()
    w = C.c_int64()
    _hip.check(L.oasr_mp3_decode_mono(buf.ctypes.data, buf.size, out.ctypes.data, n.value, C.byref(w), None), "decode")
    t4 = time.perf_counter()
    x = torch.as_tensor(out).to("cuda")
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    y = eng.resample_16k(x, r.value)
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    print("run %d: %.0f s of audio: read %.3f, frame scan %.3f, allocate %.3f, decode %.3f, upload %.3f (%.1f GB/s), "
          "resample %.3f s" % (rep, n.value / r.value, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4,
                               out.nbytes / 1e9 / (t5 - t4), t6 - t5), flush=True)
os.remove(path)
