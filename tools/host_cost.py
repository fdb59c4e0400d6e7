import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/omnilingual-asr_amd")
import numpy as np, torch
from omnilingual_asr.engine_hip import HipEngine
eng = HipEngine("GTINY")
n = 32 * 480000
pcm = (0.1 * np.random.default_rng(0).standard_normal(n)).astype(np.float32)
w = eng.bucket_samples(480000)
t, batch = eng.host_batch_slot(32, w, 0)
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(32):
        s, e = i * 480000, (i + 1) * 480000
        batch[i, : e - s] = pcm[s:e]
        batch[i, e - s:] = 0.0
    t1 = time.perf_counter()
    print(f"assemble 32 x 30 s into page-locked memory: {(t1 - t0) * 1e3:.1f} ms")
