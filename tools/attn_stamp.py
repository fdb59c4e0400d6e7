#!/usr/bin/env python3
"""Per-segment cycle sums of the diagnostic attention build (tuning aid)."""
import ctypes as C, sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ["OASR_HIP_LIBRARY"] = str(ROOT / "omnilingual-asr_amd/lib/abl_astamp.so")
import numpy as np
sys.argv = [sys.argv[0], "3"]
exec(open(ROOT / "tools/attn_one.py").read())
buf = np.zeros(8192, dtype=np.uint64)
lib.oasr_debug_read.argtypes = [C.c_void_p, C.c_int64]
assert lib.oasr_debug_read(buf.ctypes.data, buf.nbytes) == 0
print("FAR tiles, ticks per tile: stage  S-mfma  max/raise  exp  PV-issue  barrier   sum")
for wg in range(2):
    for wv in range(4):
        b = buf[(wg * 4 + wv) * 16:(wg * 4 + wv + 1) * 16].astype(np.float64)
        n = max(b[10], 1)
        print(f"wg{wg} wave{wv} n={int(b[10]):3d}: " + " ".join(f"{b[q]/n:7.0f}" for q in range(6)) + f"  {sum(b[:6])/n:7.0f}   | stage split: vmwait {b[6]/n:6.0f} ds_write+drain {b[7]/n:6.0f} load-issue {(b[0]-b[6]-b[7])/n:6.0f}")
