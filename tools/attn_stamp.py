#!/usr/bin/env python3
"""Per-segment cycle sums of a diagnostic attention build with s_memtime stamps (tuning aid; the build
is made by hand into lib/abl_astamp.so and is not part of the library)."""
import ctypes as C, sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ["OASR_HIP_LIBRARY"] = str(ROOT / "omnilingual-asr_amd/lib/abl_astamp.so")
import numpy as np
sys.argv = [sys.argv[0], "3"]
exec(open(ROOT / "tools/attn_one.py").read())
buf = np.zeros(64 * 8 * 8, dtype=np.uint64)
lib.oasr_debug_read.argtypes = [C.c_void_p, C.c_int64]
assert lib.oasr_debug_read(buf.ctypes.data, buf.nbytes) == 0
print("FAR tiles, ticks per tile:  to-S-issued  S-drain+max+raise  exp+PV-issue  vmcnt-wait  barrier    sum")
b = buf.reshape(64, 8, 8).astype(np.float64)
for wg in (0, 1, 2, 3, 17, 40):
    for wv in range(4):
        r = b[wg, wv]; n = max(r[5], 1)
        print(f"wg{wg:2d} wave{wv} n={int(r[5]):3d}: " + " ".join(f"{r[q]/n:9.0f}" for q in range(5)) + f"  {r[:5].sum()/n:8.0f}")
tot = b[:, :4, :5].sum(axis=(0, 1)); n = b[:, :4, 5].sum()
print("mean over 64 work-groups:     " + " ".join(f"{v/n:9.0f}" for v in tot) + f"  {tot.sum()/n:8.0f}")
