#!/usr/bin/env python3
"""Read the per-tile cycle stamps of the diagnostic GEMM build (tuning aid)."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import numpy as np, torch
from omnilingual_asr import _hip
lib = C.CDLL(str(ROOT / "omnilingual-asr_amd/lib/abl_stamp.so"))
m, n, k, epi = 48128, 4096, 1024, 1
if len(sys.argv) > 1: m, n, k, epi = [int(v) for v in sys.argv[1:5]]
dev = torch.device("cuda"); g = torch.Generator().manual_seed(1)
a = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn((n, k), generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
bias = torch.randn((n,), generator=g).to(dev)
c = torch.zeros((m, n), dtype=torch.float32 if epi in (3, 4) else torch.bfloat16, device=dev)
vp = C.c_void_p
lib.oasr_gemm_bf16.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_int32, vp]
for _ in range(3):
    rc = lib.oasr_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr() if epi == 3 else None, c.data_ptr(), m, n, k, 1.0, epi, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
buf = np.zeros(8192, dtype=np.uint64)
lib.oasr_debug_read.argtypes = [vp, C.c_int64]
assert lib.oasr_debug_read(buf.ctypes.data, buf.nbytes) == 0
for wg in range(2):
    for wv in range(2):
        b = buf[(wg * 2 + wv) * 256:(wg * 2 + wv + 1) * 256].astype(np.int64).reshape(-1, 8)
        t0 = b[0, 0]
        print(f"wg{wg} group{'AB'[wv]}: tile: start  main  biaswait  epi   (ticks)")
        for i in range(13):
            if b[i, 0] == 0: break
            if epi in (0, 1):
                print(f"   {i:2d}: {b[i,0]-t0:8d} {b[i,1]-b[i,0]:6d} {b[i,2]-b[i,1]:6d} {b[i,3]-b[i,2]:6d}")
            else:
                print(f"   {i:2d}: {b[i,0]-t0:8d} {b[i,1]-b[i,0]:6d}      - {b[i,3]-b[i,1]:6d}")
