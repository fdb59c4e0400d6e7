#!/usr/bin/env python3
"""Read the per-segment cycle sums of the diagnostic GEMM build (tuning aid)."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
import numpy as np, torch
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
print("per phase averages (ticks): ds_read+drain  dma-issue  vmwait  bar1  mfma  bar2")
for wg in range(2):
    for wv in (0, 4):
        b = buf[(wg * 8 + wv) * 16:(wg * 8 + wv + 1) * 16].astype(np.float64)
        steps = b[10]
        print(f"wg{wg} wave{wv}: " + " ".join(f"{b[q]/steps:7.0f}" for q in range(6)) + f"   sum {sum(b[:6])/steps:7.0f}")
