#!/usr/bin/env python3
"""Cost of the log-mel spectrum as a matrix product: one bf16 pass of [frames x 400 (padded 416)] x [2 x 257
(padded 768) x 416] through the GEMM kernel; a product accurate enough for the power spectrum needs three
(bf16 high and low parts of both operands)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "omnilingual-asr_amd"))
import torch
from omnilingual_asr import _hip
lib = _hip.lib()
m, n, k = 96000, 768, 416
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
a = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
w = torch.randn((n, k), generator=g).to(torch.bfloat16).to(dev)
bias = torch.zeros((n,)).to(dev)
c = torch.zeros((m, n), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    rc = lib.oasr_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), None, c.data_ptr(), m, n, k, 1.0, 4, st)
    assert rc == 0, lib.oasr_last_error()
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"DFT as one bf16 GEMM {m}x{n}x{k} (float32 out): {ms*1e3:.1f} us -> {2.0*m*n*k/ms/1e9:.0f} TFLOP/s")
