#!/usr/bin/env python3
"""Time the row quantiser at the 7B geometry's two widths (tuning aid): quant_one.py [rows]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import torch
from omnilingual_asr import _hip
L = _hip.lib()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12032
st = torch.cuda.current_stream().cuda_stream
for k in (4096, 16384, 1536):
    x = torch.randn((rows, k), device="cuda").to(torch.bfloat16)
    y = torch.empty((rows, k), dtype=torch.uint8, device="cuda")
    s = torch.empty((rows,), dtype=torch.float32, device="cuda")
    junk = torch.empty((64 << 20,), dtype=torch.float32, device="cuda")
    ts = []
    for it in range(12):
        junk.fill_(1.0)                       # push x out of the caches, as the producing GEMM's output stream does
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert L.oasr_quant_fp8_rows(x.data_ptr(), rows, k, y.data_ptr(), s.data_ptr(), st) == 0
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts[2:])[len(ts[2:]) // 2]
    print(f"quant {rows} x {k}: {us:.1f} us -> {rows * k * 3 / us / 1e6:.2f} TB/s")
