#!/usr/bin/env python3
"""Timing of the fp8 GEMM beside the bf16 GEMM on the 7B geometry's shapes (tuning aid)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import torch
from omnilingual_asr import _hip
L = _hip.lib()
dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
cases = [("7B ffn_up swish", 12288, 16384, 4096, 1), ("7B ffn_down resid", 12288, 4096, 16384, 3),
         ("7B qkv", 12288, 12288, 4096, 0), ("7B out resid", 12288, 4096, 4096, 3), ("7B glu", 12288, 8192, 4096, 2),
         ("300M ffn_up", 48128, 4096, 1024, 1)]
for name, m, n, k, epi in cases:
    g = torch.Generator().manual_seed(1)
    a = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((n, k), generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
    a8 = torch.zeros((m, k), dtype=torch.uint8, device=dev); sa = torch.zeros((m,), device=dev)
    w8 = torch.zeros((n, k), dtype=torch.uint8, device=dev); sw = torch.zeros((n,), device=dev)
    _hip.check(L.oasr_quant_fp8_rows(a.data_ptr(), m, k, a8.data_ptr(), sa.data_ptr(), st))
    _hip.check(L.oasr_quant_fp8_rows(w.data_ptr(), n, k, w8.data_ptr(), sw.data_ptr(), st))
    bias = torch.zeros((n,), device=dev)
    c = torch.zeros((m, n), dtype=torch.float32, device=dev) if epi == 3 else \
        torch.zeros((m, n // 2 if epi == 2 else n), dtype=torch.bfloat16, device=dev)
    r = c.data_ptr() if epi == 3 else None
    def t(fn, it=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    tb = t(lambda: _hip.check(L.oasr_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), r, c.data_ptr(), m, n, k, 1.0, epi, st)))
    t8 = t(lambda: _hip.check(L.oasr_gemm_fp8(a8.data_ptr(), w8.data_ptr(), sa.data_ptr(), sw.data_ptr(), bias.data_ptr(), r, c.data_ptr(), m, n, k, 1.0, epi, st)))
    tq = t(lambda: _hip.check(L.oasr_quant_fp8_rows(a.data_ptr(), m, k, a8.data_ptr(), sa.data_ptr(), st)))
    fl = 2.0 * m * n * k
    print(f"{name:18s} {m}x{n}x{k}: bf16 {tb:8.1f} us ({fl/tb/1e6:6.0f} TF)   fp8 {t8:8.1f} us ({fl/t8/1e6:6.0f} TF)   quant(A) {tq:6.1f} us")
