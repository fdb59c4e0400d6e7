#!/usr/bin/env python3
"""Timing of the GEMM entry point over the encoder's shapes and epilogues (tuning aid; not part of
the product or the tests).  HIP events on the launch stream, random bf16 operands, several rounds."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import torch
from omnilingual_asr import _hip

lib = _hip.lib()
cases = [("ffn_up swish", 48128, 4096, 1024, 1), ("ffn_down resid", 48128, 1024, 4096, 3),
         ("qkv bias", 48128, 3072, 1024, 0), ("out resid", 48128, 1024, 1024, 3),
         ("glu", 48128, 2048, 1024, 2), ("b1 ffn_up", 1536, 4096, 1024, 1)]
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
tot = 0.0
for name, m, n, k, epi in cases:
    g = torch.Generator().manual_seed(1)
    a = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((n, k), generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn((n,), generator=g).to(dev)
    if epi in (3, 4):
        c = torch.zeros((m, n), dtype=torch.float32, device=dev)
    else:
        c = torch.zeros((m, n // 2 if epi == 2 else n), dtype=torch.bfloat16, device=dev)
    resid = c if epi == 3 else None
    args = (a.data_ptr(), w.data_ptr(), None if epi == 2 else bias.data_ptr(), None if resid is None else resid.data_ptr(),
            c.data_ptr(), m, n, k, 1.0, epi, st)
    times = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            rc = lib.oasr_gemm_bf16(*args)
        e1.record()
        e1.synchronize()
        assert rc == 0
        times.append(e0.elapsed_time(e1) / 10)
    t = sorted(times)
    med = t[len(t) // 2]
    print(f"{name:16s} {m}x{n}x{k}: median {med*1e3:7.1f} us  min {t[0]*1e3:7.1f} us  -> {2*m*n*k/med/1e9:6.0f} TFLOP/s")
