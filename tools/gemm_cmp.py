#!/usr/bin/env python3
"""A/B timing of several builds of the GEMM entry point in one process (tuning aid).
Usage: gemm_cmp.py [--blas] lib1.so lib2.so ...   (--blas adds a column for the library GEMM behind
torch.matmul on the same operands: bf16 output, no bias, no activation, no residual).  Builds are interleaved launch by launch so that clock and
temperature drift hits them equally; each timed launch follows a kernel that rewrites the A operand
(as the LayerNorm does in the encoder), with HIP events around the GEMM only."""
import ctypes as C
import sys
from pathlib import Path
import torch

libs = []
vp = C.c_void_p
with_blas = "--blas" in sys.argv
for path in [a for a in sys.argv[1:] if a != "--blas"]:
    L = C.CDLL(str(Path(path).resolve()))
    L.oasr_gemm_bf16.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_int32, vp]
    libs.append((Path(path).stem, L))
cases = [("ffn_up swish", 48128, 4096, 1024, 1), ("ffn_down resid", 48128, 1024, 4096, 3),
         ("qkv bias", 48128, 3072, 1024, 0), ("out resid", 48128, 1024, 1024, 3),
         ("glu", 48128, 2048, 1024, 2), ("b1 ffn_up", 1536, 4096, 1024, 1)]
if with_blas:
    libs.append(("torch.matmul", None))
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
iters = 12
print("%-16s" % "shape" + "".join("%16s" % n for n, _ in libs) + "   (median us; in-context)")
tot = [0.0] * len(libs)
for name, m, n, k, epi in cases:
    g = torch.Generator().manual_seed(1)
    a_src = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
    a = torch.empty_like(a_src)
    w = (torch.randn((n, k), generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn((n,), generator=g).to(dev)
    c = torch.zeros((m, n), dtype=torch.float32, device=dev) if epi in (3, 4) else \
        torch.zeros((m, n // 2 if epi == 2 else n), dtype=torch.bfloat16, device=dev)
    args = (a.data_ptr(), w.data_ptr(), None if epi == 2 else bias.data_ptr(), c.data_ptr() if epi == 3 else None,
            c.data_ptr(), m, n, k, 1.0, epi, st)
    out_blas = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    times = [[] for _ in libs]
    for it in range(iters + 2):
        for li, (_, L) in enumerate(libs):
            a.copy_(a_src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.oasr_gemm_bf16(*args) if L is not None else (torch.matmul(a, w.t(), out=out_blas), 0)[1]
            e1.record()
            assert rc == 0
            e1.synchronize()
            if it >= 2:
                times[li].append(e0.elapsed_time(e1) * 1e3)
    meds = [sorted(t)[len(t) // 2] for t in times]
    for i, v in enumerate(meds):
        tot[i] += v * (2 if name.startswith("ffn") or name.startswith("out") else 1) if not name.startswith("b1") else 0
    print("%-16s" % name + "".join("%16.1f" % v for v in meds))
print("%-16s" % "block total" + "".join("%16.1f" % v for v in tot))
