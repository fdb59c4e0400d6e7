#!/usr/bin/env python3
"""In-process A/B timing of GEMM kernel variants (libraries built by tools/build_gemm_variants.sh):
rounds interleave the variants on the same operands; median and minimum per variant and shape, and a
check that every variant's output equals the first one's bit for bit.
Usage: gemm_ab2.py [--rows M] [--packed] lib1.so lib2.so ...   (--rows 1536: the shapes of a single 30 s utterance;
--packed: the weights in the packed layout the encoder keeps them in, through oasr_gemm_bf16_pw)"""
import ctypes as C
import sys
from pathlib import Path

import torch

libs = []
vp = C.c_void_p
argv = sys.argv[1:]
M = 48128
packed = False
while argv and argv[0] in ("--rows", "--packed"):
    if argv[0] == "--rows":
        M = int(argv[1])
        argv = argv[2:]
    else:
        packed = True
        argv = argv[1:]
extra = []
while argv and argv[0] == "--case":          # --case m n k epilogue
    extra.append(("case", int(argv[1]), int(argv[2]), int(argv[3]), int(argv[4])))
    argv = argv[5:]
for path in argv:
    h = C.CDLL(path)
    h.oasr_gemm_bf16.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_int32, vp]
    h.oasr_gemm_bf16_pw.argtypes = h.oasr_gemm_bf16.argtypes
    h.oasr_gemm_pack_w_bf16.argtypes = [vp, C.c_int64, C.c_int64, vp, vp]
    libs.append((Path(path).stem.replace("abl_gemm_", ""), h))
# the first case is repeated at the end: the first launches of a fresh process on a fresh machine can run slow
cases = [("ffn_up swish", M, 4096, 1024, 1), ("ffn_down resid", M, 1024, 4096, 3),
         ("qkv bias", M, 3072, 1024, 0), ("out resid", M, 1024, 1024, 3), ("glu", M, 2048, 1024, 2),
         ("ctc f32", M, 256, 1024, 4), ("fproj", M, 1024, 192, 4),
         ("ffn_up again", M, 4096, 1024, 1), ("glu again", M, 2048, 1024, 2)]
if extra:
    cases = extra
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
for name, m, n, k, epi in cases:
    g = torch.Generator().manual_seed(1)
    a = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((n, k), generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn((n,), generator=g).to(dev)
    odt = torch.float32 if epi in (3, 4) else torch.bfloat16
    oshape = (m, n // 2 if epi == 2 else n)
    resid = torch.randn(oshape, generator=g).to(dev) if epi == 3 else None
    if packed:
        wp = torch.empty_like(w)
        assert libs[0][1].oasr_gemm_pack_w_bf16(w.data_ptr(), n, k, wp.data_ptr(), st) == 0
        torch.cuda.synchronize()
        w = wp
    outs = []
    times = {tag: [] for tag, _ in libs}
    for rnd in range(7):
        for tag, h in libs:
            c = torch.zeros(oshape, dtype=odt, device=dev)
            args = (a.data_ptr(), w.data_ptr(), None if epi == 2 else bias.data_ptr(), None if resid is None else resid.data_ptr(),
                    c.data_ptr(), m, n, k, 0.5, epi, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                rc = (h.oasr_gemm_bf16_pw if packed else h.oasr_gemm_bf16)(*args)
            e1.record()
            e1.synchronize()
            assert rc == 0
            times[tag].append(e0.elapsed_time(e1) / 10)
            if rnd == 0:
                outs.append(c)
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    line = f"{name:15s} {m}x{n}x{k}: "
    for tag, _ in libs:
        t = sorted(times[tag])
        line += f"{tag} med {t[len(t)//2]*1e3:6.1f} min {t[0]*1e3:6.1f} | "
    print(line + ("outputs equal" if same else "OUTPUTS DIFFER"))
