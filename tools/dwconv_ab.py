#!/usr/bin/env python3
"""In-process A/B timing of the fused depthwise-conv + LayerNorm + Swish kernel across library variants
(tools/build_full_variant.sh with SRC=dwconv): interleaved rounds on the same operands, outputs compared bit for bit.
Usage: dwconv_ab.py lib1.so lib2.so ..."""
import ctypes as C
import sys
from pathlib import Path

import torch

vp = C.c_void_p
libs = []
for path in sys.argv[1:]:
    h = C.CDLL(path)
    h.oasr_dwconv_ln_swish.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_float, vp]
    libs.append((Path(path).stem, h))
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
for batch, t_pad, d in ((32, 1504, 1024), (8, 1504, 2048), (1, 1504, 1024)):
    g = torch.Generator().manual_seed(3)
    x = torch.randn((batch * t_pad, d), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((31, d), generator=g) / 5).to(dev)
    lw, lb = (1 + 0.1 * torch.randn((d,), generator=g)).to(dev), (0.1 * torch.randn((d,), generator=g)).to(dev)
    outs, times = [], {tag: [] for tag, _ in libs}
    for rnd in range(9):
        for tag, h in libs:
            y = torch.zeros_like(x)
            args = (x.data_ptr(), w.data_ptr(), lw.data_ptr(), lb.data_ptr(), y.data_ptr(), batch, t_pad, d, 31, 1e-5, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            assert h.oasr_dwconv_ln_swish(*args) == 0
            e0.record()
            for _ in range(10):
                rc = h.oasr_dwconv_ln_swish(*args)
            e1.record()
            e1.synchronize()
            assert rc == 0
            times[tag].append(e0.elapsed_time(e1) / 10)
            if rnd == 0:
                outs.append(y)
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    line = f"{batch}x{t_pad}x{d}: "
    for tag, _ in libs:
        t = sorted(times[tag])
        line += f"{tag} med {t[len(t)//2]*1e3:6.1f} min {t[0]*1e3:6.1f} | "
    print(line + ("outputs equal" if same else "OUTPUTS DIFFER"))
