#!/usr/bin/env python3
"""Launch the fused depthwise-conv + LayerNorm + Swish kernel a few times at the benchmark shape
(32 x 1504 x 1024): profiling target for rocprofv3 --pmc."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import torch
from omnilingual_asr import _hip

L = _hip.lib()
batch, t_pad, d = 32, 1504, 1024
g = torch.Generator().manual_seed(3)
x = torch.randn((batch * t_pad, d), generator=g).to(torch.bfloat16).cuda()
w = (torch.randn((31, d), generator=g) / 5).cuda()
lw, lb = torch.ones((d,), device="cuda"), torch.zeros((d,), device="cuda")
y = torch.empty_like(x)
st = _hip.current_stream_ptr()
for _ in range(5):
    _hip.check(L.oasr_dwconv_ln_swish(x.data_ptr(), w.data_ptr(), lw.data_ptr(), lb.data_ptr(), y.data_ptr(),
                                      batch, t_pad, d, 31, 1e-5, st))
torch.cuda.synchronize()
