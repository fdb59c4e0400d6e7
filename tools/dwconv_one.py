#!/usr/bin/env python3
"""Time the fused depthwise-convolution + LayerNorm + Swish kernel at the encoder's shape (tuning aid)."""
import math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import torch
from omnilingual_asr import _hip
L = _hip.lib()
batch, t_pad, d = 32, 1504, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator().manual_seed(1)
x = torch.randn((batch * t_pad, d), generator=g).to(torch.bfloat16).cuda()
w = (torch.randn((31, d), generator=g) / math.sqrt(31)).cuda()
lw, lb = torch.ones(d).cuda(), torch.zeros(d).cuda()
y = torch.zeros((batch * t_pad, d), dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run():
    _hip.check(L.oasr_dwconv_ln_swish(x.data_ptr(), w.data_ptr(), lw.data_ptr(), lb.data_ptr(), y.data_ptr(), batch, t_pad, d, 31, 1e-5, st))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    e0.record()
    for _ in range(20): run()
    e1.record(); e1.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print(f"dwconv_ln_swish {batch}x{t_pad}x{d}: {best*1e3:.1f} us -> {batch*t_pad*d*4/best/1e6:.0f} GB/s")
