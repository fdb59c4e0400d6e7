#!/usr/bin/env python3
"""Launch / time the attention kernel at the encoder's shape (profiling and tuning target)."""
import sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import torch
from omnilingual_asr import _hip
lib = _hip.lib()
head = int(sys.argv[2]) if len(sys.argv) > 2 else 64      # attn_one.py [iters] [head size] [batch] [heads]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
heads = int(sys.argv[4]) if len(sys.argv) > 4 else 1024 // head
t_pad, T = 1504, 1499
d = heads * head
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
rows = (batch * t_pad + 255) // 256 * 256
qkv = torch.randn((rows, 3 * d), generator=g)
qkv[:, :d] *= 1.4426950408889634 / math.sqrt(head)   # the kernel takes q in log2 units
qkv = qkv.to(torch.bfloat16).to(dev)
emb = torch.zeros((96, head)); emb[:73] = torch.randn((73, head), generator=g)
emb = emb.to(torch.bfloat16).to(dev)
lens = torch.full((batch,), T, dtype=torch.int32, device=dev)
ctx = torch.zeros((rows, d), dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    rc = lib.oasr_attention_relkey(qkv.data_ptr(), emb.data_ptr(), lens.data_ptr(), ctx.data_ptr(), batch, t_pad, heads, head, 64, 8, st)
    assert rc == 0, lib.oasr_last_error()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): run()
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / n
flops = batch * heads * (4.0 * T * T * head + 2.0 * T * 73 * head)
print(f"attention {batch}x{heads}x{T} head {head}: {ms*1e3:.1f} us -> {flops/ms/1e9:.0f} TFLOP/s")
