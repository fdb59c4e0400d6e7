#!/usr/bin/env python3
"""Launch one GEMM shape a few times (profiling target for rocprofv3 --pmc): gemm_one.py [M N K EPI [FLAGS]].
FLAGS are the layout flags of oasr_gemm_bf16_ex (default 1: weights packed, as the encoder keeps them; 3 adds a
packed A operand, as the FFN down-projection gets it)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import torch
from omnilingual_asr import _hip
lib = _hip.lib()
m, n, k, epi = 48128, 4096, 1024, 1
flags = 1
if len(sys.argv) > 1:
    m, n, k, epi = [int(v) for v in sys.argv[1:5]]
if len(sys.argv) > 5:
    flags = int(sys.argv[5])
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
a = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn((n, k), generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
bias = torch.randn((n,), generator=g).to(dev)
c = torch.zeros((m, n), dtype=torch.float32 if epi in (3, 4) else torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
def packed(t):
    out = torch.empty_like(t)
    assert lib.oasr_gemm_pack_w_bf16(t.data_ptr(), t.shape[0], t.shape[1], out.data_ptr(), st) == 0
    return out
if flags & 1:
    w = packed(w)
if flags & 2:
    a = packed(a)
for _ in range(5):
    rc = lib.oasr_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr() if epi == 3 else None, c.data_ptr(), m, n, k, 1.0, epi, flags, st)
    assert rc == 0, lib.oasr_last_error()
torch.cuda.synchronize()
