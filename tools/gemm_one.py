#!/usr/bin/env python3
"""Launch the FFN up-projection GEMM a few times (profiling target for rocprofv3 --pmc)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import torch
from omnilingual_asr import _hip
lib = _hip.lib()
m, n, k, epi = 48128, 4096, 1024, 1
if len(sys.argv) > 1:
    m, n, k, epi = [int(v) for v in sys.argv[1:5]]
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
a = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn((n, k), generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
bias = torch.randn((n,), generator=g).to(dev)
c = torch.zeros((m, n), dtype=torch.float32 if epi in (3, 4) else torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    rc = lib.oasr_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr() if epi == 3 else None, c.data_ptr(), m, n, k, 1.0, epi, st)
    assert rc == 0
torch.cuda.synchronize()
