#!/usr/bin/env python3
"""Launch one fp8 GEMM shape a few times (profiling target for rocprofv3 --pmc): gemm8_one.py [M N K EPI].
Weights packed, as the encoder keeps them; default: the FFN up-projection of BASELINE configs[4]
(batch 8 x 30 s at d = 4096: 12032 rows)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import torch
from omnilingual_asr import _hip
L = _hip.lib()
m, n, k, epi = 12032, 16384, 4096, 1
if len(sys.argv) > 1:
    m, n, k, epi = [int(v) for v in sys.argv[1:5]]
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(1)
a = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn((n, k), generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
a8 = torch.zeros((m, k), dtype=torch.uint8, device=dev); sa = torch.zeros((m,), device=dev)
w8 = torch.zeros((n, k), dtype=torch.uint8, device=dev); sw = torch.zeros((n,), device=dev)
_hip.check(L.oasr_quant_fp8_rows(a.data_ptr(), m, k, a8.data_ptr(), sa.data_ptr(), st))
_hip.check(L.oasr_quant_fp8_rows(w.data_ptr(), n, k, w8.data_ptr(), sw.data_ptr(), st))
wp = torch.empty_like(w8)
_hip.check(L.oasr_gemm_pack_w_fp8(w8.data_ptr(), n, k, wp.data_ptr(), st))
bias = torch.randn((n,), generator=g).to(dev)
c = torch.zeros((m, n), dtype=torch.float32, device=dev) if epi == 3 else \
    torch.zeros((m, n // 2 if epi == 2 else n), dtype=torch.bfloat16, device=dev)
for _ in range(5):
    _hip.check(L.oasr_gemm_fp8_pw(a8.data_ptr(), wp.data_ptr(), sa.data_ptr(), sw.data_ptr(), bias.data_ptr(),
                                  c.data_ptr() if epi == 3 else None, c.data_ptr(), m, n, k, 1.0, epi, st))
torch.cuda.synchronize()
