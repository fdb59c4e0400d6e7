#!/usr/bin/env python3
"""In-process A/B timing of attention kernel variants (stand-alone libraries given on the command line)
at the encoder's shape, on q/k/v with the statistics of a LayerNorm output through a random projection.
Rounds interleave the variants; prints median and minimum per variant and whether outputs are equal."""
import ctypes as C, math, sys
from pathlib import Path
import torch
vp = C.c_void_p
libs = []
for path in sys.argv[1:]:
    h = C.CDLL(path)
    h.oasr_attention_relkey.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
    libs.append((Path(path).stem.replace("abl_attn_", ""), h))
dev = torch.device("cuda")
for head, batch, heads in ((64, 32, 16), (128, 8, 32)):
    t_pad, T = 1504, 1499
    d = heads * head
    g = torch.Generator().manual_seed(1)
    rows = (batch * t_pad + 255) // 256 * 256
    qkv = torch.randn((rows, 3 * d), generator=g)
    qkv[:, :d] *= 1.4426950408889634 / math.sqrt(head)
    qkv = qkv.to(torch.bfloat16).to(dev)
    emb = torch.zeros((96, head)); emb[:73] = torch.randn((73, head), generator=g)
    emb = emb.to(torch.bfloat16).to(dev)
    lens = torch.full((batch,), T, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    outs, times = [], {tag: [] for tag, _ in libs}
    for rnd in range(7):
        for tag, h in libs:
            ctx = torch.zeros((rows, d), dtype=torch.bfloat16, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                rc = h.oasr_attention_relkey(qkv.data_ptr(), emb.data_ptr(), lens.data_ptr(), ctx.data_ptr(), batch, t_pad, heads, head, 64, 8, st)
            e1.record(); e1.synchronize()
            assert rc == 0
            times[tag].append(e0.elapsed_time(e1) / 10)
            if rnd == 0:
                outs.append(ctx)
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    line = f"head {head} batch {batch} x {heads} heads: "
    for tag, _ in libs:
        t = sorted(times[tag])
        line += f"{tag} med {t[len(t)//2]*1e3:6.1f} min {t[0]*1e3:6.1f} | "
    print(line + ("outputs equal" if same else "OUTPUTS DIFFER"))
