"""Why does the FFN up-projection read 330 us inside the step and sometimes 410 us when launched alone?
Times the same launch (same weights, encoder layouts) in several surroundings, in one process:
  a) back to back, A = LayerNorm output of the step's final states, fresh buffers
  b) LayerNorm + up alternating (what bench.py times)
  c) LayerNorm + up + down (the FFN as the step runs it)
  d) as (a) with Gaussian A
  e) as (a) with the output buffer of the engine's own workspace
Each figure: HIP events around every launch, 100 untimed rounds in front, mean / min / max over 40."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
sys.path.insert(0, str(ROOT))
import torch
from omnilingual_asr import _hip
from omnilingual_asr.engine_hip import HipEngine
from omnilingual_asr.weights import init_random_weights, resolve_geometry
from bench import synth_batch

torch.cuda.set_device(0)
eng = HipEngine("G300", weights=init_random_weights(resolve_geometry("G300"), 0))
g = eng.geometry
L = _hip.lib()
B, n = 32, 480000
pcm = synth_batch(B, n, 1234).cuda()
nv = torch.full((B,), n, dtype=torch.int32, device="cuda")
for _ in range(3):
    hidden, lengths = eng.step_graph(pcm, nv)
torch.cuda.synchronize()
rows = int(L.oasr_encoder_rows(B, hidden.shape[1]))
d, f = g.d_model, g.ffn_dim
st = _hip.current_stream_ptr()
lw = eng._packed["layers"][0]
x = torch.zeros((rows, d), dtype=torch.float32, device="cuda")
x[:hidden.numel() // d] = hidden.reshape(-1, d)
xn = torch.empty((rows, d), dtype=torch.bfloat16, device="cuda")
xr = torch.randn((rows, d), device="cuda").to(torch.bfloat16)
big = torch.empty((rows, f), dtype=torch.bfloat16, device="cuda")
resid = torch.zeros((rows, d), dtype=torch.float32, device="cuda")
W, A, C = _hip.GEMM_W_PACKED, _hip.GEMM_A_PACKED, _hip.GEMM_C_PACKED


def ln():
    _hip.check(L.oasr_layernorm(x.data_ptr(), lw["ffn1_ln_w"].data_ptr(), lw["ffn1_ln_b"].data_ptr(), xn.data_ptr(),
                                rows, d, d, 1e-5, 1, 0, st))


def up(a=xn, c=big):
    _hip.check(L.oasr_gemm_bf16_ex(a.data_ptr(), lw["ffn1_w1"].data_ptr(), lw["ffn1_b1"].data_ptr(), None, c.data_ptr(),
                                   rows, f, d, 1.0, _hip.EPI_BIAS_SWISH_BF16, W | C, st))


def down():
    _hip.check(L.oasr_gemm_bf16_ex(big.data_ptr(), lw["ffn1_w2"].data_ptr(), lw["ffn1_b2"].data_ptr(), resid.data_ptr(),
                                   resid.data_ptr(), rows, d, f, 0.5, _hip.EPI_RESIDUAL_F32, W | A, st))


def timed(label, before, launch, after=None, lead=100, iters=40):
    for _ in range(lead):
        before(); launch()
        if after:
            after()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in ev:
        before()
        e0.record(); launch(); e1.record()
        if after:
            after()
    torch.cuda.synchronize()
    t = [e0.elapsed_time(e1) * 1e3 for e0, e1 in ev]
    print("%-44s mean %6.1f  min %6.1f  max %6.1f us" % (label, sum(t) / len(t), min(t), max(t)), flush=True)


ln(); torch.cuda.synchronize()
nop = lambda: None
timed("b) LayerNorm, up   [before the upload path ran]", ln, up)
host = pcm.cpu().numpy()
tickets = []
for which in (0, 1):
    t, arr = eng.host_batch_slot(B, n, which)
    arr[:, :n] = host
    tickets.append(t)
for k in range(12):
    t = tickets[k & 1]
    if k >= 2:
        eng.collect(t)
    eng.submit(t, [n] * B)
for t in tickets:
    eng.collect(t)
torch.cuda.synchronize()
timed("b) LayerNorm, up   [after the upload path ran]", ln, up)
for rep in range(2):
    timed("a) up alone, A = LayerNorm(states)", nop, up)
    timed("b) LayerNorm, up", ln, up)
    timed("c) LayerNorm, up, down", ln, up, down)
    timed("d) up alone, Gaussian A", nop, lambda: up(xr))
    print("   A stats: LayerNorm rows std %.3f absmax %.1f | Gaussian std %.3f" %
          (float(xn.float().std()), float(xn.float().abs().max()), float(xr.float().std())))
    big2 = torch.empty((rows, f), dtype=torch.bfloat16, device="cuda")
    timed("e) up alone, another output buffer", nop, lambda: up(xn, big2))
    del big2
