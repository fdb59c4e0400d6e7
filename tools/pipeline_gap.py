#!/usr/bin/env python3
"""Where the per-batch difference between the resident step and the host chunk loop's two-deep path goes:
times, in one process, (a) the graph step alone, (b) step + CTC stage, (c) step + CTC stage + copies of the
results to page-locked memory, (d) the full submit/collect path with the upload."""
import sys, time, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
sys.path.insert(0, str(ROOT))
import torch
from omnilingual_asr.engine_hip import HipEngine
from omnilingual_asr.weights import init_random_weights, resolve_geometry
from bench import synth_batch

torch.cuda.set_device(0)
eng = HipEngine("G300", weights=init_random_weights(resolve_geometry("G300"), 0))
B, n, K = 32, 480000, 20
pcm = synth_batch(B, n, 1234).cuda()
nv = torch.full((B,), n, dtype=torch.int32, device="cuda")


def timed(label, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    print("%-52s %7.3f ms per batch" % (label, (time.perf_counter() - t0) / K * 1e3), flush=True)


timed("a) graph step", lambda: eng.step_graph(pcm, nv))


def with_ctc():
    h, l = eng.step_graph(pcm, nv)
    return eng.decode(h, l) + (l,)


timed("b) + CTC stage", with_ctc)
pins = None


def with_copies():
    global pins
    tok, frm, cnt, l = with_ctc()
    if pins is None:
        pins = [torch.zeros(t.shape, dtype=t.dtype).pin_memory() for t in (tok, frm, cnt, l)]
    for d, s in zip(pins, (tok, frm, cnt, l)):
        d.copy_(s, non_blocking=True)


timed("c) + results to page-locked memory", with_copies)
host = pcm.cpu().numpy()
tickets = []
for which in (0, 1):
    t, arr = eng.host_batch_slot(B, n, which)
    arr[:, :n] = host
    tickets.append(t)
state = {"k": 0}


def two_deep():
    k = state["k"]
    t = tickets[k & 1]
    if k >= 2:
        eng.collect(t)
    eng.submit(t, [n] * B)
    state["k"] = k + 1


timed("d) submit / collect, two in flight", two_deep)
for t in tickets:
    eng.collect(t)
timed("a) graph step (again)", lambda: eng.step_graph(pcm, nv))


# ---- elimination: the same loop with parts of submit() left out
def make_loop(upload=True, sync_host=True, two_graphs=True):
    st = {"k": 0}
    copy_stream = torch.cuda.Stream()

    def body():
        k = st["k"]
        t = tickets[(k & 1) if two_graphs else 0]
        if k >= 2 and sync_host:
            t["done"].synchronize()
        main = torch.cuda.current_stream()
        if upload:
            copy_stream.wait_event(t["done"])
            with torch.cuda.stream(copy_stream):
                t["pcm_d"].copy_(t["stage"], non_blocking=True)
                t["nv_d"].copy_(t["nv_pin"], non_blocking=True)
                t["copied"].record()
            main.wait_event(t["copied"])
        hidden, lengths = eng.step_graph(t["pcm_d"], t["nv_d"])
        tok, frm, cnt = eng.decode(hidden, lengths)
        for name, src in (("tokens", tok), ("frames", frm), ("counts", cnt), ("lengths", lengths)):
            t[name].copy_(src, non_blocking=True)
        t["done"].record()
        st["k"] = k + 1
    return body


for t in tickets:
    t["nv_pin"].numpy()[:] = n
timed("e) as d, inline", make_loop())
timed("f) as e without the upload", make_loop(upload=False))
timed("g) as e without the host waiting per batch", make_loop(sync_host=False))
timed("h) as e, one slot (one graph)", make_loop(two_graphs=False))
timed("i) as f, one slot", make_loop(upload=False, two_graphs=False))
