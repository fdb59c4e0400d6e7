#!/usr/bin/env python3
"""Benchmark of the transcribe() device path on MI355X.

A step is one pass of the whole device path (log-mel -> feature normalisation -> 12-block conformer
encoder, geometry G300, bf16 MFMA) over one resident batch of 32 x 30 s of synthetic 16 kHz audio,
replayed from a captured HIP graph.  Multi-GPU: one process per GPU (launched by
torch.distributed.run), each rank works on its own batches (weak scaling, no data-path collective
per step) and the fixed-size per-chunk records are exchanged with ONE all-gather at the end, inside
the timed region.  Rank 0 prints one JSON line.
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
MFMA_FP8_PEAK_TFLOPS = 5000.0   # dense fp8


def synth_batch(batch: int, n: int, seed: int) -> torch.Tensor:
    """0.1*N(0,1) smoothed by a 2-tap average plus a 440 Hz tone at -20 dBFS, clipped to [-1, 1]."""
    g = torch.Generator().manual_seed(seed)
    x = 0.1 * torch.randn((batch, n + 1), generator=g)
    x = 0.5 * (x[:, 1:] + x[:, :-1])
    t = torch.arange(n, dtype=torch.float64) / 16000.0
    x = x + (0.1 * torch.sin(2 * math.pi * 440.0 * t)).to(torch.float32)[None, :]
    return x.clamp_(-1.0, 1.0).contiguous()


def encoder_flops_per_utt(g, t: int) -> float:
    d, f, L = g.d_model, g.ffn_dim, g.n_layers
    per = 8.0 * t * d * f + 8.0 * t * d * d + 4.0 * t * t * d + 2.0 * t * g.n_dist * d \
        + 6.0 * t * d * d + 2.0 * t * d * g.conv_kernel
    return per * L + 2.0 * t * g.feat_dim * d


def time_events(fn, iters: int) -> float:
    """Average milliseconds per call of `fn`, bracketed by HIP events on the current stream."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / iters


def measured_traffic(kernel: str, shape: str):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE are collected in separate profiler runs, never inside this script); None when the file
    holds no entry for this kernel at this shape."""
    found = sorted((ROOT / "profiles").glob("r*_traffic.json"))      # the newest round that holds the kernel
    entry = None
    for path in reversed(found):
        entry = json.loads(path.read_text()).get(kernel)
        if entry:
            break
    if not entry or entry.get("shape") != shape:
        return None
    return entry["traffic_bytes_per_launch"]


def _time_launches(prep, launch, iters: int = 20, lead: int = 100, max_groups: int = 40) -> float:
    """Average milliseconds of `launch()` over a group of `iters` launches, each bracketed by its own pair of
    HIP events on the stream the kernel runs on (torch's current stream) and preceded, outside the events,
    by `prep()` — the kernel that produces its input in the encoder, so operands are as cache-warm as
    in the step.  `lead` untimed rounds of the same pair are queued in front, and groups are repeated until
    two in a row agree within 2 % (the later one is returned): launches of a kernel on buffers allocated
    just before have been seen to run up to a third slower for tens of milliseconds (first process on a
    freshly started machine; the same kernel inside the step, and the same call a little later, do not),
    and a figure taken inside that transient describes the machine's start-up, not the kernel."""
    for _ in range(lead):
        if prep is not None:
            prep()
        launch()
    prev = None
    for _ in range(max_groups):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters + 1)]
        for e0, e1 in ev:
            if prep is not None:
                prep()
            e0.record()
            launch()
            e1.record()
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for e0, e1 in ev[1:]) / iters
        if prev is not None and abs(ms - prev) <= 0.02 * prev:
            break
        prev = ms
    return ms


def kernel_rooflines(engine, batch: int, t_pad: int, pcm, nv, hidden) -> dict:
    """HIP-event timings of the kernels that bound the path.  `roofline` is the dominant kernel, the FFN
    up-projection GEMM (MFMA-bound); `roofline_more` holds the other GEMM shapes of a block (FFN down
    and the d x d projection, both with the float32 residual epilogue), the attention kernel and the
    feature kernel (HBM-bound)."""
    from omnilingual_asr import _hip
    g = engine.geometry
    L = _hip.lib()
    rows = int(L.oasr_encoder_rows(batch, t_pad))
    dev = engine.device
    lw = engine._packed["layers"][0]
    fp8 = engine.fp8
    peak = MFMA_FP8_PEAK_TFLOPS if fp8 else MFMA_BF16_PEAK_TFLOPS
    d, f = g.d_model, g.ffn_dim
    x = torch.zeros((rows, d), dtype=torch.float32, device=dev)
    x[:hidden.numel() // d] = hidden.reshape(-1, d)
    st = _hip.current_stream_ptr()
    bf = lambda r, c: torch.empty((r, c), dtype=torch.bfloat16, device=dev)
    xn, big, ctx = bf(rows, d), bf(rows, max(f, 3 * d)), bf(rows, d)
    a8 = torch.empty((rows, max(f, d)), dtype=torch.uint8, device=dev) if fp8 else None
    sa = torch.empty((rows,), dtype=torch.float32, device=dev) if fp8 else None
    resid = torch.zeros((rows, d), dtype=torch.float32, device=dev)
    lengths = torch.full((batch,), min(t_pad, int(_hip.num_stacked(pcm.shape[1]))), dtype=torch.int32, device=dev)

    def ln():          # LayerNorm -> A operand of the up-projection (bf16 rows, or e4m3 rows + scales)
        if fp8:
            _hip.check(L.oasr_layernorm_fp8(x.data_ptr(), lw["ffn1_ln_w"].data_ptr(), lw["ffn1_ln_b"].data_ptr(),
                                            a8.data_ptr(), sa.data_ptr(), rows, d, 1e-5, 0, st))
        else:
            _hip.check(L.oasr_layernorm(x.data_ptr(), lw["ffn1_ln_w"].data_ptr(), lw["ffn1_ln_b"].data_ptr(),
                                        xn.data_ptr(), rows, d, d, 1e-5, 1, 0, st))

    w_packed = engine._packed["gemm_weights_packed"]

    def gemm(a, wname, bias, res, c, n, k, alpha, epi, layout=0):
        """One projection as the encoder launches it; `layout` adds the packed-activation flags the encoder
        uses for the FFN hidden rows (bf16 path with packed weights only)."""
        flags = (_hip.GEMM_W_PACKED | layout) if w_packed else 0

        def run():
            if fp8:
                fn8 = L.oasr_gemm_fp8_pw if w_packed else L.oasr_gemm_fp8
                _hip.check(fn8(a8.data_ptr(), lw[wname + "_q"].data_ptr(), sa.data_ptr(),
                               lw[wname + "_s"].data_ptr(), bias, res, c.data_ptr(), rows, n, k, alpha, epi, st))
            else:
                _hip.check(L.oasr_gemm_bf16_ex(a.data_ptr(), lw[wname].data_ptr(), bias, res, c.data_ptr(), rows, n, k,
                                               alpha, epi, flags, st))
        return run

    def quant(a, k):   # fp8: the row quantiser that feeds a projection whose input a GEMM / attention produced
        def run():
            if fp8:
                _hip.check(L.oasr_quant_fp8_rows(a.data_ptr(), rows, k, a8.data_ptr(), sa.data_ptr(), st))
        return run

    def entry(name, ms, flops, shape):
        tf = flops / (ms * 1e-3) / 1e12
        return {"kernel": name + " " + shape, "bound": "mfma", "achieved": round(tf, 1), "peak": peak,
                "unit": "TFLOP/s", "frac": round(tf / peak, 4), "flop_per_launch": flops, "avg_launch_ms": round(ms, 4)}

    kern = "gemm_fp8_kernel" if fp8 else "gemm_bf16_kernel"
    up = gemm(xn, "ffn1_w1", lw["ffn1_b1"].data_ptr(), None, big, f, d, 1.0, _hip.EPI_BIAS_SWISH_BF16, _hip.GEMM_C_PACKED)
    ms_up = _time_launches(ln, up)
    down = gemm(big, "ffn1_w2", lw["ffn1_b2"].data_ptr(), resid.data_ptr(), resid, d, f, 0.5, _hip.EPI_RESIDUAL_F32,
                _hip.GEMM_A_PACKED)
    ms_down = _time_launches(lambda: (ln(), up(), quant(big, f)()), down)
    qkv = gemm(xn, "wqkv", lw["bqkv"].data_ptr(), None, big, 3 * d, d, 1.0, _hip.EPI_BIAS_BF16)
    ln(); qkv()

    def attn():
        _hip.check(L.oasr_attention_relkey(big.data_ptr(), lw["dist_emb"].data_ptr(), lengths.data_ptr(), ctx.data_ptr(),
                                           batch, t_pad, g.n_heads, g.head_dim, g.left_max, g.right_max, st))
    ms_attn = _time_launches(None, attn)
    wo = gemm(ctx, "wo", lw["bo"].data_ptr(), resid.data_ptr(), resid, d, d, 1.0, _hip.EPI_RESIDUAL_F32)
    ms_wo = _time_launches(lambda: (attn(), quant(ctx, d)()), wo)
    t_valid = int(lengths[0])
    attn_flops = batch * (4.0 * t_valid * t_valid * d + 2.0 * t_valid * g.n_dist * d)

    buf = engine._get_buffers(batch, pcm.shape[1])

    def logmel():
        _hip.check(L.oasr_logmel_f32(pcm.data_ptr(), pcm.shape[1], nv.data_ptr(), batch, buf["frame_cap"],
                                     engine._tables.data_ptr(), buf["logmel"].data_ptr(),
                                     buf["stats"].data_ptr(), st))

    ms_l = _time_launches(None, logmel)
    frames = _hip.num_frames(pcm.shape[1])
    lm_bytes = batch * (pcm.shape[1] * 4.0 + frames * 80 * 4.0)
    gbs = lm_bytes / (ms_l * 1e-3) / 1e9
    up_shape = "%dx%dx%d" % (rows, f, d)
    lm_shape = "%dx%d" % (batch, pcm.shape[1])
    main = entry(kern + "<BIAS_SWISH>", ms_up, 2.0 * rows * f * d, up_shape)
    main["traffic"] = measured_traffic("gemm_fp8_kernel<1>" if fp8 else "gemm_bf16_kernel<1>", up_shape)
    attn_e = entry("attention_relkey_kernel<%d>" % g.head_dim, ms_attn, attn_flops,
                   "%dx%dx%dx%d" % (batch, g.n_heads, t_valid, g.head_dim))
    attn_e["peak"], attn_e["frac"] = MFMA_BF16_PEAK_TFLOPS, round(attn_e["achieved"] / MFMA_BF16_PEAK_TFLOPS, 4)
    return {
        "roofline": main,
        "roofline_more": {
            "ffn_down": entry(kern + "<RESIDUAL_F32>", ms_down, 2.0 * rows * d * f, "%dx%dx%d" % (rows, d, f)),
            "dxd_residual": entry(kern + "<RESIDUAL_F32>", ms_wo, 2.0 * rows * d * d, "%dx%dx%d" % (rows, d, d)),
            "attention": attn_e,
            "logmel": {"kernel": "logmel_kernel " + lm_shape, "bound": "hbm", "achieved": round(gbs, 1),
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                       "traffic": measured_traffic("logmel_kernel", lm_shape),
                       "bytes_per_launch": lm_bytes, "avg_launch_ms": round(ms_l, 4)},
        },
    }


# Tolerances of the parity tests (tests/test_gpu_encoder.py, tests/test_gpu_fp8.py): per utterance, relative
# L2 and largest absolute difference of the encoder states against the CPU oracle.
PARITY_TOL = {"bf16": (2e-2, 0.15), "fp8": (3e-2, None)}     # the fp8 test states no absolute bound


def cpu_baseline(engine, weights, pcm_rows: np.ndarray, seconds: float, hip_hidden, n_layers: int, runs: int) -> dict:
    """The CPU oracle (float32 PyTorch restatement of the same geometry; with --fp8 its emulation of the
    device's quantisation) on this host's cores, on a bounded sample of the benchmarked batch: the rows
    `pcm_rows` copied back from the device tensor the step ran on, through the first `n_layers` blocks.
    One warm-up on a single row, then the median of `runs` timed passes.  The oracle's output is then
    compared with the device's states for the same rows and depth (`hip_hidden`)."""
    from oracle import conformer as OC
    from oracle import logmel as OL

    g = engine.geometry
    og = OC.Geometry(name=g.name, d_model=g.d_model, ffn_dim=g.ffn_dim, n_heads=g.n_heads, n_layers=n_layers)
    w = dict(weights)
    w["layers"] = weights["layers"][:n_layers]
    threads = torch.get_num_threads()
    out = {}

    def one_pass(x):
        t0 = time.perf_counter()
        feats, lens = OL.features_batch(x, None)
        out["hidden"] = OC.forward(torch.from_numpy(feats), torch.from_numpy(lens), w, og, fp8=engine.fp8)
        out["lens"] = lens
        return time.perf_counter() - t0

    if runs > 1:
        one_pass(pcm_rows[:1, : pcm_rows.shape[1] // 4])
    times = sorted(one_pass(pcm_rows) for _ in range(runs))
    dt = times[len(times) // 2] * (g.n_layers / float(n_layers))      # scaled to the full depth
    want, lens = out["hidden"], out["lens"]
    rel = mx = 0.0
    for i in range(pcm_rows.shape[0]):
        a, b = hip_hidden[i, : lens[i]].double(), want[i, : lens[i]].double()
        rel = max(rel, float((a - b).norm() / b.norm()))
        mx = max(mx, float((a - b).abs().max()))
    tol = PARITY_TOL["fp8" if engine.fp8 else "bf16"]
    depth = "all %d blocks" % n_layers if n_layers == g.n_layers else         "the first %d of %d blocks, time scaled by %d/%d" % (n_layers, g.n_layers, g.n_layers, n_layers)
    return {"value": round(pcm_rows.shape[0] * seconds / dt, 3), "unit": "x real time (audio-s/wall-s)",
            "cores": threads, "kind": "port",
            "sample": "rows 0..%d of the benchmarked batch (%.0f s each), float32 oracle (log-mel + %s), median of %d "
                      "pass(es), %.1f s of CPU work in all" % (pcm_rows.shape[0] - 1, seconds, depth, runs, sum(times)),
            "parity": {"rows": int(pcm_rows.shape[0]), "blocks": n_layers, "rel_l2": round(rel, 5), "max_abs": round(mx, 4),
                       "tol_rel_l2": tol[0], "tol_max_abs": tol[1], "ok": bool(rel <= tol[0] and (tol[1] is None or mx <= tol[1]))}}


def run_long_audio(args, world: int, rank: int) -> None:
    """BASELINE configs[3]: `--hours` of synthetic audio, planned into 30 s chunks by the C-ABI planner,
    contiguous chunk shards per rank, each rank running the host pipeline's chunk loop (batches of
    `--batch` assembled in page-locked memory, upload, device path, CTC stage) over its shard, then the
    path's ONE all-gather of records.  Each rank synthesises only its own shard (seed 1234 + chunk index)
    before the timed region; host memory -> device copies are inside it."""
    from omnilingual_asr import distributed as odist
    from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline, plan_chunks

    chunk_s = args.seconds
    total = int(round(args.hours * 3600 * 16000))
    bounds = plan_chunks(total, chunk_s, chunk_s)
    lo, hi = odist.shard_bounds(len(bounds), world, rank)
    base = int(bounds[lo][0]) if hi > lo else 0
    pcm = np.zeros((int(bounds[hi - 1][1]) - base) if hi > lo else 0, dtype=np.float32)
    for i in range(lo, hi):
        s, e = int(bounds[i][0]) - base, int(bounds[i][1]) - base
        pcm[s:e] = synth_batch(1, e - s, 1234 + i)[0].numpy()
    mine = bounds[lo:hi] - base
    pipe = GeminiASRPipeline(model=args.model, batch_size=args.batch, fp8=args.fp8)
    # warm-up: both slots of the full-batch shape, three batches each (the first replays of a freshly captured
    # graph carry one-off launch costs), and the tail shape (buffers, graphs, clocks)
    full = min(len(mine), 6 * args.batch)
    pipe._encode_chunks(pcm, mine[:full])
    tail = len(mine) % args.batch
    if tail:
        pipe._encode_chunks(pcm, mine[len(mine) - tail:])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    recs = pipe._encode_chunks(pcm, mine, first_idx=lo)
    for r in recs:                                   # back to file time
        r.start_sample += base
        r.end_sample += base
    if world > 1:
        cap = max(1, int(chunk_s * 16000) // 320)
        gathered = odist.unpack_records(odist.all_gather_records(
            odist.pack_records(recs, odist.max_shard(len(bounds), world), cap)))
    else:
        gathered = recs
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert [r.chunk_idx for r in gathered] == list(range(len(bounds))), "records missing or out of order"
    assert all(r.start_sample == int(b[0]) and r.end_sample == int(b[1]) for r, b in zip(gathered, bounds))
    if rank == 0:
        g = pipe.engine.geometry
        batches = -(-(hi - lo) // args.batch)
        print(json.dumps({
            "metric": "real-time factor (audio-s/wall-s), %.2f h auto-chunked into %.0f s chunks" % (args.hours, chunk_s),
            "value": round(total / 16000.0 / elapsed, 1), "unit": "x real time", "n_gpus": world,
            "steps": batches, "warmup": -(-full // args.batch) + (1 if tail else 0), "ms_per_step": round(elapsed / batches * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "fp8 (e4m3) projections, bf16 elsewhere" if args.fp8 else "bf16",
            "data": "synthetic (seeded noise + 440 Hz tone per chunk, seed 1234 + chunk index), random-init weights seed 0",
            "config": {"workload": "BASELINE configs[3]: %.2f h of 16 kHz audio -> %d chunks of %.0f s, host chunk loop "
                                   "(page-locked assembly, upload, log-mel + %s encoder + CTC stage per batch of %d), "
                                   "one all-gather of records" % (args.hours, len(bounds), chunk_s, g.name, args.batch),
                       "chunks": len(bounds), "chunks_per_rank": odist.max_shard(len(bounds), world),
                       "parallelism": "chunks sharded over %d rank(s), one all-gather of records" % world},
            "wall_s": round(elapsed, 4),
        }), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--model", default="G300")
    ap.add_argument("--fp8", action="store_true", help="e4m3 weights and activations in the large projections")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + --share-gpu rehearses the multi-rank path on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses device 0 (rehearsal only)")
    ap.add_argument("--hours", type=float, default=0.0,
                    help="long-audio mode: this many hours of audio through the chunker and the host chunk loop "
                         "(9.5 = BASELINE configs[3]) instead of the resident-batch step")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    if args.hours > 0:
        run_long_audio(args, world, rank)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from omnilingual_asr import distributed as odist
    from omnilingual_asr.engine_hip import HipEngine, pad_frames
    from omnilingual_asr.models.inference.asr_pipeline import ChunkRecord
    from omnilingual_asr import _hip

    from omnilingual_asr.weights import init_random_weights, resolve_geometry
    weights = init_random_weights(resolve_geometry(args.model), 0)
    engine = HipEngine(args.model, weights=weights, fp8=args.fp8)
    g = engine.geometry
    n = int(args.seconds * 16000)
    dev = engine.device
    pcm = synth_batch(args.batch, n, 1234 + rank).to(dev)
    nv = torch.full((args.batch,), n, dtype=torch.int32, device=dev)
    step = engine.step if args.no_graph else engine.step_graph

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        hidden, lengths = step(pcm, nv)
    if world > 1:   # warm-up of the record exchange too (first use sets up the collective's channels)
        lens_w = lengths.cpu().numpy()
        warm = [ChunkRecord(rank * args.batch + i, 0, n, int(lens_w[i])) for i in range(args.batch)]
        odist.all_gather_records(odist.pack_records(warm, args.batch))
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hidden, lengths = step(pcm, nv)
    # end of job: per-chunk records, exchanged by the path's single all-gather
    lens_h = lengths.cpu().numpy()
    recs = [ChunkRecord(rank * args.batch + i, 0, n, int(lens_h[i])) for i in range(args.batch)]
    if world > 1:
        gathered = odist.all_gather_records(odist.pack_records(recs, args.batch))
        assert gathered.shape[0] == world * args.batch
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        audio_s = world * args.batch * args.seconds * args.steps
        t_frames = _hip.num_stacked(n)
        flops_step = args.batch * encoder_flops_per_utt(g, t_frames)
        line = {
            "metric": "real-time factor (audio-s/wall-s), batch=%dx%.0fs@16kHz" % (args.batch, args.seconds),
            "value": round(audio_s / elapsed, 1),
            "unit": "x real time",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp8 (e4m3) projections, bf16 elsewhere" if args.fp8 else "bf16",
            "data": "synthetic (seeded noise + 440 Hz tone), random-init weights seed 0",
            "config": {"workload": (("BASELINE configs[1]" if args.batch == 1 else "BASELINE configs[2]" if args.batch == 32
                                     else "geometry of BASELINE configs[2] at another batch size") if g.name == "G300"
                                    else "BASELINE configs[4]" if args.fp8 and args.batch == 8 else
                                    "geometry of BASELINE configs[4]" + ("" if args.fp8 else " in bf16")) +
                                   (" fp8" if args.fp8 and g.name == "G300" else "") +
                                   ": batch=%d x %.0f s chunks @16 kHz, log-mel + %s encoder "
                                   "(%d blocks, d=%d, %.0fM params), %s" %
                                   (args.batch, args.seconds, g.name, g.n_layers, g.d_model, g.n_params() / 1e6,
                                    "eager launches" if args.no_graph else "hipGraph-captured step"),
                       "global_batch": world * args.batch, "frames_per_utterance": int(t_frames),
                       "parallelism": "chunks sharded over %d rank(s), one all-gather of records" % world},
            "encoder_tflops": round(flops_step / (elapsed / args.steps) / 1e12, 1),
            "encoder_mfma_frac": round(flops_step / (elapsed / args.steps) / 1e12 /
                                       (MFMA_FP8_PEAK_TFLOPS if args.fp8 else MFMA_BF16_PEAK_TFLOPS), 4),
            "encoder_mfma_peak": MFMA_FP8_PEAK_TFLOPS if args.fp8 else MFMA_BF16_PEAK_TFLOPS,
        }
        # The headline value is the resident-batch step.  Beside it: the same batch through the host chunk loop's
        # two-deep path (page-locked assembly is skipped, but the upload over PCIe, the step, the CTC stage and the
        # copies back to the host are inside) - what a caller of the pipeline sees per batch.
        if world == 1:
            host = pcm.cpu().numpy()
            n_valid = [n] * args.batch
            tickets = []
            for which in (0, 1):
                t, arr = engine.host_batch_slot(args.batch, n, which)
                arr[:, :n] = host
                tickets.append(t)
            # warm-up of both slots, three batches each: the first replays of a slot's freshly captured graph
            # carry one-off launch costs (about 1 ms each, visible as gaps in a kernel trace)
            for k in range(6):
                t = tickets[k & 1]
                if k >= 2:
                    engine.collect(t)
                engine.submit(t, n_valid)
            for t in tickets:
                engine.collect(t)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(args.steps):
                t = tickets[k & 1]
                if k >= 2:
                    engine.collect(t)
                engine.submit(t, n_valid)
            for t in tickets:
                engine.collect(t)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / args.steps
            line["with_upload_and_ctc"] = {"ms_per_batch": round(dt * 1e3, 3),
                                           "rtf": round(args.batch * args.seconds / dt, 1),
                                           "what": "upload from page-locked memory, step, vocabulary projection, greedy CTC "
                                                   "decode, results to the host; two batches in flight"}
        # kernel timings first: the device is still at its working clocks right after the timed steps
        if not args.no_kernel_timing:
            line.update(kernel_rooflines(engine, args.batch, pad_frames(t_frames), pcm, nv, hidden))
        parity_ok = True
        if world == 1 and not args.no_cpu_baseline:
            # bounded CPU sample: G300 = 4 rows at full depth; wider models = 1 row through 2 blocks
            small = g.d_model <= 1024
            n_rows, depth, runs = (min(4, args.batch), g.n_layers, 3) if small else (1, min(2, g.n_layers), 1)
            if depth == g.n_layers:
                hip_rows = hidden[:n_rows].float().cpu()          # the states the timed steps produced
            else:
                feats, lens_d = engine.features(pcm, nv)
                hip_rows = engine.encode_features(feats, lens_d, depth)[:n_rows].float().cpu()
            line["cpu_baseline"] = cpu_baseline(engine, weights, pcm[:n_rows].cpu().numpy(), args.seconds, hip_rows,
                                                depth, runs)
            parity_ok = line["cpu_baseline"]["parity"]["ok"]
        print(json.dumps(line), flush=True)
        if not parity_ok:
            print("bench.py: device states differ from the CPU oracle beyond the test tolerance", file=sys.stderr)
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit(3)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
