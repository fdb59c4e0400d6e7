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
    path = ROOT / "profiles" / "r01_traffic.json"
    if not path.exists():
        return None
    entry = json.loads(path.read_text()).get(kernel)
    if not entry or entry.get("shape") != shape:
        return None
    return entry["traffic_bytes_per_launch"]


def kernel_rooflines(engine, batch: int, t_pad: int, pcm, nv, hidden) -> dict:
    """HIP-event timings of the two kernels that bound the path: the FFN up-projection GEMM
    (MFMA-bound, the dominant kernel) and the log-mel kernel (HBM-bound)."""
    from omnilingual_asr import _hip
    g = engine.geometry
    L = _hip.lib()
    rows = int(L.oasr_encoder_rows(batch, t_pad))
    dev = engine.device
    lw = engine._packed["layers"][0]
    # Same operands as in the step: the GEMM reads the bf16 output of the LayerNorm kernel that runs
    # right before it (so its A operand is as cache-warm as in the encoder) and the block-0 weights.
    x = torch.zeros((rows, g.d_model), dtype=torch.float32, device=dev)
    x[:hidden.numel() // g.d_model] = hidden.reshape(-1, g.d_model)
    a = torch.empty((rows, g.d_model), dtype=torch.bfloat16, device=dev)
    w, bias = lw["ffn1_w1"], lw["ffn1_b1"]
    c = torch.empty((rows, g.ffn_dim), dtype=torch.bfloat16, device=dev)
    st = _hip.current_stream_ptr()
    iters = 20
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters + 1)]
    for e0, e1 in ev:
        _hip.check(L.oasr_layernorm(x.data_ptr(), lw["ffn1_ln_w"].data_ptr(), lw["ffn1_ln_b"].data_ptr(),
                                    a.data_ptr(), rows, g.d_model, g.d_model, 1e-5, 1, 0, st))
        e0.record()
        _hip.check(L.oasr_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), None, c.data_ptr(), rows,
                                    g.ffn_dim, g.d_model, 1.0, _hip.EPI_BIAS_SWISH_BF16, st))
        e1.record()
    torch.cuda.synchronize()
    ms = sum(e0.elapsed_time(e1) for e0, e1 in ev[1:]) / iters
    flops = 2.0 * rows * g.ffn_dim * g.d_model
    tf = flops / (ms * 1e-3) / 1e12
    buf = engine._get_buffers(batch, pcm.shape[1])

    def logmel():
        _hip.check(L.oasr_logmel_f32(pcm.data_ptr(), pcm.shape[1], nv.data_ptr(), batch, buf["frame_cap"],
                                     engine._tables.data_ptr(), buf["logmel"].data_ptr(),
                                     buf["stats"].data_ptr(), st))

    ms_l = time_events(logmel, 20)
    frames = _hip.num_frames(pcm.shape[1])
    lm_bytes = batch * (pcm.shape[1] * 4.0 + frames * 80 * 4.0)
    gbs = lm_bytes / (ms_l * 1e-3) / 1e9
    gemm_shape = "%dx%dx%d" % (rows, g.ffn_dim, g.d_model)
    lm_shape = "%dx%d" % (batch, pcm.shape[1])
    return {
        "roofline": {"kernel": "gemm_bf16_kernel<BIAS_SWISH> " + gemm_shape,
                     "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                     "traffic": measured_traffic("gemm_bf16_kernel<1>", gemm_shape),
                     "flop_per_launch": flops, "avg_launch_ms": round(ms, 4)},
        "roofline_logmel": {"kernel": "logmel_kernel " + lm_shape, "bound": "hbm", "achieved": round(gbs, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                            "traffic": measured_traffic("logmel_kernel", lm_shape),
                            "bytes_per_launch": lm_bytes, "avg_launch_ms": round(ms_l, 4)},
    }


def cpu_baseline(seconds: float, seed: int, utterances: int = 4, runs: int = 3) -> dict:
    """The CPU oracle (float32 PyTorch restatement of the same geometry) on this host's cores, on a
    bounded sample of the same synthetic workload: the first `utterances` rows of the batch, one
    warm-up on a single utterance, then the median of `runs` timed passes."""
    from omnilingual_asr.weights import GEOMETRIES, init_random_weights
    from oracle import conformer as OC
    from oracle import logmel as OL

    g = GEOMETRIES["G300"]
    w = init_random_weights(g, 0)
    pcm = synth_batch(utterances, int(seconds * 16000), seed).numpy()
    threads = torch.get_num_threads()

    def one_pass(x):
        t0 = time.perf_counter()
        feats, lens = OL.features_batch(x, None)
        OC.forward(torch.from_numpy(feats), torch.from_numpy(lens), w, OC.G300)
        return time.perf_counter() - t0

    one_pass(pcm[:1])
    times = sorted(one_pass(pcm) for _ in range(runs))
    dt = times[len(times) // 2]
    return {"value": round(utterances * seconds / dt, 3), "unit": "x real time (audio-s/wall-s)",
            "cores": threads, "kind": "port",
            "sample": "%d x %.0f s utterances of the batch, float32 oracle (log-mel + 12 blocks), median of %d "
                      "passes = %.1f s wall, total %.1f s" % (utterances, seconds, runs, dt, sum(times))}


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
    # warm-up: both slots of the full-batch shape and the tail shape (buffers, graphs, clocks)
    full = min(len(mine), 2 * args.batch)
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
            "steps": batches, "warmup": 2 + (1 if tail else 0), "ms_per_step": round(elapsed / batches * 1e3, 3),
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

    engine = HipEngine(args.model, seed=0, fp8=args.fp8)
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
            "config": {"workload": ("BASELINE configs[2]" if g.name == "G300" else "BASELINE configs[4]" if args.fp8 else
                                    "geometry of BASELINE configs[4] in bf16") + (" fp8" if args.fp8 and g.name == "G300" else "") +
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
        if not args.no_kernel_timing:
            line.update(kernel_rooflines(engine, args.batch, pad_frames(t_frames), pcm, nv, hidden))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.seconds, 1234)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
