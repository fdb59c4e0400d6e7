#!/usr/bin/env python3
"""Rehearsal of the sharded path on a compressed file: run under torch.distributed.run with N ranks (gloo; the ranks
may share GPU 0).  Every rank calls the public long-audio entry on the same MP3; each decodes only the samples of its
own shard of chunks, the records cross in the one all-gather, and every rank checks the merged result against a
pipeline without a process group.  Usage: python -m torch.distributed.run --nproc-per-node 2 tools/sharded_mp3.py"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import torch
import torch.distributed as dist
from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
path = ROOT / "tests" / "golden" / "reference_demo_40s.mp3"
kw = dict(model="GTINY", seed=4, batch_size=2, chunk_seconds=2.5, min_chunking_seconds=2.5)
sharded = GeminiASRPipeline(process_group=dist.group.WORLD, **kw)
got = sharded.transcribe_chunked(path, word_timestamps=True)
share = sharded.decoded_source_samples
alone = GeminiASRPipeline(**kw)
want = alone._encode_file(alone._load_audio(path), True)
same = len(got.segments) == len(want.segments) and all(
    (a.start, a.end, a.text, a.words) == (b.start, b.end, b.text, b.words) for a, b in zip(got.segments, want.segments))
total = 1531 * 1152 - 576 - 1190
ok = same and share < total * (1.0 / world + 0.05) and any(s.text for s in got.segments)
print("rank %d of %d: %d segments, equal to the unsharded result: %s, decoded %d of %d source samples (%.0f %%)"
      % (rank, world, len(got.segments), same, share, total, 100.0 * share / total), flush=True)
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if int(flag) == 1 else 1)
