"""N > 1 path on CPU: two gloo ranks shard the chunk plan, each encodes its shard (CPU oracle engine
injected), one all-gather of records, identical merged result on both ranks and equal to the
single-process result."""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _write_wav(path, seconds, rate=16000, seed=0):
    import wave
    rng = np.random.default_rng(seed)
    x = (0.1 * rng.standard_normal(int(seconds * rate))).clip(-1, 1)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes((x * 32767).astype("<i2").tobytes())


def _make_pipeline(group=None):
    for p in (str(ROOT), str(ROOT / "omnilingual-asr_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from omnilingual_asr import Pipeline
    from omnilingual_asr.weights import GEOMETRIES, init_random_weights
    from oracle import conformer as OC
    from oracle.engine import OracleEngine

    g = GEOMETRIES["GTINY"]
    og = OC.Geometry(name=g.name, d_model=g.d_model, ffn_dim=g.ffn_dim, n_heads=g.n_heads, n_layers=1)
    w = init_random_weights(g, 0)
    w["layers"] = w["layers"][:1]
    eng = OracleEngine(w, og)
    return Pipeline(engine=eng, chunk_seconds=1.0, min_chunking_seconds=1.5, batch_size=2, process_group=group), eng


def _worker(rank, world, port, wav, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipe, eng = _make_pipeline(dist.group.WORLD)
    progress = []
    segs = pipe.transcribe(wav, progress_callback=lambda s, i: progress.append((s, i)))
    torch.save({"segs": [(s.start, s.end, s.speaker, s.text) for s in segs], "progress": progress,
                "engine_calls": eng.calls}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_single_process(tmp_path):
    wav = tmp_path / "long.wav"
    _write_wav(wav, 7.4)
    single, eng = _make_pipeline(None)
    want = [(s.start, s.end, s.speaker, s.text) for s in single.transcribe(str(wav))]
    assert len(want) == 8 and eng.calls == 4
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(wav), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["segs"] == want and r1["segs"] == want           # every rank holds the merged result
    assert r0["engine_calls"] == 2 and r1["engine_calls"] == 2  # 4 chunks each, batches of 2: work is sharded
    assert r0["progress"] == r1["progress"] == [("uploading", 0), ("transcribing", 1), ("processing", 2), ("done", 3)]


def test_record_pack_roundtrip_and_order():
    from omnilingual_asr import distributed as D
    from omnilingual_asr.models.inference.asr_pipeline import ChunkRecord

    a = [ChunkRecord(4, 400, 500, 3, 0.25), ChunkRecord(5, 500, 600, 2, 1e-7)]
    b = [ChunkRecord(0, 0, 100, 7, 3.5)]
    gathered = torch.cat([D.pack_records(a, 3), D.pack_records(b, 3)])
    got = D.unpack_records(gathered)
    assert [r.chunk_idx for r in got] == [0, 4, 5]
    assert got[1] == a[0] and got[2] == a[1] and got[0] == b[0]   # float64 activity survives bit-exactly
    assert D.max_shard(1140, 8) == 143
