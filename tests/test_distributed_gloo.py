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

    a = [ChunkRecord(4, 400, 500, 3), ChunkRecord(5, 500, 600, 2)]
    b = [ChunkRecord(0, 0, 100, 7)]
    gathered = torch.cat([D.pack_records(a, 3), D.pack_records(b, 3)])
    got = D.unpack_records(gathered)
    assert [r.chunk_idx for r in got] == [0, 4, 5]
    assert got[1] == a[0] and got[2] == a[1] and got[0] == b[0]
    assert D.max_shard(1140, 8) == 143


class _PlanEngine:
    """Stands in for the device: frame counts from the sample counts, tokens that name the chunk (its
    first sample in 30 s units), so the merged order and the token transport can be checked without
    running an encoder 1 140 times."""

    def __init__(self):
        self.batches = []

    def transcribe_batch(self, pcm, n_valid):
        n_valid = np.asarray(n_valid)
        self.batches.append(len(n_valid))
        lengths = np.array([max(0, (int(n) - 400) // 160 + 1) // 2 if n >= 400 else 0 for n in n_valid], dtype=np.int32)
        toks = []
        for i in range(len(n_valid)):
            tag = int(round(float(pcm[i, 0])))             # each chunk's first sample holds its index
            toks.append((np.array([tag % 29 + 2, tag // 29 + 2, 5], dtype=np.int32), np.array([0, 1, 7], dtype=np.int32)))
        return None, lengths, toks


def _worker8(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    for p in (str(ROOT), str(ROOT / "omnilingual-asr_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from omnilingual_asr import distributed as D
    from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline, plan_chunks

    calls = []
    real = dist.all_gather
    dist.all_gather = lambda *a, **k: (calls.append("all_gather"), real(*a, **k))[1]
    chunk, n_chunks = 4000, 1140                  # the 9.5 h plan's shape at 0.25 s per chunk
    bounds = plan_chunks(chunk * n_chunks - 1500, chunk / 16000.0, 0.0)
    pcm = np.zeros(chunk * n_chunks, dtype=np.float32)
    pcm[::chunk] = np.arange(n_chunks, dtype=np.float32)
    eng = _PlanEngine()
    pipe = GeminiASRPipeline(engine=eng, chunk_seconds=chunk / 16000.0, min_chunking_seconds=0.0, batch_size=32)
    recs = D.encode_sharded(pipe, pcm, bounds, dist.group.WORLD)
    dist.all_gather = real
    torch.save({"n_bounds": len(bounds), "batches": eng.batches, "calls": calls,
                "idx": [r.chunk_idx for r in recs], "starts": [r.start_sample for r in recs],
                "ends": [r.end_sample for r in recs], "frames": [r.n_frames for r in recs],
                "tags": [int((r.tokens[0] - 2) + 29 * (r.tokens[1] - 2)) for r in recs],
                "tok_frames": [r.frames.tolist() for r in recs[:3]]}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_shard_the_1140_chunk_plan_with_one_all_gather(tmp_path):
    """World size 8 on the 1 140-chunk plan (shards of 143/142 chunks, four full batches of 32 and a tail
    of 15 or 14 per rank, token-carrying records): every rank ends with all records in chunk order, and
    the data path issues exactly one all_gather."""
    port = _free_port()
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    outs = [torch.load(tmp_path / f"r{r}.pt") for r in range(8)]
    for r, o in enumerate(outs):
        assert o["n_bounds"] == 1140 and o["calls"] == ["all_gather"]
        assert o["batches"] == [32, 32, 32, 32, 15 if r < 4 else 14]
        assert o["idx"] == list(range(1140)) and o["tags"] == list(range(1140))
        assert o["starts"] == [4000 * i for i in range(1140)]
        assert o["ends"][-1] == 4000 * 1140 - 1500 and o["ends"][:-1] == [4000 * (i + 1) for i in range(1139)]
        assert o["frames"][0] == 11 and o["frames"][-1] == ((2500 - 400) // 160 + 1) // 2
        assert o["tok_frames"] == [[0, 1, 7]] * 3


def _write_pcm_wav(path, seconds, rate, channels, width, seed=0):
    """Integer PCM WAV of any width (1-4 bytes) by hand: the `wave` module is fine too, but this keeps 24 / 32 bits simple."""
    import struct
    rng = np.random.default_rng(seed)
    n = int(seconds * rate)
    t = np.arange(n) / rate
    x = 0.2 * np.sin(2 * np.pi * 220.0 * t)[:, None] + 0.05 * rng.standard_normal((n, channels))
    full = 1 << (8 * width - 1)
    v = np.clip(np.round(x * (full - 1)), -full, full - 1).astype(np.int64)
    if width == 1:
        body = (v + 128).astype(np.uint8).tobytes()
    else:
        b = np.zeros((n, channels, width), dtype=np.uint8)
        for k in range(width):
            b[:, :, k] = (v >> (8 * k)) & 0xFF
        body = b.tobytes()
    fmt = struct.pack("<HHIIHH", 1, channels, rate, rate * channels * width, channels * width, 8 * width)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt
                + b"data" + struct.pack("<I", len(body)) + body)


def _worker_spans(rank, world, port, files, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipe, eng = _make_pipeline(dist.group.WORLD)
    out = {}
    for f in files:
        progress = []
        segs = pipe.transcribe(f, progress_callback=lambda s, i: progress.append((s, i)))
        out[f] = {"segs": [(s.start, s.end, s.speaker, s.text) for s in segs], "progress": progress,
                  "read": pipe.gemini.decoded_source_samples}
    torch.save(out, os.path.join(out_dir, f"spans{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_read_only_their_shards_of_wav_flac_and_ogg_files(tmp_path, golden_dir):
    """Under a process group a rank reads and converts only the source samples of its own shard of chunks: a 48 kHz
    stereo 16-bit WAV, a 44.1 kHz mono 24-bit WAV (ratio 160 / 441, so the converter's reach matters) and the
    reference's FLAC upload (frames decoded by range), an Ogg Vorbis stream (packets decoded by range) and the FLAC
    upload wrapped in Ogg pages (unwrapped by every rank, frames decoded by range).  Each rank
    touches about half of every file (its chunks plus
    the converter's reach), and the merged result on both ranks equals the single-process result, which loads the
    file whole."""
    from omnilingual_asr import audio
    files = [tmp_path / "stereo48.wav", tmp_path / "mono441.wav", golden_dir / "reference_upload.flac", tmp_path / "stereo22.ogg",
             tmp_path / "flac_inside.ogg"]
    _write_pcm_wav(files[0], 7.4, 48000, 2, 2, seed=1)
    _write_pcm_wav(files[1], 6.3, 44100, 1, 3, seed=2)
    # an Ogg Vorbis stream from the test writer (coupled stereo, both block sizes), its packets repeated to ~ 9 s:
    # a rank decodes only the packets that cover its shard
    import vorbis_tools as vt
    rng = np.random.default_rng(6)
    enc = vt.encode(0.1 * rng.standard_normal((22050 * 3, 2)), vt.Config(channels=2, rate=22050, pattern="LLSL", seed=6,
                                                                      residue_type=2, coupled=True, bs0=128, bs1=1024))
    files[3].write_bytes(vt.repeat_stream(enc, 3))
    import flac_tools as ft
    files[4].write_bytes(ft.wrap_in_ogg(files[2].read_bytes()))
    totals = [int(7.4 * 48000), int(6.3 * 44100), 312060, audio.vorbis_info(files[3])[2], 312060]
    single, _ = _make_pipeline(None)
    want = {str(f): [(s.start, s.end, s.speaker, s.text) for s in single.transcribe(str(f))] for f in files}
    assert [len(want[str(f)]) for f in files] == [8, 7, 20, -(-totals[3] // 22050), 20]
    assert want[str(files[4])] == want[str(files[2])]
    port = _free_port()
    mp.spawn(_worker_spans, args=(2, port, [str(f) for f in files], str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"spans{k}.pt") for k in range(2)]
    for f, total in zip(files, totals):
        a, b = r[0][str(f)], r[1][str(f)]
        assert a["segs"] == want[str(f)] and b["segs"] == want[str(f)], f
        assert a["progress"] == b["progress"] == [("uploading", 0), ("transcribing", 1), ("processing", 2), ("done", 3)]
        # rank 0 holds the first ceil(n / 2) one-second chunks, rank 1 the rest (the last one short); each reads
        # those source samples and the converter's reach (at most 1 500 samples a side), nothing more
        n_chunks = len(want[str(f)])
        rate = {0: 48000, 1: 44100, 2: 16000, 3: 22050, 4: 16000}[files.index(f)]
        own0 = (n_chunks + 1) // 2 * rate
        reach = 1500 if rate == 22050 else 1000      # (320 / 441: the halo is a whole number of 441-sample periods)
        assert own0 <= a["read"] <= own0 + reach, (f, a["read"], own0)
        assert total - own0 <= b["read"] <= total - own0 + reach, (f, b["read"], total - own0)


def _worker8_wav(rank, world, port, wav, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    for p in (str(ROOT), str(ROOT / "omnilingual-asr_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline

    calls = []
    real = dist.all_gather
    dist.all_gather = lambda *a, **k: (calls.append("all_gather"), real(*a, **k))[1]
    eng = _PlanEngine()
    pipe = GeminiASRPipeline(engine=eng, chunk_seconds=0.25, min_chunking_seconds=0.0, batch_size=32,
                             process_group=dist.group.WORLD)
    res = pipe.transcribe_chunked(wav)
    dist.all_gather = real
    torch.save({"calls": calls, "batches": eng.batches, "read": pipe.decoded_source_samples,
                "starts": [s.start for s in res.segments], "ends": [s.end for s in res.segments]},
               os.path.join(out_dir, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_each_read_an_eighth_of_a_48khz_stereo_wav(tmp_path):
    """The 1 140-chunk plan (0.25 s chunks) on a 48 kHz stereo file of 285 s, world size 8: every rank reads its
    143 or 142 chunks' worth of source samples plus the converter's reach per segment (3 segments of up to 64
    chunks), nothing else; one all_gather; the merged plan is whole on every rank."""
    wav = tmp_path / "long48.wav"
    _write_pcm_wav(wav, 285.0 - 1500 / 16000.0, 48000, 2, 2, seed=4)
    port = _free_port()
    mp.spawn(_worker8_wav, args=(8, port, str(wav), str(tmp_path)), nprocs=8, join=True)
    outs = [torch.load(tmp_path / f"w{r}.pt") for r in range(8)]
    per_chunk = 12000                                   # source samples per channel in 0.25 s
    for r, o in enumerate(outs):
        n_chunks = 143 if r < 4 else 142
        assert o["calls"] == ["all_gather"]
        assert o["batches"] == [32, 32, 32, 32, 15 if r < 4 else 14]
        assert len(o["starts"]) == 1140 and o["starts"] == [0.25 * i for i in range(1140)]
        assert abs(o["ends"][-1] - (285.0 - 1500 / 16000.0)) < 1e-3
        own = n_chunks * per_chunk - (4500 if r == 7 else 0)
        assert own <= o["read"] <= own + 3 * 2 * 200, (r, o["read"], own)     # + the reach (<= 200 samples a side)
    assert sum(o["read"] for o in outs) < 1.01 * 48000 * 285
