"""GPU tests of the boundary (Pipeline.transcribe with the default HIP engine) and size-independent
properties of the device path at the BASELINE batch size."""

import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_wav(path, seconds, rate=16000, seed=0):
    rng = np.random.default_rng(seed)
    x = (0.1 * rng.standard_normal(int(seconds * rate))).clip(-1, 1)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes((x * 32767).astype("<i2").tobytes())


def test_transcribe_gettysburg_on_hip_matches_oracle_records(golden_dir):
    from omnilingual_asr import Pipeline, audio
    from omnilingual_asr.weights import GEOMETRIES, init_random_weights
    from oracle import conformer as OC
    from oracle.engine import OracleEngine

    pipe = Pipeline(model="GTINY", seed=4)
    progress = []
    segs = pipe.transcribe(str(golden_dir / "gettysburg.wav"), progress_callback=lambda s, i: progress.append((s, i)))
    assert progress == [("uploading", 0), ("transcribing", 1), ("processing", 2), ("done", 3)]
    assert len(segs) == 1 and segs[0].start == 0.0 and segs[0].end == 281233 / 16000.0
    # same audio through the CPU oracle: frame count bit-exact, activity within the bf16 tolerance
    pcm = audio.load_audio_16k(golden_dir / "gettysburg.wav")
    g = GEOMETRIES["GTINY"]
    og = OC.Geometry(name=g.name, d_model=g.d_model, ffn_dim=g.ffn_dim, n_heads=g.n_heads, n_layers=g.n_layers)
    hidden, lens = OracleEngine(init_random_weights(g, 4), og).encode_batch(pcm[None, :], [len(pcm)])
    rec = pipe.gemini._encode_chunks(pcm, np.array([[0, len(pcm)]]))[0]
    assert rec.n_frames == int(lens[0]) == 878
    assert rec.activity == pytest.approx(float(hidden[0, : lens[0]].abs().mean()), rel=5e-3)


def test_chunked_transcribe_on_hip(tmp_path):
    from omnilingual_asr import Pipeline

    wav = tmp_path / "long.wav"
    _write_wav(wav, 100.5)
    pipe = Pipeline(model="GTINY", batch_size=3)
    segs = pipe.transcribe(str(wav))
    assert [s.start for s in segs] == [0.0, 30.0, 60.0, 90.0]
    assert [s.end for s in segs] == [30.0, 60.0, 90.0, 100.5]
    recs = pipe.gemini._encode_chunks(np.zeros(16000 * 100 + 8000, dtype=np.float32),
                                      np.array([[0, 480000], [480000, 960000], [960000, 1440000], [1440000, 1608000]]))
    assert [r.n_frames for r in recs] == [1499, 1499, 1499, 524]


def test_full_size_batch_properties():
    """Batch 32 x 30 s, G300 with 2 blocks (the per-block work is what scales; depth does not change
    the properties): (1) an utterance's states do not depend on its batch neighbours or position,
    (2) junk beyond an utterance's valid samples does not change its valid frames."""
    import dataclasses
    from omnilingual_asr.engine_hip import HipEngine
    from omnilingual_asr.weights import GEOMETRIES
    from tests.test_gpu_ops import synth_pcm

    g = dataclasses.replace(GEOMETRIES["G300"], n_layers=2)
    eng = HipEngine(g, seed=0)
    dev = torch.device("cuda")
    n = 480000
    pcm = synth_pcm(32, n, seed=21).to(dev)
    n_valid = [n] * 32
    n_valid[5], n_valid[31] = 250000, 400
    nv = torch.tensor(n_valid, dtype=torch.int32, device=dev)
    hidden, lengths = eng.step(pcm, nv)
    hidden = hidden.clone()
    lens = lengths.cpu().tolist()
    assert lens[0] == 1499 and lens[5] == 780 and lens[31] == 0
    assert torch.isfinite(hidden).all()
    for i in (0, 5, 17):
        alone, l1 = eng.step(pcm[i:i + 1].contiguous(), nv[i:i + 1].contiguous())
        assert int(l1[0]) == lens[i]
        torch.testing.assert_close(alone[0, : lens[i]], hidden[i, : lens[i]], atol=0, rtol=0)
    junk = pcm.clone()
    junk[5, 250000:] = 1e3
    h2, _ = eng.step(junk, nv)
    torch.testing.assert_close(h2[5, : lens[5]], hidden[5, : lens[5]], atol=0, rtol=0)


def test_two_deep_chunk_loop_equals_batch_at_a_time():
    """The submit/collect loop (uploads overlapped with device work, two slots) returns exactly the
    records of one synchronous `transcribe_batch` call per batch: 7 chunks of ragged length in batches of
    2 (three full batches, then a tail batch of another shape)."""
    from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline, _mean_abs

    rng = np.random.default_rng(3)
    pcm = (0.1 * rng.standard_normal(16000 * 40)).astype(np.float32)
    cuts = [0, 96000, 192000, 250000, 346000, 442000, 538000, 600000]
    bounds = np.array(list(zip(cuts[:-1], cuts[1:])), dtype=np.int64)
    pipe = GeminiASRPipeline(model="GTINY", seed=2, batch_size=2)
    for _ in range(2):                      # second pass: slots and graphs are reused
        recs = pipe._encode_chunks(pcm, bounds, first_idx=10)
        assert [r.chunk_idx for r in recs] == list(range(10, 17))
        for lo in range(0, 7, 2):
            part = bounds[lo: lo + 2]
            width = int(max(e - s for s, e in part))
            batch = np.zeros((len(part), width), dtype=np.float32)
            for i, (s, e) in enumerate(part):
                batch[i, : e - s] = pcm[s:e]
            hidden, lengths, toks = pipe.engine.transcribe_batch(batch, [int(e - s) for s, e in part])
            acts = _mean_abs(hidden, lengths)
            for i in range(len(part)):
                r = recs[lo + i]
                assert (r.start_sample, r.end_sample, r.n_frames) == (part[i][0], part[i][1], int(lengths[i]))
                assert r.activity == acts[i]
                np.testing.assert_array_equal(r.tokens, toks[i][0])
                np.testing.assert_array_equal(r.frames, toks[i][1])


@pytest.mark.parametrize("n,end", [(0, 1.0), (100, 100 / 16000.0), (400, 0.025), (720, 0.045)])
def test_audio_shorter_than_the_first_stacked_frame_on_hip(tmp_path, n, end):
    """Empty and sub-frame files through the HIP engine: zero encoder frames, no exception, empty text."""
    from omnilingual_asr import Pipeline

    path = tmp_path / "short.wav"
    rng = np.random.default_rng(n)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((rng.standard_normal(n) * 3000).astype("<i2").tobytes())
    pipe = Pipeline(model="GTINY", seed=1)
    segs = pipe.transcribe(str(path), word_timestamps=True)
    assert len(segs) == 1 and (segs[0].start, segs[0].end) == (0.0, end)
    if n < 720:
        assert segs[0].text == "" and segs[0].words == []


def test_bench_long_audio_mode_runs(tmp_path):
    """`bench.py --hours` (the auto-chunked long-audio configuration) end to end in a child process:
    planner, shard synthesis, two-deep chunk loop, record checks, one JSON line."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--hours", "0.06", "--model", "GTINY", "--batch", "4"],
                         capture_output=True, text=True, timeout=180, cwd=str(root))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["config"]["chunks"] == 8 and line["n_gpus"] == 1 and line["value"] > 0      # 216 s -> 7 x 30 s + 6 s
    assert line["unit"] == "x real time" and line["steps"] == 2


def test_four_threads_share_one_pipeline(tmp_path):
    """The reference's server runs up to four `transcribe()` calls at once on one shared instance
    (ref:workflows/wav2elan_web/app.py:383-399): results must equal the one-at-a-time results."""
    from concurrent.futures import ThreadPoolExecutor
    from omnilingual_asr import Pipeline

    paths = []
    for i, (seconds, rate) in enumerate([(3.0, 16000), (41.0, 16000), (7.5, 22050), (65.2, 8000), (0.5, 16000), (33.3, 44100)]):
        p = tmp_path / f"f{i}.wav"
        _write_wav(p, seconds, rate=rate, seed=20 + i)
        paths.append(str(p))
    pipe = Pipeline(model="GTINY", seed=5, batch_size=2)
    key = lambda segs: [(s.start, s.end, s.text, None if s.words is None else [(w.word, w.start, w.end) for w in s.words]) for s in segs]
    want = [key(pipe.transcribe(p, word_timestamps=True)) for p in paths]
    assert [len(w) for w in want] == [1, 2, 1, 3, 1, 2]
    for _ in range(2):
        with ThreadPoolExecutor(max_workers=4) as ex:
            got = list(ex.map(lambda p: key(pipe.transcribe(p, word_timestamps=True)), paths))
        assert got == want
