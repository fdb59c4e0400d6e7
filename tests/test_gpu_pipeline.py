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
    # same audio through the CPU oracle: frame count bit-exact
    pcm = audio.load_audio_16k(golden_dir / "gettysburg.wav")
    g = GEOMETRIES["GTINY"]
    og = OC.Geometry(name=g.name, d_model=g.d_model, ffn_dim=g.ffn_dim, n_heads=g.n_heads, n_layers=g.n_layers)
    hidden, lens = OracleEngine(init_random_weights(g, 4), og).encode_batch(pcm[None, :], [len(pcm)])
    rec = pipe.gemini._encode_chunks(pcm, np.array([[0, len(pcm)]]))[0]
    assert rec.n_frames == int(lens[0]) == 878


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


def test_int16_samples_kept_as_read_give_the_same_tokens(tmp_path):
    """A mono 16-bit WAV at 16 kHz stays int16 until batch assembly (scaled while copied); the device
    results equal those of the float samples bit for bit."""
    from omnilingual_asr import Pipeline, audio

    wav = tmp_path / "speechlike.wav"
    _write_wav(wav, 70.25, seed=3)
    pipe = Pipeline(model="GTINY", batch_size=2).gemini
    kept = pipe._load_audio(wav)
    assert kept.dtype == np.int16
    flt = audio.load_audio_16k(wav)
    bounds = np.array([[0, 480000], [480000, 960000], [960000, kept.shape[0]]])
    a, b = pipe._encode_chunks(kept, bounds), pipe._encode_chunks(flt, bounds)
    assert len(a) == len(b) == 3
    for ra, rb in zip(a, b):
        assert ra.n_frames == rb.n_frames
        assert np.array_equal(ra.tokens, rb.tokens) and np.array_equal(ra.frames, rb.frames)
    assert sum(len(r.tokens) for r in a) > 0


def test_reference_chunk_constants_run_on_the_device(tmp_path, golden_dir):
    """The reference's own plan — 300 s chunks behind the 360 s gate (ref gemini_pipeline.py:217-219,
    :243-310, :713-714) — through the HIP engine: a 601 s file gives the chunk starts the reference
    produced for that duration (golden plan), full chunks have 14 999 encoder frames (t_pad 15 008), and
    a chunk's states and tokens do not depend on its batch neighbours or on the padding (bit-exact)."""
    import json
    from omnilingual_asr import Pipeline, audio

    gold = json.loads((golden_dir / "reference_host.json").read_text())
    plan = next(p for p in gold["chunk_plans"] if p["chunk"] == 300.0 and p["duration"] == 601.0)
    assert plan["use_chunking"] and plan["starts"] == [0.0, 300.0, 600.0]
    wav = tmp_path / "ten_minutes.wav"
    _write_wav(wav, 601.0, seed=9)
    pipe = Pipeline(model="GTINY", seed=3, chunk_seconds=300, min_chunking_seconds=360, batch_size=2)
    segs = pipe.transcribe(str(wav), word_timestamps=True)
    assert [s.start for s in segs] == plan["starts"]
    assert [s.end for s in segs] == [300.0, 600.0, 601.0]
    assert all(s.text for s in segs[:2])
    assert all(w.start >= s.start and w.end <= s.end + 0.02 for s in segs for w in s.words)
    eng = pipe.gemini.engine
    pcm = audio.load_audio_16k(wav)
    bounds = np.array([[0, 4800000], [4800000, 9600000], [9600000, 9616000]])
    recs = pipe.gemini._encode_chunks(pcm, bounds)
    assert [r.n_frames for r in recs] == [14999, 14999, 49]
    assert [len(r.tokens) > 0 for r in recs] == [True, True, True]
    # chunk 1 alone against chunk 1 next to chunk 0 (batch of 2), and against itself padded with junk
    both = np.stack([pcm[:4800000], pcm[4800000:9600000]])
    h2, l2, t2 = eng.transcribe_batch(both, [4800000, 4800000])
    assert h2.shape[1] == 15008 and list(l2) == [14999, 14999]
    h2 = h2[1, :14999].clone()
    h1, l1, t1 = eng.transcribe_batch(both[1:2], [4800000])
    torch.testing.assert_close(h1[0, :14999], h2, atol=0, rtol=0)
    np.testing.assert_array_equal(t1[0][0], t2[1][0])
    np.testing.assert_array_equal(t1[0][0], recs[1].tokens)
    short = both.copy()
    short[0, 160000:] = 1e3                                   # 10 s valid, junk behind it
    h3, l3, _ = eng.transcribe_batch(short, [160000, 4800000])
    assert list(l3) == [499, 14999]
    torch.testing.assert_close(h3[1, :14999], h2, atol=0, rtol=0)
    h4, _, _ = eng.transcribe_batch(both[:1, :160000], [160000])
    torch.testing.assert_close(h4[0, :499], h3[0, :499], atol=0, rtol=0)


def test_every_sample_count_fits_its_bucket_and_statistics_cover_all_frames():
    """(1) a batch assembled at `bucket_samples(n)` is wide enough for n at the lengths where the frame
    count is odd and the stacked count a multiple of 32; (2) at such a length the utterance's features do
    not depend on what it is batched with (the normalisation runs over all its frames)."""
    from omnilingual_asr.engine_hip import HipEngine
    from tests.test_gpu_ops import synth_pcm

    eng = HipEngine("GTINY", seed=1)
    n = 400 + 160 * 64                      # 65 frames -> 32 stacked, one frame left over
    assert eng.bucket_samples(n) >= n
    pcm = synth_pcm(2, 48000, seed=8)
    alone, la = eng.encode_batch(pcm[:1, :n].contiguous(), [n])
    alone = alone[0, :32].clone()
    mixed, lm = eng.encode_batch(pcm, [n, 48000])
    assert int(la[0]) == int(lm[0]) == 32
    torch.testing.assert_close(mixed[0, :32], alone, atol=0, rtol=0)
    recs = GeminiPipelineForBuckets(eng)._encode_chunks(pcm[0].numpy(), np.array([[0, n], [n, n + 10640], [0, 400]]))
    assert [r.n_frames for r in recs] == [32, 32, 0]


def GeminiPipelineForBuckets(engine):
    from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline
    return GeminiASRPipeline(engine=engine, batch_size=1)


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
    from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline

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
            for i in range(len(part)):
                r = recs[lo + i]
                assert (r.start_sample, r.end_sample, r.n_frames) == (part[i][0], part[i][1], int(lengths[i]))
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


def test_two_ranks_share_the_gpu_through_the_long_audio_mode():
    """The N > 1 path on the device: two ranks (gloo, both on GPU 0, as `--share-gpu` allows for rehearsal)
    run `bench.py --hours`: each plans the same chunks, takes its contiguous shard, runs the chunk loop on
    the device, and the records of both (tokens included) cross in the one all-gather; rank 0 checks that
    every chunk is present, in order, with the planned bounds."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--hours", "0.06", "--model", "GTINY",
           "--batch", "3", "--backend", "gloo", "--share-gpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(root))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["chunks"] == 8 and line["config"]["chunks_per_rank"] == 4
    assert line["steps"] == 2 and line["value"] > 0            # 4 chunks per rank in batches of 3 + 1


def test_two_ranks_decode_only_their_shards_of_an_mp3():
    """Sharding a compressed file: two ranks (gloo, both on GPU 0) call the public long-audio entry on the reference's
    demo recording; each decodes about half of it (its chunks plus the rate converter's reach), and the merged
    result on every rank equals the unsharded one (tools/sharded_ingest.py)."""
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "tools" / "sharded_ingest.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(root))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("rank ")]
    assert len(lines) == 2 and all("equal to the unsharded result: True" in l for l in lines), lines


def test_two_ranks_read_only_their_shards_of_a_stereo_wav_a_flac_and_an_ogg():
    """The same for uncompressed and lossless input, and for Ogg Vorbis: a 48 kHz 16-bit stereo WAV (memory-mapped; a
    rank uploads and converts only the bytes of its shard, on the device), the reference's FLAC upload (a rank decodes
    only the frames that cover its shard) and a 22.05 kHz stereo Vorbis stream from the test writer (a rank decodes
    only the packets that cover its shard).  Each of the two ranks reads about half of the source samples and the
    merged result equals the unsharded one, which loads the file whole (tools/sharded_ingest.py wav flac ogg)."""
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "tools" / "sharded_ingest.py"), "wav", "flac", "ogg"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(root))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("rank ")]
    assert len(lines) == 6 and all("equal to the unsharded result: True" in l for l in lines), lines
    for l in lines:
        pct = int(l.rsplit("(", 1)[1].split()[0])
        assert 40 <= pct <= 60, l


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
    # and the compressed kinds, whose native decoders run on the calling threads: an Ogg Vorbis stream from the test
    # writer (5.6 s, coupled stereo at 22.05 kHz) and the first 40 s of the reference's MP3 recording
    from pathlib import Path
    import vorbis_tools as vt
    rng = np.random.default_rng(8)
    enc = vt.encode(0.1 * rng.standard_normal((22050 * 2, 2)), vt.Config(channels=2, rate=22050, pattern="LLSL", seed=8,
                                                                      residue_type=2, coupled=True))
    (tmp_path / "f6.ogg").write_bytes(vt.repeat_stream(enc, 3))
    paths += [str(tmp_path / "f6.ogg"), str(Path(__file__).parent / "golden" / "reference_demo_40s.mp3")]
    pipe = Pipeline(model="GTINY", seed=5, batch_size=2)
    key = lambda segs: [(s.start, s.end, s.text, None if s.words is None else [(w.word, w.start, w.end) for w in s.words]) for s in segs]
    want = [key(pipe.transcribe(p, word_timestamps=True)) for p in paths]
    assert [len(w) for w in want] == [1, 2, 1, 3, 1, 2, 1, 2]
    for _ in range(2):
        with ThreadPoolExecutor(max_workers=4) as ex:
            got = list(ex.map(lambda p: key(pipe.transcribe(p, word_timestamps=True)), paths))
        assert got == want


def test_http_shell_on_the_hip_engine(tmp_path, golden_dir):
    """The server of ref:workflows/wav2elan_web/app.py's contract with the default (HIP) engine behind it:
    single upload, event stream with the four progress steps, and a batch of four files transcribed
    concurrently on the one shared pipeline (ref app.py:383-399) — results equal the direct call."""
    import io
    import json
    import zipfile
    starlette_testclient = pytest.importorskip("starlette.testclient")
    from omnilingual_asr import Pipeline
    from omnilingual_asr.server import create_app

    pipe = Pipeline(model="GTINY", seed=4)
    want = pipe.transcribe(str(golden_dir / "gettysburg.wav"), word_timestamps=True)
    client = starlette_testclient.TestClient(create_app(pipe, upload_dir=tmp_path / "uploads"))
    wav = (golden_dir / "gettysburg.wav").read_bytes()
    entry = client.post("/api/transcribe", files={"file": ("gettysburg.wav", wav, "audio/wav")}).json()
    seg = entry["segments"][0]
    assert (seg["start"], seg["end"], seg["text"]) == (want[0].start, want[0].end, want[0].text)
    assert [w["word"] for w in seg["words"]] == [w.word for w in want[0].words]

    def events(text):
        out = []
        for block in text.replace("\r\n", "\n").strip().split("\n\n"):
            lines = dict(l.split(": ", 1) for l in block.split("\n"))
            out.append((lines["event"], json.loads(lines["data"])))
        return out

    r = client.post("/api/transcribe-stream", files={"file": ("g.wav", wav, "audio/wav")})
    ev = events(r.text)
    assert [(e, d["step"], d["index"]) for e, d in ev[:-1]] == \
        [("progress", s, i) for s, i in (("uploading", 0), ("transcribing", 1), ("processing", 2), ("done", 3))]
    assert ev[-1][1]["segments"][0]["text"] == want[0].text
    long_wav = tmp_path / "long.wav"
    _write_wav(long_wav, 70.0, seed=3)
    want_long = pipe.transcribe(str(long_wav))
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w") as zf:
        zf.writestr("a/one.wav", wav)
        zf.writestr("a/two.wav", long_wav.read_bytes())
    r = client.post("/api/transcribe-batch-stream", files=[("files", ("pack.zip", buf.getvalue(), "application/zip")),
                                                           ("files", ("three.wav", wav, "audio/wav")),
                                                           ("files", ("four.wav", long_wav.read_bytes(), "audio/wav"))])
    results = events(r.text)[-1][1]["results"]
    assert [x["file_name"] for x in results] == ["a/one.wav", "a/two.wav", "three.wav", "four.wav"]
    for x, ref in zip(results, (want, want_long, want, want_long)):
        assert [(s["start"], s["end"], s["text"]) for s in x["segments"]] == [(s.start, s.end, s.text) for s in ref]


def test_soak_of_mixed_files_through_one_pipeline():
    """Thirty files of random length, rate, width and channel count through one pipeline (graphs per shape bucket,
    buffer eviction, device- and host-resident samples): every transcript equals an eager pipeline's on freshly
    decoded host samples (tools/soak.py)."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / "tools" / "soak.py"), "30"], capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "30 files, 0 mismatches" in out.stdout
