"""GPU test of the device polyphase resampler through the C ABI, against the host implementation
(which tests/test_audio.py pins to scipy.signal.resample_poly)."""

from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rate_in,n_in", [(22050, 387574), (44100, 100001), (48000, 48000), (8000, 12345),
                                          (11025, 5), (32000, 1)])
def test_resample_matches_host(rate_in, n_in):
    from omnilingual_asr import audio
    from omnilingual_asr.engine_hip import HipEngine

    eng = HipEngine("GTINY")
    rng = np.random.default_rng(rate_in)
    x = (0.3 * rng.standard_normal((2, n_in))).astype(np.float32)
    y = eng.resample_16k(x, rate_in).cpu().numpy()
    for b in range(2):
        ref = audio.resample_poly(x[b], rate_in)
        assert y[b].shape == ref.shape                       # integer length: exact
        np.testing.assert_allclose(y[b], ref, rtol=0, atol=1.2e-7)   # float64 accumulate on both sides: <= 1 ulp of 1.0
    one = eng.resample_16k(x[0], rate_in)
    assert one.dim() == 1 and torch.equal(one.cpu(), torch.from_numpy(y[0]))


def test_resample_length_and_argument_checks():
    from omnilingual_asr import _hip
    L = _hip.lib()
    assert L.oasr_resample_length(387574, 320, 441) == 281233     # the reference's sample file at 16 kHz
    assert L.oasr_resample_length(0, 1, 3) == 0 and L.oasr_resample_length(-1, 1, 3) == -1
    z = torch.zeros((8,), device="cuda")
    h = torch.zeros((4,), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert L.oasr_resample_poly(z.data_ptr(), 8, 8, 1, h.data_ptr(), 1, 2, 4, 1, z.data_ptr(), 5, 8, st) == -1
    assert b"n_out" in L.oasr_last_error()


def test_pipeline_uses_device_resampler(golden_dir):
    from omnilingual_asr import Pipeline, audio

    pipe = Pipeline(model="GTINY", seed=4)
    got = pipe.gemini._load_audio(golden_dir / "gettysburg.wav")
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got     # left on the device for the chunk loop
    ref = audio.load_audio_16k(golden_dir / "gettysburg.wav")
    assert got.shape == ref.shape == (281233,)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1.2e-7)


@pytest.mark.parametrize("channels,rate,width", [(2, 48000, 2), (1, 44100, 2), (2, 16000, 2), (3, 22050, 2),
                                                 (2, 48000, 3), (1, 16000, 3), (2, 44100, 4), (1, 8000, 1), (2, 11025, 1)])
def test_pcm_files_decoded_on_the_device_equal_the_host_decode(tmp_path, channels, rate, width):
    """Integer PCM WAVs (8, 16, 24, 32 bits) go to the device as the file holds them (conversion, down-mix and
    rate conversion there); the 16 kHz signal equals host decoding + the same device resampler bit for bit."""
    import struct
    from omnilingual_asr import Pipeline, audio

    rng = np.random.default_rng(11 + width)
    frames = int(rate * 2.37)
    n = frames * channels
    if width == 1:
        payload = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    elif width == 2:
        payload = rng.integers(-32768, 32768, size=n, dtype=np.int16).tobytes()
    elif width == 3:
        v = rng.integers(-(1 << 23), 1 << 23, size=n, dtype=np.int32)
        payload = v.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    else:
        payload = rng.integers(-(1 << 31), 1 << 31, size=n, dtype=np.int64).astype("<i4").tobytes()
    fmt = struct.pack("<HHIIHH", 1, channels, rate, rate * channels * width, channels * width, 8 * width)
    path = tmp_path / "x.wav"
    path.write_bytes(b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVEfmt " + struct.pack("<I", 16) + fmt +
                     b"data" + struct.pack("<I", len(payload)) + payload)
    pipe = Pipeline(model="GTINY").gemini
    got = pipe._load_audio(path)
    want = audio.load_audio_16k(path, lambda mono, r: pipe.engine.resample_16k(mono, r).cpu().numpy())
    if isinstance(got, torch.Tensor):   # converted on the device and left there for the chunk loop
        assert got.is_cuda and got.dtype == torch.float32
        got = got.cpu().numpy()
    elif got.dtype == np.int16:         # mono 16-bit at 16 kHz stays as read
        got = got.astype(np.float32) / 32768.0
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want)


def test_24_bit_flac_decoded_on_the_device_equals_the_host_decode(tmp_path):
    from test_audio_formats import _flac
    from omnilingual_asr import Pipeline, audio

    rng = np.random.default_rng(2)
    l = rng.integers(-(1 << 22), 1 << 22, 700).astype(np.int64)
    r = (l // 3 + rng.integers(-1000, 1000, 700)).astype(np.int64)
    data, _ = _flac(2, 24, 48000, [[l[:400], r[:400]], [l[400:], r[400:]]], 10, [10], rice=14)
    path = tmp_path / "deep.flac"
    path.write_bytes(data)
    pipe = Pipeline(model="GTINY").gemini
    got = pipe._load_audio(path).cpu().numpy()
    want = audio.load_audio_16k(path, lambda mono, rr: pipe.engine.resample_16k(mono, rr).cpu().numpy())
    assert got.dtype == np.float32 and np.array_equal(got, want)


def test_gettysburg_fixture_through_the_device_decode(golden_dir):
    from omnilingual_asr import Pipeline, audio

    pipe = Pipeline(model="GTINY").gemini
    path = golden_dir / "gettysburg.wav"
    got = pipe._load_audio(path).cpu().numpy()
    want = audio.load_audio_16k(path, lambda mono, r: pipe.engine.resample_16k(mono, r).cpu().numpy())
    assert np.array_equal(got, want)


def test_flac_upload_through_the_device_decode(golden_dir):
    """The reference's .flac upload: 16-bit samples from the native decoder, kept int16 (it is mono at 16 kHz);
    the pipeline's records equal those of the float samples."""
    from omnilingual_asr import Pipeline, audio

    pipe = Pipeline(model="GTINY").gemini
    path = golden_dir / "reference_upload.flac"
    kept = pipe._load_audio(path)
    assert kept.dtype == np.int16
    flt = audio.load_audio_16k(path)
    bounds = np.array([[0, kept.shape[0]]])
    a, b = pipe._encode_chunks(kept, bounds), pipe._encode_chunks(flt, bounds)
    assert a[0].n_frames == b[0].n_frames and np.array_equal(a[0].tokens, b[0].tokens)


def test_flac_in_ogg_through_the_device_path_equals_the_native_file(golden_dir, tmp_path):
    """The same upload wrapped in Ogg pages: the device path takes the unwrapped stream's samples as the native file's."""
    import sys
    sys.path.insert(0, str(Path(__file__).parent))
    import flac_tools as ft
    from omnilingual_asr import Pipeline

    src = golden_dir / "reference_upload.flac"
    p = tmp_path / "upload.ogg"
    p.write_bytes(ft.wrap_in_ogg(src.read_bytes()))
    pipe = Pipeline(model="GTINY")
    kept = pipe.gemini._load_audio(p)
    assert kept.dtype == np.int16 and np.array_equal(kept, pipe.gemini._load_audio(src))
    a, b = pipe.transcribe(str(src)), pipe.transcribe(str(p))
    assert [(s.start, s.end, s.text) for s in a] == [(s.start, s.end, s.text) for s in b] and a[0].text


def test_mp3_demo_through_the_device_path(golden_dir):
    """The reference's demo recording (first 40 s, 44.1 kHz joint stereo): host threads decode and down-mix, the rate
    conversion runs on the device and the 16 kHz signal stays there; it equals host decoding + the same resampler
    bit for bit, and the public call returns the two 30 s-plan segments with text."""
    from omnilingual_asr import Pipeline, audio

    pipe = Pipeline(model="GTINY", seed=4)
    path = golden_dir / "reference_demo_40s.mp3"
    got = pipe.gemini._load_audio(path)
    assert isinstance(got, torch.Tensor) and got.is_cuda and got.dtype == torch.float32
    x, rate = audio.read_mp3(path)
    want = pipe.gemini.engine.resample_16k(x.mean(axis=1, dtype=np.float64).astype(np.float32), rate).cpu().numpy()
    assert rate == 44100 and got.shape[0] == -(-x.shape[0] * 160 // 441) and np.array_equal(got.cpu().numpy(), want)
    np.testing.assert_allclose(want, audio.load_audio_16k(path), rtol=0, atol=1.2e-7)    # the host resampler
    segs = pipe.transcribe(str(path))
    assert [s.start for s in segs] == [0.0, 30.0] and segs[0].end == 30.0 and all(s.text for s in segs)
    assert segs[1].end == pytest.approx(got.shape[0] / 16000.0, abs=1e-6)


@pytest.mark.parametrize("chunk_seconds", [5.0, 2.5])
def test_long_mp3_in_segments_equals_the_one_pass_path(golden_dir, chunk_seconds):
    """Decode of segment k + 1 overlapping the device's work on segment k (`_encode_streamed`): the same segments,
    texts and times as decoding the whole file first, because every segment starts on a whole period of the rate
    converter and carries the filter's reach on both sides."""
    from omnilingual_asr import Pipeline

    path = golden_dir / "reference_demo_40s.mp3"
    pipe = Pipeline(model="GTINY", seed=4, batch_size=2, chunk_seconds=chunk_seconds,
                    min_chunking_seconds=chunk_seconds).gemini
    streamed = pipe._encode_streamed(path, True)
    assert streamed is not None and len(streamed.segments) == -(-39953 // int(chunk_seconds * 1000))
    pcm = pipe._load_audio(path)
    whole = pipe._encode_file(pcm, True)
    assert len(whole.segments) == len(streamed.segments)
    for a, b in zip(streamed.segments, whole.segments):
        assert (a.start, a.end, a.text) == (b.start, b.end, b.text) and a.words == b.words
    assert any(s.text for s in whole.segments)
    assert pipe.transcribe_chunked(path).segments == pipe._encode_streamed(path, False).segments   # the public entry takes it
    assert Pipeline(model="GTINY", batch_size=32).gemini._encode_streamed(path, False) is None     # too short to split


@pytest.mark.parametrize("rate", [8000, 11025, 16000, 22050, 32000, 48000])
def test_segments_equal_one_pass_at_every_rate_ratio(tmp_path, rate):
    """The segment path's arithmetic (a segment starts on a whole period of the rate converter and carries the filter's
    reach on both sides) at the other conversion ratios: 2/1, 640/441, 1/1, 320/441, 1/2, 1/3.  Files from the test
    encoder (tests/mp3_tools.py), 9 s, one-second chunks, segments of four chunks."""
    import mp3_tools as mt
    from omnilingual_asr import Pipeline

    rng = np.random.default_rng(rate)
    spf = 576 if rate < 32000 else 1152
    frames = 9 * rate // spf + 1
    pcm = rng.standard_normal((frames * spf, 1))
    pcm = np.cumsum(pcm, axis=0) * 0.02 + pcm * 0.1
    data, _ = mt.encode(pcm / np.abs(pcm).max() * 0.5, rate, peak_is=[7, 15], seed=rate)
    path = tmp_path / "r.mp3"
    path.write_bytes(data)
    pipe = Pipeline(model="GTINY", seed=4, batch_size=2, chunk_seconds=1.0, min_chunking_seconds=1.0).gemini
    streamed = pipe._encode_streamed(path, True)
    whole = pipe._encode_file(pipe._load_audio(path), True)
    assert streamed is not None and len(streamed.segments) == len(whole.segments) >= 9
    for a, b in zip(streamed.segments, whole.segments):
        assert (a.start, a.end, a.text, a.words) == (b.start, b.end, b.text, b.words)
    assert any(s.text for s in whole.segments)


def test_host_only_call_before_the_engine_leaves_the_device_usable(golden_dir):
    """A host-only entry point (the FLAC decoder) used before any engine exists must not leave the process
    with a second HIP runtime: the library loader brings torch's in first.  Run in a child process, where
    nothing has touched torch or the library yet."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from omnilingual_asr import audio\n"
        "x, r = audio.read_flac(%r)\n"
        "from omnilingual_asr import Pipeline\n"
        "segs = Pipeline(model='GTINY').transcribe(%r)\n"
        "print('segments', len(segs), r)\n"
    ) % (str(root / "omnilingual-asr_amd"), str(golden_dir / "reference_upload.flac"), str(golden_dir / "reference_upload.flac"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "segments 1 16000" in out.stdout


def test_batches_gathered_on_the_device_equal_batches_assembled_on_the_host(tmp_path):
    """A 48 kHz stereo file stays on the device after conversion and its chunks are gathered there
    (oasr_gather_chunks); the records equal those of the same samples taken through the host."""
    import wave
    from omnilingual_asr import Pipeline

    rng = np.random.default_rng(21)
    rate, seconds = 48000, 75.3
    n = int(rate * seconds)
    pcm = (3000 * rng.standard_normal(n) + 5000 * np.sin(np.arange(n) * 2 * np.pi * 220 / rate)).astype(np.int16)
    path = tmp_path / "stereo48.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(rate); w.writeframes(np.repeat(pcm, 2).tobytes())
    pipe = Pipeline(model="GTINY", batch_size=2).gemini
    dev = pipe._load_audio(path)
    assert isinstance(dev, torch.Tensor) and dev.is_cuda
    host = dev.cpu().numpy()
    total = host.shape[0]
    bounds = np.array([[0, 480000], [480000, 960000], [960000, total]])
    a, b = pipe._encode_chunks(dev, bounds), pipe._encode_chunks(host, bounds)
    assert len(a) == len(b) == 3
    for ra, rb in zip(a, b):
        assert (ra.chunk_idx, ra.start_sample, ra.end_sample, ra.n_frames) == (rb.chunk_idx, rb.start_sample, rb.end_sample, rb.n_frames)
        assert np.array_equal(ra.tokens, rb.tokens) and np.array_equal(ra.frames, rb.frames)
    # and through the public call
    segs = Pipeline(model="GTINY", batch_size=2).transcribe(str(path))
    assert [s.start for s in segs] == [0.0, 30.0, 60.0]
    with pytest.raises(ValueError):
        t, _ = pipe.engine.host_batch_slot(1, 480000, 0)
        pipe.engine.submit(t, [480000], source=(dev, [total - 10]))     # runs past the end of the signal


def test_back_to_back_device_decodes_do_not_share_staging_memory_too_early():
    """`pcm_to_16k(on_device=True)` returns while its last uploads may still be reading the two page-locked
    staging blocks; the next call (another thread's file in the server) must wait for them before it refills the
    blocks.  Two multi-block inputs (three and two blocks of 48 MiB) back to back, no synchronisation between the
    calls: each result must be the host decode of its own input (16 kHz mono 16-bit, so the device result is
    exactly sample / 32768)."""
    from omnilingual_asr.engine_hip import HipEngine
    eng = HipEngine("GTINY", seed=0)
    rng = np.random.default_rng(5)
    block = 12 << 22
    a = rng.integers(-32768, 32768, (2 * block + 1000) // 2, dtype=np.int16)          # 2 blocks and a bit
    b = (rng.integers(-32768, 32768, (block + 4096) // 2, dtype=np.int16) | 1)       # 1 block and a bit, all odd
    ya = eng.pcm_to_16k(a, 1, 16000, 2, 16, on_device=True)
    yb = eng.pcm_to_16k(b, 1, 16000, 2, 16, on_device=True)
    yc = eng.pcm_to_16k(a[: block // 2 + 7], 1, 16000, 2, 16, on_device=True)
    torch.cuda.synchronize()
    assert np.array_equal(ya.cpu().numpy(), a.astype(np.float32) / 32768.0)
    assert np.array_equal(yb.cpu().numpy(), b.astype(np.float32) / 32768.0)
    assert np.array_equal(yc.cpu().numpy(), a[: block // 2 + 7].astype(np.float32) / 32768.0)


def test_progress_steps_of_the_segmented_path_arrive_around_the_work(golden_dir):
    """The reference reports "transcribing" before it fans its chunks out and "processing" before it merges
    (ref gemini_pipeline.py:619, :644).  On the segmented path (a long MP3) step 1 must arrive before the first
    batch is submitted to the engine and step 2 after the last batch was collected, not all at the end."""
    import time
    from omnilingual_asr import Pipeline

    pipe = Pipeline(model="GTINY", seed=4, batch_size=2, chunk_seconds=2.5, min_chunking_seconds=2.5).gemini
    eng = pipe.engine
    stamps = {"submit": [], "collect": []}
    real_submit, real_collect = eng.submit, eng.collect

    def submit(*a, **k):
        stamps["submit"].append(time.perf_counter())
        return real_submit(*a, **k)

    def collect(*a, **k):
        out = real_collect(*a, **k)
        stamps["collect"].append(time.perf_counter())
        return out
    eng.submit, eng.collect = submit, collect
    steps = []
    res = pipe.transcribe_chunked(golden_dir / "reference_demo_40s.mp3",
                                  progress_callback=lambda s, i: steps.append((s, i, time.perf_counter())))
    assert [(s, i) for s, i, _ in steps] == [("uploading", 0), ("transcribing", 1), ("processing", 2), ("done", 3)]
    assert len(res.segments) == 16 and len(stamps["submit"]) == 8 == len(stamps["collect"])
    t = {s: at for s, _, at in steps}
    assert t["uploading"] < t["transcribing"] < min(stamps["submit"])
    assert max(stamps["collect"]) < t["processing"] < t["done"]


@pytest.mark.parametrize("rate", [44100, 16000])
def test_long_ogg_vorbis_in_segments_equals_the_one_pass_path(tmp_path, rate):
    """An Ogg Vorbis file goes the MP3's way: host threads decode it to mono, the device converts the rate, and a long
    file is taken in segments whose decode overlaps the device's work (`_encode_streamed` on a `VorbisStream`).  Same
    segments, texts and word times as decoding the whole file first.  Stereo stream from the test writer
    (tests/vorbis_tools.py), its packets repeated to 20 s."""
    import vorbis_tools as vt
    from omnilingual_asr import Pipeline, audio

    rng = np.random.default_rng(3)
    n = rate * 2
    t = np.arange(n) / rate
    x = (0.2 * np.sin(2 * np.pi * 170.0 * t) * (np.sin(2 * np.pi * 1.3 * t) > 0))[:, None] + 0.05 * rng.standard_normal((n, 2))
    enc = vt.encode(x, vt.Config(channels=2, rate=rate, pattern="LLSL", seed=3, residue_type=2, coupled=True))
    path = tmp_path / "long.ogg"
    path.write_bytes(vt.repeat_stream(enc, 10))
    total = audio.vorbis_info(path)[2]
    assert audio.audio_duration(path) == total / rate > 19.0
    pipe = Pipeline(model="GTINY", seed=4, batch_size=2, chunk_seconds=2.5, min_chunking_seconds=2.5).gemini
    streamed = pipe._encode_streamed(path, True)
    whole = pipe._encode_file(pipe._load_audio(path), True)
    assert streamed is not None and len(streamed.segments) == len(whole.segments) >= 8
    for a, b in zip(streamed.segments, whole.segments):
        assert (a.start, a.end, a.text, a.words) == (b.start, b.end, b.text, b.words)
    assert any(s.text for s in whole.segments)
    assert pipe.transcribe_chunked(path).segments == pipe._encode_streamed(path, False).segments
    # the device path's 16 kHz signal against the host's converter on the same decode
    mono, r = audio.read_ogg_mono(path)
    np.testing.assert_allclose(pipe._load_audio(path).cpu().numpy(), audio.resample_poly(mono, r, 16000), atol=2e-6)
