"""Host-side logic of the product package against golden vectors captured from the reference
(tests/golden/reference_host.json) — parsing, chunk plan through the C ABI, offsetting, merge,
progress protocol, retry, dataclass surface — and BASELINE config 1: the reference's own
gettysburg.wav through Pipeline.transcribe() with the CPU oracle injected as the engine (plumbing
check; no GPU)."""

import dataclasses
import json
import threading
from pathlib import Path

import numpy as np
import pytest
import torch

import omnilingual_asr
from omnilingual_asr import GeminiDiarizedTranscriptionPipeline, Pipeline
from omnilingual_asr.diarization import DiarizedTranscriptSegment, WordTimestamp
from omnilingual_asr.models.inference import GeminiASRPipeline, GeminiTranscriptionResult, GeminiTranscriptSegment
from omnilingual_asr.models.inference import asr_pipeline as AP
from omnilingual_asr.models.inference import gemini_pipeline as GP_COMPAT


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.loads((golden_dir / "reference_host.json").read_text())


def test_import_surface(gold):
    assert omnilingual_asr.__version__ == gold["reference_version"]
    assert GeminiDiarizedTranscriptionPipeline is Pipeline
    assert GP_COMPAT.GeminiASRPipeline is GeminiASRPipeline
    assert (GP_COMPAT.CHUNK_DURATION_SECONDS, GP_COMPAT.MIN_DURATION_FOR_CHUNKING, GP_COMPAT.MAX_PARALLEL_CHUNKS) == (
        gold["constants"]["CHUNK_DURATION_SECONDS"], gold["constants"]["MIN_DURATION_FOR_CHUNKING"],
        gold["constants"]["MAX_PARALLEL_CHUNKS"])


def test_segment_dataclass_surface(gold):
    assert [f.name for f in dataclasses.fields(DiarizedTranscriptSegment)] == gold["segment_fields"]
    defaults = {f.name: f.default for f in dataclasses.fields(DiarizedTranscriptSegment)
                if f.default is not dataclasses.MISSING}
    assert defaults == gold["segment_defaults"]
    seg = DiarizedTranscriptSegment(0.0, 1.0, "Speaker 1", "hi")
    with pytest.raises(dataclasses.FrozenInstanceError):
        seg.start = 2.0
    assert gold["segment_frozen"] is True
    assert dataclasses.asdict(WordTimestamp("a", 0.0, 0.5)) == {"word": "a", "start": 0.0, "end": 0.5}


def test_parse_timestamp_and_mime(gold):
    for text, want in gold["parse_timestamp"]:
        assert AP.parse_timestamp(text) == want, text
    for name, want in gold["get_mime_type"]:
        assert AP.get_mime_type(Path(name)) == want


def test_parse_response(gold):
    for name, text in gold["parse_response_inputs"].items():
        assert dataclasses.asdict(AP.parse_response(text)) == gold["parse_response"][name], name


def test_chunk_plan_through_c_abi_is_bit_exact(gold):
    sr = 16000
    for plan in gold["chunk_plans"]:
        total = int(round(plan["duration"] * sr))
        got = AP.plan_chunks(total, plan["chunk"], 0.0)       # gate open
        if total == 0:
            assert got.tolist() == [[0, 0]]
            continue
        assert [int(s) for s in got[:, 0]] == [int(round(s * sr)) for s in plan["starts"]]
        assert int(got[-1, 1]) == total
        gated = AP.plan_chunks(total, plan["chunk"], float(gold["constants"]["MIN_DURATION_FOR_CHUNKING"]))
        assert (len(gated) > 1) == (plan["use_chunking"] and plan["duration"] > plan["chunk"])
    from oracle import chunker
    for total in (1, 399, 400, 480000, 480001, 16000 * 601, 547_200_000):
        for chunk, gate in ((30.0, 30.0), (300.0, 360.0)):
            np.testing.assert_array_equal(AP.plan_chunks(total, chunk, gate),
                                          chunker.plan_chunks_samples(total, sr, chunk, gate))


def test_offset_and_merge(gold):
    cm = gold["chunked_merge"]
    results = []
    for off_s, text in sorted(cm["per_chunk_inputs"].items(), key=lambda kv: float(kv[0])):
        if text != "RAISE":
            results.append(AP.offset_result(AP.parse_response(text), float(off_s)))
    merged = AP.merge_results(list(reversed(results)))
    assert dataclasses.asdict(merged) == cm["merged"]


def test_gettysburg_short_path_plumbing(gold, golden_dir, tiny_engine):
    """BASELINE config 1."""
    wav = golden_dir / "gettysburg.wav"
    assert AP.get_audio_duration(wav) == pytest.approx(17.577, abs=1e-3)
    pipe = Pipeline(api_key=None, engine=tiny_engine)
    progress = []
    segs = pipe.transcribe(str(wav), word_timestamps=True, progress_callback=lambda s, i: progress.append([s, i]),
                           language="en", speaker_count="1", some_future_kwarg=True)
    assert progress == gold["short_path"]["progress"]
    assert len(segs) == 1 and isinstance(segs[0], DiarizedTranscriptSegment)
    s = segs[0]
    assert s.start == 0.0 and s.end == 281233 / 16000.0      # 387574 frames @22.05 kHz -> 281233 @16 kHz
    assert (s.speaker, s.emotion) == ("Speaker 1", "neutral")
    # text comes from the CTC stage (random weights: gibberish, but well formed and deterministic)
    assert s.text and set(s.text) <= set("abcdefghijklmnopqrstuvwxyz' ") and s.text == " ".join(s.text.split())
    assert s.words is not None and " ".join(w.word for w in s.words) == s.text
    assert all(0.0 <= w.start < w.end <= s.end + 0.02 for w in s.words)
    assert all(a.end <= b.start for a, b in zip(s.words, s.words[1:]))
    assert pipe.summary is None and pipe.detected_languages is None
    # the reference's default: no word list (ref diarization/pipeline.py:117), same text
    again = pipe.transcribe(str(wav))
    assert again[0].words is None and again[0].text == s.text


def test_missing_file_retries_then_raises_like_reference(gold, tiny_engine, monkeypatch):
    import time
    sleeps = []
    monkeypatch.setattr(time, "sleep", lambda s: sleeps.append(s))
    pipe = Pipeline(engine=tiny_engine)
    with pytest.raises(RuntimeError) as ei:
        pipe.transcribe("/nonexistent/file.wav")
    assert sleeps == gold["retry"]["all_fail"]["sleeps"]
    assert str(ei.value).startswith("Failed to transcribe after 3 attempts: ")


def test_retry_recovers_after_transient_failure(gold, tiny_engine, golden_dir, monkeypatch):
    import time
    sleeps = []
    monkeypatch.setattr(time, "sleep", lambda s: sleeps.append(s))
    eng = GeminiASRPipeline(engine=tiny_engine)
    real, calls = eng.transcribe, []

    def flaky(*a, **k):
        calls.append(1)
        if len(calls) == 1:
            raise ValueError("transient")
        return real(*a, **k)

    eng.transcribe = flaky
    res = eng.transcribe_with_retry(golden_dir / "gettysburg.wav")
    assert isinstance(res, GeminiTranscriptionResult) and len(calls) == 2
    assert sleeps == gold["retry"]["fail_once"]["sleeps"]


def _write_wav(path, seconds, rate=16000, seed=0):
    import wave
    rng = np.random.default_rng(seed)
    x = (0.1 * rng.standard_normal(int(seconds * rate))).clip(-1, 1)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes((x * 32767).astype("<i2").tobytes())


def test_chunked_path_offsets_order_and_progress(gold, tiny_engine, tmp_path):
    wav = tmp_path / "long.wav"
    _write_wav(wav, 7.3)
    pipe = Pipeline(engine=tiny_engine, chunk_seconds=2.0, min_chunking_seconds=3.0, batch_size=3)
    progress = []
    segs = pipe.transcribe(str(wav), progress_callback=lambda s, i: progress.append([s, i]))
    assert progress == gold["chunked_merge"]["progress"]          # each step exactly once
    assert [s.start for s in segs] == [0.0, 2.0, 4.0, 6.0]
    assert [s.end for s in segs] == [2.0, 4.0, 6.0, 7.3]
    assert all(s.languages is None and s.words is None for s in segs)


def test_chunked_fallthrough_repeats_progress_like_reference(gold, tiny_engine, tmp_path):
    wav = tmp_path / "short.wav"
    _write_wav(wav, 1.0)
    eng = GeminiASRPipeline(engine=tiny_engine)
    progress = []
    eng.transcribe_chunked(wav, progress_callback=lambda s, i: progress.append([s, i]))
    assert progress == gold["chunked_fallthrough_progress"]


def test_failed_batch_is_skipped_like_a_failed_chunk(tiny_engine, tmp_path, capsys):
    class Flaky:
        def __init__(self, inner):
            self.inner, self.n = inner, 0

        def encode_batch(self, pcm, n_valid):
            self.n += 1
            if self.n == 2:
                raise RuntimeError("boom")               # the device hiccup kind: tolerated per batch
            return self.inner.encode_batch(pcm, n_valid)

    wav = tmp_path / "long.wav"
    _write_wav(wav, 6.0)
    pipe = Pipeline(engine=Flaky(tiny_engine), chunk_seconds=2.0, min_chunking_seconds=3.0, batch_size=1)
    segs = pipe.transcribe(str(wav))
    assert [s.start for s in segs] == [0.0, 4.0]
    assert "Chunk transcription failed: boom" in capsys.readouterr().out


def test_deterministic_engine_errors_are_not_swallowed(tiny_engine, tmp_path, monkeypatch):
    """A rejected kernel argument (OasrError) or a malformed batch is not a failed network call: it
    propagates (and ends as the reference's final RuntimeError), and a file whose every chunk fails
    raises instead of returning an empty transcript."""
    import time
    from omnilingual_asr._hip import OasrError
    monkeypatch.setattr(time, "sleep", lambda s: None)

    class Rejecting:
        def encode_batch(self, pcm, n_valid):
            raise OasrError("oasr_ctc_greedy failed (-1): t_pad out of range")

    class Down:
        def encode_batch(self, pcm, n_valid):
            raise RuntimeError("device lost")

    wav = tmp_path / "long.wav"
    _write_wav(wav, 6.0)
    for eng, text in ((Rejecting(), "t_pad out of range"), (Down(), "none of the 3 chunk(s) could be transcribed")):
        pipe = Pipeline(engine=eng, chunk_seconds=2.0, min_chunking_seconds=3.0, batch_size=1)
        with pytest.raises(RuntimeError) as ei:
            pipe.transcribe(str(wav))
        assert str(ei.value).startswith("Failed to transcribe after 3 attempts: ") and text in str(ei.value)
    with pytest.raises(OasrError):
        GeminiASRPipeline(engine=Rejecting(), chunk_seconds=2.0, min_chunking_seconds=3.0)._encode_chunks(
            np.zeros(64000, dtype=np.float32), np.array([[0, 32000], [32000, 64000]]))


def test_unreadable_header_still_chunks_on_the_decoded_length(tiny_engine, tmp_path, monkeypatch):
    """The duration probe returns 0.0 when it cannot read the header (ref :222-240), which sends the file
    down the single-utterance entry; the plan is made from the decoded samples, so the file is still cut
    into chunks and decoded once."""
    wav = tmp_path / "long.wav"
    _write_wav(wav, 7.3)
    monkeypatch.setattr(AP, "get_audio_duration", lambda p: 0.0)
    loads = []
    real = AP._audio.load_audio_16k
    monkeypatch.setattr(AP._audio, "load_audio_16k", lambda *a, **k: (loads.append(1), real(*a, **k))[1])
    pipe = Pipeline(engine=tiny_engine, chunk_seconds=2.0, min_chunking_seconds=3.0, batch_size=3)
    progress = []
    segs = pipe.transcribe(str(wav), progress_callback=lambda s, i: progress.append((s, i)))
    assert [s.start for s in segs] == [0.0, 2.0, 4.0, 6.0] and len(loads) == 1
    assert progress == [("uploading", 0), ("transcribing", 1), ("processing", 2), ("done", 3)]
    loads.clear()
    GeminiASRPipeline(engine=tiny_engine).transcribe_chunked(wav)      # fall-through: still one decode
    assert len(loads) == 1


def test_concurrent_calls_on_one_instance(tiny_engine, golden_dir):
    """The reference's server runs up to 4 transcribe() calls at once on one shared instance."""
    pipe = Pipeline(engine=tiny_engine)
    out, errs = {}, []

    def work(i):
        try:
            out[i] = pipe.transcribe(str(golden_dir / "gettysburg.wav"))
        except Exception as e:  # pragma: no cover
            errs.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errs and len(out) == 4 and all(out[i] == out[0] for i in out)


def test_no_api_key_needed_and_no_silent_cpu_fallback(monkeypatch):
    pipe = Pipeline()                                     # the reference raises ValueError here
    if not torch.cuda.is_available():
        from omnilingual_asr._hip import OasrError
        with pytest.raises(OasrError):
            pipe.gemini.engine                            # default engine = HIP; must fail loudly


def test_split_audio_into_chunks_contract(tmp_path):
    wav = tmp_path / "a.wav"
    _write_wav(wav, 5.0)
    parts = AP.split_audio_into_chunks(wav, chunk_duration=2, output_dir=tmp_path)
    assert [(p.name, s) for p, s in parts] == [("chunk_0000.wav", 0.0), ("chunk_0001.wav", 2.0), ("chunk_0002.wav", 4.0)]
    assert AP.get_audio_duration(parts[-1][0]) == pytest.approx(1.0)
    assert AP.split_audio_into_chunks(tmp_path / "missing.wav") == [(tmp_path / "missing.wav", 0.0)]


@pytest.mark.parametrize("n,end", [(0, 1.0), (100, 100 / 16000.0), (400, 0.025), (720, 0.045)])
def test_audio_shorter_than_the_first_stacked_frame(tmp_path, tiny_engine, n, end):
    """Empty and sub-frame files: no exception; one segment with empty text until two 25 ms frames fit
    (720 samples); a zero-length file gets the reference's end <= start fix-up (ref :396-397)."""
    import wave
    path = tmp_path / "short.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((np.random.default_rng(n).standard_normal(n) * 3000).astype("<i2").tobytes())
    segs = Pipeline(engine=tiny_engine).transcribe(str(path), word_timestamps=True)
    assert len(segs) == 1 and (segs[0].start, segs[0].end) == (0.0, end)
    if n < 720:
        assert segs[0].text == "" and segs[0].words == []
