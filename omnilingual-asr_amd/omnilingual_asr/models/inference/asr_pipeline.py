"""Local ASR engine front: duration gate, chunk plan, batched device encode, merge, retry.

Host-side mirror of ref:src/omnilingual_asr/models/inference/gemini_pipeline.py.  Same public
names, argument meaning, progress protocol and error behaviour; the remote model call
(ref :512-530) is replaced by an engine object that turns PCM batches into encoder states on the
GPU (`omnilingual_asr.engine_hip.HipEngine`).  Where the reference fans chunks out to 4 threads of
network calls (ref :623-641), this module packs them into batches for one device, or shards them
over the ranks of a `torch.distributed` job (`omnilingual_asr.distributed`).
"""

from __future__ import annotations

import contextlib
import json
import re
import math
import threading
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Callable, List, Optional, Sequence

import numpy as np

from omnilingual_asr import audio as _audio
from omnilingual_asr import tokenizer as _tok

SAMPLE_RATE = 16000

# Reference constants (ref gemini_pipeline.py:217-219), kept for callers that import them.
CHUNK_DURATION_SECONDS = 300
MIN_DURATION_FOR_CHUNKING = 360
MAX_PARALLEL_CHUNKS = 4

# Defaults of the local engine: 30 s utterances (the encoder's design length), gate at 30 s.
DEFAULT_CHUNK_SECONDS = 30.0
DEFAULT_MIN_CHUNKING_SECONDS = 30.0
DEFAULT_BATCH = 32


@dataclass(frozen=True)
class WordTimestamp:
    """ref gemini_pipeline.py:39-45"""
    word: str
    start: float
    end: float


@dataclass
class GeminiTranscriptSegment:
    """Engine-level segment record; field order and defaults of ref gemini_pipeline.py:48-61."""
    start: float
    end: float
    speaker: str
    text: str
    language: Optional[str] = None
    language_code: Optional[str] = None
    languages: Optional[List[dict]] = None
    emotion: Optional[str] = None
    translation: Optional[str] = None
    words: Optional[List[WordTimestamp]] = None


@dataclass
class GeminiTranscriptionResult:
    """ref gemini_pipeline.py:64-70"""
    summary: Optional[str] = None
    segments: List[GeminiTranscriptSegment] = field(default_factory=list)
    detected_languages: Optional[List[dict]] = None


TranscriptSegment = GeminiTranscriptSegment
TranscriptionResult = GeminiTranscriptionResult

_FIELD_SECONDS = {2: (60.0, 1.0), 3: (3600.0, 60.0, 1.0)}
_MIME_BY_SUFFIX = {".wav": "audio/wav", ".mp3": "audio/mp3", ".aiff": "audio/aiff", ".aac": "audio/aac",
                   ".ogg": "audio/ogg", ".flac": "audio/flac", ".m4a": "audio/mp4"}


def parse_timestamp(timestamp_str: str) -> float:
    """'MM:SS', 'HH:MM:SS' or raw seconds -> seconds; anything else -> 0.0
    (behaviour of ref gemini_pipeline.py:73-103)."""
    if not timestamp_str:
        return 0.0
    parts = timestamp_str.strip().split(":")
    try:
        weights = _FIELD_SECONDS.get(len(parts))
        if weights is None:
            return float(timestamp_str)
        total = 0.0
        for i, (txt, w) in enumerate(zip(parts, weights)):
            total = total + (float(txt) if i == len(parts) - 1 else int(txt)) * w
        return total
    except ValueError:
        return 0.0


def get_mime_type(file_path: Path) -> str:
    """ref gemini_pipeline.py:106-118"""
    return _MIME_BY_SUFFIX.get(Path(file_path).suffix.lower(), "audio/wav")


def get_audio_duration(audio_path: Path) -> float:
    """Duration in seconds from the file header; 0.0 on any failure, as the reference's ffprobe
    wrapper does (ref gemini_pipeline.py:222-240)."""
    try:
        return float(_audio.audio_duration(audio_path))
    except Exception:
        return 0.0


def plan_chunks(total_samples: int, chunk_seconds: float = DEFAULT_CHUNK_SECONDS,
                min_chunking_seconds: float = DEFAULT_MIN_CHUNKING_SECONDS,
                sample_rate: int = SAMPLE_RATE) -> np.ndarray:
    """Integer chunk plan [n, 2] (start_sample, end_sample) computed by liboasr_hip.so
    (`oasr_plan_chunks`): the reference's fixed-length plan (ref :256-310) behind its strict
    duration gate (ref :713-714)."""
    from omnilingual_asr import _hip
    return _hip.plan_chunks(total_samples, sample_rate, chunk_seconds, min_chunking_seconds)


def split_audio_into_chunks(audio_path: Path, chunk_duration: float = CHUNK_DURATION_SECONDS,
                            output_dir: Optional[Path] = None) -> List[tuple]:
    """Same contract as ref gemini_pipeline.py:243-310 — a list of (chunk_path, start_offset) with
    chunk files `chunk_0000<ext>` of `chunk_duration` seconds written to `output_dir` — but the
    slices are cut from the decoded PCM and written as 16-bit WAV instead of by ffmpeg.  A file
    whose duration cannot be determined is returned whole."""
    import tempfile
    import wave

    audio_path = Path(audio_path)
    if output_dir is None:
        output_dir = Path(tempfile.mkdtemp(prefix="audio_chunks_"))
    total = get_audio_duration(audio_path)
    if total <= 0:
        return [(audio_path, 0.0)]
    ext = audio_path.suffix or ".wav"
    try:
        pcm, rate = _audio.read_wav(audio_path)
    except Exception:
        return [(audio_path, 0.0)]
    out, start, idx = [], 0.0, 0
    while start < total:
        lo = int(round(start * rate))
        hi = min(int(round((start + chunk_duration) * rate)), pcm.shape[0])
        path = Path(output_dir) / f"chunk_{idx:04d}{ext}"
        try:
            with wave.open(str(path), "wb") as w:
                w.setnchannels(pcm.shape[1])
                w.setsampwidth(2)
                w.setframerate(rate)
                w.writeframes((np.clip(pcm[lo:hi], -1.0, 32767.0 / 32768.0) * 32768.0).astype("<i2").tobytes())
            out.append((path, start))
        except Exception:
            pass
        start += chunk_duration
        idx += 1
    return out if out else [(audio_path, 0.0)]


def parse_response(response_text: str) -> GeminiTranscriptionResult:
    """JSON transcript -> records, with the reference's defaults and fix-ups
    (ref gemini_pipeline.py:364-453): fenced ```json blocks accepted, `end <= start` becomes
    `start + 1.0`, speaker defaults to 'Speaker 1', emotion to 'neutral', languages unioned by
    code in first-seen order.  Kept so transcripts produced elsewhere can be ingested."""
    try:
        data = json.loads(response_text)
    except json.JSONDecodeError:
        m = re.search(r"```json\s*(.*?)\s*```", response_text, re.DOTALL)
        if m is None:
            return GeminiTranscriptionResult(summary="Failed to parse transcription", segments=[])
        data = json.loads(m.group(1))
    segments, langs, seen = [], [], set()
    for item in data.get("segments", []):
        t0 = parse_timestamp(item.get("timestamp_start", "0:00"))
        t1 = parse_timestamp(item.get("timestamp_end", "0:00"))
        if t1 <= t0:
            t1 = t0 + 1.0
        listed = item.get("languages", [])
        if listed:
            name, code = listed[0].get("name"), listed[0].get("code")
            per_seg = [{"name": x.get("name", ""), "code": x.get("code", "")} for x in listed]
            pairs = [(x.get("code", ""), x.get("name", x.get("code", ""))) for x in listed]
        else:
            name, code, per_seg = item.get("language"), item.get("language_code"), None
            pairs = [(code, name or code)] if code else []
        for c, n in pairs:
            if c and c not in seen:
                seen.add(c)
                langs.append({"code": c, "language": n})
        segments.append(GeminiTranscriptSegment(
            start=t0, end=t1, speaker=item.get("speaker", "Speaker 1"), text=item.get("content", ""),
            language=name, language_code=code, languages=per_seg,
            emotion=item.get("emotion", "neutral"), translation=item.get("translation"), words=None))
    summary = data.get("summary", "")
    return GeminiTranscriptionResult(summary=summary or None, segments=segments,
                                     detected_languages=langs or None)


def offset_result(result: GeminiTranscriptionResult, start_offset: float) -> GeminiTranscriptionResult:
    """Shift a chunk's segments by the chunk start (ref gemini_pipeline.py:555-575).  As in the
    reference the per-segment `languages` list is not carried over."""
    shifted = [GeminiTranscriptSegment(
        start=s.start + start_offset, end=s.end + start_offset, speaker=s.speaker, text=s.text,
        language=s.language, language_code=s.language_code, emotion=s.emotion,
        translation=s.translation,
        words=None if s.words is None else [WordTimestamp(w.word, w.start + start_offset, w.end + start_offset)
                                            for w in s.words]) for s in result.segments]
    return GeminiTranscriptionResult(summary=result.summary, segments=shifted,
                                     detected_languages=result.detected_languages)


def merge_results(results: Sequence[GeminiTranscriptionResult]) -> GeminiTranscriptionResult:
    """Merge per-chunk results (ref gemini_pipeline.py:646-678): stable sort by first-segment start
    (chunks without segments last), concatenate, union languages by code, join summaries."""
    ordered = sorted(results, key=lambda r: r.segments[0].start if r.segments else float("inf"))
    merged: List[GeminiTranscriptSegment] = []
    langs: List[dict] = []
    seen = set()
    for r in ordered:
        merged.extend(r.segments)
        for lang in r.detected_languages or []:
            code = lang.get("code", "")
            if code and code not in seen:
                seen.add(code)
                langs.append(lang)
    summaries = [r.summary for r in ordered if r.summary]
    return GeminiTranscriptionResult(summary=" ".join(summaries) if summaries else None,
                                     segments=merged, detected_languages=langs or None)


@dataclass
class ChunkRecord:
    """What the device path reports per chunk (also the unit gathered across ranks)."""
    chunk_idx: int
    start_sample: int
    end_sample: int
    n_frames: int          # valid encoder frames (20 ms each)
    tokens: Optional[np.ndarray] = None   # greedy CTC token ids (int32), None when the engine has no head
    frames: Optional[np.ndarray] = None   # encoder frame index of each token

    def to_result(self, sample_rate: int = SAMPLE_RATE, word_timestamps: bool = False) -> GeminiTranscriptionResult:
        """Chunk-local result: one segment spanning the chunk with the decoded text (empty when the
        engine has no CTC head); speaker and emotion take the reference's defaults (ref :435, :440).
        `words` stays None unless asked for, as in the reference (ref diarization/pipeline.py:117)."""
        dur = (self.end_sample - self.start_sample) / float(sample_rate)
        end = dur if dur > 0.0 else 1.0          # the reference's end<=start fix-up (ref :396-397)
        text, words = "", None
        if self.tokens is not None:
            text = _tok.decode_text(self.tokens)
            if word_timestamps:
                words = [WordTimestamp(word=w, start=f0 * _tok.FRAME_SECONDS, end=f1 * _tok.FRAME_SECONDS)
                         for w, f0, f1 in _tok.words_with_frames(self.tokens, self.frames)]
        seg = GeminiTranscriptSegment(start=0.0, end=end, speaker="Speaker 1", text=text,
                                      emotion="neutral", words=words)
        return GeminiTranscriptionResult(summary=None, segments=[seg], detected_languages=None)


class _Spans:
    """A file read by sample ranges: `rate`, `samples` (per channel), `read(lo, hi)` -> the range in the file's own
    form (runs on a helper thread), `to_16k(engine, payload)` -> that range as a mono 16 kHz signal, converted on the
    device when the engine offers it (the signal then stays there) and by the host's converter otherwise."""
    rate = 0
    samples = 0

    def close(self) -> None:
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class _Mp3Spans(_Spans):
    """An MP3 or an Ogg Vorbis file: the native decoders deliver any sample range as mono float32."""

    def __init__(self, path, opener=None):
        self._stream = (opener or _audio.Mp3Stream)(path)
        self.rate, self.samples = self._stream.rate, self._stream.samples

    def read(self, lo: int, hi: int):
        return self._stream.read_mono(lo, hi - lo)                 # mono float32, the channel mean rounded once

    def to_16k(self, engine, payload):
        if hasattr(engine, "resample_16k"):
            return engine.resample_16k(payload, self.rate)
        return _audio.resample_poly(payload, self.rate, SAMPLE_RATE)

    def close(self) -> None:
        self._stream.close()


class _PcmSpans(_Spans):
    """Interleaved integer samples: a memory-mapped WAV's bytes (only the pages of a range are ever touched)."""

    def __init__(self, raw, channels: int, rate: int, sample_bytes: int, bits: int):
        self._raw = raw
        self.channels, self.rate, self.sample_bytes, self.bits = channels, rate, sample_bytes, bits
        self._frame = channels * sample_bytes
        self.samples = int(raw.shape[0]) // self._frame if raw is not None else 0

    def read(self, lo: int, hi: int):
        return self._raw[lo * self._frame: hi * self._frame]

    def to_16k(self, engine, payload):
        if hasattr(engine, "pcm_to_16k"):
            return engine.pcm_to_16k(payload, self.channels, self.rate, self.sample_bytes, self.bits,
                                     on_device=hasattr(engine, "submit"))
        x = _audio.pcm_to_float(payload, self.sample_bytes, self.bits).reshape(-1, self.channels)
        mono = x.mean(axis=1, dtype=np.float64) if self.channels > 1 else x[:, 0]     # as `load_audio_16k` does
        return _audio.resample_poly(mono, self.rate, SAMPLE_RATE)


class _FlacSpans(_PcmSpans):
    """A FLAC file: `read` decodes the frames that cover the range (`oasr_flac_decode_range`)."""

    def __init__(self, path):
        self._stream = _audio.FlacStream(path)
        st = self._stream
        super().__init__(None, st.channels, st.rate, st.sample_bytes, st.bits)
        self.samples = st.samples
        self.seekable = st.seekable

    def read(self, lo: int, hi: int):
        return self._stream.read_pcm(lo, hi - lo)

    def close(self) -> None:
        self._stream.close()


def _is_deterministic(exc: BaseException) -> bool:
    """Errors that a second attempt or the next batch would meet again: a rejected kernel argument or an
    unsupported shape (OasrError), a malformed batch (ValueError / TypeError), exhausted host or device memory,
    a device context that has faulted (torch reports both of the latter as RuntimeError subclasses or
    RuntimeErrors whose text starts with the runtime's name).  The reference tolerates a failed chunk because
    its chunk is a network call (ref :635-641); none of these is that kind of failure, so they are never
    swallowed - a transcript with chunks silently missing would be the result."""
    from omnilingual_asr._hip import OasrError
    if isinstance(exc, (OasrError, ValueError, TypeError, AssertionError, MemoryError)):
        return True
    if type(exc).__name__ in ("OutOfMemoryError", "AcceleratorError"):
        return True
    if isinstance(exc, RuntimeError):
        text = str(exc).lstrip()
        return text.startswith(("HIP error", "CUDA error", "HIP out of memory", "CUDA out of memory"))
    return False


class GeminiASRPipeline:
    """Drop-in for the reference's engine class (ref gemini_pipeline.py:313-741) running locally.

    `api_key` is accepted and ignored (the reference raises without one, ref :329-334; a local
    engine needs none).  `model` names an encoder geometry ("G300", "G7B"); the reference's remote
    model names select the default.  `engine` may be injected (anything with
    `encode_batch(pcm[B,N], n_valid) -> (hidden[B,T,d], lengths)`); by default a `HipEngine` is
    created on first use and its absence (no library, no GPU) is an error.

    `chunk_seconds` / `min_chunking_seconds` default to 30 s / 30 s — the utterance length the encoder
    is sized and measured for — not to the reference's 300 s / 360 s (`CHUNK_DURATION_SECONDS`,
    `MIN_DURATION_FOR_CHUNKING`, ref :217-218).  Consequence: a file between 30 s and 360 s comes back
    as one segment per 30 s where the reference returns what its remote model segments; pass
    `chunk_seconds=300, min_chunking_seconds=360` for the reference's plan (every chunk length runs on
    the device path)."""

    def __init__(self, api_key: Optional[str] = None, model: str = "gemini-3-flash-preview", *,
                 engine: Any = None, device: Any = None, chunk_seconds: float = DEFAULT_CHUNK_SECONDS,
                 min_chunking_seconds: float = DEFAULT_MIN_CHUNKING_SECONDS,
                 batch_size: int = DEFAULT_BATCH, seed: int = 0, process_group: Any = None,
                 fp8: bool = False) -> None:
        self.api_key = api_key
        self.model = model
        self.chunk_seconds = float(chunk_seconds)
        self.min_chunking_seconds = float(min_chunking_seconds)
        self.batch_size = int(batch_size)
        self.process_group = process_group
        self._device = device
        self._seed = seed
        self._fp8 = bool(fp8)
        self._engine = engine
        self._engine_lock = threading.Lock()   # one device stream: concurrent callers serialise here
        self._per_thread = threading.local()   # diagnostics of the calling thread's last call

    @property
    def decoded_source_samples(self) -> int:
        """Source samples the calling thread's last segmented (`_encode_streamed`) call decoded: its share of the
        file under a process group, the rate converter's reach on both sides of every segment included."""
        return getattr(self._per_thread, "decoded_source_samples", 0)

    # ---- engine ---------------------------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            from omnilingual_asr.engine_hip import HipEngine
            self._engine = HipEngine(self.model, seed=self._seed, device=self._device, fp8=self._fp8)
        return self._engine

    def _encode_chunks(self, pcm: np.ndarray, bounds: np.ndarray, first_idx: int = 0) -> List[ChunkRecord]:
        """Run the chunks `bounds[i] = (start, end)` of `pcm` through the engine in batches.  An engine
        with `submit`/`collect` is driven two batches deep: batch k+1 is assembled in page-locked memory
        and uploaded while the device works on batch k.  A batch whose engine call fails with an error
        that is not deterministic (see `_is_deterministic`) is reported and skipped, as the reference skips
        a failed chunk (ref :635-641); deterministic errors propagate."""
        records: List[ChunkRecord] = []
        n = len(bounds)
        pipelined = hasattr(self.engine, "submit")

        def finish(lo, part, lengths, toks):
            for i, (s, e) in enumerate(part):
                records.append(ChunkRecord(first_idx + lo + i, int(s), int(e), int(lengths[i]),
                                           toks[i][0], toks[i][1]))

        def tolerate(exc):
            if _is_deterministic(exc):
                raise exc
            print(f"Chunk transcription failed: {exc}")

        def collect(job):
            lo, part, ticket = job
            try:
                finish(lo, part, *self.engine.collect(ticket))
            except Exception as exc:
                tolerate(exc)

        in_flight = None
        lock = self._engine_lock if pipelined else contextlib.nullcontext()
        with lock:      # the two-deep path keeps the engine for the whole file
            for k, lo in enumerate(range(0, n, self.batch_size)):
                part = bounds[lo: lo + self.batch_size]
                longest = max(int(max(e - s for s, e in part)), 400)
                width = longest
                if hasattr(self.engine, "bucket_samples"):     # nearby lengths share one device shape
                    width = self.engine.bucket_samples(longest)
                if width < longest:
                    raise ValueError("batch width %d does not cover its longest chunk (%d samples)" % (width, longest))
                n_valid = [int(e - s) for s, e in part]
                on_device = pipelined and not isinstance(pcm, np.ndarray)   # a float32 device tensor (see _load_audio)
                if pipelined:
                    ticket, batch = self.engine.host_batch_slot(len(part), width, k)
                else:
                    ticket, batch = None, np.empty((len(part), width), dtype=np.float32)
                for i, (s, e) in enumerate(() if on_device else part):
                    if pcm.dtype == np.int16:      # 16-bit samples kept as read: scaled while copied
                        np.multiply(pcm[s:e], np.float32(1.0 / 32768.0), out=batch[i, : e - s])
                    else:
                        batch[i, : e - s] = pcm[s:e]
                    batch[i, e - s:] = 0.0
                job = None
                try:
                    if on_device:       # the signal never left the device: the batch is gathered there
                        self.engine.submit(ticket, n_valid, source=(pcm, [int(s) for s, _ in part]))
                        job = (lo, part, ticket)
                    elif pipelined:
                        self.engine.submit(ticket, n_valid)
                        job = (lo, part, ticket)
                    else:
                        with self._engine_lock:
                            if hasattr(self.engine, "transcribe_batch"):
                                _, lengths, toks = self.engine.transcribe_batch(batch, n_valid)
                            else:
                                _, lengths = self.engine.encode_batch(batch, n_valid)
                                toks = [(None, None)] * len(part)
                            finish(lo, part, lengths, toks)
                except Exception as exc:
                    tolerate(exc)
                if in_flight is not None:
                    collect(in_flight)
                in_flight = job
            if in_flight is not None:
                collect(in_flight)
        return records

    def _load_audio(self, audio_path) -> np.ndarray:
        """Mono float32 at 16 kHz (int16 samples, value / 32768, for a mono 16-bit WAV already at 16 kHz: see
        `load_audio_16k`); the rate conversion runs on the device when the engine offers it."""
        if hasattr(self.engine, "pcm_to_16k"):
            mapped = _audio.map_wav_pcm(audio_path)        # integer PCM WAV: the device takes the file's samples as they are
            if mapped is None and _audio.is_flac(audio_path):
                mapped = _audio.read_flac_pcm(audio_path)  # FLAC: decoded to the same form by the native decoder
            if mapped is not None:
                samples, channels, rate, sample_bytes, bits = mapped
                if channels == 1 and rate == SAMPLE_RATE and sample_bytes == 2 and bits == 16:
                    return samples.view("<i2") if samples.dtype == np.uint8 else samples
                with self._engine_lock:
                    # the converted signal stays on the device when the chunk loop can gather batches there
                    return self.engine.pcm_to_16k(samples, channels, rate, sample_bytes, bits,
                                                  on_device=hasattr(self.engine, "submit"))
        if hasattr(self.engine, "resample_16k") and hasattr(self.engine, "submit"):
            lossy = (_audio.read_ogg_mono if _audio.is_ogg(audio_path) else
                     _audio.read_mp3_mono if _audio.is_mp3(audio_path) else None)
            if lossy is not None:
                mono, rate = lossy(audio_path)                     # host threads decode; the rate conversion runs on
                with self._engine_lock:                            # the device and the 16 kHz signal stays there
                    return self.engine.resample_16k(mono, rate)
        resampler = None
        if hasattr(self.engine, "resample_16k"):
            def resampler(mono, rate):
                with self._engine_lock:
                    return self.engine.resample_16k(mono, rate).cpu().numpy()
        return _audio.load_audio_16k(Path(audio_path), resampler, keep_int16=True)

    def _encode_file(self, pcm: np.ndarray, word_timestamps: bool) -> GeminiTranscriptionResult:
        """Plan on the decoded sample count, encode (this rank's shard when a process group is set),
        shift every chunk to file time and merge (ref :541-575, :646-678).  The header duration only
        picks the entry point (ref :713-714); a header that could not be read (duration 0.0) must not
        turn hours of audio into one utterance, so the plan is made here, from the samples."""
        bounds = plan_chunks(pcm.shape[0], self.chunk_seconds, self.min_chunking_seconds)
        if self.process_group is not None and len(bounds) > 1:
            from omnilingual_asr import distributed as _dist
            records = _dist.encode_sharded(self, pcm, bounds, self.process_group)
        else:
            records = self._encode_chunks(pcm, bounds)
        if not records:
            raise RuntimeError("none of the %d chunk(s) could be transcribed" % len(bounds))
        if len(bounds) == 1:
            return records[0].to_result(word_timestamps=word_timestamps)
        return merge_results([self._transcribe_chunk(r, word_timestamps) for r in records])

    def _open_span_source(self, audio_path):
        """A reader that hands over any sample range of the file without the rest, or None: an MP3 or an Ogg Vorbis
        file (always), and - under a process group, where a rank needs only its shard - an integer PCM WAV or a FLAC
        whose samples have to be converted (a mono 16-bit WAV at 16 kHz is memory-mapped and sliced as it is)."""
        eng = self.engine
        ogg = _audio.is_ogg(audio_path)
        if ogg or _audio.is_mp3(audio_path):
            if self.process_group is None:
                if not (hasattr(eng, "submit") and hasattr(eng, "resample_16k")):
                    return None                 # one process, host engine: nothing to overlap the decode with
                # too short to be taken in segments (the caller decodes it whole, on all host threads): say so from
                # the header walk, before the stream is opened - opening copies and checks every page of an Ogg file
                n16 = int(math.ceil(_audio.audio_duration(audio_path) * SAMPLE_RATE))
                if len(plan_chunks(n16, self.chunk_seconds, self.min_chunking_seconds)) < 4 * self.batch_size:
                    return None
            return _Mp3Spans(audio_path, _audio.VorbisStream if ogg else None)
        if self.process_group is None:
            return None
        mapped = _audio.map_wav_pcm(audio_path)
        if mapped is not None:
            samples, channels, rate, sample_bytes, bits = mapped
            if channels == 1 and rate == SAMPLE_RATE and sample_bytes == 2 and bits == 16:
                return None
            return _PcmSpans(samples, channels, rate, sample_bytes, bits)
        if _audio.is_flac(audio_path):
            try:
                src = _FlacSpans(audio_path)
            except Exception:
                return None                     # the ordinary path reports what is wrong with the file
            if src.samples > 0 and src.seekable:
                return src
            src.close()
        return None

    def _encode_streamed(self, audio_path, word_timestamps: bool, report=None) -> Optional[GeminiTranscriptionResult]:
        """A long file in segments of two batches, each read, converted to 16 kHz and encoded on its own.  For an
        MP3 on one GPU this overlaps the host decode of segment k + 1 with the device's work on segment k instead
        of decoding the whole file first.  Under a process group it is what makes ingest sharded: a rank reads and
        converts only the samples of its own contiguous shard of chunks (an MP3's frames, a FLAC's frames, the
        bytes of a memory-mapped WAV; the ordinary path has every rank load the whole file), and the records cross
        in the one all-gather.  Returns None when the file or the engine is not of that kind (the ordinary path
        then runs).  The records equal the ordinary path's bit for bit: every segment starts on a whole period of
        the rate converter and carries enough input on both sides (a halo) for the converter's filter, so each
        16 kHz sample is computed from the same inputs by the same operations as in one pass over the file.
        `report`: the progress callback's steps 1 and 2 are raised here, before the first segment is read and
        before the records are exchanged and merged (ref :619, :644)."""
        from dataclasses import replace
        from fractions import Fraction
        eng = self.engine
        source = self._open_span_source(audio_path)
        if source is None:
            return None
        with source:
            rate, total = source.rate, source.samples
            fr = Fraction(SAMPLE_RATE, rate)
            up, down = fr.numerator, fr.denominator
            n16 = -(-total * up // down)
            bounds = plan_chunks(n16, self.chunk_seconds, self.min_chunking_seconds)
            per_segment = 2 * self.batch_size
            c_lo, c_hi = 0, len(bounds)
            if self.process_group is not None:
                if len(bounds) <= 1:
                    return None
                import torch.distributed as dist
                from omnilingual_asr import distributed as _dist
                c_lo, c_hi = _dist.shard_bounds(len(bounds), dist.get_world_size(self.process_group),
                                                dist.get_rank(self.process_group))
            elif len(bounds) < 2 * per_segment:
                return None
            firsts = list(range(c_lo, c_hi, per_segment))
            if any(int(bounds[c][0]) % up for c in firsts):
                return None                      # a segment would start inside a period of the converter
            halo = 0
            if rate != SAMPLE_RATE:
                _audio.conversion_ratio(rate, SAMPLE_RATE)           # refuses rates no filter is designed for
                half = _audio.polyphase_branches(rate, SAMPLE_RATE)[2]
                halo = down * -(-(half // up + 2) // down)

            def span(c0):
                c1 = min(c0 + per_segment, c_hi)
                a_out, b_out = int(bounds[c0][0]), int(bounds[c1 - 1][1])
                a_in = a_out * down // up
                b_in = total if c1 == len(bounds) else b_out * down // up
                lo, hi = max(0, a_in - halo), min(total, b_in + halo)
                return c1, a_out, b_out, a_in, lo, hi

            if report:
                report("transcribing", 1)
            from concurrent.futures import ThreadPoolExecutor
            records: List[ChunkRecord] = []
            decoded = [0]
            with ThreadPoolExecutor(max_workers=1, thread_name_prefix="oasr-decode") as pool:
                def fetch(c0):
                    _, _, _, _, lo, hi = span(c0)
                    decoded[0] += hi - lo
                    return source.read(lo, hi)
                pending = pool.submit(fetch, firsts[0]) if firsts else None
                for i, c0 in enumerate(firsts):
                    payload = pending.result()
                    if i + 1 < len(firsts):
                        pending = pool.submit(fetch, firsts[i + 1])
                    c1, a_out, b_out, a_in, lo, hi = span(c0)
                    with self._engine_lock:
                        y = source.to_16k(eng, payload)
                    drop = (a_in - lo) * up // down
                    signal = y[drop: drop + (b_out - a_out)]
                    if signal.shape[0] != b_out - a_out:
                        raise ValueError("segment of %d samples where %d were planned" % (signal.shape[0], b_out - a_out))
                    local = bounds[c0:c1] - a_out
                    for r in self._encode_chunks(signal, local, first_idx=c0):
                        records.append(replace(r, start_sample=r.start_sample + a_out, end_sample=r.end_sample + a_out))
            self._per_thread.decoded_source_samples = decoded[0]
        if report:
            report("processing", 2)
        if self.process_group is not None:
            records = _dist.exchange_records(records, bounds, self.process_group)
        if not records:
            raise RuntimeError("none of the %d chunk(s) could be transcribed" % len(bounds))
        return merge_results([self._transcribe_chunk(r, word_timestamps) for r in records])

    # ---- single-utterance entry (ref :474-539) ----------------------------------------------------------
    def transcribe(self, audio_path, *, progress_callback: Optional[Callable[[str, int], None]] = None,
                   language: Optional[str] = None, speaker_count: Optional[str] = None,
                   word_timestamps: bool = False, _pcm: Optional[np.ndarray] = None) -> GeminiTranscriptionResult:
        def report(step, idx):
            if progress_callback:
                progress_callback(step, idx)

        report("uploading", 0)
        pcm = self._load_audio(audio_path) if _pcm is None else _pcm
        report("transcribing", 1)
        result = self._encode_file(pcm, word_timestamps)
        report("processing", 2)
        report("done", 3)
        return result

    def _transcribe_chunk(self, record: ChunkRecord, word_timestamps: bool = False) -> GeminiTranscriptionResult:
        """Chunk-local result shifted to file time (ref :541-575)."""
        return offset_result(record.to_result(word_timestamps=word_timestamps),
                             record.start_sample / float(SAMPLE_RATE))

    # ---- long-audio entry (ref :577-682) ----------------------------------------------------------------
    def transcribe_chunked(self, audio_path, *, progress_callback: Optional[Callable[[str, int], None]] = None,
                           language: Optional[str] = None, speaker_count: Optional[str] = None,
                           word_timestamps: bool = False) -> GeminiTranscriptionResult:
        def report(step, idx):
            if progress_callback:
                progress_callback(step, idx)

        report("uploading", 0)
        # a long compressed file (decode and encode overlap) or, under a process group, this rank's shard only
        streamed = self._encode_streamed(audio_path, word_timestamps, report)
        if streamed is not None:
            report("done", 3)
            return streamed
        pcm = self._load_audio(audio_path)
        bounds = plan_chunks(pcm.shape[0], self.chunk_seconds, self.min_chunking_seconds)
        if len(bounds) <= 1:
            # same fall-through as the reference (ref :609-616), including its repeated progress steps;
            # the samples decoded above are handed over, not decoded again
            return self.transcribe(audio_path, progress_callback=progress_callback,
                                   language=language, speaker_count=speaker_count,
                                   word_timestamps=word_timestamps, _pcm=pcm)
        report("transcribing", 1)
        merged = self._encode_file(pcm, word_timestamps)
        report("processing", 2)
        report("done", 3)
        return merged

    # ---- gate + retry (ref :684-741) ----------------------------------------------------------------------
    def transcribe_with_retry(self, audio_path, *, max_retries: int = 3,
                              progress_callback: Optional[Callable[[str, int], None]] = None,
                              language: Optional[str] = None, speaker_count: Optional[str] = None,
                              word_timestamps: bool = False) -> GeminiTranscriptionResult:
        import time

        audio_path = Path(audio_path)
        use_chunking = get_audio_duration(audio_path) > self.min_chunking_seconds
        run = self.transcribe_chunked if use_chunking else self.transcribe
        last_error = None
        for attempt in range(max_retries):
            try:
                return run(audio_path, progress_callback=progress_callback, language=language,
                           speaker_count=speaker_count, word_timestamps=word_timestamps)
            except Exception as exc:
                last_error = exc
                if attempt < max_retries - 1:
                    time.sleep(2 ** attempt)
        raise RuntimeError(f"Failed to transcribe after {max_retries} attempts: {last_error}")


ASRPipeline = GeminiASRPipeline
