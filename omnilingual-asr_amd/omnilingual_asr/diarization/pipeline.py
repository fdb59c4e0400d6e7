"""`Pipeline.transcribe()` — the drop-in boundary.

Same constructor keywords, `transcribe()` signature, return type, post-call properties and progress
protocol as ref:src/omnilingual_asr/diarization/pipeline.py:39-126.  Differences, all additive:
no API key is needed, extra keyword-only constructor arguments select the local engine, and the
post-call `summary` / `detected_languages` are kept per calling thread as well as on the instance,
so the up-to-4 concurrent calls the reference's server makes on one shared instance
(ref:workflows/wav2elan_web/app.py:383-399) do not read each other's values.
"""

from __future__ import annotations

import threading
from dataclasses import dataclass
from typing import Any, Callable, List, Optional


@dataclass(frozen=True)
class WordTimestamp:
    """Word-level timestamp (ref diarization/pipeline.py:15-20)."""
    word: str
    start: float
    end: float


@dataclass(frozen=True)
class DiarizedTranscriptSegment:
    """A transcribed segment; field order and defaults of ref diarization/pipeline.py:23-36."""
    start: float
    end: float
    speaker: str
    text: str
    words: list[WordTimestamp] | None = None
    language: str | None = None
    language_code: str | None = None
    languages: list[dict] | None = None
    emotion: str | None = None
    translation: str | None = None


class Pipeline:
    def __init__(self, *, api_key: Optional[str] = None, model: str = "gemini-3-flash-preview",
                 **engine_options: Any) -> None:
        """`api_key` is accepted for call-site compatibility and unused.  `model` selects an encoder
        geometry (see `omnilingual_asr.weights.GEOMETRIES`).  `engine_options` go to the engine class
        (`engine=`, `device=`, `chunk_seconds=`, `min_chunking_seconds=`, `batch_size=`, `seed=`,
        `process_group=`)."""
        from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline

        self.gemini = GeminiASRPipeline(api_key=api_key, model=model, **engine_options)
        self._summary: Optional[str] = None
        self._detected_languages: Optional[List[dict]] = None
        self._per_thread = threading.local()

    @property
    def summary(self) -> Optional[str]:
        """Summary of the last transcription made on this thread (else on any thread)."""
        return getattr(self._per_thread, "summary", self._summary)

    @property
    def detected_languages(self) -> Optional[List[dict]]:
        """Detected languages of the last transcription made on this thread (else on any thread)."""
        return getattr(self._per_thread, "languages", self._detected_languages)

    def transcribe(self, audio_path: str, *, word_timestamps: bool = False,
                   progress_callback: Optional[Callable[[str, int], None]] = None,
                   language: Optional[str] = None, speaker_count: Optional[str] = None,
                   **kwargs) -> List[DiarizedTranscriptSegment]:
        """Transcribe one file.  Unknown keyword arguments are accepted and ignored.  `words` is None
        as in the reference (ref diarization/pipeline.py:78-83, :117) unless `word_timestamps=True`,
        which the reference accepts but cannot honour; here it yields the CTC word times.  The callback
        receives ("uploading",0) ("transcribing",1) ("processing",2) ("done",3) on the calling
        thread."""
        result = self.gemini.transcribe_with_retry(
            audio_path, progress_callback=progress_callback, language=language,
            speaker_count=speaker_count, word_timestamps=bool(word_timestamps))
        self._summary = self._per_thread.summary = result.summary
        self._detected_languages = self._per_thread.languages = result.detected_languages
        return [
            DiarizedTranscriptSegment(
                start=s.start, end=s.end, speaker=s.speaker, text=s.text,
                words=None if s.words is None else [WordTimestamp(w.word, w.start, w.end) for w in s.words],
                language=s.language, language_code=s.language_code, languages=s.languages,
                emotion=s.emotion, translation=s.translation)
            for s in result.segments
        ]


GeminiDiarizedTranscriptionPipeline = Pipeline
