"""Top-level transcription API (mirrors ref:src/omnilingual_asr/diarization/__init__.py:9-19)."""

from omnilingual_asr.diarization.pipeline import (
    DiarizedTranscriptSegment,
    GeminiDiarizedTranscriptionPipeline,
    Pipeline,
    WordTimestamp,
)

__all__ = [
    "DiarizedTranscriptSegment",
    "GeminiDiarizedTranscriptionPipeline",
    "Pipeline",
    "WordTimestamp",
]
