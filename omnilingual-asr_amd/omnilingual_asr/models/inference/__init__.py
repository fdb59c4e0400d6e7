"""Engine-level API (mirrors ref:src/omnilingual_asr/models/inference/__init__.py:9-21)."""

from omnilingual_asr.models.inference.asr_pipeline import (
    ASRPipeline,
    GeminiASRPipeline,
    GeminiTranscriptionResult,
    GeminiTranscriptSegment,
    TranscriptionResult,
    TranscriptSegment,
    WordTimestamp,
)

__all__ = [
    "ASRPipeline",
    "GeminiASRPipeline",
    "GeminiTranscriptionResult",
    "GeminiTranscriptSegment",
    "TranscriptionResult",
    "TranscriptSegment",
    "WordTimestamp",
]
