"""Import-path compatibility: the reference keeps its engine in
`omnilingual_asr.models.inference.gemini_pipeline`; the local engine lives in `asr_pipeline`."""

from omnilingual_asr.models.inference.asr_pipeline import *  # noqa: F401,F403
from omnilingual_asr.models.inference.asr_pipeline import (  # noqa: F401
    CHUNK_DURATION_SECONDS,
    MAX_PARALLEL_CHUNKS,
    MIN_DURATION_FOR_CHUNKING,
    GeminiASRPipeline,
    GeminiTranscriptionResult,
    GeminiTranscriptSegment,
    WordTimestamp,
    get_audio_duration,
    get_mime_type,
    parse_timestamp,
    split_audio_into_chunks,
)
