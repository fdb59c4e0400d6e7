"""omnilingual_asr — local MI355X speech-encoder path behind the reference package's API.

Import surface of ref:src/omnilingual_asr/__init__.py:9-25 (same names, same `__version__`), served
by a local engine instead of a remote model.  `Pipeline` is the neutral name of the top-level
class; the reference's `Gemini*` names are kept as aliases so existing imports keep working.
"""

__version__ = "0.2.0"

from omnilingual_asr.diarization import (
    DiarizedTranscriptSegment,
    GeminiDiarizedTranscriptionPipeline,
    Pipeline,
    WordTimestamp,
)
from omnilingual_asr.models.inference import (
    GeminiASRPipeline,
    GeminiTranscriptionResult,
    GeminiTranscriptSegment,
)

__all__ = [
    "__version__",
    "GeminiASRPipeline",
    "GeminiTranscriptionResult",
    "GeminiTranscriptSegment",
    "GeminiDiarizedTranscriptionPipeline",
    "Pipeline",
    "DiarizedTranscriptSegment",
    "WordTimestamp",
]
