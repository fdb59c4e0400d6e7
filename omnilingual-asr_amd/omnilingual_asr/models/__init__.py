"""Model-side modules (mirrors ref:src/omnilingual_asr/models/__init__.py)."""

from omnilingual_asr.models import inference as inference

__all__ = ["inference"]
