"""Oracle for the long-audio chunk plan.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, without the ffmpeg side effects, the plan computed by
ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:243-310
(`split_audio_into_chunks`) and the gate at :217-219 / :713-714.
"""

from __future__ import annotations

import numpy as np

REF_CHUNK_SECONDS = 300.0        # ref gemini_pipeline.py:217
REF_MIN_CHUNKING_SECONDS = 360.0  # ref gemini_pipeline.py:218
REF_MAX_PARALLEL = 4             # ref gemini_pipeline.py:219


def use_chunking(duration_s: float, min_duration_s: float = REF_MIN_CHUNKING_SECONDS) -> bool:
    """Strict '>' gate, ref gemini_pipeline.py:713-714."""
    return duration_s > min_duration_s


def plan_starts_seconds(total_s: float, chunk_s: float = REF_CHUNK_SECONDS) -> list[float]:
    """Chunk start offsets in seconds, ref gemini_pipeline.py:256-310.

    `total_s <= 0` yields the single-entry plan `[0.0]` (ref :257-259).  Otherwise starts are
    produced by repeated float addition of `chunk_s` while `start < total_s` (ref :262-308):
    no overlap, no minimum tail, no empty trailing chunk on exact multiples.
    """
    if total_s <= 0:
        return [0.0]
    out = []
    t = 0.0
    while t < total_s:
        out.append(t)
        t += chunk_s
    return out


def chunk_names(n: int, ext: str = "") -> list[str]:
    """File names the reference gives its chunk files, ref gemini_pipeline.py:266-269."""
    ext = ext or ".wav"
    return ["chunk_%04d%s" % (i, ext) for i in range(n)]


def plan_chunks_samples(
    total_samples: int,
    sample_rate: int,
    chunk_seconds: float,
    min_chunking_seconds: float,
) -> np.ndarray:
    """Integer chunk plan `[n, 2]` of (start_sample, end_sample), int64.

    The sample-domain form of the reference plan: the duration gate
    (`total/sr > min_chunking_seconds`, strict) selects between one chunk covering the whole
    signal and fixed-length slices of `round(chunk_seconds*sr)` samples whose last slice is
    short.  For durations that are whole numbers of samples the starts equal
    `plan_starts_seconds(total/sr, chunk_seconds) * sr` exactly.
    """
    total = int(total_samples)
    if total <= 0:
        return np.array([[0, max(total, 0)]], dtype=np.int64)
    dur = total / float(sample_rate)
    if not use_chunking(dur, min_chunking_seconds):
        return np.array([[0, total]], dtype=np.int64)
    step = int(round(chunk_seconds * sample_rate))
    if step <= 0:
        raise ValueError("chunk_seconds too small")
    rows = []
    s = 0
    while s < total:
        rows.append((s, min(s + step, total)))
        s += step
    return np.asarray(rows, dtype=np.int64)


def partition_contiguous(n_items: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block partition used to shard chunks over ranks: the first `n % world`
    ranks own one extra item.  Returns [lo, hi)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    hi = lo + q + (1 if rank < r else 0)
    return lo, hi
