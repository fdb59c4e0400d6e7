"""CPU restatement of the CTC output stage.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Parity status: the reference has no decoder of its own (its text comes from a remote model,
ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:512-530), so this stage is
build-declared and "parity unpinned" against the reference; greedy CTC decoding itself is the
textbook rule (argmax per frame, merge repeats, drop blanks) and the integer outputs of the device
kernel are compared bit-for-bit with this file on identical logits.
"""

from __future__ import annotations

import numpy as np
import torch


def logits(hidden: torch.Tensor, ctc_w: torch.Tensor, ctc_b: torch.Tensor) -> torch.Tensor:
    """hidden [B, T, d] -> [B, T, vocab] in float32 (or the dtype of `hidden`)."""
    return hidden @ ctc_w.to(hidden.dtype).t() + ctc_b.to(hidden.dtype)


def greedy(logit_rows: np.ndarray, lengths, blank: int = 0):
    """logit_rows [B, T, V]; returns per utterance (ids, frames) int32 arrays.  Ties go to the lowest
    index (numpy argmax), a frame survives when its id is not blank and differs from the previous
    frame's id."""
    out = []
    for b in range(logit_rows.shape[0]):
        n = int(lengths[b])
        ids = np.argmax(logit_rows[b, :n], axis=-1).astype(np.int32)
        keep = ids != blank
        keep[1:] &= ids[1:] != ids[:-1]
        out.append((ids[keep], np.nonzero(keep)[0].astype(np.int32)))
    return out
