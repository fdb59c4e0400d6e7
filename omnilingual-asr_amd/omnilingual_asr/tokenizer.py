"""Output vocabulary of the CTC head and the token -> text / word-timestamp mapping.

The reference gets text from its remote model (ref:src/omnilingual_asr/models/inference/
gemini_pipeline.py:512-530) and ships no tokenizer, so the vocabulary is declared here: index 0 is
the CTC blank, index 1 the word boundary, then the Latin letters and the apostrophe.  Encoder
frames are 20 ms apart (hop 160 at 16 kHz, two frames stacked), which is the grid the reference's
historical word tier used (ref:gettysburg.eaf time slots).
"""

from __future__ import annotations

from typing import List, Sequence, Tuple

BLANK = 0
WORD_BOUNDARY = 1
SYMBOLS: Tuple[str, ...] = ("<blank>", " ") + tuple("abcdefghijklmnopqrstuvwxyz") + ("'",)
VOCAB_SIZE = 32                       # padded; ids >= len(SYMBOLS) never win (their bias is -1e30)
FRAME_SECONDS = 0.02


def decode_text(tokens: Sequence[int]) -> str:
    """Collapsed token ids (no blanks, no repeats) -> text.  Word boundaries become single spaces;
    leading / trailing / doubled boundaries are dropped."""
    chars = [SYMBOLS[t] for t in tokens if 0 < int(t) < len(SYMBOLS)]
    return " ".join("".join(chars).split())


def words_with_frames(tokens: Sequence[int], frames: Sequence[int]) -> List[Tuple[str, int, int]]:
    """Words as (text, first frame, last frame + 1): maximal runs of non-boundary tokens."""
    out: List[Tuple[str, int, int]] = []
    cur: List[str] = []
    first = last = 0
    for t, f in zip(tokens, frames):
        t, f = int(t), int(f)
        if t <= 0 or t >= len(SYMBOLS):
            continue
        if t == WORD_BOUNDARY:
            if cur:
                out.append(("".join(cur), first, last + 1))
                cur = []
            continue
        if not cur:
            first = f
        cur.append(SYMBOLS[t])
        last = f
    if cur:
        out.append(("".join(cur), first, last + 1))
    return out
