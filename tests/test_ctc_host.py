"""CPU checks of the CTC output stage: tokenizer, the oracle's greedy rule, records that carry tokens."""
import numpy as np
import torch

from omnilingual_asr import distributed as D
from omnilingual_asr import tokenizer as T
from omnilingual_asr.models.inference.asr_pipeline import ChunkRecord
from oracle import ctc


def ids(text):
    return [T.SYMBOLS.index(c) for c in text]


def test_decode_text_and_words():
    toks = ids(" four score  and'") + [99, -1]
    frames = list(range(10, 10 + len(toks)))
    assert T.decode_text(toks) == "four score and'"
    words = T.words_with_frames(toks, frames)
    assert [w for w, _, _ in words] == ["four", "score", "and'"]
    assert words[0][1:] == (11, 15) and words[1][1:] == (16, 21) and words[2][1:] == (23, 27)
    assert T.decode_text([]) == "" and T.words_with_frames([], []) == []
    assert len(T.SYMBOLS) <= T.VOCAB_SIZE and T.SYMBOLS[T.BLANK] == "<blank>" and T.SYMBOLS[T.WORD_BOUNDARY] == " "


def test_greedy_rule_known_answers():
    # frames: a a _ a b b _ _ c  (blank = 0)  ->  a a b c  at frames 0 3 4 8; ties go to the lowest index
    seq = [2, 2, 0, 2, 3, 3, 0, 0, 4]
    lg = np.full((2, 12, 8), -5.0, dtype=np.float32)
    for t, k in enumerate(seq):
        lg[0, t, k] = 1.0
    lg[0, 9:] = 7.0                      # frames past the length are ignored
    lg[1, :, :] = 0.0                    # all ties -> index 0 = blank everywhere -> nothing
    lg[1, 5, 6] = 0.5
    out = ctc.greedy(lg, [9, 12], blank=0)
    np.testing.assert_array_equal(out[0][0], [2, 2, 3, 4])
    np.testing.assert_array_equal(out[0][1], [0, 3, 4, 8])
    np.testing.assert_array_equal(out[1][0], [6])
    np.testing.assert_array_equal(out[1][1], [5])
    assert out[0][0].dtype == np.int32 and out[0][1].dtype == np.int32


def test_records_carry_tokens_through_pack_unpack():
    recs = [ChunkRecord(3, 0, 480000, 1499, np.array([5, 1, 7], np.int32), np.array([2, 900, 1498], np.int32)),
            ChunkRecord(1, 480000, 960000, 1499, np.zeros((0,), np.int32), np.zeros((0,), np.int32)),
            ChunkRecord(2, 960000, 961000, 3, None, None)]
    packed = D.pack_records(recs, capacity=5, token_capacity=1500)
    assert packed.shape == (5, D.RECORD_WORDS + 1 + 1500) and packed.dtype == torch.int64
    back = D.unpack_records(torch.cat([packed, torch.zeros_like(packed)]))
    assert [r.chunk_idx for r in back] == [1, 2, 3]
    np.testing.assert_array_equal(back[2].tokens, [5, 1, 7])
    np.testing.assert_array_equal(back[2].frames, [2, 900, 1498])
    assert back[0].tokens.size == 0 and back[1].tokens is None
    # header-only records still work
    plain = D.unpack_records(D.pack_records(recs, capacity=4))
    assert [r.chunk_idx for r in plain] == [1, 2, 3] and plain[0].tokens is None


def test_oracle_logits_shape():
    h = torch.randn((2, 5, 16))
    w, b = torch.randn((32, 16)), torch.randn((32,))
    lg = ctc.logits(h, w, b)
    assert lg.shape == (2, 5, 32)
    torch.testing.assert_close(lg[1, 3], w @ h[1, 3] + b)
