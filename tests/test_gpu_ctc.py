"""GPU tests of the CTC output stage: the cast and greedy-decode kernels through the C ABI (integer
results bit-exact against the oracle on the same logits), and the stage inside the engine/pipeline."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from omnilingual_asr import _hip
    return _hip


def _stream():
    return torch.cuda.current_stream().cuda_stream


def test_cast_bf16_rounds_to_nearest_even():
    H = _lib()
    g = torch.Generator().manual_seed(3)
    x = torch.randn((1000, 36), generator=g) * 3.0
    x[0, :4] = torch.tensor([1.00390625, 1.01171875, -1.00390625, 65280.0])   # ties and a large value
    xd = x.cuda()
    y = torch.zeros(x.shape, dtype=torch.bfloat16, device="cuda")
    H.check(H.lib().oasr_cast_bf16(xd.data_ptr(), y.data_ptr(), x.numel(), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), x.to(torch.bfloat16))


@pytest.mark.parametrize("batch,t_pad,vocab,ld,lens", [
    (3, 64, 32, 256, [64, 1, 37]),
    (2, 1504, 32, 256, [1499, 878]),
    (2, 96, 300, 304, [96, 50]),
    (1, 8192, 5, 8, [8000]),
    (2, 15008, 32, 32, [14999, 9001]),          # a 300 s chunk of the reference's plan: 15 008 frames
    (1, 2049, 70, 72, [2049]),                  # one frame past a collapse tile; 70 columns on 32 lanes
    (3, 40, 1000, 1000, [40, 0, 17]),           # several loads per lane; an empty utterance
])
def test_ctc_greedy_bit_exact(batch, t_pad, vocab, ld, lens):
    from oracle import ctc
    H = _lib()
    g = torch.Generator().manual_seed(t_pad + vocab)
    # coarse values: many exact ties, long runs of one winner, plenty of blanks
    lg = torch.randint(0, 4, (batch, t_pad, ld), generator=g).float()
    lg[:, :, 0] += (torch.rand((batch, t_pad), generator=g) < 0.4).float() * 2.0
    run = torch.randint(0, vocab, (batch, t_pad // 4 + 1), generator=g).repeat_interleave(4, dim=1)[:, :t_pad]
    lg.scatter_add_(2, run[:, :, None], torch.full((batch, t_pad, 1), 1.5))
    lg[:, :, vocab:] = 99.0                                   # columns past the vocabulary must be ignored
    lengths = torch.tensor(lens, dtype=torch.int32)
    lg_d, len_d = lg.reshape(-1, ld).contiguous().cuda(), lengths.cuda()
    tok = torch.full((batch, t_pad), 7, dtype=torch.int32, device="cuda")
    frm = torch.full((batch, t_pad), 7, dtype=torch.int32, device="cuda")
    cnt = torch.zeros((batch,), dtype=torch.int32, device="cuda")
    assert H.lib().oasr_ctc_workspace_bytes(batch, t_pad) == 4 * batch * t_pad
    ws = torch.full((batch * t_pad,), -5, dtype=torch.int32, device="cuda")
    H.check(H.lib().oasr_ctc_greedy(lg_d.data_ptr(), ld, vocab, len_d.data_ptr(), batch, t_pad, 0,
                                    tok.data_ptr(), frm.data_ptr(), cnt.data_ptr(), ws.data_ptr(), _stream()))
    torch.cuda.synchronize()
    ref = ctc.greedy(lg[:, :, :vocab].numpy(), lens, blank=0)
    tok, frm, cnt = tok.cpu().numpy(), frm.cpu().numpy(), cnt.cpu().numpy()
    for b in range(batch):
        n = len(ref[b][0])
        assert cnt[b] == n and (n > 0 or lens[b] == 0)
        np.testing.assert_array_equal(tok[b, :n], ref[b][0])
        np.testing.assert_array_equal(frm[b, :n], ref[b][1])
        assert (tok[b, n:] == -1).all() and (frm[b, n:] == -1).all()


def test_ctc_greedy_rejects_bad_arguments():
    H = _lib()
    z = torch.zeros((64, 256), device="cuda")
    i = torch.zeros((64,), dtype=torch.int32, device="cuda")
    args = lambda **kw: (z.data_ptr(), kw.get("ld", 256), kw.get("vocab", 32), i.data_ptr(), 1, kw.get("t_pad", 64),
                         kw.get("blank", 0), i.data_ptr(), i.data_ptr(), i.data_ptr(), kw.get("ws", i.data_ptr()), _stream())
    assert H.lib().oasr_ctc_greedy(*args(t_pad=0)) == -1
    assert H.lib().oasr_ctc_greedy(*args(ws=None)) == -1
    assert H.lib().oasr_ctc_greedy(*args(blank=32)) == -1
    assert H.lib().oasr_ctc_greedy(*args(ld=30)) == -1
    assert b"ld" in H.lib().oasr_last_error()


def test_engine_decode_matches_oracle_on_its_own_logits():
    """Tokens from the device are exactly the oracle's greedy decode of the device logits, and those
    logits are the float32 projection of the device encoder states up to bf16 operand rounding."""
    from omnilingual_asr.engine_hip import HipEngine
    from omnilingual_asr.weights import GEOMETRIES, init_random_weights
    from oracle import ctc
    from tests.test_gpu_ops import synth_pcm

    g = GEOMETRIES["GTINY"]
    w = init_random_weights(g, 11)
    eng = HipEngine("GTINY", weights=w)
    pcm = synth_pcm(3, 64000, seed=2)
    n_valid = [64000, 30000, 47777]
    hidden, lengths, toks = eng.transcribe_batch(pcm, n_valid)
    buf = eng._get_buffers(3, 64000)
    t_pad = buf["t_pad"]
    logits = buf["logits"][: 3 * t_pad].cpu().reshape(3, t_pad, -1)
    ref = ctc.greedy(logits[:, :, : g.vocab_size].numpy(), lengths, blank=0)
    for b in range(3):
        np.testing.assert_array_equal(toks[b][0], ref[b][0])
        np.testing.assert_array_equal(toks[b][1], ref[b][1])
        assert len(toks[b][0]) > 0
        want = ctc.logits(hidden[b, : lengths[b]].cpu(), w["ctc_w"], w["ctc_b"])
        got = logits[b, : lengths[b], : g.vocab_size]
        # |h| ~ 1, d = 512, bf16 operands: ~2^-9 * sqrt(512) * |w| absolute
        torch.testing.assert_close(got, want, atol=2e-2, rtol=2e-2)
    assert (logits[:, :, g.vocab_size:] < -1e29).all()      # padded vocabulary entries can never win


def test_pipeline_returns_text_and_word_times(golden_dir):
    from omnilingual_asr import Pipeline

    pipe = Pipeline(model="GTINY", seed=4)
    segs = pipe.transcribe(str(golden_dir / "gettysburg.wav"), word_timestamps=True)
    s = segs[0]
    assert s.text and set(s.text) <= set("abcdefghijklmnopqrstuvwxyz' ")
    assert " ".join(w.word for w in s.words) == s.text
    assert all(0.0 <= w.start < w.end <= 878 * 0.02 for w in s.words)
    assert pipe.transcribe(str(golden_dir / "gettysburg.wav"))[0].words is None
