"""Seeded random shapes through the C ABI against the oracle: ragged feature batches, CTC decoding,
resampling ratios.  One process, small sizes; complements the fixed cases of the other GPU tests."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_features_on_random_ragged_batches():
    """Frame counts, lengths (bit-exact) and normalised features (tolerance of the fixed-shape test) for
    random batch sizes and sample counts, including counts around the framing boundaries."""
    from omnilingual_asr.engine_hip import HipEngine
    from oracle import logmel as OL
    from tests.test_gpu_ops import synth_pcm

    eng = HipEngine("GTINY")
    rng = np.random.default_rng(11)
    dev = torch.device("cuda")
    edge = [0, 1, 399, 400, 559, 560, 719, 720, 721, 879, 880, 1040]
    for trial in range(10):
        batch = int(rng.integers(1, 6))
        n_list = [int(rng.choice(edge)) if rng.random() < 0.3 else int(rng.integers(1, 70000)) for _ in range(batch)]
        n_max = max(max(n_list), 400)
        src = synth_pcm(batch, n_max, seed=100 + trial)
        pcm = torch.zeros((batch, (n_max + 3) // 4 * 4))
        for i, n in enumerate(n_list):
            pcm[i, :n] = src[i, :n]
        feats, lengths = eng.features(pcm.to(dev), torch.tensor(n_list, dtype=torch.int32, device=dev))
        f_ref, l_ref = OL.features_batch(pcm.numpy(), n_list)
        assert lengths.cpu().tolist() == list(l_ref), (trial, n_list)
        got = feats.cpu()
        for i, n in enumerate(l_ref):
            if n:
                assert float((got[i, :n] - torch.from_numpy(f_ref)[i, :n]).abs().max()) <= 5e-3, (trial, i, n_list)
            assert (got[i, n:] == 0).all()                      # rows past the valid length are zero


def test_ctc_greedy_on_random_logits():
    from omnilingual_asr import _hip as H
    from oracle import ctc

    rng = np.random.default_rng(5)
    st = torch.cuda.current_stream().cuda_stream
    for trial in range(12):
        batch = int(rng.integers(1, 5))
        t_pad = int(rng.integers(1, 40)) * 32
        vocab = int(rng.integers(2, 200))
        ld = (vocab + 3) // 4 * 4 + 4 * int(rng.integers(0, 3))
        lens = [int(rng.integers(0, t_pad + 1)) for _ in range(batch)]
        lg = rng.integers(0, 3, (batch, t_pad, ld)).astype(np.float32)      # coarse values: ties everywhere
        lg[:, :, vocab:] = 50.0
        lg_d = torch.from_numpy(lg.reshape(-1, ld)).cuda()
        len_d = torch.tensor(lens, dtype=torch.int32, device="cuda")
        tok = torch.full((batch, t_pad), 9, dtype=torch.int32, device="cuda")
        frm = torch.full((batch, t_pad), 9, dtype=torch.int32, device="cuda")
        cnt = torch.full((batch,), -3, dtype=torch.int32, device="cuda")
        H.check(H.lib().oasr_ctc_greedy(lg_d.data_ptr(), ld, vocab, len_d.data_ptr(), batch, t_pad, 0,
                                        tok.data_ptr(), frm.data_ptr(), cnt.data_ptr(), st))
        torch.cuda.synchronize()
        ref = ctc.greedy(lg[:, :, :vocab], lens, blank=0)
        tok, frm, cnt = tok.cpu().numpy(), frm.cpu().numpy(), cnt.cpu().numpy()
        for b in range(batch):
            n = len(ref[b][0])
            assert cnt[b] == n, (trial, b)
            np.testing.assert_array_equal(tok[b, :n], ref[b][0])
            np.testing.assert_array_equal(frm[b, :n], ref[b][1])
            assert (tok[b, n:] == -1).all() and (frm[b, n:] == -1).all()


def test_resampler_on_random_rates():
    from omnilingual_asr import audio
    from omnilingual_asr.engine_hip import HipEngine

    eng = HipEngine("GTINY")
    rng = np.random.default_rng(8)
    for rate in [int(r) for r in rng.choice([7350, 11025, 12000, 24000, 32000, 37800, 44056, 47250, 88200, 96000], 6, replace=False)]:
        n = int(rng.integers(1, 30000))
        x = (0.3 * rng.standard_normal(n)).astype(np.float32)
        y = eng.resample_16k(x, rate).cpu().numpy()
        ref = audio.resample_poly(x, rate)
        assert y.shape == ref.shape, (rate, n)
        np.testing.assert_allclose(y, ref, rtol=0, atol=1.2e-7)
