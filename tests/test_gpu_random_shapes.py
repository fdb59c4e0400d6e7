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
        t_pad = int(rng.integers(1, 40)) * 32 if trial % 3 else int(rng.integers(1, 5000))   # any pitch, past a collapse tile
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
        ws = torch.full((batch * t_pad,), 123, dtype=torch.int32, device="cuda")
        H.check(H.lib().oasr_ctc_greedy(lg_d.data_ptr(), ld, vocab, len_d.data_ptr(), batch, t_pad, 0,
                                        tok.data_ptr(), frm.data_ptr(), cnt.data_ptr(), ws.data_ptr(), st))
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


def test_gemm_on_random_shapes_and_epilogues():
    """Random m, n (multiples of 256) and k (multiples of 32), every epilogue, both tile sizes, against a
    float64 product of the same bf16 operands."""
    import math
    from omnilingual_asr import _hip as H
    from omnilingual_asr.weights import glu_interleave

    rng = np.random.default_rng(21)
    st = torch.cuda.current_stream().cuda_stream
    bf = lambda t: t.to(torch.bfloat16)
    for trial in range(14):
        m = 256 * int(rng.integers(1, 26))
        n = 256 * int(rng.integers(1, 20))
        k = 32 * int(rng.integers(1, 40))
        epi = int(rng.integers(0, 5))
        g = torch.Generator().manual_seed(trial)
        a = bf(torch.randn((m, k), generator=g))
        w = bf(torch.randn((n, k), generator=g) / math.sqrt(k))
        bias = torch.randn((n,), generator=g)
        alpha = float(rng.choice([1.0, 0.5]))
        acc = a.double() @ w.double().t()
        ad, bd = a.cuda(), bias.cuda()
        if epi == H.EPI_GLU_BF16:
            wd = bf(glu_interleave(w.float())).cuda()
            c = torch.full((m, n // 2), float("nan"), device="cuda", dtype=torch.bfloat16)
            args = (ad.data_ptr(), wd.data_ptr(), None, None, c.data_ptr())
            ref = acc[:, : n // 2] * torch.sigmoid(acc[:, n // 2:])
        elif epi == H.EPI_RESIDUAL_F32:
            wd = w.cuda()
            resid = torch.randn((m, n), generator=g)
            c = resid.cuda().clone()
            args = (ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), c.data_ptr(), c.data_ptr())
            ref = resid.double() + alpha * (acc + bias.double())
        else:
            wd = w.cuda()
            c = torch.full((m, n), float("nan"), device="cuda", dtype=torch.float32 if epi == H.EPI_BIAS_F32 else torch.bfloat16)
            args = (ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, c.data_ptr())
            ref = acc + bias.double()
            if epi == H.EPI_BIAS_SWISH_BF16:
                ref = ref * torch.sigmoid(ref)
        H.check(H.lib().oasr_gemm_bf16(*args, m, n, k, alpha if epi == H.EPI_RESIDUAL_F32 else 1.0, epi, st))
        torch.cuda.synchronize()
        tol = 2e-4 if c.dtype == torch.float32 else 2.5e-2
        torch.testing.assert_close(c.float().cpu().double(), ref, atol=tol, rtol=tol, msg=lambda s: f"{(trial, m, n, k, epi)}: {s}")


def test_attention_on_random_shapes():
    """Random batch, heads, head size, padded length and ragged lengths (down to one key)."""
    from omnilingual_asr import _hip as H
    from tests.test_gpu_ops import _attention_ref, _prescale_q

    rng = np.random.default_rng(31)
    st = torch.cuda.current_stream().cuda_stream
    for trial in range(10):
        head = int(rng.choice([64, 128]))
        heads = int(rng.integers(1, 5))
        batch = int(rng.integers(1, 4))
        t_pad = 32 * int(rng.integers(1, 20))
        lens = [int(rng.integers(1, t_pad + 1)) for _ in range(batch)]
        d = heads * head
        g = torch.Generator().manual_seed(200 + trial)
        qkv, qkv_ref = _prescale_q(torch.randn((batch, t_pad, 3 * d), generator=g), d, head)
        emb = torch.zeros((96, head))
        emb[:73] = torch.randn((73, head), generator=g) * (8.0 / head ** 0.5)
        emb = emb.to(torch.bfloat16)
        lengths = torch.tensor(lens, dtype=torch.int32)
        rows = (batch * t_pad + 255) // 256 * 256
        qkv_d = torch.zeros((rows, 3 * d), dtype=torch.bfloat16, device="cuda")
        qkv_d[: batch * t_pad] = qkv.reshape(-1, 3 * d).cuda()
        ctx = torch.zeros((rows, d), dtype=torch.bfloat16, device="cuda")
        emb_d, len_d = emb.cuda(), lengths.cuda()
        H.check(H.lib().oasr_attention_relkey(qkv_d.data_ptr(), emb_d.data_ptr(), len_d.data_ptr(), ctx.data_ptr(),
                                              batch, t_pad, heads, head, 64, 8, st))
        torch.cuda.synchronize()
        ref = _attention_ref(qkv_ref, emb.float()[:73], lengths.long(), heads, 64, 8)
        got = ctx[: batch * t_pad].float().cpu().reshape(batch, t_pad, d).double()
        assert torch.isfinite(got).all()
        for i, n in enumerate(lens):
            torch.testing.assert_close(got[i, :n], ref[i, :n], atol=3e-2, rtol=3e-2,
                                       msg=lambda s: f"{(trial, head, heads, batch, t_pad, lens, i)}: {s}")


def test_encoder_on_random_ragged_batches():
    """The whole device path (features + two-block encoder) on random batch sizes and lengths against
    the float32 oracle, bf16 tolerance; lengths bit-exact."""
    from omnilingual_asr.engine_hip import HipEngine
    from omnilingual_asr.weights import GEOMETRIES, init_random_weights
    from oracle import conformer as OC
    from oracle import logmel as OL
    from tests.test_gpu_ops import synth_pcm

    g = GEOMETRIES["GTINY"]
    w = init_random_weights(g, 7)
    eng = HipEngine(g, weights=w)
    og = OC.Geometry(name=g.name, d_model=g.d_model, ffn_dim=g.ffn_dim, n_heads=g.n_heads, n_layers=g.n_layers)
    rng = np.random.default_rng(41)
    for trial in range(5):
        batch = int(rng.integers(1, 6))
        n_list = [int(rng.integers(720, 90000)) for _ in range(batch)]
        if trial == 2:
            n_list[0] = 300                                 # one row without a single frame
        n_max = max(n_list)
        src = synth_pcm(batch, n_max, seed=300 + trial)
        pcm = torch.zeros((batch, (n_max + 3) // 4 * 4))
        for i, n in enumerate(n_list):
            pcm[i, :n] = src[i, :n]
        hidden, lengths = eng.encode_batch(pcm, n_list)
        feats, lens = OL.features_batch(pcm.numpy(), n_list)
        assert list(lengths) == list(lens), (trial, n_list)
        ref = OC.forward(torch.from_numpy(feats), torch.from_numpy(lens), w, og)
        got = hidden.cpu()
        for i, n in enumerate(lens):
            if n == 0:
                continue
            d = got[i, :n] - ref[i, :n]
            assert float(d.norm() / ref[i, :n].norm()) <= 2e-2 and float(d.abs().max()) <= 0.15, (trial, i, n_list)
