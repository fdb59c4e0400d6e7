"""GPU parity tests of the individual HIP operators, called through the C ABI (ctypes), against the
CPU oracle / plain PyTorch float32 on the same seeded inputs.

Tolerances (stated per test): float32 kernels are compared with absolute tolerances that reflect
float32 round-off of the chosen algorithm; bf16-MFMA kernels are compared after rounding the same
operands to bf16 on the CPU, so the remaining difference is accumulation order plus one output
rounding.
"""

import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from omnilingual_asr import _hip
    return _hip


def _stream():
    return torch.cuda.current_stream().cuda_stream


def synth_pcm(batch, n, seed=1234):
    g = torch.Generator().manual_seed(seed)
    x = 0.1 * torch.randn((batch, n + 1), generator=g)
    x = 0.5 * (x[:, 1:] + x[:, :-1])
    t = torch.arange(n, dtype=torch.float64) / 16000.0
    x = x + (0.1 * torch.sin(2 * math.pi * 440.0 * t)).to(torch.float32)[None, :]
    return x.clamp(-1.0, 1.0).contiguous()


# --------------------------------------------------------------------------------------------------
# log-mel + normalisation

@pytest.mark.parametrize("batch,n_list", [(2, [48000, 48000]), (3, [16000, 400, 15999]), (1, [480000])])
def test_logmel_and_features_match_oracle(batch, n_list):
    from oracle import logmel as O
    from omnilingual_asr.engine_hip import HipEngine, pad_frames
    H = _lib()
    n_max = max(n_list)
    stride = (n_max + 3) // 4 * 4
    pcm = torch.zeros((batch, stride))
    src = synth_pcm(batch, n_max)
    for i, n in enumerate(n_list):
        pcm[i, :n] = src[i, :n]
    dev = torch.device("cuda")
    pcm_d = pcm.to(dev)
    nv = torch.tensor(n_list, dtype=torch.int32, device=dev)
    frames = H.num_frames(n_max)
    t_pad = pad_frames(frames // 2)
    cap = 2 * t_pad
    slots = int(H.lib().oasr_logmel_stats_slots(cap))
    tables = torch.from_numpy(H.logmel_table_blob().view(np.int32)).to(dev)
    lm = torch.full((batch, cap, 80), float("nan"), device=dev)
    stats = torch.zeros((batch, slots, 80, 2), device=dev)
    feats = torch.full((batch, t_pad, 160), float("nan"), device=dev)
    lens = torch.zeros((batch,), dtype=torch.int32, device=dev)
    H.check(H.lib().oasr_logmel_f32(pcm_d.data_ptr(), stride, nv.data_ptr(), batch, cap, tables.data_ptr(),
                                    lm.data_ptr(), stats.data_ptr(), _stream()))
    H.check(H.lib().oasr_feature_norm_stack(lm.data_ptr(), stats.data_ptr(), nv.data_ptr(), batch, cap, t_pad,
                                            feats.data_ptr(), lens.data_ptr(), _stream()))
    torch.cuda.synchronize()
    lm_h, feats_h, lens_h = lm.cpu().numpy(), feats.cpu().numpy(), lens.cpu().numpy()
    for i, n in enumerate(n_list):
        nf = O.num_frames(n)
        assert lens_h[i] == nf // 2                       # integer frame counts are bit-exact
        if nf == 0:
            continue
        ref_lm = O.logmel(pcm[i, :n].numpy())
        # float32 FFT of signals scaled to 2^15: |err| in ln(power) well below 1e-3 for non-floor bins
        np.testing.assert_allclose(lm_h[i, :nf], ref_lm, atol=2e-3, rtol=0)
        assert np.isnan(lm_h[i, nf:]).all()               # rows past the last frame are untouched
        if nf >= 2:
            ref_f = O.features(pcm[i, :n].numpy())
            np.testing.assert_allclose(feats_h[i, : nf // 2], ref_f, atol=5e-3, rtol=0)
        assert (feats_h[i, nf // 2:] == 0).all()          # padding rows are zero


def test_logmel_with_a_row_pitch_that_is_not_16_byte_aligned():
    """Row pitch 10 002 samples: odd rows start 8 bytes off a 16-byte boundary, so their spans are copied
    4 bytes per lane instead of 16; 62 frames (one full block of 32, one of 30), the batch's last row
    ending exactly at the end of the allocation.  Same tolerance as the aligned case."""
    from oracle import logmel as O
    H = _lib()
    batch, stride, n_list = 3, 10002, [10002, 9999, 10002]
    pcm = synth_pcm(batch, stride, seed=77)
    for i, n in enumerate(n_list):
        pcm[i, n:] = 0
    dev = torch.device("cuda")
    pcm_d = pcm.to(dev).contiguous()
    nv = torch.tensor(n_list, dtype=torch.int32, device=dev)
    cap = 64
    slots = int(H.lib().oasr_logmel_stats_slots(cap))
    tables = torch.from_numpy(H.logmel_table_blob().view(np.int32)).to(dev)
    lm = torch.full((batch, cap, 80), float("nan"), device=dev)
    stats = torch.zeros((batch, slots, 80, 2), device=dev)
    H.check(H.lib().oasr_logmel_f32(pcm_d.data_ptr(), stride, nv.data_ptr(), batch, cap, tables.data_ptr(),
                                    lm.data_ptr(), stats.data_ptr(), _stream()))
    torch.cuda.synchronize()
    lm_h, st_h = lm.cpu().numpy(), stats.cpu().numpy()
    for i, n in enumerate(n_list):
        nf = O.num_frames(n)
        ref = O.logmel(pcm[i, :n].numpy())
        np.testing.assert_allclose(lm_h[i, :nf], ref, atol=2e-3, rtol=0)
        assert np.isnan(lm_h[i, nf:]).all()
        # per-block partial statistics: (mean, sum of squared deviations) of the block's frames
        for blk, (lo, hi) in enumerate([(0, 32), (32, nf)]):
            np.testing.assert_allclose(st_h[i, blk, :, 0], ref[lo:hi].mean(axis=0), atol=2e-3)
            np.testing.assert_allclose(st_h[i, blk, :, 1], ((ref[lo:hi] - ref[lo:hi].mean(axis=0)) ** 2).sum(axis=0),
                                       rtol=2e-2, atol=1e-2)


# --------------------------------------------------------------------------------------------------
# LayerNorm

@pytest.mark.parametrize("d", [160, 512, 1024, 4096])
@pytest.mark.parametrize("bf16", [0, 1])
@pytest.mark.parametrize("swish", [0, 1])
def test_layernorm(d, bf16, swish):
    if d == 160 and swish:
        pytest.skip("swish epilogue is not offered for the feature LayerNorm")
    H = _lib()
    g = torch.Generator().manual_seed(d + bf16)
    rows = 1030
    x = torch.randn((rows, d), generator=g) * 3.0 + 0.5
    w = 1.0 + 0.2 * torch.randn((d,), generator=g)
    b = 0.2 * torch.randn((d,), generator=g)
    stride = 192 if d == 160 else d
    dev = "cuda"
    y = torch.full((rows, stride), float("nan"), device=dev, dtype=torch.bfloat16 if bf16 else torch.float32)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    H.check(H.lib().oasr_layernorm(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), rows, d, stride,
                                   1e-5, bf16, swish, _stream()))
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.double(), (d,), w.double(), b.double(), 1e-5)
    if swish:
        ref = ref * torch.sigmoid(ref)
    got = y.float().cpu()
    tol = 4e-2 if bf16 else 2e-5       # bf16 output: 2^-8 relative on values up to ~8
    torch.testing.assert_close(got[:, :d].double(), ref, atol=tol, rtol=tol)
    if stride > d:
        assert (got[:, d:] == 0).all()


def test_layernorm2_matches_two_layernorms():
    H = _lib()
    g = torch.Generator().manual_seed(7)
    rows, d = 515, 1024
    x = torch.randn((rows, d), generator=g) * 2.0
    p = [1.0 + 0.1 * torch.randn((d,), generator=g) if i % 2 == 0 else 0.1 * torch.randn((d,), generator=g)
         for i in range(4)]
    xd = x.cuda()
    pd = [t.cuda() for t in p]
    y = torch.zeros((rows, d), device="cuda", dtype=torch.bfloat16)
    H.check(H.lib().oasr_layernorm2(xd.data_ptr(), pd[0].data_ptr(), pd[1].data_ptr(), pd[2].data_ptr(),
                                    pd[3].data_ptr(), y.data_ptr(), rows, d, 1e-5, _stream()))
    torch.cuda.synchronize()
    r1 = torch.nn.functional.layer_norm(x.double(), (d,), p[0].double(), p[1].double(), 1e-5)
    r2 = torch.nn.functional.layer_norm(r1, (d,), p[2].double(), p[3].double(), 1e-5)
    torch.testing.assert_close(xd.cpu().double(), r1, atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(y.float().cpu().double(), r2, atol=4e-2, rtol=4e-2)


# --------------------------------------------------------------------------------------------------
# GEMM

def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("m,n,k", [(256, 256, 64), (512, 768, 192), (1024, 512, 1024)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_gemm_epilogues(m, n, k, epi):
    H = _lib()
    g = torch.Generator().manual_seed(100 * epi + k)
    # asymmetric, non-constant operands so that a transposed or permuted tile cannot pass
    a = _bf(torch.randn((m, k), generator=g))
    w = _bf(torch.randn((n, k), generator=g) / math.sqrt(k))
    bias = torch.randn((n,), generator=g)
    resid = torch.randn((m, n), generator=g)
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    acc = a.double() @ w.double().t()
    if epi == H.EPI_GLU_BF16:
        from omnilingual_asr.weights import glu_interleave
        wd = _bf(glu_interleave(w.float())).cuda()
        c = torch.full((m, n // 2), float("nan"), device="cuda", dtype=torch.bfloat16)
        H.check(H.lib().oasr_gemm_bf16(ad.data_ptr(), wd.data_ptr(), None, None, c.data_ptr(), m, n, k, 1.0, epi, _stream()))
        ref = acc[:, : n // 2] * torch.sigmoid(acc[:, n // 2:])
    elif epi == H.EPI_RESIDUAL_F32:
        c = resid.cuda().clone()
        H.check(H.lib().oasr_gemm_bf16(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), c.data_ptr(), c.data_ptr(), m, n, k, 0.5, epi, _stream()))
        ref = resid.double() + 0.5 * (acc + bias.double())
    elif epi == H.EPI_BIAS_F32:
        c = torch.full((m, n), float("nan"), device="cuda")
        H.check(H.lib().oasr_gemm_bf16(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, c.data_ptr(), m, n, k, 1.0, epi, _stream()))
        ref = acc + bias.double()
    else:
        c = torch.full((m, n), float("nan"), device="cuda", dtype=torch.bfloat16)
        H.check(H.lib().oasr_gemm_bf16(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, c.data_ptr(), m, n, k, 1.0, epi, _stream()))
        ref = acc + bias.double()
        if epi == H.EPI_BIAS_SWISH_BF16:
            ref = ref * torch.sigmoid(ref)
    torch.cuda.synchronize()
    got = c.float().cpu().double()
    assert torch.isfinite(got).all()
    # operands are exact bf16 on both sides: float32 accumulation error ~1e-5; bf16 outputs add 2^-8 relative
    tol = 2e-4 if c.dtype == torch.float32 else 2.5e-2
    torch.testing.assert_close(got, ref, atol=tol, rtol=tol)


def test_gemm_identity_asymmetric():
    """A = I against an asymmetric W catches a swapped row/column map in the C write."""
    H = _lib()
    m = n = k = 256
    a = _bf(torch.eye(m))
    w = _bf((torch.arange(n)[:, None] * 3 + torch.arange(k)[None, :] % 7).float() % 64)
    c = torch.zeros((m, n), device="cuda")
    ad, wd = a.cuda(), w.cuda()
    H.check(H.lib().oasr_gemm_bf16(ad.data_ptr(), wd.data_ptr(), None, None, c.data_ptr(), m, n, k, 1.0,
                                   H.EPI_BIAS_F32, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(c.cpu(), w.float().t().contiguous())


@pytest.mark.parametrize("m,n,k,epi", [(4352, 4096, 32, 0), (4352, 4096, 64, 4), (2304, 8192, 96, 3)])
def test_gemm_bias_with_several_tiles_per_work_group_and_short_k(m, n, k, epi):
    """More tiles than work-groups and one to three K slices per tile: every tile after a work-group's
    first takes its bias from the strip requested during the previous tile."""
    H = _lib()
    g = torch.Generator().manual_seed(m + k)
    a = _bf(torch.randn((m, k), generator=g))
    w = _bf(torch.randn((n, k), generator=g) / math.sqrt(k))
    bias = torch.randn((n,), generator=g) * 3
    resid = torch.randn((m, n), generator=g) if epi == 3 else None
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    c = resid.cuda() if epi == 3 else torch.zeros((m, n), dtype=torch.bfloat16 if epi == 0 else torch.float32, device="cuda")
    H.check(H.lib().oasr_gemm_bf16(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), c.data_ptr() if epi == 3 else None,
                                   c.data_ptr(), m, n, k, 0.5 if epi == 3 else 1.0, epi, _stream()))
    torch.cuda.synchronize()
    ref = a.double() @ w.double().t() + bias.double()
    if epi == 3:
        ref = resid.double() + 0.5 * ref
    torch.testing.assert_close(c.cpu().double(), ref, atol=2e-2 if epi == 0 else 1e-4, rtol=1e-2 if epi == 0 else 1e-5)


@pytest.mark.parametrize("shape", [(4096, 768, 4096, 512), (8192, 1024, 1024, 512), (8192, 768, 1024, 512),
                                   (8192, 1536, 1024, 4096), (8192, 2304, 2048, 2048), (4096, 1536, 4096, 1024)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_small_tiles_equal_large_tiles_bit_for_bit(epi, shape):
    """Problems with fewer than 128 tiles of 256 x 256 run on smaller tiles: the producer / consumer kernel with
    128 x 128 tiles (first shape), 128 x 64 tiles (second: 128 x 128 tiles would occupy at most half of the CUs)
    or 96 x 64 tiles (third and fourth: the row count is a multiple of 96), or - more 128 x 128 tiles than CUs,
    the last two shapes - the kernel whose waves both copy and multiply, two work-groups to a CU.  All kernels
    perform the same operations per output element, so the first rows of a large problem equal a small problem
    on those rows exactly (this is what keeps an utterance's states independent of the batch it is in, and why K
    is never split over work-groups).  The fourth and sixth shapes are the FFN down- and up-projection of a
    single 30 s utterance (128 and 32 K slices)."""
    H = _lib()
    g = torch.Generator().manual_seed(epi)
    m_big, m_small, n, k = shape
    a = _bf(torch.randn((m_big, k), generator=g)).cuda()
    w = _bf(torch.randn((n, k), generator=g) / math.sqrt(k)).cuda()
    bias = None if epi == 2 else torch.randn((n,), generator=g).cuda()
    n_out = n // 2 if epi == 2 else n
    outs = []
    for m in (m_big, m_small):
        if epi == 3:
            c = torch.randn((m_big, n_out), generator=torch.Generator().manual_seed(9))[:m].contiguous().cuda()
        else:
            c = torch.zeros((m, n_out), dtype=torch.bfloat16, device="cuda")
        H.check(H.lib().oasr_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                       c.data_ptr() if epi == 3 else None, c.data_ptr(), m, n, k, 0.5, epi, _stream()))
        outs.append(c)
    torch.cuda.synchronize()
    assert torch.equal(outs[0][:m_small], outs[1])


def test_gemm_rejects_bad_shapes():
    H = _lib()
    z = torch.zeros((256, 256), device="cuda", dtype=torch.bfloat16)
    rc = H.lib().oasr_gemm_bf16(z.data_ptr(), z.data_ptr(), None, None, z.data_ptr(), 100, 256, 64, 1.0, 0, _stream())
    assert rc == -1 and b"multiples" in H.lib().oasr_last_error()


# --------------------------------------------------------------------------------------------------
# attention

def _attention_ref(qkv, dist_emb, lengths, n_heads, left, right):
    b, t, d3 = qkv.shape
    d = d3 // 3
    dh = d // n_heads
    q, k, v = [x.reshape(b, t, n_heads, dh).transpose(1, 2).double() for x in qkv.split(d, dim=-1)]
    pos = torch.arange(t)
    dist = (pos[None, :] - pos[:, None]).clamp(-left, right) + left
    scores = q @ k.transpose(-1, -2)
    qe = q @ dist_emb.double().t()
    scores = (scores + torch.gather(qe, 3, dist[None, None].expand(b, n_heads, t, t))) / math.sqrt(dh)
    ok = pos[None, :] < lengths[:, None]
    scores = scores.masked_fill(~ok[:, None, None, :], float("-inf"))
    p = torch.softmax(scores, dim=-1)
    return (p @ v).transpose(1, 2).reshape(b, t, d)


def _prescale_q(qkv, d, head=64):
    """The kernel takes q * log2(e) / sqrt(head), rounded to bf16: returns (device qkv, the float qkv whose
    q is exactly that rounded value scaled back, for the reference)."""
    cs = 1.4426950408889634 / math.sqrt(head)
    dev = qkv.clone().float()
    dev[..., :d] = dev[..., :d] * cs
    dev = _bf(dev)
    ref = dev.float().clone()
    ref[..., :d] = ref[..., :d] / cs
    return dev, ref


@pytest.mark.parametrize("batch,t_pad,lens,heads,head", [
    (2, 256, [256, 131], 2, 64), (1, 1504, [1499], 1, 64), (3, 96, [96, 1, 70], 4, 64),
    (2, 256, [256, 131], 2, 128), (1, 1504, [1499], 1, 128), (3, 96, [96, 1, 70], 3, 128), (2, 544, [300, 544], 2, 128)])
def test_attention_relkey(batch, t_pad, lens, heads, head):
    H = _lib()
    g = torch.Generator().manual_seed(t_pad + heads)
    d = heads * head
    qkv, qkv_ref = _prescale_q(torch.randn((batch, t_pad, 3 * d), generator=g), d, head)
    emb = torch.zeros((96, head))
    emb[:73] = torch.randn((73, head), generator=g) * (8.0 / math.sqrt(head))
    emb = _bf(emb)
    lengths = torch.tensor(lens, dtype=torch.int32)
    rows = (batch * t_pad + 255) // 256 * 256
    qkv_d = torch.zeros((rows, 3 * d), dtype=torch.bfloat16, device="cuda")
    qkv_d[: batch * t_pad] = qkv.reshape(-1, 3 * d).cuda()
    ctx = torch.zeros((rows, d), dtype=torch.bfloat16, device="cuda")
    emb_d, len_d = emb.cuda(), lengths.cuda()
    H.check(H.lib().oasr_attention_relkey(qkv_d.data_ptr(), emb_d.data_ptr(), len_d.data_ptr(),
                                          ctx.data_ptr(), batch, t_pad, heads, head, 64, 8, _stream()))
    torch.cuda.synchronize()
    ref = _attention_ref(qkv_ref, emb.float()[:73], lengths.long(), heads, 64, 8)
    got = ctx[: batch * t_pad].float().cpu().reshape(batch, t_pad, d).double()
    for i, n in enumerate(lens):
        # P is rounded to bf16 before P.V and the output is bf16: 2^-8 relative on |ctx| <~ 3
        torch.testing.assert_close(got[i, :n], ref[i, :n], atol=3e-2, rtol=3e-2)
    assert torch.isfinite(got).all()


def test_attention_at_the_reference_chunk_length():
    """One head over 15 008 keys — a 300 s chunk of the reference's plan (ref gemini_pipeline.py:217) —
    against a float64 reference computed in blocks of 1024 queries.  Queries are scaled up so the
    softmax is peaked and the result is not just the mean of V.  Tolerance: relative L2 <= 1.5e-2 and
    3e-2 absolute (P and the output are rounded to bf16)."""
    H = _lib()
    t_pad, n, head = 15008, 14999, 64
    g = torch.Generator().manual_seed(77)
    qkv = torch.randn((1, t_pad, 3 * head), generator=g)
    qkv[..., :head] *= 3.0
    qkv, qkv_ref = _prescale_q(qkv, head, head)
    emb = torch.zeros((96, head))
    emb[:73] = torch.randn((73, head), generator=g) * (8.0 / math.sqrt(head))
    emb = _bf(emb)
    lengths = torch.tensor([n], dtype=torch.int32)
    rows = (t_pad + 255) // 256 * 256
    qkv_d = torch.zeros((rows, 3 * head), dtype=torch.bfloat16, device="cuda")
    qkv_d[:t_pad] = qkv.reshape(-1, 3 * head).cuda()
    ctx = torch.zeros((rows, head), dtype=torch.bfloat16, device="cuda")
    emb_d, len_d = emb.cuda(), lengths.cuda()
    H.check(H.lib().oasr_attention_relkey(qkv_d.data_ptr(), emb_d.data_ptr(), len_d.data_ptr(),
                                          ctx.data_ptr(), 1, t_pad, 1, head, 64, 8, _stream()))
    torch.cuda.synchronize()
    q, k, v = [x[0].double() for x in qkv_ref.split(head, dim=-1)]
    e = emb.double()[:73]
    want = torch.zeros((n, head), dtype=torch.float64)
    keys = torch.arange(n)
    for q0 in range(0, n, 1024):
        qb = q[q0: q0 + 1024][: n - q0]
        dist = (keys[None, :] - (q0 + torch.arange(qb.shape[0]))[:, None]).clamp(-64, 8) + 64
        sc = (qb @ k[:n].t() + torch.gather(qb @ e.t(), 1, dist)) / math.sqrt(head)
        want[q0: q0 + qb.shape[0]] = torch.softmax(sc, dim=-1) @ v[:n]
    got = ctx[:n].float().cpu().double()
    assert torch.isfinite(ctx.float()).all()
    rel = float((got - want).norm() / want.norm())
    assert rel <= 1.5e-2, rel
    torch.testing.assert_close(got, want, atol=3e-2, rtol=3e-2)
    assert float(want.abs().mean()) > 0.05          # the case is not the flat-softmax one


@pytest.mark.parametrize("d", [64, 128])
def test_attention_online_softmax_rescale_is_exercised(d):
    """A key late in the sequence that dominates every row forces the running maximum to jump at a
    later tile, so the rescale of the accumulated output is actually taken."""
    H = _lib()
    batch, t_pad, heads = 1, 256, 1
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn((batch, t_pad, 3 * d), generator=g) * 0.5
    qkv[0, :, :d] = qkv[0, :, :d].abs()                 # positive queries
    qkv[0, 200, d:2 * d] = 6.0                          # one key with a large positive dot product
    qkv, qkv_ref = _prescale_q(qkv, d, d)
    emb = _bf(torch.zeros((96, d)))
    lengths = torch.tensor([t_pad], dtype=torch.int32)
    ctx = torch.zeros((256, d), dtype=torch.bfloat16, device="cuda")
    qkv_d, emb_d, len_d = qkv.reshape(-1, 3 * d).cuda(), emb.cuda(), lengths.cuda()
    H.check(H.lib().oasr_attention_relkey(qkv_d.data_ptr(), emb_d.data_ptr(), len_d.data_ptr(), ctx.data_ptr(),
                                          batch, t_pad, heads, d, 64, 8, _stream()))
    torch.cuda.synchronize()
    ref = _attention_ref(qkv_ref, emb.float()[:73], lengths.long(), heads, 64, 8)
    torch.testing.assert_close(ctx.float().cpu().reshape(1, t_pad, d).double(), ref, atol=3e-2, rtol=3e-2)


# --------------------------------------------------------------------------------------------------
# depthwise convolution

@pytest.mark.parametrize("batch,t_pad,d", [(2, 64, 512), (1, 1504, 1024)])
def test_dwconv_causal(batch, t_pad, d):
    H = _lib()
    g = torch.Generator().manual_seed(d)
    x = _bf(torch.randn((batch, t_pad, d), generator=g))
    w = torch.randn((d, 31), generator=g) / math.sqrt(31)
    y = torch.full((batch * t_pad, d), float("nan"), device="cuda")
    xd, wd = x.reshape(-1, d).cuda(), w.t().contiguous().cuda()
    H.check(H.lib().oasr_dwconv_causal(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), batch, t_pad, d, 31, _stream()))
    torch.cuda.synchronize()
    xt = torch.nn.functional.pad(x.double().transpose(1, 2), (30, 0))
    ref = torch.nn.functional.conv1d(xt, w.double()[:, None, :], groups=d).transpose(1, 2)
    torch.testing.assert_close(y.cpu().double().reshape(batch, t_pad, d), ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("batch,t_pad,d", [(2, 64, 512), (3, 1504, 1024), (1, 48, 2048)])
def test_dwconv_ln_swish_fused(batch, t_pad, d):
    """The fused kernel against float64 conv -> LayerNorm -> Swish, and against the two separate kernels."""
    H = _lib()
    g = torch.Generator().manual_seed(d + 1)
    x = _bf(torch.randn((batch, t_pad, d), generator=g))
    w = torch.randn((d, 31), generator=g) / math.sqrt(31)
    ln_w = 1.0 + 0.1 * torch.randn((d,), generator=g)
    ln_b = 0.1 * torch.randn((d,), generator=g)
    rows = batch * t_pad
    xd, wd, lw, lb = x.reshape(-1, d).cuda(), w.t().contiguous().cuda(), ln_w.cuda(), ln_b.cuda()
    y = torch.zeros((rows, d), dtype=torch.bfloat16, device="cuda")
    H.check(H.lib().oasr_dwconv_ln_swish(xd.data_ptr(), wd.data_ptr(), lw.data_ptr(), lb.data_ptr(), y.data_ptr(),
                                         batch, t_pad, d, 31, 1e-5, _stream()))
    tmp = torch.zeros((rows, d), device="cuda")
    y2 = torch.zeros((rows, d), dtype=torch.bfloat16, device="cuda")
    H.check(H.lib().oasr_dwconv_causal(xd.data_ptr(), wd.data_ptr(), tmp.data_ptr(), batch, t_pad, d, 31, _stream()))
    H.check(H.lib().oasr_layernorm(tmp.data_ptr(), lw.data_ptr(), lb.data_ptr(), y2.data_ptr(), rows, d, d, 1e-5, 1, 1,
                                   _stream()))
    torch.cuda.synchronize()
    assert torch.equal(y, y2)          # same arithmetic in the same order: bit-identical
    xt = torch.nn.functional.pad(x.double().transpose(1, 2), (30, 0))
    conv = torch.nn.functional.conv1d(xt, w.double()[:, None, :], groups=d).transpose(1, 2)
    ref = torch.nn.functional.layer_norm(conv, (d,), ln_w.double(), ln_b.double(), 1e-5)
    ref = ref * torch.sigmoid(ref)
    torch.testing.assert_close(y.float().cpu().double().reshape(batch, t_pad, d), ref, atol=2e-2, rtol=1e-2)


def _pack_w_reference(w: torch.Tensor) -> torch.Tensor:
    """The packed weight layout restated with tensor operations: pieces of 16 rows x 32 k ordered
    [row group][k slice]; inside a piece, 16-byte chunk `pos` of row r holds source chunk pos ^ (-(r >> 2) & 3)."""
    n, k = w.shape
    v = w.view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 1, 3, 4)            # [group][slice][row][chunk][8]
    rows = torch.arange(16).view(16, 1)
    src = torch.arange(4).view(1, 4) ^ ((4 - ((rows >> 2) & 3)) & 3)          # [row][pos] -> source chunk
    return v[:, :, rows, src].reshape(n, k).contiguous()


@pytest.mark.parametrize("m,n,k,epi", [(4352, 4096, 1024, 1), (4352, 1024, 4096, 3), (768, 512, 64, 0),
                                       (2304, 2048, 96, 2), (48128, 1024, 192, 4)])
def test_gemm_packed_weights_equal_row_major_bit_for_bit(m, n, k, epi):
    """oasr_gemm_pack_w_bf16 against its restatement, and oasr_gemm_bf16_pw on the packed matrix against
    oasr_gemm_bf16 on the row-major one: both tile sizes, every epilogue, one to 128 K slices."""
    H = _lib()
    g = torch.Generator().manual_seed(m + n + k)
    a = _bf(torch.randn((m, k), generator=g)).cuda()
    w = _bf(torch.randn((n, k), generator=g) / math.sqrt(k))
    wd = w.cuda()
    wp = torch.empty_like(wd)
    H.check(H.lib().oasr_gemm_pack_w_bf16(wd.data_ptr(), n, k, wp.data_ptr(), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(wp.cpu().view(torch.int16), _pack_w_reference(w).view(torch.int16))
    bias = None if epi == 2 else torch.randn((n,), generator=g).cuda()
    resid = torch.randn((m, n), generator=g).cuda() if epi == 3 else None
    odt = torch.float32 if epi in (3, 4) else torch.bfloat16
    outs = []
    for fn, wmat in ((H.lib().oasr_gemm_bf16, wd), (H.lib().oasr_gemm_bf16_pw, wp)):
        c = torch.zeros((m, n // 2 if epi == 2 else n), dtype=odt, device="cuda")
        H.check(fn(a.data_ptr(), wmat.data_ptr(), None if bias is None else bias.data_ptr(),
                   None if resid is None else resid.data_ptr(), c.data_ptr(), m, n, k, 0.5, epi, _stream()))
        outs.append(c)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert H.lib().oasr_gemm_pack_w_bf16(wd.data_ptr(), n, k, wd.data_ptr(), _stream()) != 0      # in place: refused
    assert H.lib().oasr_gemm_pack_w_bf16(wd.data_ptr(), n + 8, k, wp.data_ptr(), _stream()) != 0   # n not a multiple of 16


@pytest.mark.parametrize("m,n,k,epi", [(4352, 4096, 1024, 1), (768, 512, 64, 0), (2304, 2048, 96, 1)])
def test_gemm_packed_output_then_packed_input(m, n, k, epi):
    """OASR_GEMM_C_PACKED writes the bf16 product in the packed layout (equal to packing the row-major
    product), and a second product taking it as packed A equals the one on the row-major matrix."""
    H = _lib()
    L = H.lib()
    g = torch.Generator().manual_seed(m * 3 + n + k)
    a = _bf(torch.randn((m, k), generator=g)).cuda()
    w = _bf(torch.randn((n, k), generator=g) / math.sqrt(k)).cuda()
    bias = torch.randn((n,), generator=g).cuda()
    c_row = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    c_pk = torch.zeros_like(c_row)
    H.check(L.oasr_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), bias.data_ptr(), None, c_row.data_ptr(), m, n, k, 1.0, epi, 0, _stream()))
    H.check(L.oasr_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), bias.data_ptr(), None, c_pk.data_ptr(), m, n, k, 1.0, epi,
                                H.GEMM_C_PACKED, _stream()))
    want = torch.empty_like(c_row)
    H.check(L.oasr_gemm_pack_w_bf16(c_row.data_ptr(), m, n, want.data_ptr(), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(c_pk, want)
    # second product: [m][n] x [n2][n]^T with the float32 residual epilogue, A row-major against A packed
    n2 = 512
    w2 = _bf(torch.randn((n2, n), generator=g) / math.sqrt(n)).cuda()
    b2 = torch.randn((n2,), generator=g).cuda()
    resid = torch.randn((m, n2), generator=g).cuda()
    outs = []
    for amat, flags in ((c_row, 0), (c_pk, H.GEMM_A_PACKED)):
        c2 = torch.zeros((m, n2), dtype=torch.float32, device="cuda")
        H.check(L.oasr_gemm_bf16_ex(amat.data_ptr(), w2.data_ptr(), b2.data_ptr(), resid.data_ptr(), c2.data_ptr(), m, n2, n,
                                    0.5, H.EPI_RESIDUAL_F32, flags, _stream()))
        outs.append(c2)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert L.oasr_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), bias.data_ptr(), None, c_pk.data_ptr(), m, n, k, 1.0,
                               H.EPI_BIAS_F32, H.GEMM_C_PACKED, _stream()) != 0   # float32 outputs have no packed form
    assert L.oasr_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), bias.data_ptr(), None, c_pk.data_ptr(), m, n, k, 1.0, epi, 8, _stream()) != 0


@pytest.mark.parametrize("n,k,epi,flags_c", [(4096, 1024, 1, 1), (4096, 1024, 3, 0), (2048, 1024, 2, 0), (4096, 256, 0, 0)])
def test_gemm_with_a_short_last_round_equals_the_rows_of_a_whole_number_of_rounds(n, k, epi, flags_c):
    """36 row tiles (six 30 s utterances) by 16 or 8 column tiles are 2.25 or 1.125 rounds of the 256 persistent
    work-groups: the launcher cuts the last four row tiles off and runs them on the small-tile kernels.  48 row tiles
    are whole rounds and run undivided; their first 9216 rows must equal the divided problem bit for bit - with the
    product written packed and read back as a packed left operand as well, whose offsets the cut has to get right."""
    H = _lib()
    L = H.lib()
    m_big, m = 12288, 9216
    g = torch.Generator().manual_seed(n + k + epi)
    a = _bf(torch.randn((m_big, k), generator=g)).cuda()
    w = _bf(torch.randn((n, k), generator=g) / math.sqrt(k)).cuda()
    bias = None if epi == 2 else torch.randn((n,), generator=g).cuda()
    n_out = n // 2 if epi == 2 else n
    resid = torch.randn((m_big, n_out), generator=g).cuda() if epi == 3 else None
    flags = H.GEMM_C_PACKED if flags_c else 0
    outs = []
    for rows in (m_big, m):
        c = torch.zeros((rows, n_out), dtype=torch.float32 if epi == 3 else torch.bfloat16, device="cuda")
        H.check(L.oasr_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(),
                                    None if resid is None else resid.data_ptr(), c.data_ptr(), rows, n, k, 0.5, epi, flags, _stream()))
        outs.append(c)
    torch.cuda.synchronize()
    assert torch.equal(outs[0][:m], outs[1])
    assert outs[1][m - 1024:].float().abs().max() > 0
    if flags_c:
        # the packed product as the left operand of a second product, cut the same way (36 x 16 tiles again)
        w2 = _bf(torch.randn((4096, n), generator=g) / math.sqrt(n)).cuda()
        row_major = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
        H.check(L.oasr_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), bias.data_ptr(), None, row_major.data_ptr(), m, n, k, 0.5, epi, 0, _stream()))
        second = []
        for amat, fl in ((row_major, 0), (outs[1], H.GEMM_A_PACKED)):
            c2 = torch.zeros((m, 4096), dtype=torch.bfloat16, device="cuda")
            H.check(L.oasr_gemm_bf16_ex(amat.data_ptr(), w2.data_ptr(), None, None, c2.data_ptr(), m, 4096, n, 1.0, 0, fl, _stream()))
            second.append(c2)
        torch.cuda.synchronize()
        assert torch.equal(second[0], second[1])
