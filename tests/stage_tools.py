"""One encoder block on the device, operator by operator, each fed the ORACLE's input for that operator.

The block-level distance between the device and `oracle.conformer.forward(device_rounding=True)` cannot be
made small however faithful the emulation is: a rounding step turns a perturbation of relative size e into
one of size ~sqrt(e * q) (q = the step of the rounding: 2^-8 for bfloat16, 2^-3 for e4m3), because the few
values that land on the other side of a rounding boundary move by a whole step, so after a handful of
rounding stages two correct implementations that differ only in summation order sit near the rounding
noise itself.  Feeding every operator the oracle's input removes that chain: what is left is the
operator's own arithmetic, and the comparison is sharp (a few outputs in ten thousand differ, by one unit
in the last place).  Used by `tests/test_gpu_stages.py` and `tools/stage_parity.py`.
"""

from __future__ import annotations

import dataclasses
import math

import torch
import torch.nn.functional as F

F8 = torch.float8_e4m3fn


def block_stages(geometry, fp8: bool, n_list=(33000, 21000), seed: int = 21, weight_scale_error: float = 1.0):
    """Runs the stages of one block (geometry cut to one layer) and returns a list of
    (stage name, relative L2 of device against oracle over valid rows, share of outputs that differ).
    `weight_scale_error` != 1 hands the device's fp8 projections row scales that are wrong by that factor
    (a negative control: the comparison must notice)."""
    from omnilingual_asr import _hip as H
    from omnilingual_asr.weights import glu_interleave, init_random_weights, quantize_rows_e4m3
    from oracle import conformer as OC
    from oracle import logmel as OL
    from tests.test_gpu_ops import synth_pcm

    g = dataclasses.replace(geometry, n_layers=1)
    og = OC.Geometry(name=g.name, d_model=g.d_model, ffn_dim=g.ffn_dim, n_heads=g.n_heads, n_layers=1)
    w = init_random_weights(g, seed)
    L = w["layers"][0]
    n_list = list(n_list)
    n_max = max(n_list)
    src = synth_pcm(len(n_list), n_max, seed=6)
    pcm = torch.zeros((len(n_list), n_max))
    for i, n in enumerate(n_list):
        pcm[i, :n] = src[i, :n]
    feats, lens = OL.features_batch(pcm.numpy(), n_list)
    ft, lt = torch.from_numpy(feats), torch.from_numpy(lens)
    b, t, _ = ft.shape
    t_pad = (t + 31) // 32 * 32
    d, f, h, dh = g.d_model, g.ffn_dim, g.n_heads, g.head_dim
    rows = (b * t_pad + 255) // 256 * 256
    lib, st = H.lib(), H.current_stream_ptr
    dev, bf = "cuda", torch.bfloat16
    keep = []                    # device copies must outlive the launches that read them
    out = []

    def D(x, dtype=None):
        y = (x if dtype is None else x.to(dtype)).to(dev).contiguous()
        keep.append(y)
        return y.data_ptr()

    def pad_rows(x, dtype):      # [b, t, c] oracle tensor -> [rows, c] device tensor at pitch t_pad
        y = torch.zeros((rows, x.shape[-1]), dtype=dtype, device=dev)
        y[: b * t_pad].view(b, t_pad, -1)[:, :t] = x.to(dtype).to(dev)
        return y

    def report(stage, got_dev, want):
        v = got_dev[: b * t_pad].float().cpu().view(b, t_pad, -1)
        num = den = 0.0
        diff = tot = 0
        for i in range(b):
            gv, wv = v[i, : lens[i]], want[i, : lens[i]].float()
            num += float((gv - wv).pow(2).sum())
            den += float(wv.pow(2).sum())
            diff += int((gv != wv).sum())
            tot += wv.numel()
        out.append((stage, math.sqrt(num / max(den, 1e-30)), diff / max(tot, 1)))

    r = lambda x: x.to(bf).to(torch.float32)
    fq = None                    # set below: the oracle's e4m3 fake quantisation

    def layernorm_stage(name, x_in, lw, lb, swish=0):
        """LayerNorm (+ Swish) feeding a projection: bf16 rows, or (fp8) e4m3 rows + scales.  Returns the
        oracle's output (the next stage's input)."""
        y = F.layer_norm(x_in, (d,), lw, lb, og.ln_eps)
        want = r(OC._swish(y) if swish else y)
        xd = pad_rows(x_in, torch.float32)
        if not fp8:
            yd = torch.zeros((rows, d), dtype=bf, device=dev)
            H.check(lib.oasr_layernorm(xd.data_ptr(), D(lw), D(lb), yd.data_ptr(), rows, d, d, og.ln_eps, 1, swish, st()))
            report(name + " -> bf16", yd, want)
        else:
            q = torch.zeros((rows, d), dtype=torch.uint8, device=dev)
            s = torch.zeros((rows,), dtype=torch.float32, device=dev)
            H.check(lib.oasr_layernorm_fp8(xd.data_ptr(), D(lw), D(lb), q.data_ptr(), s.data_ptr(), rows, d, og.ln_eps, swish, st()))
            report(name + " -> e4m3 rows", q.view(F8).float() * s[:, None], fq(want))
        return want

    def project_stage(name, a_in, wmat, bias, epi, alpha=1.0, resid=None, n_out=None):
        """One projection on the oracle's (16-bit) input `a_in`; returns the oracle's output."""
        k, n = a_in.shape[-1], wmat.shape[0]
        if fp8:
            acc = F.linear(fq(a_in), fq(wmat), None)
        else:
            acc = F.linear(a_in, r(wmat), None)
        z = acc if bias is None else acc + bias
        if epi == H.EPI_BIAS_BF16:
            want = r(z)
        elif epi == H.EPI_BIAS_SWISH_BF16:
            want = r(OC._swish(z))
        elif epi == H.EPI_GLU_BF16:
            want = r(z[..., : n // 2] * torch.sigmoid(z[..., n // 2:]))
        else:
            want = resid + alpha * z
        w_dev = glu_interleave(wmat) if epi == H.EPI_GLU_BF16 else wmat
        c_cols = n // 2 if epi == H.EPI_GLU_BF16 else n
        if epi == H.EPI_RESIDUAL_F32:
            c = pad_rows(resid, torch.float32)
        else:
            c = torch.zeros((rows, c_cols), dtype=bf, device=dev)
        a = pad_rows(a_in, bf)
        bptr = None if bias is None else D(bias)
        rptr = c.data_ptr() if epi == H.EPI_RESIDUAL_F32 else None
        if not fp8:
            H.check(lib.oasr_gemm_bf16(a.data_ptr(), D(w_dev, bf), bptr, rptr, c.data_ptr(), rows, n, k, alpha, epi, st()))
        else:
            a8 = torch.zeros((rows, k), dtype=torch.uint8, device=dev)
            sa = torch.zeros((rows,), dtype=torch.float32, device=dev)
            H.check(lib.oasr_quant_fp8_rows(a.data_ptr(), rows, k, a8.data_ptr(), sa.data_ptr(), st()))
            codes, scale = quantize_rows_e4m3(w_dev)
            scale = scale * weight_scale_error
            H.check(lib.oasr_gemm_fp8(a8.data_ptr(), D(codes), sa.data_ptr(), D(scale), bptr, rptr, c.data_ptr(),
                                      rows, n, k, alpha, epi, st()))
        report(name + (" -> f32" if epi == H.EPI_RESIDUAL_F32 else " -> bf16"), c, want)
        return want

    OC._DEV = True
    try:
        fq = OC._fake_quant_rows
        x0 = OC.input_projection(ft, {k: v for k, v in w.items() if k != "layers"}, og)
        # half-step feed-forward
        xn = layernorm_stage("layernorm", x0, L["ffn1_ln_w"], L["ffn1_ln_b"])
        hid = project_stage("FFN up (bias + Swish)", xn, L["ffn1_w1"], L["ffn1_b1"], H.EPI_BIAS_SWISH_BF16)
        x1 = project_stage("FFN down (residual)", hid, L["ffn1_w2"], L["ffn1_b2"], H.EPI_RESIDUAL_F32, 0.5, x0)
        # attention
        xa = r(F.layer_norm(x1, (d,), L["attn_ln_w"], L["attn_ln_b"], og.ln_eps))
        qs = OC.LOG2E / math.sqrt(dh)
        wqkv = torch.cat([L["wq"] * qs, L["wk"], L["wv"]], 0)
        bqkv = torch.cat([L["bq"] * qs, L["bk"], L["bv"]], 0)
        qkv = project_stage("qkv projection", xa, wqkv, bqkv, H.EPI_BIAS_BF16)
        emb = torch.zeros((96, dh))
        emb[: og.n_dist] = L["dist_emb"]
        ctx_d = torch.zeros((rows, d), dtype=bf, device=dev)
        qkv_d = pad_rows(qkv, bf)
        H.check(lib.oasr_attention_relkey(qkv_d.data_ptr(), D(emb, bf), D(lt, torch.int32), ctx_d.data_ptr(), b, t_pad, h, dh,
                                          og.left_max, og.right_max, st()))
        # the oracle's context: what its attention branch hands to the output projection (q, k, v taken
        # from the qkv above through a spy on its projections)
        captured = {}
        real_proj = OC._proj
        parts = dict(zip(("q", "k", "v"), qkv.split(d, dim=-1)))

        def spy(xx, ww, bb=None):
            if ww is L["wo"]:
                captured["ctx"] = xx
                return torch.zeros(xx.shape[:-1] + (d,))
            for nm in ("k", "v"):
                if ww is L["w" + nm]:
                    return parts[nm]
            return parts["q"]                      # the q rows arrive as a product (wq * scale): a new tensor
        OC._proj = spy
        try:
            OC.attention(xa, lt, L, og)
        finally:
            OC._proj = real_proj
        ctx = captured["ctx"]
        report("attention -> bf16", ctx_d, ctx)
        x2 = project_stage("output projection (residual)", ctx, L["wo"], L["bo"], H.EPI_RESIDUAL_F32, 1.0, x1)
        # convolution module
        xc = r(F.layer_norm(x2, (d,), L["conv_ln_w"], L["conv_ln_b"], og.ln_eps))
        glu = project_stage("GLU projection", xc, L["pw1"], None, H.EPI_GLU_BF16)
        yt = F.conv1d(F.pad(glu.transpose(1, 2), (og.conv_kernel - 1, 0)), L["dw"][:, None, :], groups=d).transpose(1, 2)
        a = pad_rows(glu, bf)
        if d in (512, 1024, 2048):
            cv = r(OC._swish(F.layer_norm(yt, (d,), L["dw_ln_w"], L["dw_ln_b"], og.ln_eps)))
            c = torch.zeros((rows, d), dtype=bf, device=dev)
            H.check(lib.oasr_dwconv_ln_swish(a.data_ptr(), D(L["dw"].t().contiguous()), D(L["dw_ln_w"]), D(L["dw_ln_b"]),
                                             c.data_ptr(), b, t_pad, d, og.conv_kernel, og.ln_eps, st()))
            report("convolution + LayerNorm + Swish -> bf16", c, cv)
        else:
            c = torch.zeros((rows, d), dtype=torch.float32, device=dev)
            H.check(lib.oasr_dwconv_causal(a.data_ptr(), D(L["dw"].t().contiguous()), c.data_ptr(), b, t_pad, d, og.conv_kernel, st()))
            report("convolution -> f32", c, yt)
            cv = layernorm_stage("LayerNorm + Swish", yt, L["dw_ln_w"], L["dw_ln_b"], swish=1)
        x3 = project_stage("pointwise 2 (residual)", cv, L["pw2"], None, H.EPI_RESIDUAL_F32, 1.0, x2)
        # block output: LayerNorm in float32, fused with the next block's first LayerNorm
        want1 = F.layer_norm(x3, (d,), L["final_ln_w"], L["final_ln_b"], og.ln_eps)
        want2 = r(F.layer_norm(want1, (d,), L["ffn1_ln_w"], L["ffn1_ln_b"], og.ln_eps))
        xd = pad_rows(x3, torch.float32)
        if not fp8:
            yd = torch.zeros((rows, d), dtype=bf, device=dev)
            H.check(lib.oasr_layernorm2(xd.data_ptr(), D(L["final_ln_w"]), D(L["final_ln_b"]), D(L["ffn1_ln_w"]), D(L["ffn1_ln_b"]),
                                        yd.data_ptr(), rows, d, og.ln_eps, st()))
            report("double LayerNorm, second -> bf16", yd, want2)
        else:
            q = torch.zeros((rows, d), dtype=torch.uint8, device=dev)
            s = torch.zeros((rows,), dtype=torch.float32, device=dev)
            H.check(lib.oasr_layernorm2_fp8(xd.data_ptr(), D(L["final_ln_w"]), D(L["final_ln_b"]), D(L["ffn1_ln_w"]), D(L["ffn1_ln_b"]),
                                            q.data_ptr(), s.data_ptr(), rows, d, og.ln_eps, st()))
            report("double LayerNorm, second -> e4m3 rows", q.view(F8).float() * s[:, None], fq(want2))
        report("double LayerNorm, first -> f32", xd, want1)
        torch.cuda.synchronize()
    finally:
        OC._DEV = False
    return out
