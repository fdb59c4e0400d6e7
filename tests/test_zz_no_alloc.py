"""GPU test of the allocation-free CTC entry point (`oasr_ctc_greedy_static`) through the C ABI: its
integer outputs equal `oasr_ctc_greedy`'s bit for bit (and the oracle's) on two calls in a row, the second
with a smaller input, so stale contents of the library-owned array from the first call cannot leak
into the second; sizes beyond the built-in array are rejected."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _decode(H, entry, lg_d, ld, vocab, len_d, batch, t_pad, ws=None):
    tok = torch.full((batch, t_pad), 7, dtype=torch.int32, device="cuda")
    frm = torch.full((batch, t_pad), 7, dtype=torch.int32, device="cuda")
    cnt = torch.full((batch,), -9, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    args = [lg_d.data_ptr(), ld, vocab, len_d.data_ptr(), batch, t_pad, 0, tok.data_ptr(), frm.data_ptr(), cnt.data_ptr()]
    if ws is not None:
        args.append(ws.data_ptr())
    H.check(getattr(H.lib(), entry)(*args, st), entry)
    torch.cuda.synchronize()
    return tok.cpu().numpy(), frm.cpu().numpy(), cnt.cpu().numpy()


def _logits(batch, t_pad, vocab, ld, seed):
    g = torch.Generator().manual_seed(seed)
    lg = torch.randint(0, 4, (batch, t_pad, ld), generator=g).float()          # coarse: ties, runs, blanks
    lg[:, :, 0] += (torch.rand((batch, t_pad), generator=g) < 0.4).float() * 2.0
    run = torch.randint(0, vocab, (batch, t_pad // 3 + 1), generator=g).repeat_interleave(3, dim=1)[:, :t_pad]
    lg.scatter_add_(2, run[:, :, None], torch.full((batch, t_pad, 1), 1.5))
    lg[:, :, vocab:] = 99.0
    return lg


def test_static_entry_point_equals_workspace_entry_point_twice():
    from omnilingual_asr import _hip as H
    from oracle import ctc

    assert H.lib().oasr_ctc_static_max_frames() == 1 << 20
    # first call: 32 x 1504 frames (the benchmark batch); second call: smaller in both dimensions
    for batch, t_pad, vocab, ld, lens in ((32, 1504, 32, 32, [1499] * 30 + [700, 0]),
                                          (3, 416, 29, 32, [416, 33, 250])):
        lg = _logits(batch, t_pad, vocab, ld, seed=batch)
        lg_d = lg.reshape(-1, ld).contiguous().cuda()
        len_d = torch.tensor(lens, dtype=torch.int32, device="cuda")
        ws = torch.full((batch * t_pad,), -1, dtype=torch.int32, device="cuda")
        want = _decode(H, "oasr_ctc_greedy", lg_d, ld, vocab, len_d, batch, t_pad, ws)
        got = _decode(H, "oasr_ctc_greedy_static", lg_d, ld, vocab, len_d, batch, t_pad)
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b)
        ref = ctc.greedy(lg[:, :, :vocab].numpy(), lens, blank=0)
        for b in range(batch):
            n = len(ref[b][0])
            assert got[2][b] == n
            np.testing.assert_array_equal(got[0][b, :n], ref[b][0])
            np.testing.assert_array_equal(got[1][b, :n], ref[b][1])
    # a further call with every buffer already in place leaves the device's free memory as it was
    tok = torch.zeros((batch, t_pad), dtype=torch.int32, device="cuda")
    frm = torch.zeros((batch, t_pad), dtype=torch.int32, device="cuda")
    cnt = torch.zeros((batch,), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    H.check(H.lib().oasr_ctc_greedy_static(lg_d.data_ptr(), ld, vocab, len_d.data_ptr(), batch, t_pad, 0,
                                           tok.data_ptr(), frm.data_ptr(), cnt.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before


def test_static_entry_point_rejects_sizes_beyond_its_array():
    from omnilingual_asr import _hip as H

    z = torch.zeros((64, 32), device="cuda")
    i = torch.zeros((64,), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = H.lib().oasr_ctc_greedy_static(z.data_ptr(), 32, 32, i.data_ptr(), 1024, 1025, 0, i.data_ptr(), i.data_ptr(),
                                        i.data_ptr(), st)
    assert rc == -2 and b"exceeds the built-in scratch" in H.lib().oasr_last_error()
    assert H.lib().oasr_ctc_greedy_static(z.data_ptr(), 30, 32, i.data_ptr(), 1, 64, 0, i.data_ptr(), i.data_ptr(),
                                          i.data_ptr(), st) == -1
