"""The C-ABI library loads without a GPU and exports every symbol include/oasr.h declares; host-only
entry points behave as documented (no compute calls here)."""

import ctypes as C
import re
from pathlib import Path

import numpy as np

from omnilingual_asr import _hip

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "oasr.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oasr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _hip.lib()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_hip.EXPORTED_SYMBOLS) == names       # the binding covers the whole header


def test_version_and_error_text():
    lib = _hip.lib()
    assert lib.oasr_version() == 100
    n = C.c_int64(0)
    rc = lib.oasr_plan_chunks(100, 0, 30.0, 30.0, None, 0, C.byref(n))
    assert rc == -1 and b"sample_rate" in lib.oasr_last_error()


def test_plan_chunks_capacity_protocol():
    lib = _hip.lib()
    n = C.c_int64(0)
    buf = (C.c_int64 * 4)()
    rc = lib.oasr_plan_chunks(16000 * 100, 16000, 30.0, 30.0, buf, 2, C.byref(n))
    assert rc == -2 and n.value == 4
    assert _hip.plan_chunks(16000 * 100, 16000, 30.0, 30.0).tolist() == [
        [0, 480000], [480000, 960000], [960000, 1440000], [1440000, 1600000]]
    assert _hip.plan_chunks(0, 16000, 30.0, 30.0).tolist() == [[0, 0]]
    assert _hip.plan_chunks(-5, 16000, 30.0, 30.0).tolist() == [[0, 0]]
    assert len(_hip.plan_chunks(547_200_000, 16000, 30.0, 30.0)) == 1140      # 9.5 h
    assert len(_hip.plan_chunks(547_200_000, 16000, 300.0, 360.0)) == 114


def test_shard_range_and_frames():
    from oracle import chunker, logmel
    for n in (0, 5, 114, 1140):
        for world in (1, 2, 8):
            for r in range(world):
                assert _hip.shard_range(n, world, r) == chunker.partition_contiguous(n, world, r)
    for ns in (0, 399, 400, 559, 560, 16000, 480000, 281233):
        assert _hip.num_frames(ns) == logmel.num_frames(ns)
        assert _hip.num_stacked(ns) == logmel.num_stacked(ns)
    assert _hip.num_frames(480000) == 2998 and _hip.num_stacked(480000) == 1499


def test_feature_tables_match_oracle():
    from oracle import logmel
    win, mel = _hip.logmel_tables()
    np.testing.assert_allclose(win, logmel.povey_window(), atol=6e-8)
    np.testing.assert_allclose(mel, logmel.mel_matrix(), atol=6e-8)
    blob = _hip.logmel_table_blob()
    assert blob.dtype == np.uint32 and blob.size == int(_hip.lib().oasr_logmel_table_words())
    np.testing.assert_array_equal(blob[:400].view(np.float32), win)


def test_bucket_width_covers_every_sample_count():
    """`HipEngine.bucket_samples(n)` is a multiple of 4, at least n, and pads to the same number of
    encoder frames as n or to the next bucket — for every n (exhaustive over 0.4 s .. 4 s, sampled
    beyond, including a 300 s chunk)."""
    from omnilingual_asr import _hip
    from omnilingual_asr.engine_hip import HipEngine, pad_frames

    t_of = lambda n: pad_frames(_hip.num_frames(n) // 2)
    ns = list(range(1, 64000)) + [480000, 4800000, 4800001, 4802556, 4802557, 4802559, 4802560]
    for n in ns:
        w = HipEngine.bucket_samples(n)
        assert w % 4 == 0 and w >= n, n
        assert t_of(w) in (t_of(max(n, 400)), t_of(max(n, 400)) + 32), n
        assert HipEngine.bucket_samples(w) == w, n          # a bucket is its own bucket
