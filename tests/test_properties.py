"""Randomised properties of the integer host path (hypothesis): the C-ABI chunk planner against the
oracle and against the reference's loop in seconds, shard ranges as a partition, the record packing
round trip."""

import math

import numpy as np
from hypothesis import given, settings, strategies as st

from omnilingual_asr import _hip, distributed as odist
from omnilingual_asr.models.inference.asr_pipeline import ChunkRecord
from oracle import chunker

RATES = st.sampled_from([8000, 16000, 22050, 44100, 48000])
CHUNKS = st.sampled_from([0.5, 1.0, 7.25, 30.0, 300.0])


@settings(max_examples=300, deadline=None)
@given(total=st.integers(-5, 16000 * 4000), rate=RATES, chunk=CHUNKS, gate=st.sampled_from([0.0, 1.0, 30.0, 360.0]))
def test_plan_chunks_matches_oracle_and_tiles_the_signal(total, rate, chunk, gate):
    got = _hip.plan_chunks(total, rate, chunk, gate)
    want = chunker.plan_chunks_samples(total, rate, chunk, gate)
    assert got.dtype == np.int64 and np.array_equal(got, want)
    if total > 0:
        assert got[0, 0] == 0 and got[-1, 1] == total
        assert np.array_equal(got[1:, 0], got[:-1, 1])                      # no gap, no overlap
        step = int(round(chunk * rate))
        if len(got) > 1:
            assert (np.diff(got, axis=1)[:-1] == step).all() and 0 < got[-1, 1] - got[-1, 0] <= step
            assert len(got) == math.ceil(total / step)                      # ref: n = ceil(total / chunk)


@settings(max_examples=200, deadline=None)
@given(seconds=st.integers(1, 40000), chunk=st.sampled_from([30.0, 300.0]))
def test_plan_in_samples_equals_the_reference_loop_in_seconds(seconds, chunk):
    """Whole-second durations: the sample plan's starts are the reference's float starts times the rate."""
    rate = 16000
    got = _hip.plan_chunks(seconds * rate, rate, chunk, 0.0)
    starts = chunker.plan_starts_seconds(float(seconds), chunk)
    assert [int(s * rate) for s in starts] == got[:, 0].tolist()


@settings(max_examples=300, deadline=None)
@given(n=st.integers(0, 5000), world=st.integers(1, 64))
def test_shard_ranges_partition_the_chunks(n, world):
    edges = [_hip.shard_range(n, world, r) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    sizes = [hi - lo for lo, hi in edges]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert max(sizes) == odist.max_shard(n, world) or n == 0


@settings(max_examples=100, deadline=None)
@given(data=st.data())
def test_record_packing_round_trip(data):
    n = data.draw(st.integers(0, 6))
    recs = []
    for i in range(n):
        k = data.draw(st.integers(0, 12))
        toks = data.draw(st.lists(st.integers(0, 31), min_size=k, max_size=k))
        frames = sorted(data.draw(st.lists(st.integers(0, 1498), min_size=k, max_size=k)))
        recs.append(ChunkRecord(data.draw(st.integers(0, 10 ** 6)), data.draw(st.integers(0, 2 ** 40)),
                                data.draw(st.integers(0, 2 ** 40)), data.draw(st.integers(0, 1499)),
                                np.asarray(toks, dtype=np.int32), np.asarray(frames, dtype=np.int32)))
    back = odist.unpack_records(odist.pack_records(recs, n + 2, 12))
    want = sorted(recs, key=lambda r: r.chunk_idx)
    assert len(back) == n
    for a, b in zip(back, sorted(want, key=lambda r: r.chunk_idx)):
        if [r.chunk_idx for r in recs].count(a.chunk_idx) > 1:
            continue                                   # duplicate indices: order among equals is the input order
        assert (a.chunk_idx, a.start_sample, a.end_sample, a.n_frames) == \
               (b.chunk_idx, b.start_sample, b.end_sample, b.n_frames)
        assert a.tokens.tolist() == b.tokens.tolist() and a.frames.tolist() == b.frames.tolist()
