"""The oracle's host-side restatements against golden vectors captured from the reference itself
(tests/golden/reference_host.json, made by oracle/gen_golden.py part A)."""

import json
import math

import numpy as np
import pytest

from oracle import chunker, response


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.loads((golden_dir / "reference_host.json").read_text())


def test_constants(gold):
    c = gold["constants"]
    assert chunker.REF_CHUNK_SECONDS == c["CHUNK_DURATION_SECONDS"]
    assert chunker.REF_MIN_CHUNKING_SECONDS == c["MIN_DURATION_FOR_CHUNKING"]
    assert chunker.REF_MAX_PARALLEL == c["MAX_PARALLEL_CHUNKS"]


def test_parse_timestamp_bit_exact(gold):
    for text, want in gold["parse_timestamp"]:
        assert response.parse_timestamp(text) == want, text


def test_mime(gold):
    from pathlib import Path
    for name, want in gold["get_mime_type"]:
        assert response.get_mime_type(Path(name).suffix) == want


def test_parse_response(gold):
    for name, text in gold["parse_response_inputs"].items():
        assert response.parse_response(text) == gold["parse_response"][name], name


def test_chunk_plans_bit_exact(gold):
    for plan in gold["chunk_plans"]:
        starts = chunker.plan_starts_seconds(plan["duration"], plan["chunk"])
        assert starts == plan["starts"], (plan["duration"], plan["chunk"])
        assert chunker.use_chunking(plan["duration"]) == plan["use_chunking"]
        if plan["duration"] > 0:
            assert chunker.chunk_names(len(starts), ".mp3") == plan["names"]
            # the reference asks ffmpeg for -ss <start> -t <chunk> per chunk
            assert [float(c["ss"]) for c in plan["ffmpeg"]] == starts
            assert all(float(c["t"]) == plan["chunk"] for c in plan["ffmpeg"])


def test_sample_plan_equals_seconds_plan(gold):
    sr = 16000
    for plan in gold["chunk_plans"]:
        total = int(round(plan["duration"] * sr))
        if total <= 0:
            continue
        got = chunker.plan_chunks_samples(total, sr, plan["chunk"], 0.0)   # gate open: always slice
        want = chunker.plan_starts_seconds(total / sr, plan["chunk"])
        assert [int(s) for s in got[:, 0]] == [int(round(s * sr)) for s in want]
        assert int(got[-1, 1]) == total and (got[1:, 0] == got[:-1, 1]).all()
    one = chunker.plan_chunks_samples(360 * sr, sr, 300.0, 360.0)
    assert one.tolist() == [[0, 360 * sr]]                # gate is strict: exactly 360 s is not chunked
    two = chunker.plan_chunks_samples(360 * sr + 1, sr, 300.0, 360.0)
    assert two.tolist() == [[0, 300 * sr], [300 * sr, 360 * sr + 1]]


def test_offset_and_merge_match_reference(gold):
    cm = gold["chunked_merge"]
    results = []
    for off_s, text in sorted(cm["per_chunk_inputs"].items(), key=lambda kv: float(kv[0])):
        if text == "RAISE":
            continue                                       # the reference drops a failed chunk
        results.append(response.offset_result(response.parse_response(text), float(off_s)))
    # completion order must not matter: feed the chunks in reverse
    merged = response.merge_results(list(reversed(results)))
    assert merged == cm["merged"]
    assert cm["progress"] == [list(p) for p in response.PROGRESS_SEQUENCE]
    assert "Chunk transcription failed: boom" in cm["stdout"]


def test_retry_schedule(gold):
    assert response.retry_sleeps(3) == gold["retry"]["all_fail"]["sleeps"] == [1, 2]
    assert gold["retry"]["all_fail"]["attempts"] == 3
    assert gold["retry"]["fail_once"]["sleeps"] == [1]


def test_partition_contiguous():
    for n in (0, 1, 7, 8, 9, 114, 1140):
        spans = [chunker.partition_contiguous(n, 8, r) for r in range(8)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert [chunker.partition_contiguous(1140, 8, r)[1] - chunker.partition_contiguous(1140, 8, r)[0]
            for r in range(8)] == [143, 143, 143, 143, 142, 142, 142, 142]
