"""CPU oracle for the transcribe() hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under this package is part of the shipped product path.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and only
as the checker.  The product package (`omnilingual-asr_amd/omnilingual_asr`) never imports
it and raises when its HIP library is missing.

Pinning status, per stage:

* chunk plan, timestamp parsing, response parsing, chunk offsetting, merge order, retry and
  progress protocol (`chunker.py`, `response.py`): restated from
  ref:src/omnilingual_asr/models/inference/gemini_pipeline.py and PINNED against golden
  vectors produced by importing that module in the build container
  (`oracle/gen_golden.py` -> `tests/golden/reference_host.json`).
* log-mel features and conformer encoder forward (`logmel.py`, `conformer.py`): the
  reference contains no implementation, weights, tests or fixtures for these stages, so
  they are PARITY UNPINNED against the reference.  The geometry is declared by this build
  and cross-checked against an independent third-party implementation that happens to be
  installed in the build container (transformers' SeamlessM4T feature extractor and
  Wav2Vec2-BERT encoder with seeded random weights); those cross-check vectors are
  committed under `tests/golden/standin_*.npz`.
"""
