"""CPU engine built from the oracle stages.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Has the same `encode_batch` contract as the product's HipEngine so that tests can run the
host-side pipeline logic on a machine without a GPU (BASELINE config 1: `gettysburg.wav` through
`transcribe()` as a plumbing check) and can time the CPU baseline.  The product never imports it:
it is handed to the pipeline by the test through the `engine=` argument.
"""

from __future__ import annotations

import numpy as np
import torch

from . import conformer, ctc, logmel


class OracleEngine:
    def __init__(self, weights: dict, geometry: conformer.Geometry = conformer.G300, n_layers: int = -1,
                 dtype=torch.float32):
        self.weights = weights
        self.geometry = geometry
        self.dtype = dtype
        self.n_layers = n_layers
        self.calls = 0

    def encode_batch(self, pcm, n_valid):
        self.calls += 1
        pcm = np.asarray(pcm, dtype=np.float32)
        feats, lengths = logmel.features_batch(pcm, n_valid)
        w = self.weights
        if self.n_layers >= 0:
            w = dict(w)
            w["layers"] = w["layers"][: self.n_layers]
        hidden = conformer.forward(torch.from_numpy(feats), torch.from_numpy(lengths), w, self.geometry,
                                   dtype=self.dtype)
        return hidden.to(torch.float32), lengths

    def transcribe_batch(self, pcm, n_valid):
        hidden, lengths = self.encode_batch(pcm, n_valid)
        lg = ctc.logits(hidden, self.weights["ctc_w"], self.weights["ctc_b"]).numpy()
        return hidden, lengths, ctc.greedy(lg, lengths, blank=0)
