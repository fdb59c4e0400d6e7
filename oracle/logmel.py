"""Oracle for the log-mel feature extractor.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED against the reference: ref:src/omnilingual_asr contains no feature
extractor (recognition is a remote call, ref gemini_pipeline.py:512-530).  The geometry
below is declared by this build.  It mirrors Kaldi-style fbank as restated by the
third-party stand-in tf:models/seamless_m4t/feature_extraction_seamless_m4t.py:73-85,
:123-137, :257-260, :282-289 (tf: = site-packages/transformers), against which
`tests/golden/standin_logmel.npz` cross-checks it.

Declared geometry (all arithmetic in float64 here):
  x        = pcm_f32 * 2**15
  frames   : n = 1 + (N - 400) // 160 for N >= 400 else 0; frame t = x[160 t : 160 t + 400]
  per frame: subtract the frame mean; pre-emphasis y[i] = f[i] - 0.97 f[i-1], y[0] = 0.03 f[0];
             multiply by the Povey window hanning(400)**0.85; zero-pad to 512; rfft;
             power |X|^2 over bins 0..256
  mel      : 80 triangles, Kaldi mel scale 1127 ln(1 + f/700), edges linear in mel between
             mel(20 Hz) and mel(8000 Hz), triangles evaluated in mel space at the FFT bin
             centres, no area normalisation
  log      : ln(max(mel, 1.192092955078125e-07))
  norm     : per utterance, per mel bin over all n frames: (v - mean) / sqrt(var_ddof1 + 1e-7)
  stack    : frames (2u, 2u+1) concatenated -> [n // 2, 160]; an odd trailing frame is
             dropped (the stand-in pads it with zeros and masks the stacked frame out, so
             the valid part is identical)
"""

from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
FRAME_LEN = 400
FRAME_HOP = 160
NFFT = 512
NBINS = NFFT // 2 + 1
NMEL = 80
PREEMPH = 0.97
MEL_FLOOR = 1.192092955078125e-07
MEL_LOW_HZ = 20.0
MEL_HIGH_HZ = 8000.0
NORM_EPS = 1e-7
STACK = 2


def num_frames(n_samples: int) -> int:
    return 0 if n_samples < FRAME_LEN else 1 + (n_samples - FRAME_LEN) // FRAME_HOP


def num_stacked(n_samples: int) -> int:
    return num_frames(n_samples) // STACK


def povey_window() -> np.ndarray:
    n = np.arange(FRAME_LEN, dtype=np.float64)
    hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / (FRAME_LEN - 1))
    return hann ** 0.85


def _mel(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_matrix() -> np.ndarray:
    """[257, 80] float64 triangle weights."""
    edges = np.linspace(_mel(MEL_LOW_HZ), _mel(MEL_HIGH_HZ), NMEL + 2)
    bin_mel = _mel(np.arange(NBINS) * (SAMPLE_RATE / NFFT))
    left, centre, right = edges[:-2], edges[1:-1], edges[2:]
    up = (bin_mel[:, None] - left[None, :]) / (centre - left)[None, :]
    down = (right[None, :] - bin_mel[:, None]) / (right - centre)[None, :]
    return np.maximum(0.0, np.minimum(up, down))


def frame_signal(pcm: np.ndarray) -> np.ndarray:
    """[n_frames, 400] float64 frames of pcm * 2**15 (no processing)."""
    x = np.asarray(pcm, dtype=np.float64) * 32768.0
    n = num_frames(x.shape[0])
    idx = np.arange(n)[:, None] * FRAME_HOP + np.arange(FRAME_LEN)[None, :]
    return x[idx] if n else np.zeros((0, FRAME_LEN))


def power_spectrum(pcm: np.ndarray) -> np.ndarray:
    """[n_frames, 257] float64."""
    f = frame_signal(pcm)
    f = f - f.mean(axis=1, keepdims=True)
    y = np.empty_like(f)
    y[:, 1:] = f[:, 1:] - PREEMPH * f[:, :-1]
    y[:, 0] = f[:, 0] * (1.0 - PREEMPH)
    y *= povey_window()[None, :]
    spec = np.fft.rfft(y, n=NFFT, axis=1)
    return spec.real ** 2 + spec.imag ** 2


def logmel(pcm: np.ndarray) -> np.ndarray:
    """[n_frames, 80] float64 un-normalised log-mel energies."""
    p = power_spectrum(pcm)
    return np.log(np.maximum(MEL_FLOOR, p @ mel_matrix()))


def normalise(lm: np.ndarray) -> np.ndarray:
    mean = lm.mean(axis=0, keepdims=True)
    var = lm.var(axis=0, ddof=1, keepdims=True)
    return (lm - mean) / np.sqrt(var + NORM_EPS)


def features(pcm: np.ndarray) -> np.ndarray:
    """[n_frames // 2, 160] float64 normalised, stacked features of one utterance."""
    lm = normalise(logmel(pcm))
    t2 = lm.shape[0] // STACK
    return lm[: t2 * STACK].reshape(t2, STACK * NMEL)


def features_batch(pcm: np.ndarray, n_valid=None, dtype=np.float32) -> tuple[np.ndarray, np.ndarray]:
    """Batch form: pcm [B, N]; n_valid[b] = number of valid samples (default N).
    Returns (feats [B, Tmax, 160] zero padded, lengths [B] int32)."""
    pcm = np.asarray(pcm)
    b = pcm.shape[0]
    if n_valid is None:
        n_valid = [pcm.shape[1]] * b
    lens = np.array([num_stacked(int(n)) for n in n_valid], dtype=np.int32)
    tmax = int(lens.max()) if b else 0
    out = np.zeros((b, tmax, STACK * NMEL), dtype=dtype)
    for i in range(b):
        if lens[i]:
            out[i, : lens[i]] = features(pcm[i, : int(n_valid[i])]).astype(dtype)
    return out, lens
