"""Audio ingest: RIFF/WAVE decode, down-mix and polyphase resampling to 16 kHz mono float32.

The reference hands the encoded file to its remote model and uses ffprobe/ffmpeg only for duration
and slicing (ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:222-310).  Here the host
decodes PCM itself.  Compressed formats the reference lists (:106-118: mp3, aac, ogg, flac, m4a)
need a decoder this package does not contain and raise `UnsupportedAudioError`.
"""

from __future__ import annotations

import struct
from fractions import Fraction
from pathlib import Path

import numpy as np

TARGET_RATE = 16000


class UnsupportedAudioError(RuntimeError):
    pass


def _riff_chunks(data: bytes):
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise UnsupportedAudioError("not a RIFF/WAVE file")
    pos = 12
    while pos + 8 <= len(data):
        cid = data[pos:pos + 4]
        size = struct.unpack_from("<I", data, pos + 4)[0]
        yield cid, data[pos + 8: pos + 8 + size]
        pos += 8 + size + (size & 1)


def read_wav(path) -> tuple[np.ndarray, int]:
    """Decode a WAV file -> (float32 [frames, channels] in [-1, 1), sample_rate).
    Supports PCM 8/16/24/32-bit and IEEE float 32/64 (also via WAVE_FORMAT_EXTENSIBLE)."""
    path = Path(path)
    if path.suffix.lower() not in (".wav", ".wave", ""):
        raise UnsupportedAudioError(
            f"cannot decode '{path.suffix}' audio: only RIFF/WAVE input is supported")
    data = path.read_bytes()
    fmt = payload = None
    for cid, body in _riff_chunks(data):
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            payload = body
    if fmt is None or payload is None or len(fmt) < 16:
        raise UnsupportedAudioError("WAV file lacks a fmt or data chunk")
    tag, channels, rate, _, block, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if tag == 0xFFFE and len(fmt) >= 26:
        tag = struct.unpack_from("<H", fmt, 24)[0]
    if channels < 1 or block < 1:
        raise UnsupportedAudioError("WAV header declares no channels")
    n = len(payload) // block
    raw = payload[: n * block]
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = v.astype(np.float32) / 8388608.0
        elif bits == 32:
            x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise UnsupportedAudioError(f"unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(raw, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise UnsupportedAudioError(f"unsupported WAV format tag {tag}")
    return x.reshape(n, channels), int(rate)


def wav_duration(path) -> float:
    """Duration in seconds from the header alone; raises on anything unreadable."""
    with open(path, "rb") as f:
        head = f.read(1 << 16)
    rate = block = None
    pos = 12
    if head[:4] != b"RIFF" or head[8:12] != b"WAVE":
        raise UnsupportedAudioError("not a RIFF/WAVE file")
    while pos + 8 <= len(head):
        cid = head[pos:pos + 4]
        size = struct.unpack_from("<I", head, pos + 4)[0]
        if cid == b"fmt ":
            _, _, rate, _, block, _ = struct.unpack_from("<HHIIHH", head, pos + 8)
        elif cid == b"data":
            if not rate or not block:
                break
            avail = Path(path).stat().st_size - (pos + 8)
            return min(size, max(avail, 0)) // block / float(rate)
        pos += 8 + size + (size & 1)
    raise UnsupportedAudioError("WAV header incomplete")


def resample_filter(up: int, down: int, half_width: int = 10, beta: float = 5.0) -> np.ndarray:
    """Kaiser-windowed sinc low-pass at the up-sampled rate, cut-off at the lower Nyquist, unity
    DC gain times `up`; length 2 * half_width * max(up, down) + 1 (zero phase)."""
    m = max(up, down)
    half = half_width * m
    n = np.arange(-half, half + 1, dtype=np.float64)
    h = np.sinc(n / m) / m * np.kaiser(2 * half + 1, beta)
    return h / h.sum() * up


def polyphase_branches(rate_in: int, rate_out: int = TARGET_RATE):
    """(up, down, half, hp) with hp[p, j] = h[p + j * up], float64 [up, taps]: the table both the host
    resampler below and the device kernel (`oasr_resample_poly`) read."""
    fr = Fraction(rate_out, rate_in)
    up, down = fr.numerator, fr.denominator
    h = resample_filter(up, down)
    half = (len(h) - 1) // 2
    taps = -(-len(h) // up)
    hp = np.zeros((up, taps), dtype=np.float64)
    for p in range(up):
        seg = h[p::up]
        hp[p, : len(seg)] = seg
    return up, down, half, hp


def resample_poly(x: np.ndarray, rate_in: int, rate_out: int = TARGET_RATE) -> np.ndarray:
    """Rational polyphase resampler: y[n] = sum_k x[k] h[n*down - k*up + half], output length
    ceil(len(x) * up / down).  float64 accumulate, float32 result."""
    if rate_in == rate_out:
        return np.asarray(x, dtype=np.float32)
    fr = Fraction(rate_out, rate_in)
    up, down = fr.numerator, fr.denominator
    h = resample_filter(up, down)
    half = (len(h) - 1) // 2
    x = np.asarray(x, dtype=np.float64)
    n_in = x.shape[0]
    n_out = -(-n_in * up // down)
    taps = -(-len(h) // up)                      # taps per polyphase branch
    hp = np.zeros((up, taps), dtype=np.float64)  # hp[p, j] = h[p + j*up]
    for p in range(up):
        seg = h[p::up]
        hp[p, : len(seg)] = seg
    n = np.arange(n_out, dtype=np.int64)
    pos = n * down + half                        # index into h of the tap that meets x[0]
    phase = pos % up
    k_hi = pos // up                             # newest input sample under the filter
    y = np.zeros(n_out, dtype=np.float64)
    xp = np.concatenate([np.zeros(taps, dtype=np.float64), x, np.zeros(taps, dtype=np.float64)])
    for j in range(taps):                        # tap j of branch `phase` meets x[k_hi - j]
        idx = k_hi - j + taps
        ok = (idx >= 0) & (idx < xp.shape[0])
        y += hp[phase, j] * np.where(ok, xp[np.clip(idx, 0, xp.shape[0] - 1)], 0.0)
    return y.astype(np.float32)


def load_audio_16k(path, resampler=None) -> np.ndarray:
    """File -> mono float32 at 16 kHz (channels averaged).  `resampler(mono, rate) -> float32 array`
    replaces the host polyphase filter when given (the device engine passes its own)."""
    x, rate = read_wav(path)
    mono = x.mean(axis=1, dtype=np.float64) if x.shape[1] > 1 else x[:, 0]
    if resampler is not None and rate != TARGET_RATE:
        return np.asarray(resampler(np.asarray(mono, dtype=np.float32), rate), dtype=np.float32)
    return resample_poly(mono, rate, TARGET_RATE)
