"""Audio ingest: WAV decoding variants and the polyphase resampler (against scipy)."""

import struct
import wave

import numpy as np
import pytest

from omnilingual_asr import audio


def _wav_bytes(fmt_tag, bits, channels, rate, payload):
    block = channels * bits // 8
    fmt = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * block, block, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 4) + b"abcd" \
        + b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


@pytest.mark.parametrize("bits", [8, 16, 24, 32])
def test_pcm_widths(tmp_path, bits):
    vals = np.array([0.0, 0.5, -0.5, 0.25])
    if bits == 8:
        raw = (vals * 128 + 128).astype(np.uint8).tobytes()
    elif bits == 16:
        raw = (vals * 32768).astype("<i2").tobytes()
    elif bits == 24:
        ints = (vals * 8388608).astype(np.int32)
        raw = b"".join(int(v).to_bytes(4, "little", signed=True)[:3] for v in ints)
    else:
        raw = (vals * 2147483648).astype("<i4").tobytes()
    p = tmp_path / "x.wav"
    p.write_bytes(_wav_bytes(1, bits, 1, 8000, raw))
    x, rate = audio.read_wav(p)
    assert rate == 8000 and x.shape == (4, 1)
    np.testing.assert_allclose(x[:, 0], vals, atol=1e-6)
    assert audio.wav_duration(p) == pytest.approx(4 / 8000)


def test_float_stereo_downmix(tmp_path):
    st = np.stack([np.linspace(-1, 1, 100), np.zeros(100)], axis=1).astype("<f4")
    p = tmp_path / "f.wav"
    p.write_bytes(_wav_bytes(3, 32, 2, 16000, st.tobytes()))
    y = audio.load_audio_16k(p)
    np.testing.assert_allclose(y, st.mean(axis=1), atol=1e-7)


def test_compressed_formats_raise(tmp_path):
    p = tmp_path / "a.mp3"
    p.write_bytes(b"ID3....")
    with pytest.raises(audio.UnsupportedAudioError):
        audio.load_audio_16k(p)


@pytest.mark.parametrize("rate", [22050, 44100, 8000, 48000])
def test_resampler_matches_scipy(rate):
    from fractions import Fraction
    from scipy.signal import resample_poly
    rng = np.random.default_rng(rate)
    x = rng.standard_normal(5000)
    fr = Fraction(16000, rate)
    want = resample_poly(x, fr.numerator, fr.denominator)
    got = audio.resample_poly(x, rate, 16000)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=2e-6)


def test_gettysburg_fixture(golden_dir):
    x = audio.load_audio_16k(golden_dir / "gettysburg.wav")
    assert x.shape == (281233,) and x.dtype == np.float32 and np.abs(x).max() <= 1.0
