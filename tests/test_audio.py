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


@pytest.mark.parametrize("tag,law", [(7, "ulaw"), (6, "alaw")])
def test_g711_wav_matches_the_standard_library_expansion(tmp_path, tag, law):
    """Telephone-speech WAVs (8-bit mu-law / A-law): every code against Python's own G.711 expansion (audioop)."""
    import audioop
    import warnings
    codes = bytes(range(256)) * 2
    p = tmp_path / "g711.wav"
    p.write_bytes(_wav_bytes(tag, 8, 2, 8000, codes))
    x, rate = audio.read_wav(p)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lin = audioop.ulaw2lin(codes, 2) if law == "ulaw" else audioop.alaw2lin(codes, 2)
    want = np.frombuffer(lin, dtype="<i2").astype(np.float32) / 32768.0
    assert rate == 8000 and x.shape == (256, 2) and np.array_equal(x.reshape(-1), want)
    y = audio.load_audio_16k(p)
    assert y.shape == (512,) and np.isfinite(y).all()


@pytest.mark.parametrize("channels", [1, 2])
def test_ima_adpcm_wav_matches_the_standard_library_decoder(tmp_path, channels):
    """4-bit IMA / DVI ADPCM WAVs: blocks built from Python's own encoder (audioop.lin2adpcm, nibbles re-packed low
    first as WAV wants them), decoded by the native library, compared with audioop.adpcm2lin sample for sample."""
    import audioop
    import warnings
    rng = np.random.default_rng(channels)
    block, rate = 256 * channels, 8000
    per_block = 1 + (block // channels - 4) * 2
    n_blocks = 9
    t = np.arange(n_blocks * per_block)
    sig = [(np.sin(t * (0.02 + 0.01 * c)) * np.linspace(50, 30000, t.size) + rng.integers(-40, 40, t.size)).astype("<i2")
           for c in range(channels)]
    payload, want = bytearray(), np.zeros((t.size, channels), dtype=np.int16)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        index = [int(rng.integers(0, 89)) for _ in range(channels)]
        for b in range(n_blocks):
            seg = [s[b * per_block:(b + 1) * per_block] for s in sig]
            coded = []
            for c in range(channels):
                state = (int(seg[c][0]), index[c])
                payload += struct.pack("<hBB", state[0], state[1], 0)
                adpcm, new_state = audioop.lin2adpcm(seg[c][1:].tobytes(), 2, state)
                lin, _ = audioop.adpcm2lin(adpcm, 2, state)
                want[b * per_block, c] = seg[c][0]
                want[b * per_block + 1:(b + 1) * per_block, c] = np.frombuffer(lin, dtype="<i2")
                coded.append(bytes(((v >> 4) | (v << 4)) & 0xFF for v in adpcm))     # first code into the low nibble
                index[c] = new_state[1]
            for g in range(len(coded[0]) // 4):
                for c in range(channels):
                    payload += coded[c][4 * g: 4 * g + 4]
    fmt = struct.pack("<HHIIHHHH", 0x11, channels, rate, rate * block // per_block, block, 4, 2, per_block)
    stated = t.size - 5                                                  # the writer's count: the last block is padded
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<II", 4, stated) \
        + b"data" + struct.pack("<I", len(payload)) + bytes(payload)
    p = tmp_path / "ima.wav"
    p.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    x, r = audio.read_wav(p)
    assert r == rate and x.shape == (stated, channels)
    assert np.array_equal(x, want[:stated].astype(np.float32) / 32768.0)
    assert audio.audio_duration(p) == stated / rate and audio.load_audio_16k(p).shape == (stated * 2,)


@pytest.mark.parametrize("magic,trailer", [(b"RF64", b""), (b"BW64", b"LIST" + struct.pack("<I", 4) + b"abcd"),
                                           (b"RIFF", b"")])
def test_wav_beyond_four_gigabytes_forms(tmp_path, magic, trailer):
    """A WAV of many hours exceeds RIFF's 32-bit sizes: RF64 / BW64 keep the real sizes in a 'ds64' chunk and put
    0xFFFFFFFF in the 32-bit fields (chunks may follow the data); a RIFF written as a stream carries 0xFFFFFFFF
    and runs to the end of the file.  All readers (decode, memory map, duration probe) take them."""
    rng = np.random.default_rng(5)
    pcm = rng.integers(-20000, 20000, size=(3000, 2), dtype=np.int16)
    payload = pcm.tobytes()
    fmt = struct.pack("<HHIIHH", 1, 2, 16000, 64000, 4, 16)
    ds64 = b"" if magic == b"RIFF" else b"ds64" + struct.pack("<IQQQI", 28, 0, len(payload), 3000, 0)
    p = tmp_path / "big.wav"
    p.write_bytes(magic + struct.pack("<I", 0xFFFFFFFF) + b"WAVE" + ds64 + b"fmt " + struct.pack("<I", 16) + fmt +
                  b"data" + struct.pack("<I", 0xFFFFFFFF) + payload + trailer)
    x, rate = audio.read_audio(p)
    assert rate == 16000 and np.array_equal(x, pcm.astype(np.float32) / 32768.0)
    mapped = audio.map_wav_pcm16(p)
    assert mapped is not None and np.array_equal(mapped[0].reshape(-1, 2), pcm) and mapped[1:] == (2, 16000)
    assert audio.audio_duration(p) == 3000 / 16000.0
    assert np.array_equal(audio.load_audio_16k(p), pcm.astype(np.float32).mean(axis=1) / 32768.0)


def test_damaged_headers_are_refused_not_obeyed(tmp_path, golden_dir):
    """A header may state anything.  An absurd sampling rate would make the rate converter design a filter of up to
    billions of taps, a frame size that contradicts the channel count breaks the sample layout, a FLAC sample count
    beyond what the file can hold asks for terabytes: each is refused with UnsupportedAudioError, at once."""
    import time
    pcm = np.zeros(4000, dtype="<i2").tobytes()
    t0 = time.perf_counter()
    for rate in (1_124_089_472, 47_999_999, 3, 0, 999, 1_000_003):      # the last: prime, shares no factor with 16 000
        p = tmp_path / "r.wav"
        p.write_bytes(_wav_bytes(1, 16, 1, rate, pcm))
        with pytest.raises(audio.UnsupportedAudioError):
            audio.load_audio_16k(p)
    for rate in (8000, 11025, 44056, 47999, 96000, 192000, 352800, 768000):   # real rates, the odd ones included
        assert max(audio.conversion_ratio(rate)) <= audio.MAX_CONVERSION_TERM
    p = tmp_path / "b.wav"
    blob = bytearray(_wav_bytes(1, 16, 1, 16000, pcm))
    struct.pack_into("<H", blob, blob.index(b"fmt ") + 8 + 12, 60)          # 60 bytes per frame for one 16-bit channel
    p.write_bytes(bytes(blob))
    with pytest.raises(audio.UnsupportedAudioError):
        audio.read_wav(p)
    flac = bytearray((golden_dir / "reference_upload.flac").read_bytes())
    flac[21] |= 0x0F                                                        # top bits of the 36-bit sample count
    flac[22] = 0xFF
    p = tmp_path / "f.flac"
    p.write_bytes(bytes(flac))
    with pytest.raises(audio.UnsupportedAudioError):
        audio.read_flac(p)
    with pytest.raises(audio.UnsupportedAudioError):
        audio.read_flac_pcm(p)
    assert time.perf_counter() - t0 < 15.0          # (obeying the first header alone took 36 s, or the machine's memory)


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


def test_int16_fast_path_equals_the_float_path(tmp_path):
    """A mono 16-bit WAV at 16 kHz may come back as its int16 samples; scaled by 1/32768 they are the float
    path's samples bit for bit, and the pipeline's batch assembly does exactly that."""
    rng = np.random.default_rng(5)
    pcm = rng.integers(-32768, 32768, size=48000, dtype=np.int16)
    path = tmp_path / "mono16.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    ref = audio.load_audio_16k(path)
    fast = audio.load_audio_16k(path, keep_int16=True)
    assert ref.dtype == np.float32 and fast.dtype == np.int16 and fast.shape == ref.shape
    out = np.empty(ref.shape, dtype=np.float32)
    np.multiply(fast, np.float32(1.0 / 32768.0), out=out)
    assert np.array_equal(out, ref)
    # other layouts keep the float path: stereo, another rate, 8-bit
    for ch, rate, width in ((2, 16000, 2), (1, 8000, 2), (1, 16000, 1)):
        p2 = tmp_path / f"v{ch}_{rate}_{width}.wav"
        with wave.open(str(p2), "wb") as w:
            w.setnchannels(ch); w.setsampwidth(width); w.setframerate(rate)
            w.writeframes(pcm[:3200 * ch].tobytes() if width == 2 else (pcm[:3200] >> 8).astype(np.int8).view(np.uint8).tobytes())
        assert audio.load_audio_16k(p2, keep_int16=True).dtype == np.float32


def test_pipeline_records_do_not_depend_on_the_sample_container(tmp_path, tiny_engine):
    """The same file through the pipeline with int16 samples kept (default) and with float samples handed in."""
    from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline
    rng = np.random.default_rng(6)
    t = np.arange(16000 * 3)
    pcm = (2000 * rng.standard_normal(t.size) + 8000 * np.sin(t * 2 * np.pi * 300 / 16000)).astype(np.int16)
    path = tmp_path / "three_seconds.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    pipe = GeminiASRPipeline(model="GTINY", engine=tiny_engine, chunk_seconds=1.0, min_chunking_seconds=1.0)
    kept = pipe._load_audio(path)
    assert kept.dtype == np.int16
    a = pipe._encode_chunks(kept, np.array([[0, 16000], [16000, 32000], [32000, 48000]]))
    b = pipe._encode_chunks(audio.load_audio_16k(path), np.array([[0, 16000], [16000, 32000], [32000, 48000]]))
    assert len(a) == len(b) == 3
    for ra, rb in zip(a, b):
        assert (ra.chunk_idx, ra.start_sample, ra.end_sample, ra.n_frames) == (rb.chunk_idx, rb.start_sample, rb.end_sample, rb.n_frames)
        assert (ra.tokens is None and rb.tokens is None) or (np.array_equal(ra.tokens, rb.tokens) and np.array_equal(ra.frames, rb.frames))


def test_map_wav_pcm16_layouts(tmp_path):
    """16-bit PCM WAVs of any channel count are mapped as interleaved int16; other widths are not."""
    rng = np.random.default_rng(7)
    pcm = rng.integers(-32768, 32768, size=3 * 4410, dtype=np.int16)
    p3 = tmp_path / "three.wav"
    with wave.open(str(p3), "wb") as w:
        w.setnchannels(3); w.setsampwidth(2); w.setframerate(44100); w.writeframes(pcm.tobytes())
    samples, channels, rate = audio.map_wav_pcm16(p3)
    assert (channels, rate) == (3, 44100) and np.array_equal(samples, pcm)
    x, r = audio.read_wav(p3)
    assert np.array_equal(samples.reshape(-1, 3).astype(np.float32) / 32768.0, x) and r == rate
    p8 = tmp_path / "eight.wav"
    with wave.open(str(p8), "wb") as w:
        w.setnchannels(1); w.setsampwidth(1); w.setframerate(16000); w.writeframes(bytes(range(200)))
    assert audio.map_wav_pcm16(p8) is None
    (tmp_path / "short.wav").write_bytes(b"RIFF")
    assert audio.map_wav_pcm16(tmp_path / "short.wav") is None
