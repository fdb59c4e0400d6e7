"""Ogg Vorbis decoder (`csrc/vorbis.cpp`), host code reached through the C ABI.

Parity unpinned against the reference (it forwards `.ogg` uploads to its remote model and holds no stream of the
kind, ref gemini_pipeline.py:106-118).  What pins the decoder instead:
(1) a stream a real encoder wrote - MathJax's `invalid_keypress.ogg` (libVorbis 2007, 44.1 kHz stereo, 26 packets,
    block sizes 256 / 2048, floor 1, coupled residue), found in this image's Python packages and kept as
    `tests/golden/mathjax_invalid_keypress.ogg`: every packet's residue decode ends inside the packet's last byte (an
    encoder writes no more than it needs; a wrong codeword table, floor or residue layout loses that within a few
    symbols), the page checksums hold, the decoded sound is a clean low tone - time-domain aliasing cancels to
    1e-9 of the energy, where a decoder with the alias sign of the inverse MDCT flipped leaves 5e-5 - and every sample
    equals the one `oracle/vorbis.py` computes, a second decoder written from the specification alone (codewords
    matched bit by bit in a dictionary, the inverse MDCT as its cosine sum in float64);
(2) the format's forward definitions: `tests/vorbis_tools.py` writes streams (codebooks of both lookup kinds, ordered
    and sparse length lists, floor-1 posts, residues 0 / 1 / 2 in cascaded passes, coupling, submaps, both block
    sizes with every window shape) and states the PCM each stands for, computed in float64 from the definitions."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).parent))
import vorbis_tools as vt  # noqa: E402
from omnilingual_asr import _hip, audio  # noqa: E402
from oracle import vorbis as OV  # noqa: E402


def _decode(data: bytes, threads: int = 1, mono: bool = False):
    L = _hip.lib()
    d = np.frombuffer(data, dtype=np.uint8)
    rate, ch, tot = C.c_int32(), C.c_int32(), C.c_int64()
    _hip.check(L.oasr_vorbis_info(d.ctypes.data, d.size, C.byref(rate), C.byref(ch), C.byref(tot)))
    out = np.zeros((tot.value,) if mono else (tot.value, ch.value), dtype=np.float32)
    w = C.c_int64()
    fn = L.oasr_vorbis_decode_mono if mono else L.oasr_vorbis_decode
    _hip.check(fn(d.ctypes.data, d.size, out.ctypes.data, tot.value, C.byref(w), threads))
    assert w.value == tot.value
    return out, rate.value


def _tails(data: bytes) -> np.ndarray:
    d = np.frombuffer(data, dtype=np.uint8)
    tails = np.zeros(200000, dtype=np.int32)
    n = C.c_int64()
    _hip.check(_hip.lib().oasr_vorbis_packet_tails(d.ctypes.data, d.size, tails.ctypes.data, tails.size, C.byref(n)))
    return tails[: n.value]


def test_a_stream_from_a_real_encoder(golden_dir):
    path = golden_dir / "mathjax_invalid_keypress.ogg"
    data = path.read_bytes()
    assert b"Xiph.Org libVorbis" in data[:400]
    assert audio.vorbis_info(path) == (44100, 2, 22050)
    assert audio.audio_duration(path) == 0.5
    tails = _tails(data)
    assert len(tails) == 26 and tails.min() >= 0 and tails.max() < 8, tails
    x, rate = audio.read_ogg(path)
    assert rate == 44100 and x.shape == (22050, 2) and np.isfinite(x).all()
    assert 0.5 < np.abs(x).max() < 0.7
    # a thud of ~0.2 s, then digital silence
    assert np.sqrt((x[:8192] ** 2).mean()) > 0.15 and np.abs(x[11000:]).max() < 1e-3
    seg = x[:8192, 0].astype(np.float64) * np.hanning(8192)
    power = np.abs(np.fft.rfft(seg)) ** 2
    hz = np.arange(power.size) * 44100.0 / 8192
    assert 100 < hz[power.argmax()] < 250
    # aliasing between blocks cancels: a flipped alias sign in the inverse MDCT measures 4.8e-5 here, this decoder 1e-9
    assert power[hz > 1500].sum() / power.sum() < 1e-7
    mono, _ = audio.read_ogg_mono(path)
    np.testing.assert_array_equal(mono, x.mean(axis=1, dtype=np.float64).astype(np.float32))
    # sample by sample against the restatement: float32 butterflies against a float64 cosine sum (measured 1.1e-7)
    want, want_rate = OV.decode(data)
    assert want_rate == rate and want.shape == x.shape
    assert np.abs(x - want).max() < 1e-6


def _signal(n, ch, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = 0.15 * rng.standard_normal((n, ch)) + 0.2 * np.sin(2 * np.pi * 330.0 * t)[:, None]
    if ch == 2:
        x[:, 1] = 0.6 * x[:, 0] + 0.05 * rng.standard_normal(n)
        x[n // 3: n // 2, 1] = 0.0          # a stretch with one channel silent: "unused" floors next to coded ones
    return x


CONFIGS = {
    "mono residue 1, lattice books, block switching": vt.Config(pattern="LLSSL", seed=0),
    "mono residue 0, begin > 0": vt.Config(pattern="LSL", seed=1, residue_type=0),
    "mono table books": vt.Config(pattern="LLSSSL", seed=2, book_kind="table"),
    "mono table books with running sums, floor multiplier 4": vt.Config(pattern="L", seed=3, book_kind="table", seq_books=True,
                                                                     floor_multiplier=4),
    "mono floor multiplier 1 and 3": vt.Config(pattern="SLL", seed=5, floor_multiplier=1),
    "floor multiplier 3": vt.Config(pattern="LS", seed=6, floor_multiplier=3, book_kind="table"),
    "stereo coupled residue 2": vt.Config(channels=2, pattern="LLSSL", seed=0, residue_type=2, coupled=True),
    "stereo coupled residue 1, table books": vt.Config(channels=2, pattern="LSSL", seed=1, residue_type=1, coupled=True,
                                                       book_kind="table"),
    "stereo, a submap per channel (residues 0 and 1)": vt.Config(channels=2, pattern="LLS", seed=2, residue_type=0, submaps=2),
    "stereo residue 2, blocks 64 / 512, short first, end trimmed": vt.Config(channels=2, pattern="SL", seed=3, residue_type=2,
                                                                          trim_end=True, bs0=64, bs1=512, partition_size=16),
    "stereo, start and end trimmed by granule positions": vt.Config(channels=2, pattern="LLSL", seed=4, residue_type=2,
                                                                  coupled=True, start_offset=300, trim_end=True),
    "both block sizes equal": vt.Config(channels=2, pattern="LS", seed=5, residue_type=1, coupled=True, bs0=2048, bs1=2048),
    "largest blocks": vt.Config(pattern="LSSL", seed=7, bs0=1024, bs1=8192, partition_size=64),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_written_streams_decode_to_the_pcm_their_spectra_stand_for(name):
    cfg = CONFIGS[name]
    n = 20000 if cfg.bs1 == 8192 else 9000
    x = _signal(n, cfg.channels, cfg.seed)
    enc = vt.encode(x if cfg.channels > 1 else x[:, 0], cfg)
    out, rate = _decode(enc.data)
    assert rate == cfg.rate and out.shape == (enc.total, cfg.channels)
    if cfg.trim_end:
        assert enc.total == n - cfg.start_offset
    tails = _tails(enc.data)
    assert len(tails) == len(enc.blocks) and tails.min() >= 0 and tails.max() < 8
    # float32 transforms against the float64 definition
    assert np.abs(out - enc.expected).max() < 2e-6 * max(1.0, np.abs(enc.expected).max())
    assert np.abs(enc.expected).max() > 0.1
    mono, _ = _decode(enc.data, mono=True)
    np.testing.assert_array_equal(mono, out.mean(axis=1, dtype=np.float64).astype(np.float32))


@pytest.mark.parametrize("name", ["mono residue 0, begin > 0", "mono table books with running sums, floor multiplier 4",
                                  "stereo coupled residue 2", "stereo, a submap per channel (residues 0 and 1)",
                                  "stereo residue 2, blocks 64 / 512, short first, end trimmed"])
def test_the_restatement_reads_written_streams_too(name):
    """The checker checked: `oracle/vorbis.py` against the forward definitions (streams whose start is trimmed by the
    first page's granule position are left out: the restatement does not do that, and the real stream does not ask)."""
    cfg = CONFIGS[name]
    x = _signal(3000, cfg.channels, cfg.seed)
    enc = vt.encode(x if cfg.channels > 1 else x[:, 0], cfg)
    got, rate = OV.decode(enc.data)
    assert rate == cfg.rate
    assert got.shape == enc.expected.shape
    assert np.abs(got - enc.expected).max() < 2e-6 * max(1.0, np.abs(enc.expected).max())


def test_long_stream_any_thread_count_and_any_sample_range():
    cfg = vt.Config(channels=2, pattern="LSSLL", seed=8, residue_type=2, coupled=True, bs0=128, bs1=1024)
    enc = vt.encode(_signal(6000, 2, 8), cfg)
    data = vt.repeat_stream(enc, 120)
    assert len(enc.blocks) * 120 > 1100          # more than two threads' worth of packets
    one, rate = _decode(data, threads=1)
    assert one.shape[0] > 300000 and np.abs(one).max() > 0.1
    for t in (0, 2, 3, 7):
        np.testing.assert_array_equal(_decode(data, threads=t)[0], one)
    mono = one.mean(axis=1, dtype=np.float64).astype(np.float32)
    L = _hip.lib()
    d = np.frombuffer(data, dtype=np.uint8)
    h, r, c, tot = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int64()
    _hip.check(L.oasr_vorbis_open(d.ctypes.data, d.size, C.byref(h), C.byref(r), C.byref(c), C.byref(tot)))
    try:
        assert (r.value, c.value, tot.value) == (rate, 2, one.shape[0])
        rng = np.random.default_rng(0)
        spans = [(0, 1), (0, 5000), (tot.value - 777, 777), (tot.value - 1, 1), (12345, 0)]
        spans += [(0, tot.value), (1001, tot.value - 3003)]        # long ranges: several threads inside the call
        spans += [(int(a), int(rng.integers(1, 9000))) for a in rng.integers(0, tot.value - 9000, size=12)]
        for first, count in spans:
            got = np.full(count, np.nan, dtype=np.float32)
            _hip.check(L.oasr_vorbis_decode_mono_range(h, first, count, got.ctypes.data))
            np.testing.assert_array_equal(got, mono[first: first + count])
        assert L.oasr_vorbis_decode_mono_range(h, tot.value - 5, 6, got.ctypes.data) != 0
    finally:
        L.oasr_vorbis_close(h)


def test_chained_streams_come_out_as_one_signal(golden_dir):
    """Logical streams one behind the other (a chained file: concatenated recordings, captured broadcasts) are one
    signal: every link's samples in order, whole or by ranges that cross the joins, on any thread count; links with
    different setups (block sizes, residue types, books) are fine, links that differ in rate or channel count are
    refused by name rather than cut off; the real stream behind itself decodes to itself twice."""
    a_cfg = vt.Config(channels=2, pattern="LSSL", seed=1, residue_type=2, coupled=True)
    b_cfg = vt.Config(channels=2, pattern="LL", seed=2, residue_type=1, coupled=True, book_kind="table", bs0=128, bs1=512,
                      trim_end=True)
    a, b = vt.encode(_signal(7000, 2, 1), a_cfg), vt.encode(_signal(5000, 2, 2), b_cfg)
    want = np.concatenate([a.expected, b.expected, a.expected])
    for eos in (True, False):
        data = vt.with_serial(a.data, 11, eos) + vt.with_serial(b.data, 12, eos) + vt.with_serial(a.data, 13)
        out, rate = _decode(data)
        assert rate == 16000 and out.shape == want.shape
        assert np.abs(out - want).max() < 2e-6
        assert len(_tails(data)) == 2 * len(a.blocks) + len(b.blocks) and _tails(data).min() >= 0
    for threads in (2, 5):
        np.testing.assert_array_equal(_decode(data, threads=threads)[0], out)
    mono, _ = _decode(data, mono=True)
    L = _hip.lib()
    d = np.frombuffer(data, dtype=np.uint8)
    h, r, c, t = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int64()
    _hip.check(L.oasr_vorbis_open(d.ctypes.data, d.size, C.byref(h), C.byref(r), C.byref(c), C.byref(t)))
    try:
        assert t.value == want.shape[0]
        for first, count in ((0, t.value), (a.total - 10, 20), (a.total, b.total), (a.total - 1, b.total + 2), (a.total + b.total - 300, 301),
                             (a.total + b.total, 1), (t.value - 1, 1), (3, 0)):
            got = np.zeros(count, dtype=np.float32)
            _hip.check(L.oasr_vorbis_decode_mono_range(h, first, count, got.ctypes.data))
            np.testing.assert_array_equal(got, mono[first: first + count])
    finally:
        L.oasr_vorbis_close(h)
    # a link of another shape: named, not dropped
    other = vt.encode(_signal(3000, 1, 3)[:, 0], vt.Config(pattern="L", seed=3))
    with pytest.raises(_hip.OasrError, match="link 2 of the chained file is 16000 Hz with 1 channel"):
        _decode(a.data + vt.with_serial(other.data, 99))
    # junk after the last link ends the chain quietly
    np.testing.assert_array_equal(_decode(a.data + b"\x00" * 100 + b"OggS" + b"\x01" * 10)[0], _decode(a.data)[0])
    # the real stream twice (the same serial number again: tolerated)
    real = (golden_dir / "mathjax_invalid_keypress.ogg").read_bytes()
    x, _ = _decode(real)
    xx, _ = _decode(real + real)
    np.testing.assert_array_equal(xx, np.concatenate([x, x]))


def test_damaged_and_foreign_streams_are_refused_or_cut_short(tmp_path):
    enc = vt.encode(_signal(9000, 1, 9)[:, 0], vt.Config(pattern="SSL", seed=9, bs0=64, bs1=256, partition_size=16))
    data = vt.ogg_pages(enc.packets, enc.granules, per_page=[1, 1, 1] + [10] * len(enc.packets))   # several audio pages
    whole, _ = _decode(data)
    np.testing.assert_array_equal(whole, _decode(enc.data)[0])

    def refuses(data, text):
        p = tmp_path / "x.ogg"
        p.write_bytes(data)
        with pytest.raises(audio.UnsupportedAudioError, match=text):
            audio.read_ogg(p)

    # a file that ends inside a page is decoded up to the last whole page
    pages = [i for i in range(len(data)) if data[i:i + 4] == b"OggS"]
    assert len(pages) >= 6
    cut, _ = _decode(data[: pages[-1] + 40])
    assert 0 < cut.shape[0] < whole.shape[0]
    np.testing.assert_array_equal(cut, whole[: cut.shape[0]])
    # one flipped bit in a page body: the checksum says so
    bad = bytearray(data)
    bad[pages[-1] + 60] ^= 0x10
    refuses(bytes(bad), "checksum")
    # a page removed: the sequence numbers say so
    refuses(data[: pages[-2]] + data[pages[-1]:], "missing")
    # other codecs in an Ogg container are named
    head = b"OpusHead" + bytes(11)
    refuses(vt.ogg_pages([head], [0]), "Opus")
    refuses(vt.ogg_pages([b"\x80theora" + bytes(30)], [0]), "another codec")
    refuses(bytes(300), "no Vorbis stream")
    # headers only: a stream with no audio packet is an empty signal
    empty, _ = _decode(vt.ogg_pages(enc.packets[:3], [0, 0, 0]))
    assert empty.shape == (0, 1)
    # a setup header cut short
    refuses(vt.ogg_pages(enc.packets[:2] + [enc.packets[2][:200]] + enc.packets[3:], [0] * len(enc.packets)), "vorbis:")
    # random bytes after the headers: packets decode to something finite or to silence, never past a buffer
    rng = np.random.default_rng(1)
    junk = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 400)), dtype=np.uint8)) for _ in range(60)]
    out, _ = _decode(vt.ogg_pages(enc.packets[:3] + junk, [0, 0, 0] + [-1] * 59 + [4000]))
    assert np.isfinite(out).all()


def test_ogg_through_the_pipeline(tmp_path, tiny_engine):
    """`.ogg` end to end on the CPU (oracle engine injected): the duration probe, the chunker and the merge see an
    Ogg Vorbis file as they see a WAV."""
    from omnilingual_asr import Pipeline
    x = _signal(16000 * 7, 1, 10)[:, 0]
    enc = vt.encode(x, vt.Config(pattern="LLLS", seed=10, rate=16000, trim_end=True))
    p = tmp_path / "speech.ogg"
    p.write_bytes(enc.data)
    assert audio.audio_duration(p) == 7.0
    pcm = audio.load_audio_16k(p)
    assert pcm.shape == (16000 * 7,) and pcm.dtype == np.float32
    np.testing.assert_allclose(pcm, enc.expected[:, 0], atol=2e-6)
    pipe = Pipeline(engine=tiny_engine, chunk_seconds=2.5, min_chunking_seconds=2.5)
    segs = pipe.transcribe(str(p))
    assert len(segs) == 3 and segs[0].start == 0.0 and segs[-1].end == 7.0
