"""FLAC and AIFF ingest.

FLAC: the decoder is pinned by a file a real encoder wrote — the .flac upload kept in the reference's
repository (ref:workflows/wav2elan_web/uploads/becdcbb9…flac, copied as data to
tests/golden/reference_upload.flac) — through the MD5 of the decoded samples that the format stores in
STREAMINFO.  That file is mono, so the stereo decorrelations, wasted bits, escape partitions and the
CONSTANT / VERBATIM subframes are exercised with streams built by the small encoder below (same spec,
written for the test: self-consistency only)."""

import ctypes as C
import hashlib
import struct

import numpy as np
import pytest

from omnilingual_asr import _hip, audio


def _info(data: bytes):
    buf = np.frombuffer(data, dtype=np.uint8)
    r, c, b, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    md5 = (C.c_uint8 * 16)()
    _hip.check(_hip.lib().oasr_flac_info(buf.ctypes.data, buf.size, C.byref(r), C.byref(c), C.byref(b), C.byref(n), md5))
    return r.value, c.value, b.value, n.value, bytes(md5)


def test_flac_reference_upload_matches_its_streaminfo_md5(golden_dir):
    path = golden_dir / "reference_upload.flac"
    rate, channels, bits, total, md5 = _info(path.read_bytes())
    assert (rate, channels, bits, total) == (16000, 1, 16, 312060)
    x, r = audio.read_flac(path)
    assert r == 16000 and x.shape == (312060, 1) and x.dtype == np.float32
    pcm = np.round(x * 32768.0).astype("<i2")
    assert hashlib.md5(pcm.tobytes()).digest() == md5
    assert audio.audio_duration(path) == 312060 / 16000.0
    y = audio.load_audio_16k(path)                      # dispatch is by content
    assert y.shape == (312060,) and np.array_equal(y, x[:, 0])


# ---- a small FLAC writer (test only) ---------------------------------------------------------------------

class _Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value, width):
        if width:
            self.v = (self.v << width) | (int(value) & ((1 << width) - 1))
            self.n += width

    def unary(self, q):
        self.put(1, q + 1)

    def pad(self):
        self.put(0, (-self.n) % 8)

    def bytes(self):
        assert self.n % 8 == 0
        return self.v.to_bytes(self.n // 8, "big")


def _crc(data, width, poly):
    c, top, mask = 0, 1 << (width - 1), (1 << width) - 1
    for byte in data:
        c ^= byte << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


def _subframe(bw, samples, bps, kind, wasted=0, rice=3, escape=False):
    s = [int(v) >> wasted for v in samples]
    bps -= wasted
    bw.put(0, 1)
    order = kind - 8 if 8 <= kind <= 12 else 0
    bw.put(kind, 6)
    bw.put(1 if wasted else 0, 1)
    if wasted:
        bw.unary(wasted - 1)
    if kind == 0:
        bw.put(s[0], bps)
        return
    if kind == 1:
        for v in s:
            bw.put(v, bps)
        return
    for v in s[:order]:
        bw.put(v, bps)
    coef = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
    res = [s[i] - sum(c * s[i - 1 - j] for j, c in enumerate(coef)) for i in range(order, len(s))]
    bw.put(0, 2)          # 4-bit Rice parameters
    bw.put(0, 4)          # one partition
    if escape:
        width = max(1, max(abs(r) for r in res).bit_length() + 1)
        bw.put(15, 4)
        bw.put(width, 5)
        for r in res:
            bw.put(r, width)
    else:
        bw.put(rice, 4)
        for r in res:
            u = (r << 1) if r >= 0 else ((-r) << 1) - 1
            bw.unary(u >> rice)
            bw.put(u & ((1 << rice) - 1), rice)


def _flac(channels, bps, rate, blocks, mode, kinds, **kw):
    """blocks: list of [channels][n] integer arrays.  mode 0: independent, 8/9/10: left-side, side-right,
    mid-side (two channels)."""
    frames = b""
    for fi, blk in enumerate(blocks):
        n = len(blk[0])
        bw = _Bits()
        bw.put(0x3FFE, 14); bw.put(0, 1); bw.put(0, 1)
        bw.put(7, 4); bw.put(0, 4)                                     # 16-bit block size, rate from STREAMINFO
        bw.put(mode if mode else channels - 1, 4)
        bw.put({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}[bps], 3); bw.put(0, 1)
        bw.put(fi, 8)                                                   # frame number < 128: one byte
        bw.put(n - 1, 16)
        hdr = bw.bytes()
        bw.put(_crc(hdr, 8, 0x07), 8)
        l = np.asarray(blk[0], dtype=np.int64)
        r = np.asarray(blk[1], dtype=np.int64) if channels == 2 else None
        if mode == 8:
            chans, extra = [l, l - r], [0, 1]
        elif mode == 9:
            chans, extra = [l - r, r], [1, 0]
        elif mode == 10:
            chans, extra = [(l + r) >> 1, l - r], [0, 1]
        else:
            chans, extra = [np.asarray(c, dtype=np.int64) for c in blk], [0] * channels
        for ci, (c, e) in enumerate(zip(chans, extra)):
            _subframe(bw, c, bps + e, kinds[ci % len(kinds)], **kw)
        bw.pad()
        body = bw.bytes()
        frames += body + struct.pack(">H", _crc(body, 16, 0x8005))
    total = sum(len(b[0]) for b in blocks)
    inter = np.stack([np.concatenate([np.asarray(b[c]) for b in blocks]) for c in range(channels)], axis=1)
    width = (bps + 7) // 8
    raw = b"".join(int(v).to_bytes(width, "little", signed=True) for v in inter.reshape(-1))
    x = (rate << 44) | ((channels - 1) << 41) | ((bps - 1) << 36) | total
    si = struct.pack(">HH", 16, 65535) + b"\x00" * 6 + x.to_bytes(8, "big") + hashlib.md5(raw).digest()
    return b"fLaC" + bytes([0x80]) + len(si).to_bytes(3, "big") + si + frames, inter


@pytest.mark.parametrize("mode,kinds,kw", [
    (0, [1, 0], {}),                       # verbatim + constant, independent channels
    (8, [10, 9], {}), (9, [12, 8], {}), (10, [11, 10], {"rice": 5}),
    (10, [10], {"escape": True}), (0, [9, 1], {"wasted": 3}),
])
def test_flac_stereo_modes_subframe_kinds_wasted_bits(tmp_path, mode, kinds, kw):
    rng = np.random.default_rng(mode + len(kinds))
    blocks = []
    for n in (192, 77):
        t = np.arange(n)
        l = (3000 * np.sin(t / 9.0) + rng.integers(-40, 40, n)).astype(np.int64)
        r = (l * 0.7 + rng.integers(-25, 25, n)).astype(np.int64)
        if kinds == [1, 0]:
            r[:] = -123
        if kw.get("wasted"):
            l, r = (l >> 3) << 3, (r >> 3) << 3
        blocks.append([l, r])
    data, want = _flac(2, 16, 22050, blocks, mode, kinds, **kw)
    rate, channels, bits, total, md5 = _info(data)
    assert (rate, channels, bits, total) == (22050, 2, 16, 269)
    p = tmp_path / "t.flac"
    p.write_bytes(data)
    x, r = audio.read_flac(p)
    got = np.round(x * 32768.0).astype(np.int64)
    assert np.array_equal(got, want)
    assert hashlib.md5(got.astype("<i2").tobytes()).digest() == md5


def test_flac_24_bit_and_damage_detection(tmp_path):
    rng = np.random.default_rng(1)
    l = rng.integers(-(1 << 22), 1 << 22, 300)
    data, want = _flac(1, 24, 48000, [[l]], 0, [9], rice=14)
    p = tmp_path / "t.flac"
    p.write_bytes(data)
    x, r = audio.read_flac(p)
    assert r == 48000 and np.array_equal(np.round(x[:, 0].astype(np.float64) * (1 << 23)).astype(np.int64), want[:, 0])
    bad = bytearray(data)
    bad[len(bad) // 2] ^= 0x10                      # one flipped bit inside the frame
    p.write_bytes(bytes(bad))
    with pytest.raises(_hip.OasrError, match="CRC|malformed|sync"):
        audio.read_flac(p)
    p.write_bytes(b"fLaC" + data[4:20])
    with pytest.raises(_hip.OasrError, match="truncated|STREAMINFO"):
        audio.read_flac(p)


@pytest.mark.parametrize("kind,bits", [(b"NONE", 16), (b"sowt", 16), (b"NONE", 24), (b"NONE", 8), (b"fl32", 32), (None, 16)])
def test_aiff_variants(tmp_path, kind, bits):
    vals = np.array([[0.0, 0.5], [-0.5, 0.25], [0.125, -1.0]])
    width = bits // 8
    if kind == b"fl32":
        raw = vals.astype(">f4").tobytes()
    else:
        ints = np.round(vals * (1 << (bits - 1))).astype(np.int64).clip(-(1 << (bits - 1)), (1 << (bits - 1)) - 1)
        order = "little" if kind == b"sowt" else "big"
        raw = b"".join(int(v).to_bytes(width, order, signed=True) for v in ints.reshape(-1))
    rate80 = bytes([0x40, 0x0E, 0xAC, 0x44, 0, 0, 0, 0, 0, 0])            # 44100 as 80-bit extended
    comm = struct.pack(">hIh", 2, 3, bits) + rate80
    form = b"AIFF"
    if kind is not None:
        comm += kind + b"\x00\x00"
        form = b"AIFC"
    ssnd = struct.pack(">II", 0, 0) + raw
    body = form + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
    p = tmp_path / "x.aiff"
    p.write_bytes(b"FORM" + struct.pack(">I", len(body)) + body)
    x, rate = audio.read_audio(p)
    assert rate == 44100 and x.shape == (3, 2) and audio.audio_duration(p) == 3 / 44100.0
    np.testing.assert_allclose(x, vals.clip(-1, 1 - 2.0 ** -(bits - 1)) if kind != b"fl32" else vals, atol=2.0 ** -(bits - 1) if bits < 32 else 1e-7)


def test_lossy_formats_raise(tmp_path):
    for name, head in (("a.mp3", b"ID3\x03" + b"\x00" * 20), ("a.ogg", b"OggS" + b"\x00" * 20), ("a.m4a", b"\x00\x00\x00\x20ftypM4A ")):
        p = tmp_path / name
        p.write_bytes(head)
        with pytest.raises(audio.UnsupportedAudioError):
            audio.load_audio_16k(p)


def test_flac_through_transcribe(golden_dir, tiny_engine):
    """The reference's .flac upload end to end on the CPU (oracle engine injected): 19.5 s, one segment."""
    from omnilingual_asr import Pipeline
    segs = Pipeline(engine=tiny_engine).transcribe(str(golden_dir / "reference_upload.flac"))
    assert len(segs) == 1 and segs[0].start == 0.0 and segs[0].end == 312060 / 16000.0 and segs[0].text


def test_flac_long_stream_decodes_the_same_on_any_number_of_threads(golden_dir, tmp_path, monkeypatch):
    """A stream of several MiB (the reference upload's frames repeated with fresh frame numbers, CRCs
    recomputed: tests/flac_tools.py) is decoded by byte-range workers whose ranges must chain; the samples are
    the short file's, repeated, for every thread count - and for the 16-bit output form."""
    import flac_tools as ft

    src = golden_dir / "reference_upload.flac"
    x, rate = audio.read_flac(src)
    ints = np.round(x * 32768.0).astype(np.int32)
    stream, want = ft.repeat_stream(src.read_bytes(), 14, ints)
    assert len(stream) > 4 << 20
    p = tmp_path / "long.flac"
    p.write_bytes(stream)
    for threads in ("1", "2", "3", "4", "7"):
        monkeypatch.setenv("OASR_FLAC_THREADS", threads)
        y, r = audio.read_flac(p)
        assert r == rate and np.array_equal(np.round(y * 32768.0).astype(np.int32), want), threads
        s16, ch, r16 = audio.read_flac_pcm16(p)
        assert (ch, r16) == (1, rate) and np.array_equal(s16.astype(np.int32), want[:, 0]), threads
    # damage in the middle of the stream is reported whatever the thread count (the ranges no longer chain,
    # the in-order decode names the frame)
    bad = bytearray(stream)
    bad[len(bad) // 2] ^= 0x40
    p.write_bytes(bytes(bad))
    for threads in ("1", "4"):
        monkeypatch.setenv("OASR_FLAC_THREADS", threads)
        with pytest.raises(Exception):
            audio.read_flac(p)


def test_flac_pcm16_form_of_short_and_narrow_streams(tmp_path):
    """8-bit streams come out scaled to 16 bits; 24-bit streams have no 16-bit form."""
    rng = np.random.default_rng(3)
    v = rng.integers(-128, 128, 300).astype(np.int64)
    data, want = _flac(1, 8, 8000, [[v]], 0, [9], rice=4)
    p = tmp_path / "eight.flac"
    p.write_bytes(data)
    s16, ch, rate = audio.read_flac_pcm16(p)
    assert (ch, rate) == (1, 8000) and np.array_equal(s16.astype(np.int64), want[:, 0] * 256)
    x, _ = audio.read_flac(p)
    assert np.array_equal(s16.astype(np.float32) / 32768.0, x[:, 0])
    data24, _ = _flac(1, 24, 48000, [[rng.integers(-(1 << 20), 1 << 20, 100).astype(np.int64)]], 0, [9], rice=14)
    p.write_bytes(data24)
    assert audio.read_flac_pcm16(p) is None


def test_flac_sample_ranges_equal_the_whole_decode(golden_dir, tmp_path, monkeypatch):
    """`FlacStream.read_pcm(first, count)` — only the frames that cover the range are decoded — against the
    whole decode of the same long stream: ranges at the stream's start and end, inside one frame, across many
    frames, empty; split over 1, 3 and 16 host threads; the 32-bit form of a 24-bit stereo stream; a range that
    reaches past the end is refused; damage inside the range is reported, damage outside it is not seen."""
    import flac_tools as ft

    src = golden_dir / "reference_upload.flac"
    x, rate = audio.read_flac(src)
    ints = np.round(x * 32768.0).astype(np.int32)
    stream, want = ft.repeat_stream(src.read_bytes(), 40, ints)
    p = tmp_path / "long.flac"
    p.write_bytes(stream)
    total = want.shape[0]
    rng = np.random.default_rng(11)
    with audio.FlacStream(p) as fs:
        assert (fs.rate, fs.channels, fs.samples, fs.sample_bytes, fs.bits) == (rate, 1, total, 2, 16)
        ranges = [(0, 1), (0, 5000), (total - 1, 1), (total - 70000, 70000), (4096, 4096), (4095, 2), (1234567, 0),
                  (0, total)]
        ranges += [(int(a), int(rng.integers(1, min(total - a, 3_000_000) + 1))) for a in rng.integers(0, total - 1, 12)]
        for threads in ("1", "3", "16"):
            monkeypatch.setenv("OASR_FLAC_THREADS", threads)
            for first, count in ranges:
                got = fs.read_pcm(first, count)
                assert got.dtype == np.int16 and np.array_equal(got.astype(np.int32), want[first: first + count, 0]), (threads, first, count)
        with pytest.raises(Exception, match="outside the stream"):
            fs.read_pcm(total - 10, 11)
    # damage: one flipped bit at 3/4 of the file
    bad = bytearray(stream)
    at = len(bad) * 3 // 4
    bad[at] ^= 0x10
    p.write_bytes(bytes(bad))
    monkeypatch.setenv("OASR_FLAC_THREADS", "1")
    with audio.FlacStream(p) as fs:
        head = fs.read_pcm(0, total // 2)
        assert np.array_equal(head.astype(np.int32), want[: total // 2, 0])
        with pytest.raises(Exception):
            fs.read_pcm(total // 2, total - total // 2)
    # 24-bit stereo: the int32 form
    v = [rng.integers(-(1 << 20), 1 << 20, 4096).astype(np.int64) for _ in range(2 * 6)]
    data24, want24 = _flac(2, 24, 48000, [[v[2 * i], v[2 * i + 1]] for i in range(6)], 0, [9, 9], rice=14)
    p.write_bytes(data24)
    with audio.FlacStream(p) as fs:        # the little writer states block sizes 16..65535: positions are unknown
        with pytest.raises(Exception, match="frame numbers do not give sample positions"):
            fs.read_pcm(0, 10)
    p.write_bytes(data24[:8] + struct.pack(">HH", 4096, 4096) + data24[12:])
    with audio.FlacStream(p) as fs:
        assert (fs.channels, fs.sample_bytes, fs.bits, fs.samples) == (2, 4, 24, 6 * 4096)
        for first, count in ((0, 6 * 4096), (4000, 200), (8192, 4096), (24575, 1)):
            got = fs.read_pcm(first, count).reshape(-1, 2)
            assert np.array_equal(got.astype(np.int64), want24[first: first + count])


def test_aiff_that_states_no_sample_width_is_refused(tmp_path):
    """A COMM chunk with 0 (or more than 32) bits per sample: refused, not divided by."""
    for bits in (0, 40, -8):
        comm = struct.pack(">hIh", 1, 4, bits) + b"\x40\x0e\xac\x44" + bytes(6)
        ssnd = struct.pack(">II", 0, 0) + bytes(16)
        body = b"AIFF" + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
        p = tmp_path / "z.aiff"
        p.write_bytes(b"FORM" + struct.pack(">I", len(body)) + body)
        with pytest.raises(audio.UnsupportedAudioError):
            audio.read_audio(p)

def test_flac_in_an_ogg_container_is_unwrapped_and_read_like_the_native_stream(golden_dir, tmp_path, tiny_engine):
    """The reference's FLAC upload wrapped page by page as the FLAC format's Ogg mapping lays it out
    (tests/flac_tools.py: `wrap_in_ogg`; parity unpinned - no encoder for that mapping exists here): the unwrapped
    bytes are the native stream's signature, metadata and frames; samples, duration, sample ranges and the transcript
    equal the native file's; a stream that states no length is counted by decoding; a damaged page, a frame that does
    not start with the synchronisation code and a truncated file behave like their Vorbis counterparts."""
    import flac_tools as ft
    from omnilingual_asr import Pipeline

    src = golden_dir / "reference_upload.flac"
    native = src.read_bytes()
    x, rate = audio.read_flac(src)
    wrapped = ft.wrap_in_ogg(native, comment=b"TITLE=upload")
    p = tmp_path / "upload.ogg"
    p.write_bytes(wrapped)
    assert audio.is_flac(p) and not audio.is_ogg(p) and not audio.is_mp3(p)
    # the unwrapped stream: the native file's STREAMINFO and frames around the one comment block the wrapper wrote
    plain = audio._flac_bytes(p).tobytes()
    info, frames = ft.split_stream(native)
    info2, frames2 = ft.split_stream(plain)
    assert plain[:4] == b"fLaC" and info2 == info and frames2 == frames and b"TITLE=upload" in plain[:200]
    y, r = audio.read_audio(p)
    assert r == rate and np.array_equal(y, x)
    assert audio.audio_duration(p) == audio.audio_duration(src)
    s16, ch, r16 = audio.read_flac_pcm16(p)
    assert (ch, r16) == (1, rate) and np.array_equal(s16.astype(np.float32) / 32768.0, x[:, 0])
    np.testing.assert_array_equal(audio.load_audio_16k(p), audio.load_audio_16k(src))
    with audio.FlacStream(p) as fs:
        assert fs.seekable and fs.samples == x.shape[0]
        got = fs.read_pcm(100000, 5000)
        assert np.array_equal(got.astype(np.float32) / 32768.0, x[100000:105000, 0])
    a = Pipeline(engine=tiny_engine).transcribe(str(src))
    b = Pipeline(engine=tiny_engine).transcribe(str(p))
    assert [(s.start, s.end, s.text) for s in a] == [(s.start, s.end, s.text) for s in b] and a[0].text
    # no stated length
    p.write_bytes(ft.wrap_in_ogg(native, per_page=3, unknown_total=True))
    y, _ = audio.read_audio(p)
    assert np.array_equal(y, x)
    with pytest.raises(audio.UnsupportedAudioError, match="no duration"):
        audio.audio_duration(p)
    # damage inside a page: the checksum says so
    bad = bytearray(wrapped)
    bad[len(bad) // 2] ^= 1
    p.write_bytes(bytes(bad))
    with pytest.raises(audio.UnsupportedAudioError, match="checksum"):
        audio.read_audio(p)
    # a page whose packet is not a frame (checksum made right again)
    import vorbis_tools as vt
    at = wrapped.index(b"OggS", wrapped.index(b"OggS", 100) + 4)      # the first audio page
    nseg = wrapped[at + 26]
    body = at + 27 + nseg
    bad = bytearray(wrapped)
    bad[body] = 0x00
    size = 27 + nseg + sum(wrapped[at + 27: at + 27 + nseg])
    page = bytes(bad[at: at + 22]) + bytes(4) + bytes(bad[at + 26: at + size])
    bad[at + 22: at + 26] = struct.pack("<I", vt.ogg_crc(page))
    p.write_bytes(bytes(bad))
    with pytest.raises(audio.UnsupportedAudioError, match="does not start with a frame header"):
        audio.read_audio(p)
    # a file cut inside a page: the frames of the whole pages, as a native file cut short gives its whole frames
    for data in (wrapped, ft.wrap_in_ogg(native, unknown_total=True)):
        p.write_bytes(data[: len(data) * 2 // 3])
        y, _ = audio.read_audio(p)
        assert 0.5 * x.shape[0] < y.shape[0] < 0.75 * x.shape[0] and np.array_equal(y, x[: y.shape[0]])
    # a Vorbis reader asked for this file names what it holds
    d = np.frombuffer(wrapped, dtype=np.uint8)
    ra, ca, ta = C.c_int32(), C.c_int32(), C.c_int64()
    with pytest.raises(_hip.OasrError, match="holds FLAC"):
        _hip.check(_hip.lib().oasr_vorbis_info(d.ctypes.data, d.size, C.byref(ra), C.byref(ca), C.byref(ta)))
