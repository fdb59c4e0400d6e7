"""MPEG Layer III ingest (csrc/mp3.cpp through omnilingual_asr.audio.read_mp3).

The reference ships no decoder and no decoded audio (it uploads the bytes: ref gemini_pipeline.py:106-118, :512-530),
so the decoded samples are PARITY-UNPINNED against it.  What pins the decoder instead:
  * a file a real encoder wrote — the first 40 s of the demo recording in the reference's repository
    (tests/golden/reference_demo_40s.mp3, cut by oracle/gen_mp3_fixture.py): every granule's Huffman data must end on
    the bit its side information states (no overrun, no stuffing), the spectrum must be empty above the encoder's
    low-pass, and the energy must follow the speech intervals of the reference's own annotation of that recording
    (ref:samuel_lisu.eaf, tier SPEAKER_00);
  * the format's own mathematics — a small encoder in tests/mp3_tools.py, written from the forward definitions
    (analysis filter bank, MDCT, quantiser): PCM -> encoder -> decoder must come back with the signal-to-noise ratio
    of the quantised spectrum, for every block type, stereo mode, sampling frequency, code book and scale-factor form.
Intensity stereo and mixed blocks do not occur in the recording; the encoder produces them.
"""
import json

import numpy as np
import pytest

from omnilingual_asr import audio

import mp3_tools as mt


def test_code_books_are_complete_prefix_codes_and_the_include_file_is_theirs():
    """Every Huffman book: Kraft sum exactly 1 and no code a prefix of another (a wrong length or code cannot hide),
    and csrc/mp3_tables.inc holds exactly the trees built from them."""
    import re
    gen = mt.books
    for n in gen.BOOKS:
        assert gen.check_book(n) == [], n
    text = (mt.ROOT / "omnilingual-asr_amd" / "csrc" / "mp3_tables.inc").read_text()
    for n in gen.BOOKS:
        body = text[text.index("kMp3Book%d[" % n):]
        body = body[body.index("=") + 1: body.index(";")]
        nodes = [[int(a), int(b)] for a, b in re.findall(r"\{(-?\d+),(-?\d+)\}", body)]
        assert nodes == gen.build_tree(n), n


def test_second_standard_partition_table_is_consistent():
    """Every way of splitting the scale factors into four width classes must cover the same bands: 21 for long granules,
    36 for short ones, 33 for mixed ones (6 long + 27 short)."""
    for rule in mt.LSF_PARTS:
        assert [sum(row) for row in rule] == [21, 36, 33], rule


@pytest.fixture(scope="module")
def demo(golden_dir):
    x, rate, diag = audio.read_mp3(golden_dir / "reference_demo_40s.mp3", diagnostics=True)
    info = json.loads((golden_dir / "reference_demo_mp3.json").read_text())
    return x, rate, diag, info


def test_reference_demo_bit_accounting(demo, golden_dir):
    x, rate, diag, info = demo
    frames = info["fixture_audio_frames"]
    assert rate == 44100 and x.shape == (frames * 1152 - 576 - 1190, 2)      # delay and padding of its Info frame
    assert list(diag) == [frames, 0, 0, 0, 0]      # all 4 x 1531 granules end exactly where the side information says
    assert audio.audio_duration(golden_dir / "reference_demo_40s.mp3") == x.shape[0] / 44100.0
    assert 0.5 < np.abs(x).max() < 1.0 and 0.03 < np.sqrt((x.astype(np.float64) ** 2).mean()) < 0.12
    whole = info["whole_file"]                      # the whole recording, as decoded when the fixture was cut
    assert whole["diagnostics"] == [whole["audio_frames"], 0, 0, 0, 0]
    assert abs(whole["samples_per_channel"] / whole["sample_rate"] - whole["annotated_end_ms"] / 1000) < 1.0


def test_reference_demo_spectrum_and_continuity(demo):
    x = demo[0][:, 0].astype(np.float64)
    n = 4096
    seg = x[: len(x) // n * n].reshape(-1, n) * np.hanning(n)
    p = (np.abs(np.fft.rfft(seg, axis=1)) ** 2).mean(axis=0)
    f = np.fft.rfftfreq(n, 1 / 44100)
    db = lambda lo, hi: 10 * np.log10(p[(f >= lo) & (f < hi)].mean())  # noqa: E731
    assert db(0, 4000) - db(17500, 22050) > 80       # the encoder's low-pass: aliasing terms of the filter bank cancel
    assert db(0, 4000) - db(12000, 16000) < 45       # ... while the band below it is alive
    d2 = np.abs(np.diff(x, 2))
    phase = (np.arange(len(d2)) + 576 + 529 + 1) % 576
    prof = np.array([d2[phase == k].mean() for k in range(576)])
    assert prof.max() / np.median(prof) < 1.15        # nothing special happens at granule boundaries


def test_reference_demo_follows_the_reference_annotation(demo):
    x, rate, _, info = demo
    mono = x.astype(np.float64).mean(axis=1)
    hop = rate // 100
    e = 10 * np.log10((mono[: len(mono) // hop * hop].reshape(-1, hop) ** 2).mean(axis=1) + 1e-12)
    speech = np.zeros(len(e), dtype=bool)
    for a, b in info["speech_intervals_ms"]:
        speech[a // 10: b // 10] = True
    speech[info["speech_intervals_ms"][-1][1] // 10:] = True         # (the cut falls inside the next interval)
    assert speech.sum() > 2000 and (~speech).sum() > 200
    contrast = {lag: e[np.roll(speech, lag)].mean() - e[~np.roll(speech, lag)].mean() for lag in range(-60, 61, 5)}
    best = max(contrast, key=contrast.get)
    assert contrast[0] > 6.0 and abs(best) <= 15, (contrast[0], best)    # dB; lag in 10 ms steps


def test_thread_count_does_not_change_a_sample(golden_dir, monkeypatch):
    out = {}
    for th in ("1", "2", "7"):
        monkeypatch.setenv("OASR_MP3_THREADS", th)
        out[th] = audio.read_mp3(golden_dir / "reference_demo_40s.mp3", diagnostics=True)
    for th in ("2", "7"):
        assert np.array_equal(out["1"][0], out[th][0]) and list(out["1"][2]) == list(out[th][2])


def test_mono_form_equals_the_float64_mean(golden_dir, monkeypatch):
    x, rate = audio.read_mp3(golden_dir / "reference_demo_40s.mp3")
    for th in ("1", "3"):
        monkeypatch.setenv("OASR_MP3_THREADS", th)
        m, r = audio.read_mp3_mono(golden_dir / "reference_demo_40s.mp3")
        assert r == rate and np.array_equal(m, x.mean(axis=1, dtype=np.float64).astype(np.float32))
    y = audio.load_audio_16k(golden_dir / "reference_demo_40s.mp3")
    assert y.dtype == np.float32 and y.shape[0] == -(-x.shape[0] * 160 // 441)
    assert np.array_equal(y, audio.resample_poly(x.mean(axis=1, dtype=np.float64).astype(np.float32), rate))


def test_ranges_of_an_opened_stream_equal_the_whole_decode(golden_dir, monkeypatch):
    path = golden_dir / "reference_demo_40s.mp3"
    whole, rate = audio.read_mp3_mono(path)
    rng = np.random.default_rng(4)
    with audio.Mp3Stream(path) as st:
        assert (st.rate, st.channels, st.samples) == (rate, 2, whole.shape[0])
        spans = [(0, 1), (0, 5000), (whole.shape[0] - 3, 3), (1151, 2), (0, whole.shape[0]), (777777, 0)]
        spans += [(int(a), int(rng.integers(1, 300000))) for a in rng.integers(0, whole.shape[0] - 300000, 6)]
        for th in ("1", "3"):
            monkeypatch.setenv("OASR_MP3_THREADS", th)
            for first, count in spans:
                assert np.array_equal(st.read_mono(first, count), whole[first: first + count]), (th, first, count)
        from omnilingual_asr import _hip
        with pytest.raises(_hip.OasrError):
            st.read_mono(whole.shape[0] - 2, 3)


# ---- round trips through the test encoder ----------------------------------------------------------------------------
def _band_noise(rng, n, rate, lo_line, lines, ch=1, amp=0.25):
    """Noise confined to MDCT lines [lo_line, lo_line + lines) of a 576-line granule."""
    spec = np.fft.rfft(rng.standard_normal((ch, n)), axis=1)
    f = np.fft.rfftfreq(n, 1 / rate)
    hz = rate / 2 / 576
    spec[:, (f < lo_line * hz) | (f >= (lo_line + lines) * hz)] = 0
    x = np.fft.irfft(spec, n, axis=1).T
    return x / np.abs(x).max() * amp


def _pink_noise(rng, n, ch=1, amp=0.3, power=1.5):
    spec = np.fft.rfft(rng.standard_normal((ch, n)), axis=1)
    spec[:, 1:] /= np.arange(1, spec.shape[1]) ** (power / 2)
    spec[:, 0] = 0
    x = np.fft.irfft(spec, n, axis=1).T
    return x / np.abs(x).max() * amp


DELAY = 481 + 576      # analysis + synthesis filters, one granule of MDCT overlap


def _roundtrip(tmp_path, pcm, rate, **opts):
    data, info = mt.encode(pcm, rate, **opts)
    p = tmp_path / "t.mp3"
    p.write_bytes(data)
    y, r, diag = audio.read_mp3(p, diagnostics=True)
    assert r == rate and y.shape[1] == pcm.shape[1] and y.shape[0] == info["frames"] * (576 if rate < 32000 else 1152)
    assert list(diag[:4]) == [info["frames"], 0, 0, 0], diag
    spf = 576 if rate < 32000 else 1152
    a, b = 2 * spf, y.shape[0] - 2 * spf - DELAY            # interior: away from the start-up and the missing tail
    ref, got = pcm[a:b], y[a + DELAY: b + DELAY].astype(np.float64)
    snr = 10 * np.log10((ref ** 2).sum() / ((ref - got) ** 2).sum())
    return snr, info


def test_the_delay_constant(tmp_path):
    rng = np.random.default_rng(1)
    pcm = _band_noise(rng, 1152 * 12, 44100, 10, 50)
    data, _ = mt.encode(pcm, 44100, peak_is=4000)
    (tmp_path / "d.mp3").write_bytes(data)
    y, _ = audio.read_mp3(tmp_path / "d.mp3")
    a, b = 2304, 9216
    err = {lag: ((pcm[a:b, 0] - y[a + lag: b + lag, 0]) ** 2).sum() for lag in range(900, 1300)}
    assert min(err, key=err.get) == DELAY


@pytest.mark.parametrize("rate", [44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000])
@pytest.mark.parametrize("blocks", ["long", "switching"])
def test_perfect_reconstruction_at_fine_quantisation(tmp_path, rate, blocks):
    """Band-limited noise anywhere in the spectrum, quantised finely: what comes back is the signal (> 50 dB) — the
    synthesis window, both inverse MDCTs with all four window shapes, alias reduction, frequency inversion,
    short-block ordering and the power law are right together or this fails."""
    rng = np.random.default_rng(rate + len(blocks))
    spf = 576 if rate < 32000 else 1152
    for lo in (0, 37, 200, 480):
        pcm = _band_noise(rng, spf * 14, rate, lo, 45)
        bt = None if blocks == "long" else [0, 0, 1, 2, 2, 3, 0, 1, 2, 3, 1, 2, 2, 2, 3, 0]
        snr, info = _roundtrip(tmp_path, pcm, rate, peak_is=3000, block_types=bt, seed=lo)
        # short windows spread a band over three times its lines: the spectrum's own figure is the yardstick
        assert info["snr_spectrum_db"] > (50 if blocks == "long" else 35), info["snr_spectrum_db"]
        assert snr > info["snr_spectrum_db"] - 1.5, (rate, blocks, lo, snr, info["snr_spectrum_db"])


@pytest.mark.parametrize("mode", ["mono", "stereo", "ms"])
@pytest.mark.parametrize("rate", [44100, 22050, 8000])
def test_coarse_quantisation_scale_factors_reservoir(tmp_path, rate, mode):
    """Broadband noise, coarse and changing quantiser steps, random scale factors (scfsi, preflag, scalefac_scale,
    subblock gain), window switching, CRC, the bit reservoir in use: the time-domain error must be the quantiser's
    (the filter bank is orthogonal), i.e. equal the spectrum's signal-to-noise ratio to within 1.5 dB."""
    rng = np.random.default_rng(rate)
    ch = 1 if mode == "mono" else 2
    spf = 576 if rate < 32000 else 1152
    pcm = _pink_noise(rng, spf * 24, ch)
    if ch == 2:
        pcm[:, 1] = 0.6 * pcm[:, 0] + 0.4 * pcm[:, 1]
    snr, info = _roundtrip(tmp_path, pcm, rate, ms=mode == "ms", crc=True, scalefactors=True, reservoir=True,
                           peak_is=[1, 2, 3, 5, 7, 15, 40], seed=7,
                           block_types=[0, 0, 0, 1, 2, 2, 3, 0, 0, 1, 2, 3])
    assert 1 < info["snr_spectrum_db"] < 40
    assert abs(snr - info["snr_spectrum_db"]) < 1.5, (snr, info["snr_spectrum_db"])


@pytest.mark.parametrize("table", mt.VALID_TABLES)
def test_every_code_book(tmp_path, table):
    """Each pair book on its own (all three regions forced to it, every line in the pair region), at the largest
    magnitude it holds or 600, whichever is smaller; both quadruple books are drawn at random in every test."""
    rng = np.random.default_rng(table)
    pcm = _pink_noise(rng, 1152 * 8, 1, power=2.5)
    peak = min(mt.table_max(table), 600)
    snr, info = _roundtrip(tmp_path, pcm, 44100, tables=[table], peak_is=peak, all_big=table < 16, seed=table)
    assert info["tables_used"] == [table]
    # (a short, steeply coloured signal: the interior that is compared can be a little cleaner than the whole spectrum)
    assert -1.5 < snr - info["snr_spectrum_db"] < 4.0, (snr, info["snr_spectrum_db"])


@pytest.mark.parametrize("rate", [44100, 48000, 22050, 16000])
def test_mixed_blocks(tmp_path, rate):
    """Window-switching granules with the mixed flag: the two lowest subbands keep the normal long window through
    start, short and stop granules (and long scale-factor bands in the short one), the other thirty switch."""
    rng = np.random.default_rng(rate + 1)
    spf = 576 if rate < 32000 else 1152
    plan = [0, 0, 1, 2, 2, 3, 0, 1, 2, 3]
    for lo in (0, 15, 30, 300):
        pcm = _band_noise(rng, spf * 14, rate, lo, 45)
        snr, info = _roundtrip(tmp_path, pcm, rate, peak_is=3000, block_types=plan, mixed=True, seed=lo)
        assert info["snr_spectrum_db"] > 35 and snr > info["snr_spectrum_db"] - 1.5, (lo, snr, info["snr_spectrum_db"])
    pcm = _pink_noise(rng, spf * 24, 2)
    snr, info = _roundtrip(tmp_path, pcm, rate, ms=True, crc=True, scalefactors=True, reservoir=True, mixed=True,
                           peak_is=[2, 5, 15, 40], seed=5, block_types=plan)
    assert abs(snr - info["snr_spectrum_db"]) < 1.5, (snr, info["snr_spectrum_db"])


@pytest.mark.parametrize("rate,pos,ms,sel,plan", [(44100, p, m, 0, None) for p in range(7) for m in (False, True)]
                         + [(22050, p, False, sel, None) for p in (0, 1, 2, 3, 6) for sel in (0, 1)]
                         + [(48000, 2, True, 0, "switching"), (24000, 3, False, 1, "switching"),
                            (44100, 4, False, 0, "mixed")])
def test_intensity_stereo(tmp_path, rate, pos, ms, sel, plan):
    """Above a bound the right channel carries no spectrum, only a position per band; both channels are the left
    channel's spectrum times the two factors the position stands for.  Input: left = A + C, right = B with C above
    the bound; expected output: left = A + kl C, right = B + kr C."""
    rng = np.random.default_rng(rate + pos)
    spf = 576 if rate < 32000 else 1152
    n = spf * 14
    a, b = _band_noise(rng, n, rate, 100, 50)[:, 0], _band_noise(rng, n, rate, 100, 50)[:, 0]
    c = _band_noise(rng, n, rate, 250, 50)[:, 0]
    blocks = None if plan is None else [0, 0, 1, 2, 2, 3, 0, 1, 2, 3]
    data, info = mt.encode(np.stack([a + c, b], axis=1), rate, ms=ms, peak_is=600, seed=pos, block_types=blocks,
                           mixed=plan == "mixed", intensity={"bound": 200, "pos": pos, "scale_sel": sel})
    p = tmp_path / "i.mp3"
    p.write_bytes(data)
    y, _, diag = audio.read_mp3(p, diagnostics=True)
    assert list(diag[:4]) == [info["frames"], 0, 0, 0]
    kl, kr = mt.intensity_gains(pos, rate < 32000, sel)
    lo, hi = 2 * spf, n - 2 * spf - DELAY
    for ch, want in enumerate((a + kl * c, b + kr * c)):
        got = y[lo + DELAY: hi + DELAY, ch].astype(np.float64)
        snr = 10 * np.log10((want[lo:hi] ** 2).sum() / ((want[lo:hi] - got) ** 2).sum())
        assert snr > 25, (rate, pos, ms, ch, snr)


def test_damage_is_contained(tmp_path):
    rng = np.random.default_rng(3)
    pcm = _pink_noise(rng, 1152 * 16, 2)
    data, info = mt.encode(pcm, 44100, crc=True, peak_is=15, seed=3)
    fb = info["frame_bytes"]
    clean_path = tmp_path / "c.mp3"
    clean_path.write_bytes(data)
    clean, _ = audio.read_mp3(clean_path)
    # a flipped bit in frame 5's side information: the header CRC catches it, that frame is silence, the rest is intact
    bad = bytearray(data)
    bad[5 * fb + 8] ^= 0x10
    p = tmp_path / "b.mp3"
    p.write_bytes(bytes(bad))
    y, _, diag = audio.read_mp3(p, diagnostics=True)
    assert diag[3] == 1 and y.shape == clean.shape
    assert np.array_equal(y[: 5 * 1152 - 600], clean[: 5 * 1152 - 600]) and np.array_equal(y[8 * 1152:], clean[8 * 1152:])
    # junk between frames and an ID3v2 tag in front are stepped over
    junk = b"ID3\x04\x00\x00\x00\x00\x00\x0a" + b"\x00" * 10 + data[: 4 * fb] + b"\x01\x02\x03junk" * 5 + data[4 * fb:]
    p.write_bytes(junk)
    y, _, diag = audio.read_mp3(p, diagnostics=True)
    assert diag[0] == info["frames"] and diag[1] == 35 and np.array_equal(y, clean)
    # truncation: whole frames before the cut are delivered
    p.write_bytes(data[: 9 * fb + 100])
    y, _ = audio.read_mp3(p)
    assert y.shape[0] == 9 * 1152 and np.array_equal(y[: 8 * 1152], clean[: 8 * 1152])
    # padding or a foreign tag block in front of the first frame: the type is found by the frame scan, not by byte 0
    for lead in (bytes(700), b"APETAGEX" + bytes(range(256)) * 3):
        q = tmp_path / "lead.bin"
        q.write_bytes(lead + data)
        assert audio.is_mp3(q) and audio.audio_duration(q) == clean.shape[0] / 44100.0
        y, rate = audio.read_audio(q)
        assert rate == 44100 and np.array_equal(y, clean)
        assert audio.load_audio_16k(q).shape[0] == -(-clean.shape[0] * 160 // 441)
    # not an MP3 at all
    p.write_bytes(b"\xff\xfb" + bytes(200))
    with pytest.raises(audio.UnsupportedAudioError):
        audio.read_mp3(p)
    p.write_bytes(bytes(5000))
    assert not audio.is_mp3(p)
    with pytest.raises(audio.UnsupportedAudioError):
        audio.read_audio(p)


def test_mp3_inside_a_wave_container(tmp_path):
    """Some recorders wrap the stream in RIFF/WAVE with format tag 0x55: the data chunk goes to the same decoder."""
    import struct
    rng = np.random.default_rng(9)
    data, info = mt.encode(_pink_noise(rng, 1152 * 6, 2), 44100, peak_is=15, seed=9)
    plain = tmp_path / "p.mp3"
    plain.write_bytes(data)
    fmt = struct.pack("<HHIIHHH", 0x55, 2, 44100, 40000, 1, 0, 12) + bytes(12)
    wrapped = tmp_path / "w.wav"
    wrapped.write_bytes(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data)) + b"WAVE" + b"fmt " +
                        struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(data)) + data)
    a, ra = audio.read_audio(plain)
    b, rb = audio.read_audio(wrapped)
    assert ra == rb == 44100 and np.array_equal(a, b) and audio.audio_duration(wrapped) == a.shape[0] / 44100.0
    assert np.array_equal(audio.load_audio_16k(wrapped), audio.load_audio_16k(plain))


def test_mp3_through_transcribe(golden_dir, tiny_engine):
    """The reference's demo recording end to end on the CPU (oracle engine injected): 40 s at the default 30 s
    chunks -> two segments that tile the decoded duration."""
    from omnilingual_asr import Pipeline
    segs = Pipeline(engine=tiny_engine).transcribe(str(golden_dir / "reference_demo_40s.mp3"))
    n16 = -(-(1531 * 1152 - 576 - 1190) * 160 // 441)
    assert len(segs) == 2 and segs[0].start == 0.0 and segs[0].end == 30.0 and segs[1].start == 30.0
    assert segs[1].end == pytest.approx(n16 / 16000.0, abs=1e-6) and all(s.text for s in segs)


def test_low_bitrate_stream_with_a_deep_reservoir_decodes_the_same_from_any_start(tmp_path, monkeypatch):
    """8 kbit/s at 22.05 kHz (second standard, mono): a frame is 26 bytes of which 13 are main data, and
    main_data_begin reaches its maximum of 255 bytes, so a frame's data begins about twenty frames before it.  A
    worker thread or a sample range that starts inside the stream has to gather that much main data (and decode the
    two frames before its first one for the overlap and the synthesis history) before its first frame is right: the
    samples must not depend on the thread count or on where a range starts."""
    rng = np.random.default_rng(8)
    n = 576 * 400
    t = np.arange(n) / 22050.0
    pcm = (0.02 * np.sin(2 * np.pi * 300.0 * t) * (1.5 + np.sin(2 * np.pi * 1.3 * t)) + 0.0002 * rng.standard_normal(n))[:, None]
    data, info = mt.encode(pcm, 22050, kbps=8, reservoir=True, peak_is=[1, 1, 2], seed=3)
    assert info["frame_bytes"] == 26
    begins = [((data[26 * f + 4] << 8 | data[26 * f + 5]) >> 8) & 0xFF for f in range(info["frames"])]   # 8 bits in the LSF side info
    assert max(begins) == 255 and np.mean(np.array(begins[40:]) > 200) > 0.5                      # the reservoir is deep
    p = tmp_path / "low.mp3"
    p.write_bytes(data)
    monkeypatch.setenv("OASR_MP3_THREADS", "1")
    whole, rate = audio.read_mp3_mono(p)
    assert rate == 22050 and float(np.abs(whole).max()) > 0.005
    for th in ("2", "5", "9"):
        monkeypatch.setenv("OASR_MP3_THREADS", th)
        got, _ = audio.read_mp3_mono(p)
        assert np.array_equal(got, whole), th
    monkeypatch.setenv("OASR_MP3_THREADS", "1")
    with audio.Mp3Stream(p) as st:
        for first, count in ((576 * 100 + 7, 5000), (576 * 37, 576), (576 * 250 - 1, 20000), (0, 3000)):
            assert np.array_equal(st.read_mono(first, count), whole[first: first + count]), (first, count)
