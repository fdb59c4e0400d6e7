#!/usr/bin/env python3
"""What a caller of the public API sees: `Pipeline.transcribe_with_retry(path)` on a long file written to local
disk, end to end - file read, decode, chunk plan, device path, CTC stage, segments - with the time spent in audio
loading shown separately.
Usage: transcribe_e2e.py [hours [rate [channels [width]]]]   WAV of synthetic noise + tone (width 2 or 3 bytes)
       transcribe_e2e.py flac [repeats]               FLAC made of the reference upload's frames (tests/flac_tools.py)
       transcribe_e2e.py rf64 [hours]                 RF64 WAV beyond 4 GB: 48 kHz 16-bit stereo (9.5 h = 6.6 GB), one minute
                                                      of noise + tone repeated
       transcribe_e2e.py mp3 [repeats]                MP3 made of the audio frames of the reference's demo recording
                                                      (40 s at 44.1 kHz joint stereo per repeat; the frame at each
                                                      join finds another bit reservoir than it was written for)
       transcribe_e2e.py ogg [repeats]                Ogg Vorbis from the test writer (tests/vorbis_tools.py): 4 s of 44.1 kHz
                                                      coupled stereo per repeat (block sizes 256 / 2048), ~ 300 kbit/s"""
import os, sys, tempfile, time, wave
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
import numpy as np
from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline


def write_wav(hours: float, rate: int, channels: int, width: int = 2) -> str:
    n = int(hours * 3600 * rate)
    rng = np.random.default_rng(0)
    t = np.arange(n, dtype=np.float32)
    pcm = (3000 * rng.standard_normal(n).astype(np.float32) + 6000 * np.sin(t * (2 * np.pi * 440 / rate))).astype(np.int16)
    path = os.path.join(tempfile.gettempdir(), "oasr_e2e.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(channels); w.setsampwidth(width); w.setframerate(rate)
        pcm = pcm if channels == 1 else np.repeat(pcm, channels)
        if width == 3:      # 24-bit: the 16-bit signal in the upper bytes
            b = np.zeros((pcm.shape[0], 3), dtype=np.uint8)
            b[:, 1:] = pcm.astype("<i2").view(np.uint8).reshape(-1, 2)
            w.writeframes(b.tobytes())
        else:
            w.writeframes(pcm.tobytes())
    return path


def write_flac(repeats: int):
    sys.path.insert(0, str(ROOT / "tests"))
    import flac_tools as ft
    from omnilingual_asr import audio
    src = ROOT / "tests" / "golden" / "reference_upload.flac"
    x, rate = audio.read_flac(src)
    stream, want = ft.repeat_stream(src.read_bytes(), repeats, np.round(x * 32768).astype(np.int32))
    path = os.path.join(tempfile.gettempdir(), "oasr_e2e.flac")
    with open(path, "wb") as f:
        f.write(stream)
    return path, want.shape[0] / rate / 3600.0


def write_mp3(repeats: int):
    from omnilingual_asr import audio
    d = (ROOT / "tests" / "golden" / "reference_demo_40s.mp3").read_bytes()
    head = 10 + ((d[6] & 127) << 21 | (d[7] & 127) << 14 | (d[8] & 127) << 7 | (d[9] & 127))
    body = d[head + 208:]                       # past the ID3v2 tag and the 208-byte Info frame
    path = os.path.join(tempfile.gettempdir(), "oasr_e2e.mp3")
    with open(path, "wb") as f:
        f.write(d)
        for _ in range(repeats - 1):
            f.write(body)
    rate, _, total = audio.mp3_info(path)
    return path, total / rate / 3600.0


def write_ogg(repeats: int):
    sys.path.insert(0, str(ROOT / "tests"))
    import vorbis_tools as vt
    from omnilingual_asr import audio
    rate = 44100
    rng = np.random.default_rng(3)
    n = rate * 4
    t = np.arange(n) / rate
    x = (0.2 * np.sin(2 * np.pi * 170.0 * t) * (np.sin(2 * np.pi * 1.3 * t) > 0))[:, None] + 0.05 * rng.standard_normal((n, 2))
    enc = vt.encode(x, vt.Config(channels=2, rate=rate, pattern="LLLLLLLS", seed=3, residue_type=2, coupled=True))
    path = os.path.join(tempfile.gettempdir(), "oasr_e2e.ogg")
    with open(path, "wb") as f:
        f.write(vt.repeat_stream(enc, repeats))
    rate, _, total = audio.vorbis_info(path)
    return path, total / rate / 3600.0


def write_rf64(hours: float) -> str:
    import struct
    rate, channels = 48000, 2
    minutes = int(round(hours * 60))
    rng = np.random.default_rng(0)
    t = np.arange(rate * 60, dtype=np.float32)
    mono = (3000 * rng.standard_normal(t.size).astype(np.float32) + 6000 * np.sin(t * (2 * np.pi * 440 / rate))).astype("<i2")
    block = np.repeat(mono, channels).tobytes()
    data_bytes = len(block) * minutes
    path = os.path.join(tempfile.gettempdir(), "oasr_e2e_rf64.wav")
    with open(path, "wb") as f:
        f.write(b"RF64" + struct.pack("<I", 0xFFFFFFFF) + b"WAVE")
        f.write(b"ds64" + struct.pack("<IQQQI", 28, 72 + data_bytes, data_bytes, data_bytes // (2 * channels), 0))
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * channels * 2, channels * 2, 16))
        f.write(b"data" + struct.pack("<I", 0xFFFFFFFF))
        for _ in range(minutes):
            f.write(block)
    return path


if len(sys.argv) > 1 and sys.argv[1] == "rf64":
    hours = float(sys.argv[2]) if len(sys.argv) > 2 else 9.5
    path = write_rf64(hours)
elif len(sys.argv) > 1 and sys.argv[1] == "mp3":
    path, hours = write_mp3(int(sys.argv[2]) if len(sys.argv) > 2 else 180)
elif len(sys.argv) > 1 and sys.argv[1] == "ogg":
    path, hours = write_ogg(int(sys.argv[2]) if len(sys.argv) > 2 else 1800)
elif len(sys.argv) > 1 and sys.argv[1] == "flac":
    path, hours = write_flac(int(sys.argv[2]) if len(sys.argv) > 2 else 400)
else:
    hours = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    path = write_wav(hours, int(sys.argv[2]) if len(sys.argv) > 2 else 16000, int(sys.argv[3]) if len(sys.argv) > 3 else 1,
                     int(sys.argv[4]) if len(sys.argv) > 4 else 2)
print("%s: %.1f MB" % (path, os.path.getsize(path) / 1e6))
pipe = GeminiASRPipeline(model="G300")
load_s = [0.0]
orig = pipe._load_audio


def timed_load(p):
    t0 = time.perf_counter()
    out = orig(p)
    load_s[0] += time.perf_counter() - t0
    return out


pipe._load_audio = timed_load
for rep in range(3):
    load_s[0] = 0.0
    t0 = time.perf_counter()
    res = pipe.transcribe_with_retry(path)
    dt = time.perf_counter() - t0
    print("run %d: %.2f h -> %.3f s wall (audio loading %.3f s) = RTF %.0fx; %d segments" %
          (rep, hours, dt, load_s[0], hours * 3600 / dt, len(res.segments)), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "rf64":
    # the signal repeats every minute: a chunk's text must equal that of the chunk one minute earlier, also where the
    # file's byte offsets no longer fit 32 bits (past 6.2 h)
    texts = [seg.text for seg in res.segments]
    late = [i for i, seg in enumerate(res.segments) if seg.start * 48000 * 4 > 2 ** 32]
    # (the first and the last chunk see the silence outside the file through the rate converter's filter)
    differ = [i for i in range(3, len(texts) - 1) if texts[i] != texts[i - 2]]
    print("texts repeat with the one-minute period over chunks 1 .. %d (%d of them past the 4 GiB offset): %s %s; "
          "distinct non-empty texts among them: %d" % (len(texts) - 2, len(late), not differ, differ[:10],
                                                        len(set(texts[1:-1]))))
    if differ or not all(texts):
        sys.exit(1)
os.remove(path)
