#!/usr/bin/env python3
"""Soak run of the public call: many files of random length, rate, width and channel count (integer PCM WAVs, every
fifth file an MP3 from the test encoder: mono / stereo / mid-side, 8 - 48 kHz, and every seventh an Ogg Vorbis file
from the test writer: mono / coupled stereo, both block sizes) in sequence through one
pipeline (graph capture per shape bucket, buffer eviction, device-resident and host-resident sample paths), each
checked against a second pipeline that runs eager launches on freshly decoded host samples.  Usage: soak.py [files]"""
import struct, sys, tempfile, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
from omnilingual_asr import audio
from omnilingual_asr.engine_hip import HipEngine
from omnilingual_asr.models.inference.asr_pipeline import GeminiASRPipeline, plan_chunks

files = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(2024)
pipe = GeminiASRPipeline(model="GTINY", batch_size=4)
ref = GeminiASRPipeline(model="GTINY", batch_size=4, engine=HipEngine("GTINY", use_graphs=False))
path = os.path.join(tempfile.gettempdir(), "oasr_soak.wav")
bad = chars = 0
for k in range(files):
    rate = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
    channels = int(rng.choice([1, 1, 2]))
    width = int(rng.choice([2, 2, 2, 3, 1]))
    seconds = float(rng.choice([0.02, 0.3, 1.0, 7.7, 29.99, 30.0, 30.01, 61.5, 95.0])) * float(rng.uniform(0.9, 1.1))
    n = max(1, int(seconds * rate)) * channels
    if k % 5 == 4:          # an MP3: at least one frame, at most ~8 s (the test encoder is a numpy program)
        import mp3_tools
        rate = int(rng.choice([8000, 16000, 22050, 32000, 44100, 48000]))
        spf = 576 if rate < 32000 else 1152
        frames = max(1, min(int(seconds * rate) // spf, 8 * rate // spf))
        pcm = 0.2 * rng.standard_normal((frames * spf, channels))
        pcm = np.cumsum(pcm, axis=0) * 0.05 + pcm * 0.2           # coloured, so that the granules fit their budget
        data, _ = mp3_tools.encode(pcm / max(1.0, np.abs(pcm).max()), rate, ms=channels == 2 and bool(rng.integers(2)),
                                   peak_is=[3, 7, 15], seed=k, reservoir=True,
                                   block_types=[0, 0, 1, 2, 3, 0] if rng.integers(2) else None)
        with open(path, "wb") as f:
            f.write(data)
        width, seconds = 0, frames * spf / rate
    elif k % 7 == 6:        # an Ogg Vorbis file, a few seconds repeated up to ~60 s
        import vorbis_tools
        base = min(int(seconds * rate), 2 * rate)
        pcm = 0.15 * rng.standard_normal((max(base, 64), channels))
        enc = vorbis_tools.encode(pcm if channels > 1 else pcm[:, 0], vorbis_tools.Config(
            channels=channels, rate=rate, pattern=str(rng.choice(["L", "LLSL", "SSL"])), seed=k, trim_end=True,
            residue_type=2 if channels == 2 else int(rng.integers(2)), coupled=channels == 2,
            book_kind=str(rng.choice(["lattice", "table"]))))
        times = max(1, min(30, int(seconds * rate) // max(base, 64)))
        with open(path, "wb") as f:
            f.write(vorbis_tools.repeat_stream(enc, times) if times > 1 else enc.data)
        width, seconds = 0, audio.audio_duration(path)
    elif width == 1:
        payload = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    elif width == 2:
        payload = (rng.standard_normal(n) * 4000).astype("<i2").tobytes()
    else:
        payload = (rng.standard_normal(n) * 1e6).astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    if width:
        fmt = struct.pack("<HHIIHH", 1, channels, rate, rate * channels * width, channels * width, 8 * width)
        with open(path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVEfmt " + struct.pack("<I", 16) + fmt +
                    b"data" + struct.pack("<I", len(payload)) + payload)
    got = pipe.transcribe_with_retry(path)
    host = audio.load_audio_16k(path, lambda mono, r: ref.engine.resample_16k(mono, r).cpu().numpy())
    bounds = plan_chunks(host.shape[0], ref.chunk_seconds, ref.min_chunking_seconds)
    recs = ref._encode_chunks(host, bounds)
    want = [r.to_result().segments[0].text for r in recs]
    have = [s.text for s in got.segments]
    same = len(want) == len(have) and all(a == b for a, b in zip(want, have))
    chars += sum(len(t) for t in have)
    if not same:
        bad += 1
        print("MISMATCH file %d: rate %d ch %d width %d %.2f s: %d vs %d segments" % (k, rate, channels, width, seconds, len(have), len(want)), flush=True)
    if k % 20 == 19:
        print("%d files, %d mismatches, device memory %.2f GB" % (k + 1, bad, torch.cuda.memory_allocated() / 1e9), flush=True)
os.remove(path)
print("soak: %d files, %d mismatches, %d characters of text compared" % (files, bad, chars))
sys.exit(1 if bad or chars == 0 else 0)
