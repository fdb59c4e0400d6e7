#!/usr/bin/env python3
"""Fuzz of the container readers (host code only): small valid files of every kind the package decodes - WAV (16- and
24-bit PCM, mu-law, IMA ADPCM), AIFF, FLAC (native and in Ogg), MP3, Ogg Vorbis - are damaged at random (header bytes overwritten, truncation, bit
flips anywhere, size fields replaced by extreme values) and loaded through `audio.load_audio_16k` under a 12 GB
address-space limit.  Every file must either decode or be refused with UnsupportedAudioError / OasrError; any other
exception, a MemoryError (a header obeyed instead of checked) or a load of more than 1.5 s is reported.
Usage: container_fuzz.py [files]"""
import collections, math, os, resource, struct, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
sys.path.insert(0, str(ROOT / "tests"))
resource.setrlimit(resource.RLIMIT_AS, (12 << 30, 12 << 30))
import numpy as np
from omnilingual_asr import _hip, audio

files = int(sys.argv[1]) if len(sys.argv) > 1 else 2800
rng = np.random.default_rng(1)


def wav(tag, bits, ch, rate, payload, block=None, extra=b""):
    block = block or ch * bits // 8
    fmt = struct.pack("<HHIIHH", tag, ch, rate, rate * block, block, bits) + extra
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def aiff(ch, bits, rate, raw):
    e = 16383 + int(math.log2(rate))
    m = int(rate * 2 ** (63 - int(math.log2(rate))))
    comm = struct.pack(">hIh", ch, len(raw) // (ch * bits // 8), bits) + struct.pack(">HQ", e, m)
    body = b"AIFF" + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">III", len(raw) + 8, 0, 0) + raw
    return b"FORM" + struct.pack(">I", len(body)) + body


gold = ROOT / "tests" / "golden"
seeds = {
    "pcm16.wav": wav(1, 16, 2, 16000, rng.integers(-3000, 3000, 4000, dtype=np.int16).tobytes()),
    "pcm24.wav": wav(1, 24, 1, 44100, bytes(rng.integers(0, 256, 3000, dtype=np.uint8))),
    "ulaw.wav": wav(7, 8, 1, 8000, bytes(rng.integers(0, 256, 3000, dtype=np.uint8))),
    "ima.wav": wav(0x11, 4, 1, 8000, bytes(rng.integers(0, 256, 256 * 6, dtype=np.uint8)), block=256,
                   extra=struct.pack("<HH", 2, 505)),
    "x.aiff": aiff(2, 16, 22050, rng.integers(-3000, 3000, 4000, dtype=np.int16).astype(">i2").tobytes()),
    "x.flac": (gold / "reference_upload.flac").read_bytes(),
    "x.mp3": (gold / "reference_demo_40s.mp3").read_bytes()[:40000],
    "x.ogg": (gold / "mathjax_invalid_keypress.ogg").read_bytes(),
}
import flac_tools  # noqa: E402  (tests/: the Ogg wrapper for FLAC streams)
seeds["flac_in.ogg"] = flac_tools.wrap_in_ogg(seeds["x.flac"], comment=b"A=B")
for name, blob in seeds.items():          # every seed decodes as it is (and the library is loaded before timing starts)
    p = os.path.join(tempfile.gettempdir(), "oasr_fz_" + name)
    open(p, "wb").write(blob)
    assert audio.load_audio_16k(p).shape[0] > 0, name
kinds, slow, t0 = collections.Counter(), [], time.time()
for it in range(files):
    name = list(seeds)[it % len(seeds)]
    b = bytearray(seeds[name])
    k = int(rng.integers(4))
    if k == 0:
        for _ in range(int(rng.integers(1, 8))):
            b[int(rng.integers(0, min(len(b), 80)))] = int(rng.integers(256))
    elif k == 1:
        b = b[: int(rng.integers(1, len(b)))]
    elif k == 2:
        for _ in range(int(rng.integers(1, 20))):
            b[int(rng.integers(len(b)))] ^= 1 << int(rng.integers(8))
    else:
        i = int(rng.integers(0, len(b) - 8))
        b[i:i + 4] = struct.pack("<I", int(rng.choice([0, 1, 0xFFFFFFFF, 0x7FFFFFFF, len(b) * 2])))
    p = os.path.join(tempfile.gettempdir(), "oasr_fz_" + name)
    open(p, "wb").write(bytes(b))
    t1 = time.time()
    try:
        audio.load_audio_16k(p)
        kinds["decoded"] += 1
        try:
            audio.audio_duration(p)
        except audio.UnsupportedAudioError:
            kinds["decoded, duration probe refused"] += 1
    except (audio.UnsupportedAudioError, _hip.OasrError):
        kinds["refused"] += 1
    except Exception as e:  # noqa: BLE001 - the point of the run
        kinds["%s (%s)" % (type(e).__name__, name)] += 1
        print(type(e).__name__, name, "damage", k, str(e)[:90], bytes(b[:60]).hex(), flush=True)
    if time.time() - t1 > 1.5:
        slow.append((name, k, round(time.time() - t1, 1), bytes(b[:48]).hex()))
for name in seeds:
    os.remove(os.path.join(tempfile.gettempdir(), "oasr_fz_" + name))
print(dict(kinds), "in %.1f s; slow loads: %s" % (time.time() - t0, slow[:6]))
sys.exit(0 if set(kinds) <= {"decoded", "refused", "decoded, duration probe refused"} and not slow else 1)
