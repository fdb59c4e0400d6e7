"""Test infrastructure (not product code): cuts the MP3 fixtures out of the reference's own files.

Reads  /root/reference/Samuel Speaking Lisu.mp3   (the demo recording the reference ships; the same bytes as
       /root/reference/public/codeswitching-demo.mp3) and /root/reference/samuel_lisu.eaf (the reference's own
       annotation of that recording: speech intervals of tier SPEAKER_00 in milliseconds).
Writes tests/golden/reference_demo_40s.mp3     the ID3v2 tag, the Info frame and the first 1531 audio frames (40 s),
                                               byte for byte as the file holds them
       tests/golden/reference_demo_mp3.json    the annotated intervals that end inside those 40 s, and the figures
                                               the whole file gives (frame count, samples, bit accounting) as
                                               decoded in the build container.
Only data is copied (encoded audio and time stamps), no source text.  Run once in the build container:
    python oracle/gen_mp3_fixture.py
"""
import json
import re
import struct
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
KEEP_FRAMES = 1531


def frame_offsets(d: bytes):
    pos = 0
    if d[:3] == b"ID3":
        pos = 10 + ((d[6] & 127) << 21 | (d[7] & 127) << 14 | (d[8] & 127) << 7 | (d[9] & 127))
    kbps = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]
    out = []
    while pos + 4 <= len(d):
        h = struct.unpack(">I", d[pos:pos + 4])[0]
        assert h >> 21 == 0x7FF and (h >> 19) & 3 == 3 and (h >> 17) & 3 == 1, "MPEG-1 Layer III expected"
        out.append(pos)
        pos += 144000 * kbps[(h >> 12) & 15] // 44100 + ((h >> 9) & 1)
    return out


def main():
    d = (REF / "Samuel Speaking Lisu.mp3").read_bytes()
    assert d == (REF / "public" / "codeswitching-demo.mp3").read_bytes()
    offs = frame_offsets(d)
    cut = offs[1 + KEEP_FRAMES]          # offs[0] is the Info frame
    (ROOT / "tests" / "golden" / "reference_demo_40s.mp3").write_bytes(d[:cut])

    eaf = (REF / "samuel_lisu.eaf").read_text(encoding="utf-8")
    slot = {m.group(1): int(m.group(2)) for m in re.finditer(r'TIME_SLOT_ID="(\w+)" TIME_VALUE="(\d+)"', eaf)}
    tier = eaf[eaf.index('TIER_ID="SPEAKER_00"'):eaf.index('TIER_ID="SPEAKER_00_words"')]
    segs = [(slot[a], slot[b]) for a, b in re.findall(r'TIME_SLOT_REF1="(\w+)" TIME_SLOT_REF2="(\w+)"', tier)]
    limit_ms = (KEEP_FRAMES * 1152 - 576 - 1190) * 1000 // 44100
    info = {
        "source": "ref:Samuel Speaking Lisu.mp3 (= ref:public/codeswitching-demo.mp3), ref:samuel_lisu.eaf tier SPEAKER_00",
        "fixture_audio_frames": KEEP_FRAMES,
        "speech_intervals_ms": [s for s in segs if s[1] <= limit_ms],
        "whole_file": {"audio_frames": len(offs) - 1, "annotated_end_ms": max(b for _, b in segs)},
    }
    sys.path.insert(0, str(ROOT / "omnilingual-asr_amd"))
    from omnilingual_asr import audio
    x, rate, diag = audio.read_mp3(REF / "Samuel Speaking Lisu.mp3", diagnostics=True)
    info["whole_file"].update(sample_rate=rate, channels=int(x.shape[1]), samples_per_channel=int(x.shape[0]),
                              diagnostics=[int(v) for v in diag])
    (ROOT / "tests" / "golden" / "reference_demo_mp3.json").write_text(json.dumps(info, indent=1) + "\n")
    print(json.dumps(info["whole_file"]), "fixture bytes", cut)


if __name__ == "__main__":
    main()
