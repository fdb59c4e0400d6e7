"""Test infrastructure (not product code): the annotation data of ref:samuel_lisu.eaf (the reference's own annotation of
its demo recording: 36 speech intervals of tier SPEAKER_00 and 314 words of tier SPEAKER_00_words, text in several
scripts) as JSON for the exporter tests.  Only data is copied (times in milliseconds and annotation values).
Run once in the build container:  python oracle/gen_eaf_fixture.py"""
import json
import xml.etree.ElementTree as ET
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
doc = ET.parse("/root/reference/samuel_lisu.eaf").getroot()
slot = {t.get("TIME_SLOT_ID"): int(t.get("TIME_VALUE")) for t in doc.iter("TIME_SLOT")}


def tier(name):
    out = []
    for t in doc.iter("TIER"):
        if t.get("TIER_ID") == name:
            for a in t.iter("ALIGNABLE_ANNOTATION"):
                out.append((slot[a.get("TIME_SLOT_REF1")], slot[a.get("TIME_SLOT_REF2")], a.findtext("ANNOTATION_VALUE") or ""))
    return out


segments = []
words = tier("SPEAKER_00_words")
for a, b, text in tier("SPEAKER_00"):
    inside = [{"word": w, "start_ms": wa, "end_ms": wb} for wa, wb, w in words if wa >= a and wb <= b]
    segments.append({"start_ms": a, "end_ms": b, "speaker": "SPEAKER_00", "text": text, "words": inside})
out = ROOT / "tests" / "golden" / "reference_lisu_annotation.json"
out.write_text(json.dumps({"source": "ref:samuel_lisu.eaf", "segments": segments}, ensure_ascii=False, indent=0) + "\n",
               encoding="utf-8")
print(len(segments), "segments,", sum(len(s["words"]) for s in segments), "words ->", out, out.stat().st_size, "bytes")
