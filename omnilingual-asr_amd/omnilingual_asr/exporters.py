"""Server-side exporters: ELAN (.eaf), Praat TextGrid, SubRip, plain text, JSON.

The reference builds these files in the browser (ref:public/static/app.js:1381-1576 buildEAF,
:1582-1689 buildTextGrid, :1694-1736 buildPlainText, :1741-1771 buildSRT, :1364 JSON).  These are
restatements of those formats so that the product is usable without the web client.  Parity is
unpinned: the reference's functions could not be executed here to produce golden files, so the
tests hold hand-derived expectations.

`data` is the mapping the reference's server returns for a transcription:
`{"audio_url", "summary", "detected_languages", "segments": [{"start", "end", "speaker", "text",
"language", "language_code", "emotion", "translation", "words": [{"word", "start", "end"}]}]}`;
`result_to_data` builds it from the pipeline's records.
"""

from __future__ import annotations

import dataclasses
import datetime as _dt
import json
import math
from typing import Any, Dict, Iterable, List, Optional


def result_to_data(segments: Iterable[Any], *, audio_url: str = "", summary: Optional[str] = None,
                   detected_languages: Optional[List[dict]] = None) -> Dict[str, Any]:
    """Dataclass segments (engine-level or public) -> the exporters' input mapping."""
    segs = []
    for s in segments:
        d = dataclasses.asdict(s) if dataclasses.is_dataclass(s) else dict(s)
        if d.get("words") is None:
            d["words"] = []      # the reference's server sends an empty list (ref workflows/wav2elan_web/app.py:127-130)
        segs.append(d)
    return {"audio_url": audio_url, "summary": summary, "detected_languages": detected_languages, "segments": segs}


def _js_round(x: float) -> int:
    """Math.round: halves go up."""
    return int(math.floor(x + 0.5))


def _escape_xml(s: str) -> str:
    return s.replace("&", "&amp;").replace("<", "&lt;").replace(">", "&gt;").replace('"', "&quot;")


def _valid_translation(seg: dict) -> bool:
    t = seg.get("translation")
    return bool(t) and t != "null" and t != seg.get("text") and t.strip() != ""


def _ordered_unique(items: Iterable[Any]) -> List[Any]:
    seen, out = set(), []
    for it in items:
        if it not in seen:
            seen.add(it)
            out.append(it)
    return out


def build_eaf(data: dict, now: Optional[_dt.datetime] = None) -> str:
    """ELAN annotation document: one transcription tier per speaker plus `<speaker>_language`,
    `_emotion`, `_translation` tiers when any segment carries those fields; times in integer
    milliseconds, two time slots per segment (ref app.js:1381-1576)."""
    segs = data["segments"]
    ts_id = ann_id = 1
    slots: List[tuple] = []
    anns = {"transcript": [], "language": [], "emotion": [], "translation": []}
    has_language = any(s.get("language") for s in segs)
    has_emotion = any(s.get("emotion") for s in segs)
    has_translation = any(s.get("translation") and s["translation"] != "null" and s["translation"].strip() != ""
                          for s in segs)
    for s in segs:
        ts1, ts2 = "ts%d" % ts_id, "ts%d" % (ts_id + 1)
        ts_id += 2
        slots.append((ts1, _js_round(s["start"] * 1000)))
        slots.append((ts2, _js_round(s["end"] * 1000)))
        sp = s["speaker"]
        anns["transcript"].append((sp, ts1, ts2, s["text"], "a%d" % ann_id))
        ann_id += 1
        if s.get("language"):
            anns["language"].append((sp + "_language", ts1, ts2, s.get("language_code") or s["language"], "a%d" % ann_id))
            ann_id += 1
        if s.get("emotion"):
            anns["emotion"].append((sp + "_emotion", ts1, ts2, s["emotion"], "a%d" % ann_id))
            ann_id += 1
        if _valid_translation(s):
            anns["translation"].append((sp + "_translation", ts1, ts2, s["translation"], "a%d" % ann_id))
            ann_id += 1
    speakers = _ordered_unique(s["speaker"] for s in segs)

    def ann_xml(a):
        return ("            <ANNOTATION>\n"
                '                <ALIGNABLE_ANNOTATION ANNOTATION_ID="%s" TIME_SLOT_REF1="%s" TIME_SLOT_REF2="%s">\n'
                "                    <ANNOTATION_VALUE>%s</ANNOTATION_VALUE>\n"
                "                </ALIGNABLE_ANNOTATION>\n"
                "            </ANNOTATION>") % (a[4], a[1], a[2], _escape_xml(a[3]))

    def tier_xml(kind, type_ref, tier_id, drop_empty):
        body = "\n".join(ann_xml(a) for a in anns[kind] if a[0] == tier_id)
        if drop_empty and not body:
            return ""
        return '        <TIER LINGUISTIC_TYPE_REF="%s" TIER_ID="%s">\n%s\n        </TIER>' % (type_ref, tier_id, body)

    ts_xml = "\n".join('        <TIME_SLOT TIME_SLOT_ID="%s" TIME_VALUE="%d"/>' % t for t in slots)
    transcript_xml = "\n".join(tier_xml("transcript", "transcription", sp, False) for sp in speakers)
    extra = ""
    if has_language:
        extra += "\n".join(x for x in (tier_xml("language", "language", sp + "_language", True) for sp in speakers) if x) + "\n"
    if has_emotion:
        extra += "\n".join(x for x in (tier_xml("emotion", "emotion", sp + "_emotion", True) for sp in speakers) if x) + "\n"
    if has_translation:
        extra += "\n".join(x for x in (tier_xml("translation", "translation", sp + "_translation", True)
                                       for sp in speakers) if x)
    types = '    <LINGUISTIC_TYPE LINGUISTIC_TYPE_ID="transcription" TIME_ALIGNABLE="true"/>'
    for flag, name in ((has_language, "language"), (has_emotion, "emotion"), (has_translation, "translation")):
        if flag:
            types += '\n    <LINGUISTIC_TYPE LINGUISTIC_TYPE_ID="%s" TIME_ALIGNABLE="true"/>' % name
    if now is None:
        now = _dt.datetime.now(_dt.timezone.utc)
    stamp = now.astimezone(_dt.timezone.utc).strftime("%Y-%m-%dT%H:%M:%S.") + "%03dZ" % (now.microsecond // 1000)
    return ('<?xml version="1.0" encoding="UTF-8"?>\n'
            '<ANNOTATION_DOCUMENT AUTHOR="OmniTranscribe" DATE="%s" FORMAT="3.0" VERSION="3.0" '
            'xmlns:xsi="http://www.w3.org/2001/XMLSchema-instance" '
            'xsi:noNamespaceSchemaLocation="http://www.mpi.nl/tools/elan/EAFv3.0.xsd">\n'
            '    <HEADER MEDIA_FILE="" TIME_UNITS="milliseconds">\n'
            '        <MEDIA_DESCRIPTOR MEDIA_URL="%s" MIME_TYPE="audio/x-wav"/>\n'
            "    </HEADER>\n"
            "    <TIME_ORDER>\n%s\n    </TIME_ORDER>\n%s\n%s\n%s\n</ANNOTATION_DOCUMENT>") % (
        stamp, _escape_xml(data.get("audio_url") or ""), ts_xml, transcript_xml, extra, types)


def build_textgrid(data: dict) -> str:
    """Praat long-format TextGrid: one interval tier per speaker; a segment's words become intervals
    when it has any, else the segment itself; gaps wider than 1 ms are filled with empty intervals;
    times with 6 decimals (ref app.js:1582-1689)."""
    segs = data["segments"]
    max_time = 0.0
    for s in segs:
        if s["end"] > max_time:
            max_time = s["end"]
    speakers = _ordered_unique(s["speaker"] for s in segs)
    tiers: Dict[Any, list] = {sp: [] for sp in speakers}
    for s in segs:
        words = s.get("words")
        if words:
            for w in words:
                tiers[s["speaker"]].append((w["start"], w["end"], w["word"]))
        else:
            tiers[s["speaker"]].append((s["start"], s["end"], s["text"]))
    fmt = lambda t: "%.6f" % t
    esc = lambda t: t.replace('"', '""')
    out = ('File type = "ooTextFile"\nObject class = "TextGrid"\n\nxmin = 0 \nxmax = %s\n\n'
           "tiers? <exists> \nsize = %d\nitem []:\n") % (fmt(max_time), len(speakers))
    for idx, sp in enumerate(speakers):
        filled, last_end = [], 0.0
        for iv in sorted(tiers[sp], key=lambda iv: iv[0]):      # stable, as Array.prototype.sort is
            if iv[0] > last_end + 0.001:
                filled.append((last_end, iv[0], ""))
            filled.append(iv)
            last_end = iv[1]
        if last_end < max_time - 0.001:
            filled.append((last_end, max_time, ""))
        out += ('    item [%d]:\n        class = "IntervalTier" \n        name = "%s"\n        xmin = 0 \n'
                "        xmax = %s\n        intervals: size = %d\n") % (idx + 1, esc(sp), fmt(max_time), len(filled))
        for k, iv in enumerate(filled):
            out += ("        intervals [%d]:\n            xmin = %s \n            xmax = %s\n"
                    '            text = "%s"\n') % (k + 1, fmt(iv[0]), fmt(iv[1]), esc(iv[2]))
    return out


def _clock(seconds: float) -> str:
    """M:SS (ref app.js:585-589)."""
    return "%d:%02d" % (math.floor(seconds / 60), math.floor(seconds % 60))


def build_plain_text(data: dict) -> str:
    """Readable transcript: optional summary and language list, then lines `M:SS text` grouped under
    `[speaker]` headings, translations on an indented arrow line (ref app.js:1694-1736)."""
    lines: List[str] = []
    if data.get("summary"):
        lines += ["=== Summary ===", data["summary"], ""]
    langs = data.get("detected_languages")
    if langs:
        lines.append("=== Detected Languages ===")
        lines += ["- %s (%s)" % (l.get("language"), l.get("code")) for l in langs]
        lines.append("")
    lines += ["=== Transcript ===", ""]
    current = None
    for s in data["segments"]:
        if s["speaker"] != current:
            if current is not None:
                lines.append("")
            lines.append("[%s]" % s["speaker"])
            current = s["speaker"]
        lines.append("%s %s" % (_clock(s["start"]), s["text"]))
        if _valid_translation(s):
            lines.append("         → %s" % s["translation"])
    return "\n".join(lines)


def _srt_time(seconds: float) -> str:
    """HH:MM:SS,mmm with the milliseconds rounded on their own (so x.9996 prints ",1000", as the
    reference does, ref app.js:1746-1752)."""
    h = math.floor(seconds / 3600)
    m = math.floor((seconds % 3600) / 60)
    s = math.floor(seconds % 60)
    ms = _js_round((seconds % 1) * 1000)
    return "%02d:%02d:%02d,%03d" % (h, m, s, ms)


def build_srt(data: dict) -> str:
    """SubRip: numbered cues, `[speaker]` prefix only when there is more than one speaker (ref app.js:1741-1771)."""
    segs = data["segments"]
    many = len({s["speaker"] for s in segs}) > 1
    lines: List[str] = []
    for i, s in enumerate(segs):
        lines += [str(i + 1), "%s --> %s" % (_srt_time(s["start"]), _srt_time(s["end"])),
                  ("[%s] %s" % (s["speaker"], s["text"])) if many else s["text"], ""]
    return "\n".join(lines)


def _js_number(x: float) -> str:
    """A double as JSON.stringify prints it: shortest round-trip digits, no trailing `.0`, exponents
    only below 1e-6 and from 1e21 up."""
    if x != x or x in (float("inf"), float("-inf")):
        return "null"
    if x == int(x) and abs(x) < 1e21:
        return str(int(x))
    r = repr(float(x))
    if "e" not in r and "E" not in r:
        return r
    mant, exp = r.lower().split("e")
    e = int(exp)
    if -6 <= e < 21:
        digits = mant.replace(".", "").lstrip("-")
        sign = "-" if x < 0 else ""
        point = (len(mant.lstrip("-").split(".")[0])) + e
        if point <= 0:
            return sign + "0." + "0" * (-point) + digits.rstrip("0")
        if point >= len(digits):
            return sign + digits + "0" * (point - len(digits))
        return sign + digits[:point] + "." + digits[point:].rstrip("0")
    return "%se%s%d" % (mant.rstrip("0").rstrip(".") if "." in mant else mant, "+" if e > 0 else "-", abs(e))


def build_json(data: dict) -> str:
    """`JSON.stringify(data, null, 2)` (ref app.js:1364): two-space indent, insertion order, numbers in
    JavaScript's notation."""

    def enc(v, depth):
        pad, pad_in = "  " * depth, "  " * (depth + 1)
        if v is None:
            return "null"
        if v is True:
            return "true"
        if v is False:
            return "false"
        if isinstance(v, (int, float)):
            return _js_number(float(v)) if isinstance(v, float) else str(v)
        if isinstance(v, str):
            return json.dumps(v, ensure_ascii=False)
        if isinstance(v, dict):
            if not v:
                return "{}"
            return "{\n" + ",\n".join(pad_in + json.dumps(str(k), ensure_ascii=False) + ": " + enc(x, depth + 1)
                                       for k, x in v.items()) + "\n" + pad + "}"
        if isinstance(v, (list, tuple)):
            if not v:
                return "[]"
            return "[\n" + ",\n".join(pad_in + enc(x, depth + 1) for x in v) + "\n" + pad + "]"
        if dataclasses.is_dataclass(v):
            return enc(dataclasses.asdict(v), depth)
        raise TypeError("cannot export %r" % type(v))

    return enc(data, 0)


EXPORTERS = {"eaf": build_eaf, "textgrid": build_textgrid, "txt": build_plain_text, "srt": build_srt,
             "json": build_json}
