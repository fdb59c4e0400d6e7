"""Oracle for the host-side record handling.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Plain-dict restatement of the scalar/float64 behaviour of
ref:src/omnilingual_asr/models/inference/gemini_pipeline.py — timestamp parsing (:73-103),
MIME lookup (:106-118), response parsing (:364-453), chunk offsetting (:541-575), chunk
merge (:643-678) and the retry schedule (:716-741).  Segments are dicts with the field
names of `GeminiTranscriptSegment` (ref :48-61) so they compare directly with
`dataclasses.asdict` of the reference's objects.
"""

from __future__ import annotations

import json
import math
import re

_UNIT_WEIGHTS = {2: (60.0, 1.0), 3: (3600.0, 60.0, 1.0)}

_MIME = {
    ".wav": "audio/wav", ".mp3": "audio/mp3", ".aiff": "audio/aiff", ".aac": "audio/aac",
    ".ogg": "audio/ogg", ".flac": "audio/flac", ".m4a": "audio/mp4",
}


def parse_timestamp(text) -> float:
    """ref gemini_pipeline.py:73-103.  'MM:SS' / 'HH:MM:SS' / raw seconds -> float seconds;
    empty, >=4 fields, or any unparsable field -> 0.0.  All fields but the last must be
    integers ('1.5:00' -> 0.0)."""
    if not text:
        return 0.0
    fields = text.strip().split(":")
    try:
        if len(fields) in _UNIT_WEIGHTS:
            w = _UNIT_WEIGHTS[len(fields)]
            vals = [int(f) for f in fields[:-1]] + [float(fields[-1])]
            acc = 0.0
            # same association order as the reference: ((h*3600 + m*60) + s)
            for v, k in zip(vals, w):
                acc = acc + v * k
            return acc
        return float(text)
    except ValueError:
        return 0.0


def get_mime_type(suffix: str) -> str:
    """ref gemini_pipeline.py:106-118 (unknown -> audio/wav)."""
    return _MIME.get(suffix.lower(), "audio/wav")


def parse_response(text: str) -> dict:
    """ref gemini_pipeline.py:364-453.  Returns {'summary', 'segments', 'detected_languages'}."""
    try:
        data = json.loads(text)
    except json.JSONDecodeError:
        m = re.search(r"```json\s*(.*?)\s*```", text, re.DOTALL)
        if not m:
            return {"summary": "Failed to parse transcription", "segments": [],
                    "detected_languages": None}
        data = json.loads(m.group(1))

    segs, langs, seen = [], [], set()
    for raw in data.get("segments", []):
        t0 = parse_timestamp(raw.get("timestamp_start", "0:00"))
        t1 = parse_timestamp(raw.get("timestamp_end", "0:00"))
        if t1 <= t0:                       # ref :396-397
            t1 = t0 + 1.0
        lang_list = raw.get("languages", [])
        if lang_list:                      # ref :401-419
            name, code = lang_list[0].get("name"), lang_list[0].get("code")
            seg_langs = [{"name": x.get("name", ""), "code": x.get("code", "")} for x in lang_list]
            for x in lang_list:
                c = x.get("code", "")
                if c and c not in seen:
                    seen.add(c)
                    langs.append({"code": c, "language": x.get("name", c)})
        else:                              # ref :420-430 (legacy single-language form)
            name, code, seg_langs = raw.get("language"), raw.get("language_code"), None
            if code and code not in seen:
                seen.add(code)
                langs.append({"code": code, "language": name or code})
        segs.append({
            "start": t0, "end": t1,
            "speaker": raw.get("speaker", "Speaker 1"),
            "text": raw.get("content", ""),
            "language": name, "language_code": code, "languages": seg_langs,
            "emotion": raw.get("emotion", "neutral"),
            "translation": raw.get("translation"),
            "words": None,
        })
    summary = data.get("summary", "")
    return {"summary": summary if summary else None, "segments": segs,
            "detected_languages": langs if langs else None}


def offset_result(result: dict, offset: float) -> dict:
    """ref gemini_pipeline.py:555-575: add the chunk's start offset to every segment; the
    per-segment `languages` list is NOT carried over (it becomes None)."""
    out = []
    for s in result["segments"]:
        d = dict(s)
        d["start"] = s["start"] + offset
        d["end"] = s["end"] + offset
        d["languages"] = None
        out.append(d)
    return {"summary": result["summary"], "segments": out,
            "detected_languages": result["detected_languages"]}


def merge_results(results: list[dict]) -> dict:
    """ref gemini_pipeline.py:646-678: stable sort of chunk results by first-segment start
    (empty -> +inf), concatenate, union languages by code, join summaries with ' '."""
    ordered = sorted(results, key=lambda r: r["segments"][0]["start"] if r["segments"] else math.inf)
    segs = [s for r in ordered for s in r["segments"]]
    langs, seen = [], set()
    for r in ordered:
        for lang in r["detected_languages"] or []:
            c = lang.get("code", "")
            if c and c not in seen:
                seen.add(c)
                langs.append(lang)
    summaries = [r["summary"] for r in ordered if r["summary"]]
    return {"summary": " ".join(summaries) if summaries else None, "segments": segs,
            "detected_languages": langs if langs else None}


def retry_sleeps(max_retries: int = 3) -> list[int]:
    """Sleep lengths between attempts, ref gemini_pipeline.py:733-737 (none after the last)."""
    return [2 ** a for a in range(max_retries - 1)]


PROGRESS_SEQUENCE = [("uploading", 0), ("transcribing", 1), ("processing", 2), ("done", 3)]
