"""Exporter formats against hand-derived expectations (the reference's browser code could not be run
to make golden files: parity unpinned; each expectation cites the rule of ref:public/static/app.js it encodes)."""
import datetime as dt

from omnilingual_asr import exporters as X
from omnilingual_asr.diarization import DiarizedTranscriptSegment, WordTimestamp

TWO = {"audio_url": "x&y.wav", "summary": "A short talk.",
       "detected_languages": [{"language": "English", "code": "en"}, {"language": "Lisu", "code": "lis"}],
       "segments": [
           {"start": 0.5, "end": 2.25, "speaker": "Speaker 1", "text": "hello <there>", "language": "English",
            "language_code": "en", "emotion": "happy", "translation": None},
           {"start": 2.2505, "end": 3661.0395, "speaker": "Speaker 2", "text": "ngwa", "language": "Lisu",
            "emotion": "neutral", "translation": "I"},
           {"start": 3661.9996, "end": 3662.9996, "speaker": "Speaker 1", "text": "bye", "language": "English",
            "language_code": "en", "emotion": "sad", "translation": "bye"}]}


def test_srt_times_prefix_and_millisecond_quirk():
    srt = X.build_srt(TWO)
    assert srt == ("1\n00:00:00,500 --> 00:00:02,250\n[Speaker 1] hello <there>\n\n"
                   "2\n00:00:02,251 --> 01:01:01,039\n[Speaker 2] ngwa\n\n"       # 250.5 rounds up; (t % 1) * 1000 = 39.4999..
                   "3\n01:01:01,1000 --> 01:01:02,1000\n[Speaker 1] bye\n")        # ms rounded on its own (ref :1750)
    one = {"segments": [{"start": 0.0, "end": 17.577, "speaker": "Speaker 1", "text": "four score"}]}
    assert X.build_srt(one) == "1\n00:00:00,000 --> 00:00:17,577\nfour score\n"    # single speaker: no prefix
    assert X.build_srt({"segments": []}) == ""


def test_plain_text_sections_and_translation_rule():
    txt = X.build_plain_text(TWO)
    assert txt == ("=== Summary ===\nA short talk.\n\n=== Detected Languages ===\n- English (en)\n- Lisu (lis)\n\n"
                   "=== Transcript ===\n\n[Speaker 1]\n0:00 hello <there>\n\n[Speaker 2]\n0:02 ngwa\n         → I\n\n"
                   "[Speaker 1]\n61:01 bye")            # translation equal to the text is not repeated (ref :1729)
    assert X.build_plain_text({"summary": None, "detected_languages": [], "segments": []}) == "=== Transcript ===\n"


def test_textgrid_words_gaps_and_quotes():
    data = {"segments": [
        {"start": 1.0, "end": 4.0, "speaker": 'S "1"', "text": "four score",
         "words": [{"word": "four", "start": 1.0, "end": 1.4}, {"word": "score", "start": 1.6, "end": 2.0}]},
        {"start": 6.0, "end": 7.5, "speaker": 'S "1"', "text": 'and "so"', "words": []}]}
    tg = X.build_textgrid(data)
    assert tg == ('File type = "ooTextFile"\nObject class = "TextGrid"\n\nxmin = 0 \nxmax = 7.500000\n\n'
                  'tiers? <exists> \nsize = 1\nitem []:\n'
                  '    item [1]:\n        class = "IntervalTier" \n        name = "S ""1"""\n        xmin = 0 \n'
                  '        xmax = 7.500000\n        intervals: size = 6\n'
                  '        intervals [1]:\n            xmin = 0.000000 \n            xmax = 1.000000\n            text = ""\n'
                  '        intervals [2]:\n            xmin = 1.000000 \n            xmax = 1.400000\n            text = "four"\n'
                  '        intervals [3]:\n            xmin = 1.400000 \n            xmax = 1.600000\n            text = ""\n'
                  '        intervals [4]:\n            xmin = 1.600000 \n            xmax = 2.000000\n            text = "score"\n'
                  '        intervals [5]:\n            xmin = 2.000000 \n            xmax = 6.000000\n            text = ""\n'
                  '        intervals [6]:\n            xmin = 6.000000 \n            xmax = 7.500000\n            text = "and ""so"""\n')
    # gaps of at most 1 ms are not filled (ref :1628), nor is a tail within 1 ms of the end
    close = {"segments": [{"start": 0.0005, "end": 1.0, "speaker": "A", "text": "x"},
                          {"start": 1.0009, "end": 2.0, "speaker": "A", "text": "y"},
                          {"start": 0.0, "end": 2.0005, "speaker": "B", "text": "z"}]}
    tg = X.build_textgrid(close)
    assert tg.count("intervals [") == 3 and "size = 2\nitem []" in tg and "xmax = 2.000500" in tg


def test_eaf_structure_ids_rounding_and_tiers():
    eaf = X.build_eaf(TWO, now=dt.datetime(2025, 1, 2, 3, 4, 5, 678000, tzinfo=dt.timezone.utc))
    assert eaf.startswith('<?xml version="1.0" encoding="UTF-8"?>\n<ANNOTATION_DOCUMENT AUTHOR="OmniTranscribe" '
                          'DATE="2025-01-02T03:04:05.678Z" FORMAT="3.0" VERSION="3.0" ')
    assert '<MEDIA_DESCRIPTOR MEDIA_URL="x&amp;y.wav" MIME_TYPE="audio/x-wav"/>' in eaf
    # two slots per segment, integer milliseconds, halves round up
    for tsid, ms in (("ts1", 500), ("ts2", 2250), ("ts3", 2251), ("ts4", 3661040), ("ts5", 3662000), ("ts6", 3663000)):
        assert '        <TIME_SLOT TIME_SLOT_ID="%s" TIME_VALUE="%d"/>' % (tsid, ms) in eaf
    # annotation ids run per segment: transcript, language, emotion, translation (only when it differs from the text)
    assert ('<ALIGNABLE_ANNOTATION ANNOTATION_ID="a1" TIME_SLOT_REF1="ts1" TIME_SLOT_REF2="ts2">\n'
            '                    <ANNOTATION_VALUE>hello &lt;there&gt;</ANNOTATION_VALUE>') in eaf
    assert 'ANNOTATION_ID="a2" TIME_SLOT_REF1="ts1" TIME_SLOT_REF2="ts2">\n                    <ANNOTATION_VALUE>en<' in eaf
    assert 'ANNOTATION_ID="a5" TIME_SLOT_REF1="ts3" TIME_SLOT_REF2="ts4">\n                    <ANNOTATION_VALUE>Lisu<' in eaf
    assert 'ANNOTATION_ID="a7" TIME_SLOT_REF1="ts3" TIME_SLOT_REF2="ts4">\n                    <ANNOTATION_VALUE>I<' in eaf
    assert 'ANNOTATION_ID="a10"' in eaf and 'ANNOTATION_ID="a11"' not in eaf
    order = [eaf.index('TIER_ID="%s"' % t) for t in ("Speaker 1", "Speaker 2", "Speaker 1_language", "Speaker 2_language",
                                                      "Speaker 1_emotion", "Speaker 2_emotion", "Speaker 2_translation")]
    assert order == sorted(order) and 'TIER_ID="Speaker 1_translation"' not in eaf
    assert eaf.endswith('    <LINGUISTIC_TYPE LINGUISTIC_TYPE_ID="transcription" TIME_ALIGNABLE="true"/>\n'
                        '    <LINGUISTIC_TYPE LINGUISTIC_TYPE_ID="language" TIME_ALIGNABLE="true"/>\n'
                        '    <LINGUISTIC_TYPE LINGUISTIC_TYPE_ID="emotion" TIME_ALIGNABLE="true"/>\n'
                        '    <LINGUISTIC_TYPE LINGUISTIC_TYPE_ID="translation" TIME_ALIGNABLE="true"/>\n</ANNOTATION_DOCUMENT>')
    import xml.dom.minidom
    xml.dom.minidom.parseString(eaf)          # well formed
    plain = X.build_eaf({"audio_url": "", "segments": [{"start": 0, "end": 1, "speaker": "S", "text": "t"}]})
    assert "_language" not in plain and plain.count("<TIER ") == 1


def test_json_matches_javascript_number_notation():
    data = {"a": 1.0, "b": 17.577, "c": [], "d": {}, "e": None, "f": [0.5, 1e-7, 1e21, 123456789012.0, -2.0, True],
            "g": "q\"→"}
    assert X.build_json(data) == ('{\n  "a": 1,\n  "b": 17.577,\n  "c": [],\n  "d": {},\n  "e": null,\n  "f": [\n'
                                  '    0.5,\n    1e-7,\n    1e+21,\n    123456789012,\n    -2,\n    true\n  ],\n'
                                  '  "g": "q\\"→"\n}')
    assert X._js_number(0.000001) == "0.000001" and X._js_number(1.5e-7) == "1.5e-7" and X._js_number(281233 / 16000) == "17.5770625"


def test_result_to_data_from_public_segments():
    segs = [DiarizedTranscriptSegment(0.0, 1.5, "Speaker 1", "hi", words=[WordTimestamp("hi", 0.2, 0.6)], emotion="neutral"),
            DiarizedTranscriptSegment(1.5, 2.0, "Speaker 1", "yo", emotion="neutral")]
    data = X.result_to_data(segs, audio_url="/a.wav")
    assert data["segments"][0]["words"] == [{"word": "hi", "start": 0.2, "end": 0.6}] and data["segments"][1]["words"] == []
    assert "hi" in X.build_textgrid(data) and set(X.EXPORTERS) == {"eaf", "textgrid", "txt", "srt", "json"}
    assert X.build_srt(data).startswith("1\n00:00:00,000 --> 00:00:01,500\nhi\n")


def test_real_annotation_data_round_trips_through_every_format(golden_dir):
    """The reference's own annotation of its demo recording (ref:samuel_lisu.eaf as data: 36 intervals, 314 words,
    text in Arabic, Tibetan, Chinese and Latin script; tests/golden/reference_lisu_annotation.json, made by
    oracle/gen_eaf_fixture.py) through all five writers and read back: every time (whole milliseconds) and every
    text survives.  This checks the writers against themselves on real data, not against the reference's writer."""
    import json
    import re
    import xml.etree.ElementTree as ET

    src = json.loads((golden_dir / "reference_lisu_annotation.json").read_text(encoding="utf-8"))["segments"]
    data = {"audio_url": "Samuel Speaking Lisu.mp3", "summary": None, "detected_languages": [],
            "segments": [{"start": s["start_ms"] / 1000.0, "end": s["end_ms"] / 1000.0, "speaker": s["speaker"],
                          "text": s["text"],
                          "words": [{"word": w["word"], "start": w["start_ms"] / 1000.0, "end": w["end_ms"] / 1000.0}
                                    for w in s["words"]]} for s in src]}
    want = [(s["start_ms"], s["end_ms"], s["text"]) for s in src]
    assert len(want) == 36 and any(ord(c) > 0x4E00 for _, _, t in want for c in t)

    root = ET.fromstring(X.build_eaf(data).encode("utf-8"))
    slot = {t.get("TIME_SLOT_ID"): int(t.get("TIME_VALUE")) for t in root.iter("TIME_SLOT")}
    tiers = list(root.iter("TIER"))
    assert [t.get("TIER_ID") for t in tiers] == ["SPEAKER_00"]
    got = [(slot[a.get("TIME_SLOT_REF1")], slot[a.get("TIME_SLOT_REF2")], a.findtext("ANNOTATION_VALUE"))
           for a in tiers[0].iter("ALIGNABLE_ANNOTATION")]
    assert got == want

    def ms(stamp):
        h, m, rest = stamp.split(":")
        sec, milli = rest.split(",")
        return ((int(h) * 60 + int(m)) * 60 + int(sec)) * 1000 + int(milli)
    blocks = X.build_srt(data).strip().split("\n\n")
    got = []
    for i, b in enumerate(blocks):
        idx, times, text = b.split("\n", 2)
        a, z = times.split(" --> ")
        assert int(idx) == i + 1
        got.append((ms(a), ms(z), text))
    assert got == want                                   # one speaker: no prefix

    tg = X.build_textgrid(data)
    spoken = [(round(float(a) * 1000), round(float(z) * 1000), t.replace('""', '"'))
              for a, z, t in re.findall(r'xmin = ([0-9.]+) \n\s+xmax = ([0-9.]+)\n\s+text = "((?:[^"]|"")*)"', tg) if t]
    words = [(w["start_ms"], w["end_ms"], w["word"]) for s in src for w in s["words"]]
    assert len(words) == 314 and spoken == words          # segments that have words are written word by word

    assert json.loads(X.build_json(data)) == data
    txt = X.build_plain_text(data)
    assert all(t in txt for _, _, t in want) and txt.count("[SPEAKER_00]") == 1

