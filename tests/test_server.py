"""The HTTP shell against the reference server's contract (ref:workflows/wav2elan_web/app.py), run
in-process on the CPU with the oracle engine injected."""
import io
import json
import zipfile

import pytest

from omnilingual_asr import Pipeline
from omnilingual_asr.server import create_app, parse_multipart

starlette_testclient = pytest.importorskip("starlette.testclient")


@pytest.fixture()
def client(tmp_path, tiny_engine):
    app = create_app(Pipeline(engine=tiny_engine), upload_dir=tmp_path / "uploads")
    return starlette_testclient.TestClient(app)


def _events(text):
    out = []
    for block in text.replace("\r\n", "\n").strip().split("\n\n"):
        lines = dict(l.split(": ", 1) for l in block.split("\n"))
        out.append((lines["event"], json.loads(lines["data"])))
    return out


def test_transcribe_and_history_round_trip(client, golden_dir):
    wav = (golden_dir / "gettysburg.wav").read_bytes()
    r = client.post("/api/transcribe", files={"file": ("gettysburg.wav", wav, "audio/wav")})
    assert r.status_code == 200
    entry = r.json()
    assert set(entry) == {"audio_url", "file_name", "segments", "id", "created_at"}       # no summary / languages: unset
    assert entry["file_name"] == "gettysburg.wav" and entry["audio_url"].startswith("/uploads/") and entry["audio_url"].endswith(".wav")
    seg = entry["segments"][0]
    assert set(seg) == {"start", "end", "speaker", "text", "words", "emotion"} and seg["emotion"] == "neutral"
    assert seg["words"] and " ".join(w["word"] for w in seg["words"]) == seg["text"]
    assert client.get(entry["audio_url"]).content == wav                                   # the upload is served back
    listing = client.get("/api/history").json()
    assert listing == [{"id": entry["id"], "file_name": "gettysburg.wav", "created_at": entry["created_at"]}]
    assert client.get("/api/history/" + entry["id"]).json() == entry
    put = client.put("/api/history/" + entry["id"], json={"file_name": "renamed.wav", "id": "ignored"}).json()
    assert put["file_name"] == "renamed.wav" and put["id"] == entry["id"]
    srt = client.get("/api/history/%s/export/srt" % entry["id"]).text
    assert srt.startswith("1\n00:00:00,000 --> 00:00:17,577\n" + seg["text"])
    assert client.get("/api/history/%s/export/eaf" % entry["id"]).text.startswith('<?xml version="1.0"')
    assert client.get("/api/history/%s/export/nope" % entry["id"]).status_code == 400
    assert client.delete("/api/history/" + entry["id"]).json() == {"ok": True}
    assert client.get("/api/history/" + entry["id"]).status_code == 404
    assert client.get("/api/history/" + entry["id"]).json() == {"detail": "History entry not found."}


def test_mp3_upload_is_transcribed(client, golden_dir):
    """The upload type of the reference's demos (ref app.py:76-77 accepts .mp3): 40 s of its demo recording through
    POST /api/transcribe; two 30 s-plan segments that cover the decoded duration."""
    mp3 = (golden_dir / "reference_demo_40s.mp3").read_bytes()
    r = client.post("/api/transcribe", files={"file": ("demo.mp3", mp3, "audio/mpeg")})
    assert r.status_code == 200, r.text
    entry = r.json()
    assert entry["file_name"] == "demo.mp3" and entry["audio_url"].endswith(".mp3")
    segs = entry["segments"]
    assert [s["start"] for s in segs] == [0.0, 30.0] and abs(segs[1]["end"] - 39.953) < 0.01 and all(s["text"] for s in segs)


def test_upload_validation_messages(client):
    assert client.post("/api/transcribe", files={"file": ("notes.txt", b"x", "text/plain")}).json() == {"detail": "Unsupported file type."}
    assert client.post("/api/transcribe", files={"file": ("a.zip", b"PK", "application/zip")}).json() == \
        {"detail": "Use batch endpoint for zip uploads."}
    r = client.post("/api/transcribe-batch-stream", files=[("files", ("a.zip", _zip({"readme.txt": b"hi"}), "application/zip"))])
    assert r.status_code == 400 and r.json() == {"detail": "No supported audio files found."}


def _zip(members):
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w") as zf:
        for name, data in members.items():
            zf.writestr(name, data)
    return buf.getvalue()


def test_stream_reports_the_four_progress_steps_then_the_result(client, golden_dir):
    wav = (golden_dir / "gettysburg.wav").read_bytes()
    r = client.post("/api/transcribe-stream", files={"file": ("g.wav", wav, "audio/wav")}, data={"language": "en"})
    assert r.status_code == 200 and r.headers["content-type"].startswith("text/event-stream")
    ev = _events(r.text)
    assert [(e, d["step"], d["index"], d["file_name"]) for e, d in ev[:-1]] == \
        [("progress", s, i, "g.wav") for s, i in (("uploading", 0), ("transcribing", 1), ("processing", 2), ("done", 3))]
    assert ev[-1][0] == "result" and ev[-1][1]["segments"][0]["text"]


def test_batch_stream_with_zip_and_loose_file(client, golden_dir):
    wav = (golden_dir / "gettysburg.wav").read_bytes()
    z = _zip({"dir/a.wav": wav, "../evil.wav": wav, "notes.txt": b"x"})
    r = client.post("/api/transcribe-batch-stream", files=[("files", ("pack.zip", z, "application/zip")),
                                                           ("files", ("b.wav", wav, "audio/wav"))])
    ev = _events(r.text)
    progress = [d for e, d in ev if e == "progress"]
    assert {d["file_name"] for d in progress} == {"dir/a.wav", "b.wav"} and all(d["file_count"] == 2 for d in progress)
    for i in (0, 1):     # each file reports its four steps in order
        assert [(d["step"], d["index"]) for d in progress if d["file_index"] == i] == \
            [("uploading", 0), ("transcribing", 1), ("processing", 2), ("done", 3)]
    results = ev[-1][1]["results"]
    assert [x["file_name"] for x in results] == ["dir/a.wav", "b.wav"]
    assert results[0]["audio_url"].endswith("/dir/a.wav") and len(client.get("/api/history").json()) == 2


def test_parse_multipart_fields_and_files():
    boundary = "XyZ"
    body = (b"--XyZ\r\nContent-Disposition: form-data; name=\"language\"\r\n\r\nen\r\n"
            b"--XyZ\r\nContent-Disposition: form-data; name=\"file\"; filename=\"a b.wav\"\r\nContent-Type: audio/wav\r\n\r\n"
            b"\x00\x01\r\n\xff\r\n--XyZ--\r\n")
    parts = parse_multipart("multipart/form-data; boundary=" + boundary, body)
    assert parts == [("language", None, b"en"), ("file", "a b.wav", b"\x00\x01\r\n\xff")]
