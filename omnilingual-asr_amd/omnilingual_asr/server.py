"""HTTP shell around `Pipeline.transcribe()` with the routes, payloads and progress events of the
reference's web workflow (ref:workflows/wav2elan_web/app.py), so its single-page client can talk to
the local engine unchanged:

  GET  /                               index.html from the static directory when there is one
  POST /api/transcribe                 one file -> JSON entry                              (ref :157-174)
  POST /api/transcribe-stream          one file -> SSE `progress`* then `result`            (ref :177-281)
  POST /api/transcribe-batch-stream    files / zip -> SSE `progress`* then `result`         (ref :284-431)
  GET  /api/history                    [{id, file_name, created_at}] newest first           (ref :434-445)
  GET|PUT|DELETE /api/history/{id}                                                         (ref :448-476)
  GET  /api/history/{id}/export/{fmt}  eaf | textgrid | srt | txt | json (the client's download
                                       formats, built server-side by `exporters`; not in the reference)

Differences forced by this environment: multipart bodies are parsed here with the standard library
(python-multipart is not installed) and server-sent events are written directly (no sse-starlette).
History lives in memory, as in the reference.  `create_app()` returns a Starlette application.
"""

from __future__ import annotations

import asyncio
import email.parser
import email.policy
import json
import threading
import uuid
import zipfile
from datetime import datetime, timezone
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

from starlette.applications import Starlette
from starlette.exceptions import HTTPException
from starlette.requests import Request
from starlette.responses import HTMLResponse, JSONResponse, PlainTextResponse, Response, StreamingResponse
from starlette.routing import Mount, Route
from starlette.staticfiles import StaticFiles

from . import exporters

AUDIO_EXT = {".wav", ".mp3", ".flac", ".ogg", ".m4a"}            # ref :76-77
UPLOAD_EXT = AUDIO_EXT | {".zip"}                                  # ref :104-106
MAX_PARALLEL = 4                                                   # ref :395-397


def _now_iso() -> str:
    """UTC to the second with a 'Z' suffix: the `created_at` form of the reference's history entries (ref :63-64)."""
    return datetime.now(timezone.utc).replace(microsecond=0, tzinfo=None).isoformat() + "Z"


def parse_multipart(content_type: str, body: bytes) -> List[Tuple[str, Optional[str], bytes]]:
    """multipart/form-data -> [(field name, file name or None, payload)], in order."""
    if "multipart/form-data" not in (content_type or ""):
        raise HTTPException(status_code=400, detail="Expected multipart/form-data.")
    msg = email.parser.BytesParser(policy=email.policy.HTTP).parsebytes(
        b"Content-Type: " + content_type.encode("latin-1") + b"\r\nMIME-Version: 1.0\r\n\r\n" + body)
    parts = []
    for part in msg.iter_parts():
        name = part.get_param("name", header="content-disposition")
        filename = part.get_param("filename", header="content-disposition")
        parts.append((name, filename, part.get_payload(decode=True) or b""))
    return parts


def segment_dict(seg) -> Dict[str, Any]:
    """One segment as the reference's server emits it (ref :121-143): words always a list, the optional
    fields only when set."""
    d: Dict[str, Any] = {"start": seg.start, "end": seg.end, "speaker": seg.speaker, "text": seg.text,
                         "words": [{"word": w.word, "start": w.start, "end": w.end} for w in (seg.words or [])]}
    for key in ("language", "language_code", "languages", "emotion", "translation"):
        value = getattr(seg, key, None)
        if value:
            d[key] = value
    return d


def sse(event: str, data: Any) -> bytes:
    return ("event: %s\r\ndata: %s\r\n\r\n" % (event, json.dumps(data))).encode("utf-8")


def create_app(pipeline=None, upload_dir: Optional[Path] = None, static_dir: Optional[Path] = None) -> Starlette:
    """`pipeline`: an `omnilingual_asr.Pipeline` (created on first use when None)."""
    upload_dir = Path(upload_dir) if upload_dir else Path.cwd() / "uploads"
    upload_dir.mkdir(parents=True, exist_ok=True)
    state = {"pipeline": pipeline}
    history: Dict[str, Dict[str, Any]] = {}
    order: List[str] = []
    lock = threading.Lock()

    def get_pipeline():
        if state["pipeline"] is None:
            from . import Pipeline
            state["pipeline"] = Pipeline()
        return state["pipeline"]

    def store(entry: Dict[str, Any]) -> Dict[str, Any]:
        with lock:
            entry["id"] = uuid.uuid4().hex
            entry["created_at"] = _now_iso()
            history[entry["id"]] = entry
            order.insert(0, entry["id"])
        return entry

    def save_upload(filename: Optional[str], payload: bytes, dest: Path) -> Tuple[Path, str]:
        """Stores an upload under a random name that keeps its extension; the two refusals and their texts are the
        reference's (ref :98-108)."""
        suffix = Path(filename).suffix.lower() if filename else None
        problem = "Missing file name." if suffix is None else None if suffix in UPLOAD_EXT else "Unsupported file type."
        if problem:
            raise HTTPException(status_code=400, detail=problem)
        stored = dest / (uuid.uuid4().hex + suffix)
        stored.write_bytes(payload)
        return stored, filename

    def safe_extract_zip(zip_path: Path, dest: Path) -> List[Tuple[Path, str]]:
        """Unpacks an archive below `dest` and lists its audio files as (path, name inside the archive).  Members that
        would land outside `dest` (absolute names, '..' components) are skipped, as the reference skips them (ref :80-95)."""
        def inside(name: str) -> bool:
            member = Path(name)
            return not member.is_absolute() and ".." not in member.parts

        with zipfile.ZipFile(zip_path) as zf:
            wanted = [info for info in zf.infolist() if not info.is_dir() and inside(info.filename)]
            for info in wanted:
                target = dest / info.filename
                target.parent.mkdir(parents=True, exist_ok=True)
                target.write_bytes(zf.read(info))
        return [(dest / info.filename, Path(info.filename).as_posix()) for info in wanted
                if Path(info.filename).suffix.lower() in AUDIO_EXT]

    def run(audio_path: Path, **kw) -> Dict[str, Any]:
        pipe = get_pipeline()
        segments = pipe.transcribe(str(audio_path), word_timestamps=True, **kw)
        result: Dict[str, Any] = {"segments": [segment_dict(s) for s in segments]}
        if pipe.summary:
            result["summary"] = pipe.summary
        if pipe.detected_languages:
            result["detected_languages"] = pipe.detected_languages
        return result

    async def form(request: Request):
        parts = parse_multipart(request.headers.get("content-type", ""), await request.body())
        files = [(fn, data) for name, fn, data in parts if fn is not None and name in ("file", "files")]
        fields = {name: data.decode("utf-8") for name, fn, data in parts if fn is None}
        return files, fields

    async def index(request: Request):
        if static_dir and (Path(static_dir) / "index.html").exists():
            return HTMLResponse((Path(static_dir) / "index.html").read_text(encoding="utf-8"))
        return PlainTextResponse("omnilingual-asr local engine: POST audio to /api/transcribe")

    async def transcribe(request: Request):
        files, _ = await form(request)
        if not files:
            raise HTTPException(status_code=400, detail="Missing file name.")
        path, display = save_upload(files[0][0], files[0][1], upload_dir)
        if path.suffix.lower() == ".zip":
            raise HTTPException(status_code=400, detail="Use batch endpoint for zip uploads.")
        result = await asyncio.get_event_loop().run_in_executor(None, lambda: run(path))
        return JSONResponse(store({"audio_url": "/uploads/" + path.name, "file_name": display, **result}))

    async def transcribe_stream(request: Request):
        files, fields = await form(request)
        if not files:
            raise HTTPException(status_code=400, detail="Missing file name.")
        path, display = save_upload(files[0][0], files[0][1], upload_dir)
        if path.suffix.lower() == ".zip":
            raise HTTPException(status_code=400, detail="Use batch endpoint for zip uploads.")

        async def events():
            loop = asyncio.get_event_loop()
            queue: asyncio.Queue = asyncio.Queue()
            cb = lambda step, idx: loop.call_soon_threadsafe(queue.put_nowait, (step, idx))
            task = loop.run_in_executor(None, lambda: run(path, progress_callback=cb, language=fields.get("language"),
                                                          speaker_count=fields.get("speaker_count")))
            while not task.done() or not queue.empty():
                try:
                    step, idx = await asyncio.wait_for(queue.get(), timeout=0.1)
                    yield sse("progress", {"step": step, "index": idx, "file_name": display})
                except asyncio.TimeoutError:
                    continue
            result = await task
            yield sse("result", store({"audio_url": "/uploads/" + path.name, "file_name": display, **result}))

        return StreamingResponse(events(), media_type="text/event-stream")

    async def transcribe_batch_stream(request: Request):
        files, fields = await form(request)
        if not files:
            raise HTTPException(status_code=400, detail="No files uploaded.")
        batch_id = uuid.uuid4().hex
        batch_dir = upload_dir / batch_id
        batch_dir.mkdir(parents=True, exist_ok=True)
        audio: List[Tuple[Path, str]] = []
        for fn, data in files:
            saved, display = save_upload(fn, data, batch_dir)
            if saved.suffix.lower() == ".zip":
                audio.extend(safe_extract_zip(saved, batch_dir))
            elif saved.suffix.lower() in AUDIO_EXT:
                audio.append((saved, display))
        if not audio:
            raise HTTPException(status_code=400, detail="No supported audio files found.")

        async def events():
            loop = asyncio.get_event_loop()
            queue: asyncio.Queue = asyncio.Queue()
            sem = asyncio.Semaphore(min(MAX_PARALLEL, len(audio)))

            async def one(i: int, path: Path, display: str):
                def cb(step, idx):
                    loop.call_soon_threadsafe(queue.put_nowait, {"step": step, "index": idx, "file_index": i,
                                                                 "file_count": len(audio), "file_name": display})
                async with sem:
                    result = await loop.run_in_executor(
                        None, lambda: run(path, progress_callback=cb, language=fields.get("language"),
                                          speaker_count=fields.get("speaker_count")))
                rel = path.relative_to(batch_dir).as_posix()
                return store({"file_name": display, "audio_url": "/uploads/%s/%s" % (batch_id, rel), **result})

            task = asyncio.ensure_future(asyncio.gather(*[one(i, p, d) for i, (p, d) in enumerate(audio)]))
            while not task.done() or not queue.empty():
                try:
                    yield sse("progress", await asyncio.wait_for(queue.get(), timeout=0.1))
                except asyncio.TimeoutError:
                    continue
            yield sse("result", {"results": list(await task)})

        return StreamingResponse(events(), media_type="text/event-stream")

    async def list_history(request: Request):
        with lock:
            return JSONResponse([{"id": i, "file_name": history[i]["file_name"], "created_at": history[i]["created_at"]}
                                 for i in order if i in history])

    async def one_history(request: Request):
        hid = request.path_params["history_id"]
        if request.method == "DELETE":
            with lock:
                history.pop(hid, None)
                if hid in order:
                    order.remove(hid)
            return JSONResponse({"ok": True})
        if hid not in history:
            raise HTTPException(status_code=404, detail="History entry not found.")
        if request.method == "PUT":
            payload = await request.json()
            with lock:
                for key in ("file_name", "segments"):
                    if key in payload:
                        history[hid][key] = payload[key]
        return JSONResponse(history[hid])

    async def export(request: Request):
        hid, fmt = request.path_params["history_id"], request.path_params["fmt"].lower()
        if hid not in history:
            raise HTTPException(status_code=404, detail="History entry not found.")
        if fmt not in exporters.EXPORTERS:
            raise HTTPException(status_code=400, detail="Unknown export format.")
        entry = history[hid]
        data = {"audio_url": entry.get("audio_url", ""), "summary": entry.get("summary"),
                "detected_languages": entry.get("detected_languages"), "segments": entry["segments"]}
        media = {"eaf": "application/xml", "json": "application/json"}.get(fmt, "text/plain")
        return Response(exporters.EXPORTERS[fmt](data), media_type=media + "; charset=utf-8")

    async def http_error(request: Request, exc: HTTPException):
        return JSONResponse({"detail": exc.detail}, status_code=exc.status_code)

    routes = [
        Route("/", index),
        Route("/api/transcribe", transcribe, methods=["POST"]),
        Route("/api/transcribe-stream", transcribe_stream, methods=["POST"]),
        Route("/api/transcribe-batch-stream", transcribe_batch_stream, methods=["POST"]),
        Route("/api/history", list_history),
        Route("/api/history/{history_id}", one_history, methods=["GET", "PUT", "DELETE"]),
        Route("/api/history/{history_id}/export/{fmt}", export),
        Mount("/uploads", StaticFiles(directory=str(upload_dir)), name="uploads"),
    ]
    if static_dir and Path(static_dir).is_dir():
        routes.append(Mount("/static", StaticFiles(directory=str(static_dir)), name="static"))
    return Starlette(routes=routes, exception_handlers={HTTPException: http_error})
