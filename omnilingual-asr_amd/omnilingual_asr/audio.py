"""Audio ingest: WAV, AIFF/AIFF-C, FLAC, MP3 and Ogg Vorbis decode, down-mix and polyphase resampling to 16 kHz mono
float32.

The reference hands the encoded file to its remote model and uses ffprobe/ffmpeg only for duration
and slicing (ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:222-310).  Here the host
decodes the audio itself: PCM containers in this module, FLAC, MPEG Layer III and Ogg Vorbis by the native decoders
in liboasr_hip.so (`oasr_flac_decode`, `oasr_mp3_decode`, `oasr_vorbis_decode`).  The file type is taken from its
first bytes, not its name.  The remaining formats the reference lists (:106-118: aac, m4a) and other codecs in an
Ogg container (Opus, Speex) need decoders this package does not contain and raise `UnsupportedAudioError`.
"""

from __future__ import annotations

import struct
from fractions import Fraction
from pathlib import Path

import numpy as np

TARGET_RATE = 16000


class UnsupportedAudioError(RuntimeError):
    pass


_RIFF_MAGICS = (b"RIFF", b"RF64", b"BW64")   # the 64-bit forms carry their real sizes in a 'ds64' chunk


def _chunk_size(cid: bytes, size: int, ds64_data: int, remaining: int) -> int:
    """A chunk's length: 0xFFFFFFFF stands for "see ds64" in the 64-bit forms and for "to the end of the file" in
    a RIFF written as a stream; never more than the file holds."""
    if size == 0xFFFFFFFF:
        size = ds64_data if (cid == b"data" and ds64_data) else remaining
    return min(size, remaining)


def _riff_chunks(data: bytes):
    if len(data) < 12 or data[:4] not in _RIFF_MAGICS or data[8:12] != b"WAVE":
        raise UnsupportedAudioError("not a RIFF/WAVE file")
    pos, ds64_data = 12, 0
    while pos + 8 <= len(data):
        cid = data[pos:pos + 4]
        size = _chunk_size(cid, struct.unpack_from("<I", data, pos + 4)[0], ds64_data, len(data) - pos - 8)
        if cid == b"ds64" and size >= 16:
            ds64_data = struct.unpack_from("<Q", data, pos + 16)[0]
        yield cid, data[pos + 8: pos + 8 + size]
        pos += 8 + size + (size & 1)


def _g711_table(law: str) -> np.ndarray:
    """The 256 linear 16-bit values of ITU-T G.711 (telephone speech), as float32 / 32768."""
    code = np.arange(256, dtype=np.int32)
    if law == "ulaw":
        u = ~code & 0xFF
        mag = ((((u & 0x0F) << 3) + 0x84) << ((u >> 4) & 7)) - 0x84
        lin = np.where(u & 0x80, -mag, mag)
    else:
        a = code ^ 0x55
        exp, man = (a >> 4) & 7, a & 0x0F
        mag = np.where(exp == 0, (man << 4) + 8, ((man << 4) + 0x108) << np.maximum(exp - 1, 0))
        lin = np.where(a & 0x80, mag, -mag)
    return (lin / 32768.0).astype(np.float32)


def read_wav(path) -> tuple[np.ndarray, int]:
    """Decode a WAV file -> (float32 [frames, channels] in [-1, 1), sample_rate).
    Supports PCM 8/16/24/32-bit, IEEE float 32/64, 8-bit G.711 A-law / mu-law (format tags 6 / 7) and 4-bit IMA ADPCM
    (tag 0x11, decoded by the native library), also via WAVE_FORMAT_EXTENSIBLE."""
    path = Path(path)
    data = path.read_bytes()
    if data[:4] not in _RIFF_MAGICS:
        raise UnsupportedAudioError(
            f"cannot decode '{path.suffix}' audio: WAV, AIFF, FLAC (also in Ogg), MP3 and Ogg Vorbis input is supported")
    fmt = payload = None
    for cid, body in _riff_chunks(data):
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            payload = body
    if fmt is None or payload is None or len(fmt) < 16:
        raise UnsupportedAudioError("WAV file lacks a fmt or data chunk")
    tag, channels, rate, _, block, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if tag == 0xFFFE and len(fmt) >= 26:
        tag = struct.unpack_from("<H", fmt, 24)[0]
    if channels < 1 or block < 1:
        raise UnsupportedAudioError("WAV header declares no channels")
    if tag in (1, 3, 6, 7) and block != channels * ((bits + 7) // 8):
        raise UnsupportedAudioError("WAV header: %d bytes per frame for %d channel(s) of %d bits" % (block, channels, bits))
    n = len(payload) // block
    raw = payload[: n * block]
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = v.astype(np.float32) / 8388608.0
        elif bits == 32:
            x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise UnsupportedAudioError(f"unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(raw, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    elif tag in (6, 7) and bits == 8:
        x = _g711_table("alaw" if tag == 6 else "ulaw")[np.frombuffer(raw, dtype=np.uint8)]
    elif tag == 0x11 and bits == 4:
        import ctypes as C

        from . import _hip
        body = np.frombuffer(payload, dtype=np.uint8)
        blocks = len(payload) // block
        per_block = 1 + (block // channels - 4) * 2
        out = np.zeros((max(blocks * per_block, 1), channels), dtype=np.int16)
        written = C.c_int64()
        try:
            _hip.check(_hip.lib().oasr_adpcm_ima_decode(body.ctypes.data, body.size, channels, block, out.ctypes.data,
                                                        out.shape[0], C.byref(written)), "oasr_adpcm_ima_decode")
        except _hip.OasrError as e:
            raise UnsupportedAudioError(str(e)) from e
        n = written.value
        fact = next((b for cid, b in _riff_chunks(data) if cid == b"fact" and len(b) >= 4), None)
        if fact is not None:                     # the sample count the writer states (the last block may be padded)
            n = min(n, struct.unpack_from("<I", fact, 0)[0])
        return out[:n].astype(np.float32) / 32768.0, int(rate)
    else:
        raise UnsupportedAudioError(f"unsupported WAV format tag {tag}")
    return x.reshape(n, channels), int(rate)


def map_wav_pcm(path):
    """(uint8 view of the memory-mapped file's sample bytes, channels, sample_rate, bytes per sample, bits) when
    `path` is an integer PCM WAV of 8, 16, 24 or 32 bits, else None.  Samples are interleaved, little-endian
    (8-bit: unsigned); a sample's value in [-1, 1) is the integer / 2^(bits - 1).  Nothing is read, converted
    or copied here: the pages are touched when the samples are used."""
    import mmap
    with open(path, "rb") as f:
        size = f.seek(0, 2)
        if size < 44:
            return None
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    if mm[:4] not in _RIFF_MAGICS or mm[8:12] != b"WAVE":
        return None
    fmt = data_at = data_len = None
    pos, ds64_data = 12, 0
    while pos + 8 <= size:
        cid = mm[pos:pos + 4]
        n = _chunk_size(cid, struct.unpack_from("<I", mm, pos + 4)[0], ds64_data, size - pos - 8)
        if cid == b"ds64" and n >= 16:
            ds64_data = struct.unpack_from("<Q", mm, pos + 16)[0]
        if cid == b"fmt ":
            fmt = mm[pos + 8: pos + 8 + min(n, 40)]
        elif cid == b"data":
            data_at, data_len = pos + 8, min(n, size - pos - 8)
        pos += 8 + n + (n & 1)
    if fmt is None or data_at is None or len(fmt) < 16:
        return None
    tag, channels, rate, _, block, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if tag == 0xFFFE and len(fmt) >= 26:
        tag = struct.unpack_from("<H", fmt, 24)[0]
    if tag != 1 or bits not in (8, 16, 24, 32) or channels < 1 or channels > 64 or block != bits // 8 * channels or rate < 1:
        return None
    frames = data_len // block
    return np.frombuffer(mm, dtype=np.uint8, count=frames * block, offset=data_at), int(channels), int(rate), bits // 8, int(bits)


def pcm_to_float(raw, sample_bytes: int, bits: int) -> np.ndarray:
    """Integer samples as `map_wav_pcm` / `FlacStream.read_pcm` hand them over (any array over the sample bytes,
    little-endian, 8-bit unsigned) -> flat float32, value = integer / 2^(bits - 1): the numbers `read_wav` and
    `read_flac` deliver."""
    raw = np.ascontiguousarray(raw).reshape(-1).view(np.uint8)
    if sample_bytes == 1:
        return (raw.astype(np.float32) - 128.0) / 128.0
    if sample_bytes == 2:
        return raw.view("<i2").astype(np.float32) / np.float32(1 << (bits - 1))
    if sample_bytes == 3:
        b = raw.reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        return v.astype(np.float32) / 8388608.0
    if sample_bytes == 4:
        return (raw.view("<i4").astype(np.float64) / float(1 << (bits - 1))).astype(np.float32)
    raise UnsupportedAudioError(f"unsupported PCM width of {sample_bytes} bytes")


def map_wav_pcm16(path):
    """(int16 [frames * channels] interleaved view of the memory-mapped file, channels, sample_rate) when `path`
    is a 16-bit PCM WAV (see `map_wav_pcm`), else None."""
    m = map_wav_pcm(path)
    if m is None or m[3] != 2:
        return None
    return m[0].view("<i2"), m[1], m[2]


def read_wav_pcm16_mono(path):
    """(int16 [frames], sample_rate) when `path` is a mono 16-bit PCM WAV (see `map_wav_pcm16`), else None."""
    m = map_wav_pcm16(path)
    return (m[0], m[2]) if m is not None and m[1] == 1 else None


def wav_duration(path) -> float:
    """Duration in seconds from the header alone; raises on anything unreadable."""
    with open(path, "rb") as f:
        head = f.read(1 << 16)
    rate = block = None
    per_block, stated = 1, 0
    pos = 12
    if head[:4] not in _RIFF_MAGICS or head[8:12] != b"WAVE":
        raise UnsupportedAudioError("not a RIFF/WAVE file")
    file_size, ds64_data = Path(path).stat().st_size, 0
    while pos + 8 <= len(head):
        cid = head[pos:pos + 4]
        size = _chunk_size(cid, struct.unpack_from("<I", head, pos + 4)[0], ds64_data, file_size - pos - 8)
        if cid == b"ds64" and size >= 16 and pos + 24 <= len(head):
            ds64_data = struct.unpack_from("<Q", head, pos + 16)[0]
        if cid == b"fmt ":
            tag, channels, rate, _, block, _ = struct.unpack_from("<HHIIHH", head, pos + 8)
            if tag == 0xFFFE and size >= 26:
                tag = struct.unpack_from("<H", head, pos + 8 + 24)[0]
            if tag == 0x11 and channels and block >= 4 * channels:      # IMA ADPCM: a block holds many samples
                per_block = 1 + (block // channels - 4) * 2
        elif cid == b"fact" and size >= 4:
            stated = struct.unpack_from("<I", head, pos + 8)[0]
        elif cid == b"data":
            if not rate or not block:
                break
            avail = Path(path).stat().st_size - (pos + 8)
            frames = min(size, max(avail, 0)) // block * per_block
            return (min(frames, stated) if per_block > 1 and stated else frames) / float(rate)
        pos += 8 + size + (size & 1)
    raise UnsupportedAudioError("WAV header incomplete")


def audio_duration(path) -> float:
    """Duration in seconds without decoding, for every container `read_audio` decodes (WAV data chunk, AIFF
    COMM chunk, FLAC STREAMINFO - native or in an Ogg container; MP3: its frame headers are counted; Ogg Vorbis: its
    pages are walked for the block sizes and granule positions); raises on anything unreadable."""
    with open(path, "rb") as f:
        head = f.read(1 << 16)
    if head[:4] in _RIFF_MAGICS and not _looks_like_mp3(head, path):
        return wav_duration(path)
    if head[:4] == b"FORM" and head[8:12] in (b"AIFF", b"AIFC"):
        pos = 12
        while pos + 8 <= len(head):
            size = struct.unpack_from(">I", head, pos + 4)[0]
            if head[pos:pos + 4] == b"COMM" and pos + 26 <= len(head):
                frames = struct.unpack_from(">I", head, pos + 10)[0]
                rate = _extended80(head[pos + 16: pos + 26])
                if rate > 0:
                    return frames / rate
            pos += 8 + size + (size & 1)
        raise UnsupportedAudioError("AIFF header incomplete")
    if head[:4] == b"fLaC" and len(head) >= 42 and head[4] & 0x7F == 0:
        x = int.from_bytes(head[18:26], "big")
        rate, total = x >> 44, x & ((1 << 36) - 1)
        if rate > 0 and total > 0:
            return total / float(rate)
    if head[:4] == b"OggS" and _ogg_holds_flac(head):
        at = 27 + head[26] + 9                  # the native signature inside the first packet
        x = int.from_bytes(head[at + 18: at + 26], "big")
        rate, total = x >> 44, x & ((1 << 36) - 1)
        if rate > 0 and total > 0:
            return total / float(rate)
        raise UnsupportedAudioError("no duration in the header")
    if head[:4] == b"OggS":
        rate, _, total = vorbis_info(path)
        return total / float(rate)
    if _looks_like_mp3(head, path) or _is_mp3_behind_junk(head, path):
        rate, _, total = mp3_info(path)
        return total / float(rate)
    raise UnsupportedAudioError("no duration in the header")


MAX_CONVERSION_TERM = 200_000   # largest numerator / denominator of a rate ratio the converter designs a filter for


def conversion_ratio(rate_in: int, rate_out: int = TARGET_RATE) -> tuple[int, int]:
    """(up, down) of rate_out / rate_in in lowest terms.  The low-pass has 20 * max(up, down) + 1 taps, so a header
    that states an absurd rate (a damaged or hostile file: 1.1 GHz, or a value sharing no factor with the target)
    would ask for a filter of billions of taps; every real rate (8 kHz .. 768 kHz, the odd 44 056 and 47 999
    included) stays far below the bound."""
    rate_in = int(rate_in)
    if not 1000 <= rate_in <= 4_000_000:    # (below 1 kHz the converter would also inflate the file more than 16-fold)
        raise UnsupportedAudioError("sample rate %d Hz is outside 1 kHz .. 4 MHz" % rate_in)
    fr = Fraction(int(rate_out), rate_in)
    if max(fr.numerator, fr.denominator) > MAX_CONVERSION_TERM:
        raise UnsupportedAudioError("sample rate %d Hz cannot be converted to %d Hz (ratio %d/%d)"
                                    % (rate_in, rate_out, fr.numerator, fr.denominator))
    return fr.numerator, fr.denominator


def resample_filter(up: int, down: int, half_width: int = 10, beta: float = 5.0) -> np.ndarray:
    """Kaiser-windowed sinc low-pass at the up-sampled rate, cut-off at the lower Nyquist, unity
    DC gain times `up`; length 2 * half_width * max(up, down) + 1 (zero phase)."""
    m = max(up, down)
    half = half_width * m
    n = np.arange(-half, half + 1, dtype=np.float64)
    h = np.sinc(n / m) / m * np.kaiser(2 * half + 1, beta)
    return h / h.sum() * up


def polyphase_branches(rate_in: int, rate_out: int = TARGET_RATE):
    """(up, down, half, hp) with hp[p, j] = h[p + j * up], float64 [up, taps]: the table both the host
    resampler below and the device kernel (`oasr_resample_poly`) read."""
    up, down = conversion_ratio(rate_in, rate_out)
    h = resample_filter(up, down)
    half = (len(h) - 1) // 2
    taps = -(-len(h) // up)
    hp = np.zeros((up, taps), dtype=np.float64)
    for p in range(up):
        seg = h[p::up]
        hp[p, : len(seg)] = seg
    return up, down, half, hp


def resample_poly(x: np.ndarray, rate_in: int, rate_out: int = TARGET_RATE) -> np.ndarray:
    """Rational polyphase resampler: y[n] = sum_k x[k] h[n*down - k*up + half], output length
    ceil(len(x) * up / down).  float64 accumulate, float32 result."""
    if rate_in == rate_out:
        return np.asarray(x, dtype=np.float32)
    up, down = conversion_ratio(rate_in, rate_out)
    h = resample_filter(up, down)
    half = (len(h) - 1) // 2
    x = np.asarray(x, dtype=np.float64)
    n_in = x.shape[0]
    n_out = -(-n_in * up // down)
    taps = -(-len(h) // up)                      # taps per polyphase branch
    hp = np.zeros((up, taps), dtype=np.float64)  # hp[p, j] = h[p + j*up]
    for p in range(up):
        seg = h[p::up]
        hp[p, : len(seg)] = seg
    n = np.arange(n_out, dtype=np.int64)
    pos = n * down + half                        # index into h of the tap that meets x[0]
    phase = pos % up
    k_hi = pos // up                             # newest input sample under the filter
    y = np.zeros(n_out, dtype=np.float64)
    xp = np.concatenate([np.zeros(taps, dtype=np.float64), x, np.zeros(taps, dtype=np.float64)])
    for j in range(taps):                        # tap j of branch `phase` meets x[k_hi - j]
        idx = k_hi - j + taps
        ok = (idx >= 0) & (idx < xp.shape[0])
        y += hp[phase, j] * np.where(ok, xp[np.clip(idx, 0, xp.shape[0] - 1)], 0.0)
    return y.astype(np.float32)


def _extended80(b: bytes) -> float:
    """IEEE 754 80-bit extended (the sample-rate field of an AIFF COMM chunk)."""
    exp = ((b[0] & 0x7F) << 8) | b[1]
    mant = int.from_bytes(b[2:10], "big")
    if exp == 0 and mant == 0:
        return 0.0
    return (-1.0 if b[0] & 0x80 else 1.0) * mant * 2.0 ** (exp - 16383 - 63)


def read_aiff(path) -> tuple[np.ndarray, int]:
    """Decode AIFF / AIFF-C with uncompressed samples ('NONE', 'twos', 'sowt', 'fl32', 'fl64') or G.711 ('ulaw', 'alaw')
    -> (float32 [frames, channels] in [-1, 1), sample_rate)."""
    data = Path(path).read_bytes()
    if len(data) < 12 or data[:4] != b"FORM" or data[8:12] not in (b"AIFF", b"AIFC"):
        raise UnsupportedAudioError("not an AIFF file")
    comm = ssnd = None
    pos = 12
    while pos + 8 <= len(data):
        cid = data[pos:pos + 4]
        size = struct.unpack_from(">I", data, pos + 4)[0]
        body = data[pos + 8: pos + 8 + size]
        if cid == b"COMM":
            comm = body
        elif cid == b"SSND":
            ssnd = body
        pos += 8 + size + (size & 1)
    if comm is None or ssnd is None or len(comm) < 18 or len(ssnd) < 8:
        raise UnsupportedAudioError("AIFF file lacks a COMM or SSND chunk")
    channels, frames, bits = struct.unpack_from(">hIh", comm, 0)
    rate = int(round(_extended80(comm[8:18])))
    kind = comm[18:22] if data[8:12] == b"AIFC" and len(comm) >= 22 else b"NONE"
    if channels < 1 or rate <= 0:
        raise UnsupportedAudioError("AIFF header declares no channels or no sample rate")
    offset = struct.unpack_from(">I", ssnd, 0)[0]
    raw = ssnd[8 + offset:]
    width = (bits + 7) // 8
    if kind in (b"fl32", b"FL32"):
        x = np.frombuffer(raw[: len(raw) // 4 * 4], dtype=">f4").astype(np.float32)
    elif kind in (b"fl64", b"FL64"):
        x = np.frombuffer(raw[: len(raw) // 8 * 8], dtype=">f8").astype(np.float32)
    elif kind in (b"ulaw", b"ULAW", b"alaw", b"ALAW"):
        x = _g711_table(kind.decode().lower())[np.frombuffer(raw, dtype=np.uint8)]
    elif kind in (b"NONE", b"twos", b"sowt"):
        order = "<" if kind == b"sowt" else ">"
        if not 1 <= bits <= 32:
            raise UnsupportedAudioError(f"unsupported AIFF sample width {bits}")
        n = len(raw) // width
        if width == 1:
            v = np.frombuffer(raw, dtype=np.int8).astype(np.int32)
        elif width == 2:
            v = np.frombuffer(raw[: n * 2], dtype=order + "i2").astype(np.int32)
        elif width == 3:
            b = np.frombuffer(raw[: n * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = (b[:, 0] << 16 | b[:, 1] << 8 | b[:, 2]) if order == ">" else (b[:, 2] << 16 | b[:, 1] << 8 | b[:, 0])
            v = (v ^ 0x800000) - 0x800000
        elif width == 4:
            v = np.frombuffer(raw[: n * 4], dtype=order + "i4").astype(np.int64)
        else:
            raise UnsupportedAudioError(f"unsupported AIFF sample width {bits}")
        x = (v / float(1 << (8 * width - 1))).astype(np.float32)
    else:
        raise UnsupportedAudioError(f"compressed AIFF-C ({kind.decode('latin-1')}) is not supported")
    n = min(x.shape[0] // channels, frames) if frames else x.shape[0] // channels
    return x[: n * channels].reshape(n, channels), rate


def _ogg_holds_flac(head: bytes) -> bool:
    """The first page of an Ogg file opens a FLAC stream (first packet 0x7f "FLAC": the FLAC format's Ogg mapping)."""
    return head[:4] == b"OggS" and len(head) > 27 and head[27 + head[26]: 32 + head[26]] == b"\x7fFLAC"


def is_flac(path) -> bool:
    """A native FLAC stream (behind an ID3 tag only under its own suffix: the tag is an MP3's too) or FLAC in Ogg."""
    with open(path, "rb") as f:
        head = f.read(27 + 255 + 5)
    return head[:4] == b"fLaC" or (head[:3] == b"ID3" and Path(path).suffix.lower() == ".flac") or _ogg_holds_flac(head)


def _flac_bytes(path, mapped: bool = False) -> np.ndarray:
    """The file's native FLAC stream as uint8: the file itself (memory-mapped when asked), or what an Ogg container
    wraps (`oasr_ogg_flac_unwrap`: the packets' bytes in order)."""
    with open(path, "rb") as f:
        head = f.read(27 + 255 + 5)
    if not _ogg_holds_flac(head):
        return np.memmap(path, dtype=np.uint8, mode="r") if mapped else np.fromfile(path, dtype=np.uint8)
    import ctypes as C

    from . import _hip
    wrapped = np.fromfile(path, dtype=np.uint8)
    out = np.empty(wrapped.size, dtype=np.uint8)
    written = C.c_int64()
    try:
        _hip.check(_hip.lib().oasr_ogg_flac_unwrap(wrapped.ctypes.data, wrapped.size, out.ctypes.data, out.size,
                                                   C.byref(written)), "oasr_ogg_flac_unwrap")
    except _hip.OasrError as e:
        raise UnsupportedAudioError(str(e)) from e
    return out[: written.value]


def read_flac(path) -> tuple[np.ndarray, int]:
    """Decode a FLAC file (native or in an Ogg container) with the native decoder -> (float32 [frames, channels] in
    [-1, 1), rate)."""
    import ctypes as C

    from . import _hip
    buf = _flac_bytes(path)
    rate, channels, bits, total = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    L = _hip.lib()
    _hip.check(L.oasr_flac_info(buf.ctypes.data, buf.size, C.byref(rate), C.byref(channels), C.byref(bits),
                                C.byref(total), None), "oasr_flac_info")
    if total.value > buf.size * 8192:        # (a frame of 65 535 constant samples still takes about ten bytes)
        raise UnsupportedAudioError("FLAC header declares %d samples, more than %d bytes can hold" % (total.value, buf.size))
    out = np.empty((max(total.value, 1), channels.value), dtype=np.int32)
    written = C.c_int64()
    _hip.check(L.oasr_flac_decode(buf.ctypes.data, buf.size, out.ctypes.data, out.shape[0], C.byref(written)),
               "oasr_flac_decode")
    x = out[: written.value].astype(np.float32) * np.float32(1.0 / (1 << (bits.value - 1)))
    return x, int(rate.value)


def read_flac_pcm(path):
    """(interleaved integer samples, channels, sample_rate, bytes per sample, bits) of a FLAC file, or None when
    it is not one: int16 scaled to 16 bits for streams of at most 16 bits per sample (bits = 16), else int32 at
    the stream's depth.  A sample's value in [-1, 1) is the integer / 2^(bits - 1)."""
    import ctypes as C

    from . import _hip
    if not is_flac(path):
        return None
    buf = _flac_bytes(path)
    rate, channels, bits, total = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    L = _hip.lib()
    _hip.check(L.oasr_flac_info(buf.ctypes.data, buf.size, C.byref(rate), C.byref(channels), C.byref(bits),
                                C.byref(total), None), "oasr_flac_info")
    if total.value > buf.size * 8192:        # (a frame of 65 535 constant samples still takes about ten bytes)
        raise UnsupportedAudioError("FLAC header declares %d samples, more than %d bytes can hold" % (total.value, buf.size))
    narrow = bits.value <= 16
    out = np.empty((max(total.value, 1) * channels.value,), dtype=np.int16 if narrow else np.int32)
    written = C.c_int64()
    fn, what = (L.oasr_flac_decode16, "oasr_flac_decode16") if narrow else (L.oasr_flac_decode, "oasr_flac_decode")
    _hip.check(fn(buf.ctypes.data, buf.size, out.ctypes.data, max(total.value, 1), C.byref(written)), what)
    return (out[: written.value * channels.value], int(channels.value), int(rate.value),
            2 if narrow else 4, 16 if narrow else int(bits.value))


def read_flac_pcm16(path):
    """(int16 interleaved, channels, sample_rate) of a FLAC file with at most 16 bits per sample, else None."""
    if not is_flac(path):
        return None
    import ctypes as C

    from . import _hip
    buf = _flac_bytes(path)
    rate, channels, bits, total = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    _hip.check(_hip.lib().oasr_flac_info(buf.ctypes.data, buf.size, C.byref(rate), C.byref(channels), C.byref(bits),
                                         C.byref(total), None), "oasr_flac_info")
    if bits.value > 16:
        return None
    m = read_flac_pcm(path)
    return m[0], m[1], m[2]


class FlacStream:
    """A FLAC file decoded by sample ranges: `read_pcm(first, count)` returns the interleaved integer samples
    [first, first + count) per channel exactly as `read_flac_pcm` delivers them, decoding only the frames that
    cover the range.  The file is memory-mapped, so a reader that asks for one part of a long file touches only
    that part's pages.  `sample_bytes` / `bits` say how a sample is stored (int16 scaled to 16 bits for streams
    of at most 16 bits, else int32 at the stream's depth)."""

    def __init__(self, path):
        import ctypes as C

        from . import _hip
        self._lib = _hip.lib()
        self._buf = _flac_bytes(path, mapped=True)       # (FLAC in Ogg: the unwrapped stream, held in memory)
        rate, channels, bits, total = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        try:
            _hip.check(self._lib.oasr_flac_info(self._buf.ctypes.data, self._buf.size, C.byref(rate), C.byref(channels),
                                                C.byref(bits), C.byref(total), None), "oasr_flac_info")
        except _hip.OasrError as e:
            raise UnsupportedAudioError(str(e)) from e
        if total.value > self._buf.size * 8192:
            raise UnsupportedAudioError("FLAC header declares %d samples, more than %d bytes can hold"
                                        % (total.value, self._buf.size))
        self.rate, self.channels, self.samples = int(rate.value), int(channels.value), int(total.value)
        narrow = bits.value <= 16
        self.sample_bytes, self.bits = (2, 16) if narrow else (4, int(bits.value))
        # ranges can be located only when the frame headers give sample positions (a stated length, and a
        # fixed block size or sample-numbered frames): one sample is asked for to find out
        self.seekable = False
        if self.samples > 0:
            try:
                self.read_pcm(0, 1)
                self.seekable = True
            except _hip.OasrError:
                pass

    def read_pcm(self, first: int, count: int) -> np.ndarray:
        from . import _hip
        out = np.empty((int(count) * self.channels,), dtype=np.int16 if self.sample_bytes == 2 else np.int32)
        if count:
            _hip.check(self._lib.oasr_flac_decode_range(self._buf.ctypes.data, self._buf.size, int(first), int(count),
                                                        out.ctypes.data, 8 * self.sample_bytes), "oasr_flac_decode_range")
        return out

    def close(self) -> None:
        self._buf = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _looks_like_mp3(head: bytes, path) -> bool:
    """An ID3v2 tag (unless the name says FLAC, which may carry one too) or an MPEG audio Layer III frame header."""
    if head[:3] == b"ID3":
        return Path(path).suffix.lower() != ".flac"
    if head[:4] in _RIFF_MAGICS:      # MPEG Layer III in a WAVE container: the fmt chunk says so
        with open(path, "rb") as f:
            blob = f.read(4096)
        at = blob.find(b"fmt ")
        return at >= 0 and len(blob) >= at + 10 and struct.unpack_from("<H", blob, at + 8)[0] == 0x55
    return (len(head) >= 4 and head[0] == 0xFF and head[1] & 0xE0 == 0xE0 and (head[1] >> 3) & 3 != 1
            and (head[1] >> 1) & 3 == 1 and head[2] >> 4 not in (0, 15) and (head[2] >> 2) & 3 != 3)


def mp3_info(path) -> tuple[int, int, int]:
    """(sample_rate, channels, samples per channel after the encoder delay and padding are trimmed) of an MP3 file."""
    import ctypes as C

    from . import _hip
    buf = _mp3_bytes(path)
    rate, channels, total = C.c_int32(), C.c_int32(), C.c_int64()
    try:
        _hip.check(_hip.lib().oasr_mp3_info(buf.ctypes.data, buf.size, C.byref(rate), C.byref(channels),
                                            C.byref(total)), "oasr_mp3_info")
    except _hip.OasrError as e:
        raise UnsupportedAudioError(str(e)) from e
    return int(rate.value), int(channels.value), int(total.value)


def _mp3_bytes(path) -> np.ndarray:
    """The file's bytes; for MPEG Layer III wrapped in RIFF/WAVE (format tag 0x55) the payload of its data chunk."""
    buf = np.fromfile(path, dtype=np.uint8)
    if buf.size >= 12 and bytes(buf[:4]) in _RIFF_MAGICS and bytes(buf[8:12]) == b"WAVE":
        pos = 12
        while pos + 8 <= buf.size:
            n = int.from_bytes(bytes(buf[pos + 4: pos + 8]), "little")
            if bytes(buf[pos: pos + 4]) == b"data":
                return buf[pos + 8: pos + 8 + n]
            pos += 8 + n + (n & 1)
    return buf


def read_mp3(path, diagnostics: bool = False):
    """Decode an MPEG Layer III file with the native decoder -> (float32 [frames, channels], rate); the samples are
    nominally in [-1, 1) and not clipped.  With `diagnostics` a third value: int64 [5] = audio frames, bytes stepped
    over while searching for frames, granules whose spectrum overran, frames or granules delivered as silence,
    stuffing bits."""
    import ctypes as C

    from . import _hip
    buf = _mp3_bytes(path)
    rate, channels, total = C.c_int32(), C.c_int32(), C.c_int64()
    L = _hip.lib()
    try:
        _hip.check(L.oasr_mp3_info(buf.ctypes.data, buf.size, C.byref(rate), C.byref(channels), C.byref(total)),
                   "oasr_mp3_info")
        out = np.zeros((total.value, channels.value), dtype=np.float32)
        written, diag = C.c_int64(), np.zeros(5, dtype=np.int64)
        _hip.check(L.oasr_mp3_decode(buf.ctypes.data, buf.size, out.ctypes.data, total.value, C.byref(written),
                                     diag.ctypes.data), "oasr_mp3_decode")
    except _hip.OasrError as e:
        raise UnsupportedAudioError(str(e)) from e
    out = out[: written.value]
    return (out, int(rate.value), diag) if diagnostics else (out, int(rate.value))


def read_mp3_mono(path) -> tuple[np.ndarray, int]:
    """(float32 [frames], rate): `read_mp3` with the channels averaged inside the decoder - the values
    `read_mp3(path)[0].mean(axis=1, dtype=float64)` rounds to, without the two-channel array."""
    import ctypes as C

    from . import _hip
    buf = _mp3_bytes(path)
    rate, channels, total = C.c_int32(), C.c_int32(), C.c_int64()
    L = _hip.lib()
    try:
        _hip.check(L.oasr_mp3_info(buf.ctypes.data, buf.size, C.byref(rate), C.byref(channels), C.byref(total)),
                   "oasr_mp3_info")
        out = np.zeros((total.value,), dtype=np.float32)
        written = C.c_int64()
        _hip.check(L.oasr_mp3_decode_mono(buf.ctypes.data, buf.size, out.ctypes.data, total.value, C.byref(written),
                                          None), "oasr_mp3_decode_mono")
    except _hip.OasrError as e:
        raise UnsupportedAudioError(str(e)) from e
    return out[: written.value], int(rate.value)


class Mp3Stream:
    """An MP3 file opened once (frames found and counted) and decoded by sample ranges: `read_mono(first, count)`
    returns the samples [first, first + count) of what `read_mp3_mono` delivers, bit for bit, so a caller can work on
    one part of a long file while the next is being decoded.  Ranges may be read from several threads at once."""

    def __init__(self, path):
        import ctypes as C

        from . import _hip
        self._handle = None
        self._lib = _hip.lib()
        self._buf = _mp3_bytes(path)             # must outlive the handle
        rate, channels, total = C.c_int32(), C.c_int32(), C.c_int64()
        self._handle = C.c_void_p()
        try:
            _hip.check(self._lib.oasr_mp3_open(self._buf.ctypes.data, self._buf.size, C.byref(self._handle),
                                               C.byref(rate), C.byref(channels), C.byref(total)), "oasr_mp3_open")
        except _hip.OasrError as e:
            raise UnsupportedAudioError(str(e)) from e
        self.rate, self.channels, self.samples = int(rate.value), int(channels.value), int(total.value)

    def read_mono(self, first: int, count: int) -> np.ndarray:
        from . import _hip
        if not self._handle:
            raise ValueError("the stream is closed")
        out = np.zeros((int(count),), dtype=np.float32)
        _hip.check(self._lib.oasr_mp3_decode_mono_range(self._handle, int(first), int(count), out.ctypes.data),
                   "oasr_mp3_decode_mono_range")
        return out

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.oasr_mp3_close(self._handle)
        self._handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()


_OTHER_MAGICS = _RIFF_MAGICS + (b"FORM", b"fLaC", b"OggS")


def _is_mp3_behind_junk(head: bytes, path) -> bool:
    """A file whose first bytes name no container at all may still be an MP3 with padding or a foreign tag block (APE,
    a run of zeros) in front of its first frame: the native frame scan steps over such bytes and accepts a stream only
    on two chained frame headers, so it is asked."""
    if head[:4] in _OTHER_MAGICS or head[4:8] == b"ftyp":
        return False
    try:
        mp3_info(path)
        return True
    except UnsupportedAudioError:
        return False


def is_mp3(path) -> bool:
    with open(path, "rb") as f:
        head = f.read(12)
    return _looks_like_mp3(head, path) or _is_mp3_behind_junk(head, path)


def is_ogg(path) -> bool:
    """An Ogg file for the Vorbis decoder (FLAC in Ogg goes with FLAC: `is_flac`)."""
    with open(path, "rb") as f:
        head = f.read(27 + 255 + 5)
    return head[:4] == b"OggS" and not _ogg_holds_flac(head)


def _vorbis_call(path, mono: bool):
    import ctypes as C

    from . import _hip
    buf = np.fromfile(path, dtype=np.uint8)
    rate, channels, total = C.c_int32(), C.c_int32(), C.c_int64()
    L = _hip.lib()
    try:
        _hip.check(L.oasr_vorbis_info(buf.ctypes.data, buf.size, C.byref(rate), C.byref(channels), C.byref(total)),
                   "oasr_vorbis_info")
        if mono is None:
            return None, int(rate.value), int(channels.value), int(total.value)
        out = np.zeros((total.value,) if mono else (total.value, channels.value), dtype=np.float32)
        written = C.c_int64()
        fn = L.oasr_vorbis_decode_mono if mono else L.oasr_vorbis_decode
        _hip.check(fn(buf.ctypes.data, buf.size, out.ctypes.data, total.value, C.byref(written), 0), "oasr_vorbis_decode")
    except _hip.OasrError as e:
        raise UnsupportedAudioError(str(e)) from e
    return out[: written.value], int(rate.value), int(channels.value), int(total.value)


def vorbis_info(path) -> tuple[int, int, int]:
    """(sample_rate, channels, samples per channel as the granule positions state them) of an Ogg Vorbis file."""
    _, rate, channels, total = _vorbis_call(path, None)
    return rate, channels, total


def read_ogg(path) -> tuple[np.ndarray, int]:
    """Decode the first Vorbis stream of an Ogg file with the native decoder -> (float32 [frames, channels], rate);
    not clipped.  Opus and Speex in an Ogg container are refused by name (FLAC in Ogg: `read_flac`)."""
    out, rate, _, _ = _vorbis_call(path, False)
    return out, rate


def read_ogg_mono(path) -> tuple[np.ndarray, int]:
    """(float32 [frames], rate): `read_ogg` with the channels averaged inside the decoder (mean formed in double)."""
    out, rate, _, _ = _vorbis_call(path, True)
    return out, rate


class VorbisStream:
    """An Ogg Vorbis file opened once (headers parsed, packets placed on the sample axis) and decoded by sample
    ranges: `read_mono(first, count)` returns the samples [first, first + count) of what `read_ogg_mono` delivers, bit
    for bit.  Ranges may be read from several threads at once."""

    def __init__(self, path):
        import ctypes as C

        from . import _hip
        self._handle = None
        self._lib = _hip.lib()
        buf = np.fromfile(path, dtype=np.uint8)      # the opened stream keeps its own copy of the packets
        rate, channels, total = C.c_int32(), C.c_int32(), C.c_int64()
        self._handle = C.c_void_p()
        try:
            _hip.check(self._lib.oasr_vorbis_open(buf.ctypes.data, buf.size, C.byref(self._handle), C.byref(rate),
                                                  C.byref(channels), C.byref(total)), "oasr_vorbis_open")
        except _hip.OasrError as e:
            self._handle = None
            raise UnsupportedAudioError(str(e)) from e
        self.rate, self.channels, self.samples = int(rate.value), int(channels.value), int(total.value)

    def read_mono(self, first: int, count: int) -> np.ndarray:
        from . import _hip
        if not self._handle:
            raise ValueError("the stream is closed")
        out = np.zeros((int(count),), dtype=np.float32)
        _hip.check(self._lib.oasr_vorbis_decode_mono_range(self._handle, int(first), int(count), out.ctypes.data),
                   "oasr_vorbis_decode_mono_range")
        return out

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.oasr_vorbis_close(self._handle)
        self._handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()


def read_audio(path) -> tuple[np.ndarray, int]:
    """Decode by content: (float32 [frames, channels], sample_rate)."""
    with open(path, "rb") as f:
        head = f.read(27 + 255 + 5)
    if head[:4] in _RIFF_MAGICS:
        return read_mp3(path) if _looks_like_mp3(head, path) else read_wav(path)
    if head[:4] == b"FORM":
        return read_aiff(path)
    if head[:4] == b"fLaC" or head[:3] == b"ID3" and Path(path).suffix.lower() == ".flac" or _ogg_holds_flac(head):
        return read_flac(path)
    if head[:4] == b"OggS":
        return read_ogg(path)
    if _looks_like_mp3(head, path) or _is_mp3_behind_junk(head, path):
        return read_mp3(path)
    raise UnsupportedAudioError(
        f"cannot decode '{Path(path).suffix}' audio: WAV, AIFF, FLAC (also in Ogg), MP3 and Ogg Vorbis input is supported")


def load_audio_16k(path, resampler=None, keep_int16: bool = False) -> np.ndarray:
    """File -> mono float32 at 16 kHz (channels averaged).  `resampler(mono, rate) -> float32 array`
    replaces the host polyphase filter when given (the device engine passes its own).
    With `keep_int16`, a mono 16-bit PCM WAV that is already at 16 kHz comes back as the int16 samples
    themselves (value = integer / 32768, exactly what the float path computes): the caller converts while
    it copies chunks into batches, so a long file is neither converted nor copied as a whole."""
    if keep_int16:
        with open(path, "rb") as f:
            is_riff = f.read(4) in _RIFF_MAGICS
        fast = read_wav_pcm16_mono(path) if is_riff else None
        if fast is not None and fast[1] == TARGET_RATE:
            return fast[0]
    if is_ogg(path):
        mono, rate = read_ogg_mono(path)
    elif is_mp3(path):
        mono, rate = read_mp3_mono(path)
    else:
        x, rate = read_audio(path)
        mono = x.mean(axis=1, dtype=np.float64) if x.shape[1] > 1 else x[:, 0]
    if resampler is not None and rate != TARGET_RATE:
        return np.asarray(resampler(np.asarray(mono, dtype=np.float32), rate), dtype=np.float32)
    return resample_poly(mono, rate, TARGET_RATE)
