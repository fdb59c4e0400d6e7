"""Test helper: long FLAC streams made of a real encoder's frames.  The frames of a short file are repeated
with fresh frame numbers (header CRC-8 and frame CRC-16 recomputed), so a stream of any length holds the
LPC subframes and partitioned Rice codes a real encoder writes, and its decoded samples are known: the
short file's, repeated."""
import struct

import numpy as np


def _tables():
    t8, t16 = [], []
    for v in range(256):
        c, w = v, v << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
            w = ((w << 1) ^ 0x8005) & 0xFFFF if w & 0x8000 else (w << 1) & 0xFFFF
        t8.append(c)
        t16.append(w)
    return t8, t16


_T8, _T16 = _tables()


def crc8(data) -> int:
    c = 0
    for b in data:
        c = _T8[c ^ b]
    return c


def crc16(data) -> int:
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFF) ^ _T16[(c >> 8) ^ b]
    return c


def split_stream(data: bytes):
    """(STREAMINFO payload, [frame bytes ...]) of a FLAC file whose frames carry frame numbers."""
    assert data[:4] == b"fLaC"
    pos, info = 4, None
    while True:
        last, kind = data[pos] >> 7, data[pos] & 0x7F
        size = int.from_bytes(data[pos + 1: pos + 4], "big")
        if kind == 0:
            info = data[pos + 4: pos + 4 + size]
        pos += 4 + size
        if last:
            break
    frames = []
    while pos < len(data):
        assert data[pos] == 0xFF and data[pos + 1] & 0xFE == 0xF8, "lost sync"
        c, end = 0, None
        history = []                         # CRC-16 of data[pos : pos + i]
        for i in range(pos, len(data)):
            history.append(c)
            c = ((c << 8) & 0xFFFF) ^ _T16[(c >> 8) ^ data[i]]
            nxt = i + 1
            at_sync = nxt == len(data) or (nxt + 1 < len(data) and data[nxt] == 0xFF and data[nxt + 1] & 0xFE == 0xF8)
            if at_sync and nxt - pos > 8 and history[-2] == int.from_bytes(data[nxt - 2: nxt], "big"):
                end = nxt
                break
        assert end is not None, "no frame end found"
        frames.append(data[pos:end])
        pos = end
    return info, frames


def _utf8(n: int) -> bytes:
    if n < 0x80:
        return bytes([n])
    out, lead_bits = [], 6
    while n >= (1 << lead_bits):
        out.append(0x80 | (n & 0x3F))
        n >>= 6
        lead_bits -= 1
    lead = (0xFF << (lead_bits + 1)) & 0xFF
    return bytes([lead | n] + out[::-1])


def renumber(frame: bytes, number: int) -> bytes:
    """The same frame with another frame number."""
    b0 = frame[4]
    extra = 0 if b0 < 0x80 else (1 if b0 < 0xE0 else 2 if b0 < 0xF0 else 3 if b0 < 0xF8 else 4 if b0 < 0xFC else 5 if b0 < 0xFE else 6)
    after = 5 + extra
    bs_code, sr_code = frame[2] >> 4, frame[2] & 15
    tail = (1 if bs_code == 6 else 2 if bs_code == 7 else 0) + (1 if sr_code == 12 else 2 if sr_code in (13, 14) else 0)
    hdr = frame[:4] + _utf8(number) + frame[after: after + tail]
    body = hdr + bytes([crc8(hdr)]) + frame[after + tail + 1: -2]
    return body + struct.pack(">H", crc16(body))


def repeat_stream(data: bytes, repeats: int, decoded: np.ndarray):
    """(bytes of a stream holding the file's full-size frames `repeats` times and then its last frame,
    the samples it decodes to).  `decoded`: the file's samples [frames, channels]."""
    info, frames = split_stream(data)
    block = int.from_bytes(info[2:4], "big")
    assert int.from_bytes(info[0:2], "big") == block, "fixed block size expected"
    x = int.from_bytes(info[10:18], "big")
    full, tail = frames[:-1], frames[-1]
    n_full = len(full) * block
    total = repeats * n_full + (decoded.shape[0] - n_full)
    out, k = [], 0
    for _ in range(repeats):
        for f in full:
            out.append(renumber(f, k))
            k += 1
    out.append(renumber(tail, k))
    x = (x & ~((1 << 36) - 1)) | total
    new_info = info[:10] + x.to_bytes(8, "big") + bytes(16)
    stream = b"fLaC" + bytes([0x80]) + len(new_info).to_bytes(3, "big") + new_info + b"".join(out)
    want = np.concatenate([decoded[:n_full]] * repeats + [decoded[n_full:]], axis=0)
    return stream, want


def wrap_in_ogg(data: bytes, per_page: int = 6, comment: bytes = b"", unknown_total: bool = False) -> bytes:
    """The native stream `data` in an Ogg container, as the FLAC format's Ogg mapping lays it out: first packet
    0x7f "FLAC" 1.0, the count of header packets that follow, "fLaC" and the STREAMINFO block, on a page of its own;
    a VORBIS_COMMENT block as the one further header packet; then one packet per frame, `per_page` frames a page,
    granule position = samples completed.  `unknown_total`: STREAMINFO states no sample count (a streamed encode)."""
    import vorbis_tools as vt
    info, frames = split_stream(data)
    if unknown_total:
        x = int.from_bytes(info[10:18], "big") & ~((1 << 36) - 1)
        info = info[:10] + x.to_bytes(8, "big") + info[18:]
    block = int.from_bytes(info[2:4], "big")
    total = int.from_bytes(info[10:18], "big") & ((1 << 36) - 1)
    vendor = b"flac_tools"
    body = struct.pack("<I", len(vendor)) + vendor + struct.pack("<I", 1 if comment else 0)
    if comment:
        body += struct.pack("<I", len(comment)) + comment
    first = b"\x7fFLAC\x01\x00" + struct.pack(">H", 1) + b"fLaC" + bytes([0x00]) + len(info).to_bytes(3, "big") + info
    packets = [first, bytes([0x84]) + len(body).to_bytes(3, "big") + body] + frames
    granules = [0, 0] + [min((i + 1) * block, total) if total else (i + 1) * block for i in range(len(frames))]
    pages = [1, 1] + [per_page] * (len(frames) // per_page + 1)
    return vt.ogg_pages(packets, granules, serial=0x464C, per_page=pages)
