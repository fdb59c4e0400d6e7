"""CPU restatement of Ogg Vorbis decoding - test infrastructure only (imported by tests/, never by the package).

The reference accepts `.ogg` uploads (ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:106-118,
ref:workflows/wav2elan_web/app.py:76-77) and forwards them to its remote model: it holds no decoder, so this follows
the published format instead - the Vorbis I specification (Xiph.Org; section numbers below) and RFC 3533 for the
container.  Written for clarity, not speed: bits are read one integer at a time, codewords are matched in a
dictionary, floor curves are rendered point by point, the inverse MDCT is the cosine sum of its definition.  It is the
checker for `csrc/vorbis.cpp` on streams a real encoder wrote, whose samples nothing else in this repository can
state.  Parity unpinned against the reference (no such code there)."""
from __future__ import annotations

import math
import struct

import numpy as np


class Bits:
    """§2.1: least-significant bit first.  Reading past the end sets `eop` and returns 0."""

    def __init__(self, data: bytes):
        self.v = int.from_bytes(data, "little")
        self.n = 8 * len(data)
        self.p = 0
        self.eop = False

    def get(self, k: int) -> int:
        if k == 0:
            return 0
        if self.p + k > self.n:
            self.eop = True
            self.p = self.n
            return 0
        r = (self.v >> self.p) & ((1 << k) - 1)
        self.p += k
        return r


def ilog(v: int) -> int:
    return v.bit_length() if v > 0 else 0


def float32_unpack(x: int) -> float:
    """§9.2.2"""
    mant = x & 0x1FFFFF
    if x & 0x80000000:
        mant = -mant
    return math.ldexp(mant, ((x & 0x7FE00000) >> 21) - 788)


def lookup1_values(entries: int, dims: int) -> int:
    """§9.2.3: the greatest r with r ** dims <= entries."""
    r = 0
    while (r + 1) ** dims <= entries:
        r += 1
    return r


def packets_of(data: bytes):
    """RFC 3533: the packets of the first logical stream whose first packet is a Vorbis identification header;
    (packets, granule position of the last page)."""
    pos, serial, cur, packets, granule = 0, None, b"", [], 0
    while pos + 27 <= len(data) and data[pos:pos + 4] == b"OggS":
        htype, gp, ser, _, _, nseg = struct.unpack_from("<xBqIIIB", data, pos + 4)
        table = data[pos + 27:pos + 27 + nseg]
        body = pos + 27 + nseg
        if serial is None and data[body:body + 7] == b"\x01vorbis":
            serial = ser
        if ser == serial:
            for l in table:
                cur += data[body:body + l]
                body += l
                if l < 255:
                    packets.append(cur)
                    cur = b""
            granule = gp
            if htype & 4:
                break
        pos += 27 + nseg + sum(table)
    return packets, granule


class Codebook:
    """§3.2.1: lengths -> codewords (each entry takes the lowest free codeword of its length, the tree filled from the
    left), and the vector of every entry for lookup types 1 and 2."""

    def __init__(self, b: Bits):
        assert b.get(24) == 0x564342
        self.dims, self.entries = b.get(16), b.get(24)
        lengths = [0] * self.entries
        if not b.get(1):
            sparse = b.get(1)
            for e in range(self.entries):
                if sparse and not b.get(1):
                    continue
                lengths[e] = b.get(5) + 1
        else:
            cur, length = 0, b.get(5) + 1
            while cur < self.entries:
                num = b.get(ilog(self.entries - cur))
                for k in range(num):
                    lengths[cur + k] = length
                cur += num
                length += 1
        self.lookup = b.get(4)
        self.vectors = None
        if self.lookup:
            minv, delta = float32_unpack(b.get(32)), float32_unpack(b.get(32))
            bits, seq = b.get(4) + 1, b.get(1)
            n = lookup1_values(self.entries, self.dims) if self.lookup == 1 else self.entries * self.dims
            mult = [b.get(bits) for _ in range(n)]
            self.vectors = []
            for e in range(self.entries):
                last, div, vec = 0.0, 1, []
                for i in range(self.dims):
                    m = mult[(e // div) % n] if self.lookup == 1 else mult[e * self.dims + i]
                    # float32 arithmetic, as a decoder's tables are
                    val = float(np.float32(np.float32(np.float32(m) * np.float32(delta)) + np.float32(minv)) + np.float32(last))
                    vec.append(val)
                    if seq:
                        last = val
                    if self.lookup == 1:
                        div *= n
                self.vectors.append(vec)
        # codewords as bit strings, first bit read first
        self.words = {}
        used = [e for e in range(self.entries) if lengths[e]]
        if len(used) == 1:
            self.single = (used[0], lengths[used[0]])
            return
        self.single = None
        free = [""]
        for e in used:
            length = lengths[e]
            cands = [p for p in free if len(p) <= length]
            p = min(cands, key=lambda s: s + "0" * (length - len(s)))
            free.remove(p)
            w = p + "0" * (length - len(p))
            for k in range(len(p), length):
                free.append(w[:k] + "1")
            self.words[w] = e
        assert not free, "the lengths do not fill the code tree"

    def decode(self, b: Bits):
        if self.single:
            b.get(self.single[1])
            return None if b.eop else self.single[0]
        w = ""
        while True:
            w += str(b.get(1))
            if b.eop:
                return None
            if w in self.words:
                return self.words[w]
            assert len(w) <= 32


class Floor1:
    """§7.2.2 header, §7.2.3 packet decode, §7.2.4 curve computation."""

    def __init__(self, b: Bits):
        parts = b.get(5)
        self.part_class = [b.get(4) for _ in range(parts)]
        self.cdim, self.csub, self.cmaster, self.cbooks = {}, {}, {}, {}
        for c in range(max(self.part_class) + 1 if parts else 0):
            self.cdim[c] = b.get(3) + 1
            self.csub[c] = b.get(2)
            if self.csub[c]:
                self.cmaster[c] = b.get(8)
            self.cbooks[c] = [b.get(8) - 1 for _ in range(1 << self.csub[c])]
        self.mult = b.get(2) + 1
        rangebits = b.get(4)
        self.x = [0, 1 << rangebits]
        for c in self.part_class:
            for _ in range(self.cdim[c]):
                self.x.append(b.get(rangebits))

    def posts(self, b: Bits, books):
        if not b.get(1):
            return None
        rng = [256, 128, 86, 64][self.mult - 1]
        y = [b.get(ilog(rng - 1)), b.get(ilog(rng - 1))]
        for c in self.part_class:
            cbits = self.csub[c]
            cval = 0
            if cbits:
                cval = books[self.cmaster[c]].decode(b)
                if cval is None:
                    return None
            for _ in range(self.cdim[c]):
                book = self.cbooks[c][cval & ((1 << cbits) - 1)]
                cval >>= cbits
                if book >= 0:
                    v = books[book].decode(b)
                    if v is None:
                        return None
                    y.append(v)
                else:
                    y.append(0)
        if b.eop:
            return None
        return y

    @staticmethod
    def _point(x0, y0, x1, y1, x):
        dy, adx = y1 - y0, x1 - x0
        off = abs(dy) * (x - x0) // adx
        return y0 - off if dy < 0 else y0 + off

    def curve(self, y, n):
        x = self.x
        rng = [256, 128, 86, 64][self.mult - 1]
        final, flag = [y[0], y[1]], [True, True]
        for i in range(2, len(x)):
            lo = max((j for j in range(i) if x[j] < x[i]), key=lambda j: x[j])
            hi = min((j for j in range(i) if x[j] > x[i]), key=lambda j: x[j])
            pred = self._point(x[lo], final[lo], x[hi], final[hi], x[i])
            val, highroom, lowroom = y[i], rng - pred, pred
            room = min(highroom, lowroom) * 2
            if val:
                flag[lo] = flag[hi] = True
                flag.append(True)
                if val >= room:
                    final.append(val - lowroom + pred if highroom > lowroom else pred - val + highroom - 1)
                else:
                    final.append(pred - (val + 1) // 2 if val & 1 else pred + val // 2)
            else:
                flag.append(False)
                final.append(pred)
        order = sorted(range(len(x)), key=lambda i: x[i])
        out = [0] * max(n, 1)

        def line(x0, y0, x1, y1):
            dy, adx = y1 - y0, x1 - x0
            ady = abs(dy)
            base = int(dy / adx)
            sy = base - 1 if dy < 0 else base + 1
            ady -= abs(base) * adx
            xx, yy, err = x0, y0, 0
            if xx < n:
                out[xx] = yy
            for xx in range(x0 + 1, x1):
                err += ady
                if err >= adx:
                    err -= adx
                    yy += sy
                else:
                    yy += base
                if xx < n:
                    out[xx] = yy

        lx, ly = 0, final[order[0]] * self.mult
        hx, hy = 0, ly
        for i in order[1:]:
            if flag[i]:
                hy, hx = final[i] * self.mult, x[i]
                line(lx, ly, hx, hy)
                lx, ly = hx, hy
        if hx < n:
            line(hx, hy, n, hy)
        # §10.1: the table of 256 amplitudes is 10 ** ((i - 255) * 35 / 64 / 20), listed to 8 digits
        return np.array([np.float32(10.0 ** ((v - 255) * 0.02734375)) for v in out[:n]], dtype=np.float32)


class Residue:
    """§8.6.1 header, §8.6.2 packet decode (formats 0 / 1; type 2 interleaves the channels into one vector)."""

    def __init__(self, b: Bits, kind: int):
        self.kind = kind
        self.begin, self.end, self.psize = b.get(24), b.get(24), b.get(24) + 1
        self.nclass, self.classbook = b.get(6) + 1, b.get(8)
        cascade = []
        for _ in range(self.nclass):
            low = b.get(3)
            high = b.get(5) if b.get(1) else 0
            cascade.append(high * 8 + low)
        self.books = [[b.get(8) if c >> j & 1 else -1 for j in range(8)] for c in cascade]

    def decode(self, b: Bits, books, n: int, skip: list[bool]):
        ch = len(skip)
        if self.kind == 2:
            if all(skip):
                return [np.zeros(n, dtype=np.float32) for _ in range(ch)]
            big = self._decode(b, books, n * ch, [False], 1)[0]
            return [big[c::ch].copy() for c in range(ch)]
        return self._decode(b, books, n, skip, self.kind)

    def _decode(self, b: Bits, books, n: int, skip, fmt: int):
        out = [np.zeros(n, dtype=np.float32) for _ in skip]
        lb, le = min(self.begin, n), min(self.end, n)
        cw = books[self.classbook].dims
        parts = (le - lb) // self.psize
        cls = [[0] * (parts + cw) for _ in skip]
        for pas in range(8):
            pc = 0
            while pc < parts:
                if pas == 0:
                    for j in range(len(skip)):
                        if skip[j]:
                            continue
                        t = books[self.classbook].decode(b)
                        if t is None:
                            return out
                        for i in range(cw - 1, -1, -1):
                            cls[j][pc + i] = t % self.nclass
                            t //= self.nclass
                for _ in range(cw):
                    if pc >= parts:
                        break
                    for j in range(len(skip)):
                        if skip[j]:
                            continue
                        book = self.books[cls[j][pc]][pas]
                        if book < 0:
                            continue
                        bk = books[book]
                        at = lb + pc * self.psize
                        if fmt == 0:
                            step = self.psize // bk.dims
                            for k in range(step):
                                e = bk.decode(b)
                                if e is None:
                                    return out
                                for d in range(bk.dims):
                                    out[j][at + k + d * step] += np.float32(bk.vectors[e][d])
                        else:
                            k = 0
                            while k < self.psize:
                                e = bk.decode(b)
                                if e is None:
                                    return out
                                for d in range(bk.dims):
                                    out[j][at + k] += np.float32(bk.vectors[e][d])
                                    k += 1
                    pc += 1
        return out


def slope(n: int) -> np.ndarray:
    """§4.3.1: the rising half of the window over n samples."""
    i = np.arange(n)
    return np.sin(np.pi / 2 * np.sin((i + 0.5) / n * np.pi / 2) ** 2)


def decode(data: bytes) -> tuple[np.ndarray, int]:
    """The whole stream -> (float64 [samples][channels], rate): §4.2 headers, §4.3 audio packets."""
    packets, granule = packets_of(data)
    ident = packets[0]
    assert ident[:7] == b"\x01vorbis"
    _, ch, rate, _, _, _, bsz = struct.unpack_from("<IBIiiiB", ident, 7)
    bs = [1 << (bsz & 15), 1 << (bsz >> 4)]
    b = Bits(packets[2][7:])
    books = [Codebook(b) for _ in range(b.get(8) + 1)]
    for _ in range(b.get(6) + 1):
        assert b.get(16) == 0
    floors = []
    for _ in range(b.get(6) + 1):
        assert b.get(16) == 1, "floor 0 is not restated"
        floors.append(Floor1(b))
    residues = []
    for _ in range(b.get(6) + 1):
        kind = b.get(16)
        residues.append(Residue(b, kind))
    mappings = []
    for _ in range(b.get(6) + 1):
        assert b.get(16) == 0
        submaps = b.get(4) + 1 if b.get(1) else 1
        steps = []
        if b.get(1):
            for _ in range(b.get(8) + 1):
                steps.append((b.get(ilog(ch - 1)), b.get(ilog(ch - 1))))
        assert b.get(2) == 0
        mux = [b.get(4) for _ in range(ch)] if submaps > 1 else [0] * ch
        sub = []
        for _ in range(submaps):
            b.get(8)
            sub.append((b.get(8), b.get(8)))
        mappings.append((steps, mux, sub))
    modes = []
    for _ in range(b.get(6) + 1):
        flag = b.get(1)
        b.get(16)
        b.get(16)
        modes.append((flag, b.get(8)))
    assert b.get(1) == 1 and not b.eop

    cos = {}

    def imdct(spec: np.ndarray) -> np.ndarray:
        m = len(spec)
        if m not in cos:
            t = np.arange(2 * m)[:, None] + 0.5 + m / 2
            k = np.arange(m)[None, :] + 0.5
            cos[m] = np.cos(np.pi / m * t * k)
        return cos[m] @ spec.astype(np.float64)

    out = []              # finished samples [n][ch]
    prev = None           # right half of the previous windowed block
    for pkt in packets[3:]:
        b = Bits(pkt)
        if b.get(1) != 0:
            continue
        mode = b.get(ilog(len(modes) - 1))
        if b.eop or mode >= len(modes):
            continue
        flag, mp = modes[mode]
        n = bs[flag]
        prev_long = next_long = 1
        if flag:
            prev_long, next_long = b.get(1), b.get(1)
        steps, mux, sub = mappings[mp]
        ys = [floors[sub[mux[c]][0]].posts(b, books) for c in range(ch)]
        unused = [y is None for y in ys]
        nores = list(unused)
        for mag, ang in steps:
            if not nores[mag] or not nores[ang]:
                nores[mag] = nores[ang] = False
        spec = [None] * ch
        for s, (_, res) in enumerate(sub):
            chans = [c for c in range(ch) if mux[c] == s]
            vecs = residues[res].decode(b, books, n // 2, [nores[c] for c in chans])
            for c, v in zip(chans, vecs):
                spec[c] = v
        for mag, ang in reversed(steps):
            m, a = spec[mag].copy(), spec[ang].copy()
            nm = np.where(m > 0, np.where(a > 0, m, m + a), np.where(a > 0, m, m - a))
            na = np.where(m > 0, np.where(a > 0, m - a, m), np.where(a > 0, m + a, m))
            spec[mag], spec[ang] = nm.astype(np.float32), na.astype(np.float32)
        w = np.zeros(n)
        ls, ln = (n // 4 - bs[0] // 4, bs[0] // 2) if flag and not prev_long else (0, n // 2)
        rs, rn = (n * 3 // 4 - bs[0] // 4, bs[0] // 2) if flag and not next_long else (n // 2, n // 2)
        w[ls:ls + ln] = slope(ln)
        w[ls + ln:rs] = 1.0
        w[rs:rs + rn] = slope(rn)[::-1]
        block = np.zeros((n, ch))
        for c in range(ch):
            if unused[c]:
                continue
            curve = floors[sub[mux[c]][0]].curve(ys[c], n // 2)
            block[:, c] = imdct((spec[c] * curve).astype(np.float32)) * w
        if prev is not None:
            pn = 2 * prev.shape[0]
            length = pn // 4 + n // 4
            seg = np.zeros((length, ch))
            seg[:min(length, pn // 2)] += prev[:min(length, pn // 2)]
            shift = n // 2 - length
            lo = max(0, -shift)
            seg[lo:] += block[lo + shift: n // 2]
            out.append(seg)
        prev = block[n // 2:].copy()
    pcm = np.concatenate(out) if out else np.zeros((0, ch))
    if 0 <= granule < pcm.shape[0]:
        pcm = pcm[:granule]
    return pcm, rate
