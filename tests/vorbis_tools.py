"""A small Ogg Vorbis writer for the decoder's tests, written from the format's forward definitions (Vorbis I
specification, Xiph.Org; Ogg: RFC 3533): windowed MDCT, floor-1 posts and their rendered curve, residue vector
quantisation in cascaded passes, channel coupling, codebooks with Huffman lengths, page framing with CRC-32.
It makes no attempt at audio quality (no psychoacoustics); what it provides is a stream whose every field is
known, the quantised spectrum it wrote, and the PCM that spectrum stands for (`Encoded.expected`), computed in
float64 from the definitions."""
from __future__ import annotations

import heapq
import math
import struct
from dataclasses import dataclass, field

import numpy as np


def ilog(v: int) -> int:
    return int(v).bit_length() if v > 0 else 0


class BitWriter:
    """Least-significant bit first, as Vorbis packs."""

    def __init__(self):
        self.acc = 0
        self.n = 0

    def put(self, value: int, bits: int):
        assert 0 <= value < (1 << bits) or bits == 0, (value, bits)
        self.acc |= value << self.n
        self.n += bits

    def word(self, w: str):
        """A codeword: its first (most significant) bit is written first."""
        for c in w:
            self.put(int(c), 1)

    def bytes(self) -> bytes:
        return self.acc.to_bytes((self.n + 7) // 8, "little")


def float32_pack(x: float) -> int:
    """The format's own float: 21-bit integer mantissa, exponent biased by 788, sign bit."""
    if x == 0:
        return 788 << 21
    sign = 0x80000000 if x < 0 else 0
    m, e = abs(x), 0
    while m != int(m):
        m *= 2
        e -= 1
    m = int(m)
    while m >= (1 << 21):
        assert m % 2 == 0, "value needs more than 21 bits"
        m //= 2
        e += 1
    return sign | ((e + 788) << 21) | m


def huffman_lengths(freqs) -> list[int]:
    """Code lengths of a Huffman code for the entries with freq > 0 (0 for the others): a complete prefix code."""
    heap = [(f, i, (i,)) for i, f in enumerate(freqs) if f > 0]
    lengths = [0] * len(freqs)
    if len(heap) == 1:
        lengths[heap[0][1]] = 1
        return lengths
    heapq.heapify(heap)
    tick = len(freqs)
    while len(heap) > 1:
        a = heapq.heappop(heap)
        b = heapq.heappop(heap)
        for i in a[2] + b[2]:
            lengths[i] += 1
        heapq.heappush(heap, (a[0] + b[0], tick, a[2] + b[2]))
        tick += 1
    return lengths


def assign_codewords(lengths) -> list[str | None]:
    """Entries in order; each takes the leftmost free place of its length in the binary tree."""
    free = [""]          # free subtrees, as bit strings
    words: list[str | None] = []
    for L in lengths:
        if L == 0:
            words.append(None)
            continue
        cands = [p for p in free if len(p) <= L]
        assert cands, "lengths overfill the tree"
        p = min(cands, key=lambda s: s + "0" * (L - len(s)))
        free.remove(p)
        w = p + "0" * (L - len(p))
        for k in range(len(p), L):      # the right siblings along the way down stay free
            free.append(w[:k] + "1")
        words.append(w)
    return words


@dataclass
class Book:
    dims: int
    lengths: list[int]
    lookup: int = 0                 # 0: scalar book, 1: lattice, 2: explicit table
    minv: float = 0.0
    delta: float = 1.0
    value_bits: int = 1
    seq: bool = False
    mults: list[int] = field(default_factory=list)
    ordered: bool = False
    sparse: bool = False

    def __post_init__(self):
        self.entries = len(self.lengths)
        self.words = assign_codewords(self.lengths)

    def vector(self, e: int) -> np.ndarray:
        v = np.zeros(self.dims)
        last = 0.0
        if self.lookup == 1:
            lv = len(self.mults)
            div = 1
            for i in range(self.dims):
                v[i] = self.mults[(e // div) % lv] * self.delta + self.minv + last
                if self.seq:
                    last = v[i]
                div *= lv
        else:
            for i in range(self.dims):
                v[i] = self.mults[e * self.dims + i] * self.delta + self.minv + last
                if self.seq:
                    last = v[i]
        return v

    def header(self, bw: BitWriter):
        bw.put(0x564342, 24)
        bw.put(self.dims, 16)
        bw.put(self.entries, 24)
        if self.ordered:
            assert all(b >= a > 0 for a, b in zip(self.lengths, self.lengths[1:]))
            bw.put(1, 1)
            cur, L = 0, self.lengths[0]
            bw.put(L - 1, 5)
            while cur < self.entries:
                num = sum(1 for x in self.lengths[cur:] if x == L)
                bw.put(num, ilog(self.entries - cur))
                cur += num
                L += 1
        else:
            bw.put(0, 1)
            bw.put(1 if self.sparse else 0, 1)
            for L in self.lengths:
                if self.sparse:
                    bw.put(1 if L else 0, 1)
                    if not L:
                        continue
                assert L > 0
                bw.put(L - 1, 5)
        bw.put(self.lookup, 4)
        if self.lookup:
            bw.put(float32_pack(self.minv), 32)
            bw.put(float32_pack(self.delta), 32)
            bw.put(self.value_bits - 1, 4)
            bw.put(1 if self.seq else 0, 1)
            for m in self.mults:
                bw.put(m, self.value_bits)

    def write(self, bw: BitWriter, e: int):
        assert self.words[e] is not None, "unused entry"
        bw.word(self.words[e])

    @property
    def table(self) -> np.ndarray:
        """[entries][dims] vectors; unused entries far away so that a nearest-vector search never picks them."""
        if not hasattr(self, "_table"):
            self._table = np.stack([self.vector(e) if self.lengths[e] else np.full(self.dims, 1e30)
                                    for e in range(self.entries)])
        return self._table

    def nearest(self, groups: np.ndarray):
        """Entry numbers and vectors nearest to each row of `groups` [G][dims]."""
        t = self.table
        d = ((groups[:, None, :] - t[None, :, :]) ** 2).sum(-1)
        e = d.argmin(1)
        return e, t[e]


def lattice_book(levels: int, dims: int, delta: float, rng, ordered=False, drop_unused=False) -> Book:
    """Lookup type 1: `levels` (odd) values -(levels//2)*delta ... per dimension, Huffman lengths from a peaked
    random frequency table."""
    entries = levels ** dims
    half = levels // 2
    idx = np.arange(entries)
    weight = np.ones(entries)
    div = 1
    for _ in range(dims):
        weight *= np.exp(-0.7 * np.abs((idx // div) % levels - half))
        div *= levels
    freqs = weight * (0.5 + rng.random(entries)) + 1e-9
    lengths = huffman_lengths(list(freqs))
    if ordered:
        # the ordered form states lengths as ascending runs: permuting entries is not possible for a lattice, so
        # such a book gets equal lengths (entries a power of two) - only used where that holds
        assert entries & (entries - 1) == 0
        lengths = [ilog(entries - 1)] * entries
    return Book(dims=dims, lengths=lengths, lookup=1, minv=-half * delta, delta=delta,
                value_bits=max(1, ilog(levels - 1)), mults=list(range(levels)), ordered=ordered)


def table_book(dims: int, values: np.ndarray, delta: float, rng, seq=False) -> Book:
    """Lookup type 2: an explicit table [entries][dims] of integer multiples of delta (offset so they are >= 0)."""
    values = np.asarray(values, dtype=np.int64)
    entries = values.shape[0]
    lo = int(values.min())
    freqs = list(0.2 + rng.random(entries))
    stored = values - lo
    if seq:     # stored values are differences along the vector
        stored = np.concatenate([values[:, :1] - lo, np.diff(values, axis=1) - lo], axis=1)
        assert stored.min() >= 0
    return Book(dims=dims, lengths=huffman_lengths(freqs), lookup=2, minv=lo * delta, delta=delta,
                value_bits=max(1, ilog(int(stored.max()))), seq=seq, mults=[int(v) for v in stored.reshape(-1)])


def scalar_book(entries: int, rng, ordered=False, sparse_unused=()) -> Book:
    freqs = list(0.1 + rng.random(entries))
    for i in sparse_unused:
        freqs[i] = 0
    lengths = huffman_lengths(freqs)
    if ordered:
        lengths = sorted(lengths)
    return Book(dims=1, lengths=lengths, ordered=ordered, sparse=bool(sparse_unused))


# ---- floor 1 ----
@dataclass
class Floor1:
    part_class: list[int]
    class_dim: list[int]
    class_sub: list[int]
    class_master: list[int]           # book numbers
    sub_books: list[list[int]]        # per class, 1 << sub entries, -1 = none
    multiplier: int
    rangebits: int
    xs: list[int]                     # posts after the two fixed ones, in partition order

    def header(self, bw: BitWriter):
        bw.put(1, 16)
        bw.put(len(self.part_class), 5)
        for c in self.part_class:
            bw.put(c, 4)
        for c in range(max(self.part_class) + 1):
            bw.put(self.class_dim[c] - 1, 3)
            bw.put(self.class_sub[c], 2)
            if self.class_sub[c]:
                bw.put(self.class_master[c], 8)
            for j in range(1 << self.class_sub[c]):
                bw.put(self.sub_books[c][j] + 1, 8)
        bw.put(self.multiplier - 1, 2)
        bw.put(self.rangebits, 4)
        for x in self.xs:
            bw.put(x, self.rangebits)

    @property
    def x_list(self):
        return [0, 1 << self.rangebits] + list(self.xs)

    @property
    def range(self):
        return [256, 128, 86, 64][self.multiplier - 1]


def _render_point(x0, y0, x1, y1, X):
    dy, adx = y1 - y0, x1 - x0
    off = abs(dy) * (X - x0) // adx
    return y0 - off if dy < 0 else y0 + off


def _render_line(x0, y0, x1, y1, v):
    dy, adx = y1 - y0, x1 - x0
    ady = abs(dy)
    base = int(dy / adx)            # toward zero
    sy = base - 1 if dy < 0 else base + 1
    ady -= abs(base) * adx
    x, y, err = x0, y0, 0
    if x < len(v):
        v[x] = y
    for x in range(x0 + 1, x1):
        err += ady
        if err >= adx:
            err -= adx
            y += sy
        else:
            y += base
        if x < len(v):
            v[x] = y


def floor1_codes(fl: Floor1, target: list[int]):
    """The Y values to write so that the decoder's amplitude synthesis lands on `target` at every post, together
    with the values it lands on (equal to `target`) and the flags that say which posts take part in the curve."""
    X = fl.x_list
    rng_ = fl.range
    Y = [target[0], target[1]]
    final = [target[0], target[1]]
    flag = [True, True]
    for i in range(2, len(X)):
        lo = max((j for j in range(i) if X[j] < X[i]), key=lambda j: X[j])
        hi = min((j for j in range(i) if X[j] > X[i]), key=lambda j: X[j])
        pred = _render_point(X[lo], final[lo], X[hi], final[hi], X[i])
        highroom, lowroom = rng_ - pred, pred
        room = min(highroom, lowroom) * 2
        t = target[i]
        d = t - pred
        if d == 0:
            val = 0
        elif d > 0:
            val = 2 * d if 2 * d < room else t - pred + lowroom
        else:
            val = -2 * d - 1 if -2 * d - 1 < room else pred - t + highroom - 1
        # decode it again, as the format defines it
        if val:
            flag[lo] = flag[hi] = True
            flag.append(True)
            if val >= room:
                got = val - lowroom + pred if highroom > lowroom else pred - val + highroom - 1
            else:
                got = pred - (val + 1) // 2 if val & 1 else pred + val // 2
        else:
            flag.append(False)
            got = pred
        assert got == t and 0 <= val, (got, t, val)
        Y.append(val)
        final.append(got)
    return Y, final, flag


def floor1_curve(fl: Floor1, final, flag, n: int) -> np.ndarray:
    X = fl.x_list
    order = sorted(range(len(X)), key=lambda i: X[i])
    v = [0] * n
    lx, ly = 0, final[order[0]] * fl.multiplier
    hx, hy = 0, ly
    for i in order[1:]:
        if flag[i]:
            hy = final[i] * fl.multiplier
            hx = X[i]
            _render_line(lx, ly, hx, hy, v)
            lx, ly = hx, hy
    if hx < n:
        _render_line(hx, hy, n, hy, v)
    return np.array([10.0 ** ((y - 255) * 0.02734375) for y in v[:n]])


# ---- transforms ----
def slope(n: int) -> np.ndarray:
    i = np.arange(n)
    return np.sin(np.pi / 2 * np.sin((i + 0.5) / n * np.pi / 2) ** 2)


def window(n: int, bs0: int, long: bool, prev_long: bool, next_long: bool) -> np.ndarray:
    w = np.zeros(n)
    if long and not prev_long:
        ls, ln = n // 4 - bs0 // 4, bs0 // 2
    else:
        ls, ln = 0, n // 2
    if long and not next_long:
        rs, rn = n * 3 // 4 - bs0 // 4, bs0 // 2
    else:
        rs, rn = n // 2, n // 2
    w[ls:ls + ln] = slope(ln)
    w[ls + ln:rs] = 1.0
    w[rs:rs + rn] = slope(rn)[::-1]
    return w


_COS = {}


def _cos_matrix(n: int) -> np.ndarray:
    if n not in _COS:
        M = n // 2
        t = np.arange(n)[:, None] + 0.5 + M / 2
        k = np.arange(M)[None, :] + 0.5
        _COS[n] = np.cos(np.pi / M * t * k)
    return _COS[n]


def mdct(x: np.ndarray) -> np.ndarray:
    """X[k] = 2/M sum_t x[t] cos(pi/M (t + 1/2 + M/2)(k + 1/2)), M = len(x) / 2."""
    n = len(x)
    return (2.0 / (n // 2)) * (x @ _cos_matrix(n))


def imdct(X: np.ndarray) -> np.ndarray:
    return _cos_matrix(2 * len(X)) @ X


# ---- Ogg ----
def _crc_table():
    t = []
    for i in range(256):
        r = i << 24
        for _ in range(8):
            r = ((r << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if r & 0x80000000 else (r << 1) & 0xFFFFFFFF
        t.append(r)
    return t


_CRC = _crc_table()


def ogg_crc(data: bytes) -> int:
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFFFFFF) ^ _CRC[((c >> 24) & 0xFF) ^ b]
    return c


def ogg_pages(packets: list[bytes], granules: list[int], serial=0x1234, per_page=None, first_seq=0) -> bytes:
    """Packets into pages.  `granules[i]` is the sample count at the end of packet i (headers: 0).  `per_page` gives
    the number of packets started on each page (default: headers one per page... the three headers on pages of
    their own as the format requires for the first, then as many audio packets per page as fit 255 segments)."""
    out = bytearray()
    seq = first_seq
    # lacing for every packet
    segs = []       # (packet index, bytes, is_last_segment_of_packet)
    for i, p in enumerate(packets):
        k = 0
        while True:
            piece = p[k:k + 255]
            k += 255
            last = len(piece) < 255
            segs.append((i, piece, last))
            if last:
                break
    pos = 0
    page_no = 0
    limits = list(per_page) if per_page else None
    started_prev_unfinished = False
    while pos < len(segs):
        take = []
        started = 0
        while pos < len(segs) and len(take) < 255:
            i, piece, last = segs[pos]
            is_start = (pos == 0) or segs[pos - 1][2]
            if is_start:
                if limits is not None:
                    if page_no < len(limits) and started >= limits[page_no]:
                        break
                elif i <= 2 and take:         # headers: id alone on its page, comment and setup together is allowed; keep each alone
                    break
                elif i == 3 and take:
                    break
                started += 1
            take.append(segs[pos])
            pos += 1
        htype = 0
        if started_prev_unfinished:
            htype |= 1
        if page_no == 0:
            htype |= 2
        if pos >= len(segs):
            htype |= 4
        done = [i for i, _, last in take if last]
        granule = granules[done[-1]] if done else -1
        started_prev_unfinished = not take[-1][2]
        table = bytes(len(piece) for _, piece, _ in take)
        body = b"".join(piece for _, piece, _ in take)
        head = b"OggS" + struct.pack("<BBqIII", 0, htype, granule, serial, seq, 0) + bytes([len(table)]) + table
        crc = ogg_crc(head + body)
        out += head[:22] + struct.pack("<I", crc) + head[26:] + body
        seq += 1
        page_no += 1
    return bytes(out)


def with_serial(data: bytes, serial: int, eos: bool = True) -> bytes:
    """The same pages under another stream serial number (checksums recomputed) - for chained files, whose links each
    carry their own.  `eos` False: the last page loses its end-of-stream mark (a chain some writers leave that way)."""
    out, pos = bytearray(), 0
    while pos < len(data):
        assert data[pos:pos + 4] == b"OggS"
        nseg = data[pos + 26]
        size = 27 + nseg + sum(data[pos + 27: pos + 27 + nseg])
        page = bytearray(data[pos: pos + size])
        page[14:18] = struct.pack("<I", serial)
        if not eos:
            page[5] &= ~4
        page[22:26] = bytes(4)
        page[22:26] = struct.pack("<I", ogg_crc(bytes(page)))
        out += page
        pos += size
    return bytes(out)


# ---- the encoder ----
@dataclass
class Config:
    channels: int = 1
    rate: int = 16000
    bs0: int = 256
    bs1: int = 2048
    residue_type: int = 1
    coupled: bool = False
    book_kind: str = "lattice"        # "lattice" (lookup 1) or "table" (lookup 2)
    floor_multiplier: int = 2
    submaps: int = 1                  # 2: channel c uses submap c % 2 (own floor and residue)
    pattern: str = "L"                # block sizes, repeated: e.g. "LLSSSL"
    seed: int = 0
    trim_end: bool = False            # the last page's granule position states the signal's length, not the blocks'
    start_offset: int = 0             # > 0: the first audio page states this many samples fewer than it produces
    partition_size: int = 32
    seq_books: bool = False


@dataclass
class Encoded:
    data: bytes
    expected: np.ndarray              # [samples][channels] float64: what the written spectrum stands for
    blocks: list[int]
    total: int
    packets: list = field(default_factory=list)      # the three headers, then one packet per block
    granules: list = field(default_factory=list)     # sample count at the end of each packet


def encode(pcm: np.ndarray, cfg: Config) -> Encoded:
    pcm = np.asarray(pcm, dtype=np.float64)
    if pcm.ndim == 1:
        pcm = pcm[:, None]
    ch = cfg.channels
    assert pcm.shape[1] == ch
    rng = np.random.default_rng(cfg.seed)
    bs = [cfg.bs0, cfg.bs1]

    # ---- books ----
    books: list[Book] = []

    def add(b: Book) -> int:
        books.append(b)
        return len(books) - 1

    # floor books: class master books and post-value books
    fl_range = [256, 128, 86, 64][cfg.floor_multiplier - 1]
    y_book = add(scalar_book(fl_range, rng))                       # any post code 0 .. range-1
    y_book_small = add(scalar_book(fl_range, rng, ordered=True))
    master = add(scalar_book(64, rng))                              # three values of two subclass bits each
    # residue books: coarse (step 1), fine, finer
    if cfg.book_kind == "lattice":
        dims = 2
        coarse = add(lattice_book(15, dims, 1.0, rng))
        fine = add(lattice_book(9, dims, 0.125, rng))
        finer = add(lattice_book(9, dims, 1.0 / 64, rng))
        coarse_reach = 7.0
    else:
        dims = 4 if cfg.partition_size % 4 == 0 else 2
        tabs = []
        for levels, delta in [(5, 1.0), (5, 0.25), (5, 1.0 / 16)]:
            grid = np.stack(np.meshgrid(*[np.arange(levels) - levels // 2] * dims, indexing="ij"), -1).reshape(-1, dims)
            tabs.append(add(table_book(dims, np.cumsum(grid, axis=1) if cfg.seq_books else grid, delta, rng,
                                       seq=cfg.seq_books)))
        coarse, fine, finer = tabs
        coarse_reach = 2.0
    coarse_step = 1.0
    n_class = 3
    classwords = 2
    classbook = add(scalar_book(n_class ** classwords + 3, rng, sparse_unused=(9, 10, 11)))
    books[classbook].dims = classwords          # a class book's dimension is the number of partitions one codeword classifies
    res_books = [coarse, fine, finer]
    # passes used: class 0 none; class 1 coarse in pass 0; class 2 coarse, fine, finer in passes 0, 1 and 3
    cascade = [0b0000, 0b0001, 0b1011]
    class_books = [[-1] * 8, [coarse] + [-1] * 7, [coarse, fine, -1, finer] + [-1] * 4]

    # ---- floors ----
    def make_floor(rangebits: int, seed: int) -> Floor1:
        r = np.random.default_rng(seed)
        span = 1 << rangebits
        part_class = [0, 1, 1, 0]
        class_dim = [2, 3]
        xs = sorted(r.choice(np.arange(1, span), size=sum(class_dim[c] for c in part_class), replace=False).tolist())
        r.shuffle(xs)
        return Floor1(part_class=part_class, class_dim=class_dim, class_sub=[0, 2], class_master=[0, master],
                      sub_books=[[y_book], [y_book, y_book_small, -1, y_book]], multiplier=cfg.floor_multiplier,
                      rangebits=rangebits, xs=[int(x) for x in xs])

    # one floor per block size (the post positions span the block's spectrum), per submap
    floors: list[Floor1] = []
    floor_of = {}      # (submap, blockflag) -> floor number
    for sm in range(cfg.submaps):
        for w in range(2):
            floor_of[(sm, w)] = len(floors)
            floors.append(make_floor(ilog(bs[w] // 2 - 1), 100 * cfg.seed + 10 * sm + w))

    # ---- setup header ----
    bw = BitWriter()
    for c in b"\x05vorbis":
        bw.put(c, 8)
    bw.put(len(books) - 1, 8)
    for b in books:
        b.header(bw)
    bw.put(0, 6)
    bw.put(0, 16)
    bw.put(len(floors) - 1, 6)
    for f in floors:
        f.header(bw)
    n_res = cfg.submaps
    bw.put(n_res - 1, 6)
    res_types = [cfg.residue_type if sm == 0 else (cfg.residue_type + 1) % 2 for sm in range(n_res)]
    res_begin = [cfg.partition_size * (cfg.seed % 2), 0]
    res_end = [bs[1] // 2 * (ch if cfg.residue_type == 2 else 1)] * 2
    for sm in range(n_res):
        bw.put(res_types[sm], 16)
        bw.put(res_begin[sm], 24)
        bw.put(res_end[sm], 24)
        bw.put(cfg.partition_size - 1, 24)
        bw.put(n_class - 1, 6)
        bw.put(classbook, 8)
        for c in range(n_class):
            bw.put(cascade[c] & 7, 3)
            if cascade[c] >> 3:
                bw.put(1, 1)
                bw.put(cascade[c] >> 3, 5)
            else:
                bw.put(0, 1)
        for c in range(n_class):
            for j in range(8):
                if cascade[c] & (1 << j):
                    bw.put(class_books[c][j], 8)
    # mappings: one per block size (they differ in their floors)
    bw.put(2 - 1, 6)
    for w in range(2):
        bw.put(0, 16)
        if cfg.submaps > 1:
            bw.put(1, 1)
            bw.put(cfg.submaps - 1, 4)
        else:
            bw.put(0, 1)
        if cfg.coupled:
            assert ch == 2
            bw.put(1, 1)
            bw.put(0, 8)
            bw.put(0, ilog(ch - 1))
            bw.put(1, ilog(ch - 1))
        else:
            bw.put(0, 1)
        bw.put(0, 2)
        if cfg.submaps > 1:
            for c in range(ch):
                bw.put(c % cfg.submaps, 4)
        for sm in range(cfg.submaps):
            bw.put(0, 8)
            bw.put(floor_of[(sm, w)], 8)
            bw.put(sm, 8)
    # modes: 0 short, 1 long, and a third (long again) so that the mode number takes two bits
    modes = [(0, 0), (1, 1), (1, 1)]
    bw.put(len(modes) - 1, 6)
    for flag, mp in modes:
        bw.put(flag, 1)
        bw.put(0, 16)
        bw.put(0, 16)
        bw.put(mp, 8)
    bw.put(1, 1)
    setup = bw.bytes()

    ident = b"\x01vorbis" + struct.pack("<IBIiiiB", 0, ch, cfg.rate, 0, 0, 0, (ilog(bs[1]) - 1) << 4 | (ilog(bs[0]) - 1)) + b"\x01"
    vendor = b"oasr test writer"
    comment = b"\x03vorbis" + struct.pack("<I", len(vendor)) + vendor + struct.pack("<I", 0) + b"\x01"

    # ---- block sequence covering the signal ----
    flags = []
    centre = []
    n_samples = pcm.shape[0]
    i = 0
    c = None
    while True:
        w = 1 if cfg.pattern[i % len(cfg.pattern)] == "L" else 0
        c = bs[w] // 2 if c is None else c + bs[flags[-1]] // 4 + bs[w] // 4
        flags.append(w)
        centre.append(c)
        i += 1
        if c - centre[0] >= n_samples and len(flags) >= 2:
            break
    produced = centre[-1] - centre[0]
    # the signal on the block axis: sample s of the stream sits at centre[0] + s
    pad = bs[1]        # a long block after a short first one starts before the first block does
    axis = np.zeros((pad + centre[-1] + bs[1], ch))
    axis[pad + centre[0]:pad + centre[0] + n_samples] = pcm[:min(n_samples, axis.shape[0] - centre[0] - pad)]
    recon = np.zeros_like(axis)

    packets = [ident, comment, setup]
    granules = [0, 0, 0]
    for bi, w in enumerate(flags):
        n = bs[w]
        M = n // 2
        prev_long = bool(flags[bi - 1]) if bi > 0 else True
        next_long = bool(flags[bi + 1]) if bi + 1 < len(flags) else True
        win = window(n, bs[0], bool(w), prev_long, next_long)
        start = pad + centre[bi] - n // 2
        bw = BitWriter()
        bw.put(0, 1)
        bw.put((1 if w else 0) if bi % 3 else (2 if w else 0), ilog(len(modes) - 1))
        if w:
            bw.put(1 if prev_long else 0, 1)
            bw.put(1 if next_long else 0, 1)
        spectra = [mdct(axis[start:start + n, c_] * win) for c_ in range(ch)]
        # floors: a random smooth-ish set of posts per channel, raised until the residue fits the coarse book
        curves, post_codes = [], []
        silent = []
        for c_ in range(ch):
            fl = floors[floor_of[(c_ % cfg.submaps, w)]]
            peak = np.abs(spectra[c_]).max()
            if peak < 1e-9 and (bi + c_) % 2 == 0:
                silent.append(True)
                curves.append(None)
                post_codes.append(None)
                continue
            silent.append(False)
            need_db = 20 * math.log10(max(peak, 1e-7) / coarse_reach)     # amplitude of the floor's lowest point
            y_min = int(math.ceil((need_db / 0.546875 + 255) / fl.multiplier))
            y_min = min(max(y_min, 0), fl.range - 1)
            hi = min(fl.range - 1, y_min + 12)
            target = [int(v) for v in rng.integers(y_min, hi + 1, size=len(fl.x_list))]
            Y, final, flag = floor1_codes(fl, target)
            curves.append(floor1_curve(fl, final, flag, M))
            post_codes.append((fl, Y))
        # residue targets
        resid = [np.zeros(M) if silent[c_] else spectra[c_] / curves[c_] for c_ in range(ch)]
        if cfg.coupled:
            m, a = resid[0].copy(), resid[1].copy()
            # forward of the format's inverse coupling (square polar mapping)
            mag = np.where(np.abs(m) > np.abs(a), m, a)
            ang = np.where(np.abs(m) > np.abs(a), np.where(m > 0, m - a, a - m), np.where(a > 0, m - a, a - m))
            resid = [mag, ang]
        # floors into the packet
        for c_ in range(ch):
            if silent[c_]:
                bw.put(0, 1)
                continue
            fl, Y = post_codes[c_]
            bw.put(1, 1)
            yb = ilog(fl.range - 1)
            bw.put(Y[0], yb)
            bw.put(Y[1], yb)
            off = 2
            for cls in fl.part_class:
                cdim, cbits = fl.class_dim[cls], fl.class_sub[cls]
                choice = []
                if cbits:
                    # pick per value a subclass whose book can state it (subclass 2 has no book: value must be 0)
                    cval = 0
                    for j in range(cdim):
                        y = Y[off + j]
                        opts = [s for s in range(1 << cbits) if (fl.sub_books[cls][s] >= 0 or y == 0)
                                and (fl.sub_books[cls][s] < 0 or books[fl.sub_books[cls][s]].words[y] is not None)]
                        s = opts[int(rng.integers(len(opts)))]
                        choice.append(s)
                        cval |= s << (cbits * j)
                    books[fl.class_master[cls]].write(bw, cval)
                else:
                    choice = [0] * cdim
                for j in range(cdim):
                    book = fl.sub_books[cls][choice[j]]
                    if book >= 0:
                        books[book].write(bw, Y[off + j])
                    else:
                        assert Y[off + j] == 0
                off += cdim
        # residues per submap
        quant = [np.zeros(M) for _ in range(ch)]
        for sm in range(cfg.submaps):
            chans = [c_ for c_ in range(ch) if c_ % cfg.submaps == sm]
            skip = [silent[c_] for c_ in chans]
            if cfg.coupled and not all(silent):
                skip = [False] * len(chans)
            rtype = res_types[sm]
            if rtype == 2:
                if all(skip):
                    continue
                vecs = [np.stack([resid[c_] for c_ in chans], 1).reshape(-1)]
                skip = [False]
            else:
                vecs = [resid[c_] for c_ in chans]
            size = len(vecs[0])
            lb, le = min(res_begin[sm], size), min(res_end[sm], size)
            ps = cfg.partition_size
            parts = (le - lb) // ps
            # quantise every partition in cascaded passes (nearest vector of each pass's book) and choose its class
            fmt = 0 if rtype == 0 else 1
            qv = [np.zeros(size) for _ in vecs]
            cls_of = [[0] * parts for _ in vecs]
            ents = [[None] * parts for _ in vecs]

            def groups_of(seg, d_):
                # format 0: vector s takes elements s, s + step, ...; format 1: consecutive elements
                return seg.reshape(d_, ps // d_).T if fmt == 0 else seg.reshape(ps // d_, d_)

            def ungroup(g, d_):
                return g.T.reshape(-1) if fmt == 0 else g.reshape(-1)

            for vi, v in enumerate(vecs):
                if skip[vi]:
                    continue
                for p in range(parts):
                    seg = v[lb + p * ps: lb + (p + 1) * ps]
                    rem = seg.copy()
                    got, approx = [], []
                    for book_no in res_books:
                        bk = books[book_no]
                        e, vec = bk.nearest(groups_of(rem, bk.dims))
                        got.append(e)
                        approx.append(ungroup(vec, bk.dims))
                        rem = rem - approx[-1]
                    if not np.abs(approx[0]).any() and np.abs(seg).max() < coarse_step / 4:
                        cls = 0
                    elif (p + bi) % 3 == 0:
                        cls = 1
                    else:
                        cls = 2
                    cls_of[vi][p] = cls
                    ents[vi][p] = got
                    if cls == 1:
                        qv[vi][lb + p * ps: lb + (p + 1) * ps] = approx[0]
                    elif cls == 2:
                        qv[vi][lb + p * ps: lb + (p + 1) * ps] = approx[0] + approx[1] + approx[2]
            # write, in the format's order: passes outermost, class words inside pass 0
            for pas in range(8):
                pc = 0
                while pc < parts:
                    if pas == 0:
                        for vi in range(len(vecs)):
                            if skip[vi]:
                                continue
                            num = 0
                            for k in range(classwords):
                                cidx = cls_of[vi][pc + k] if pc + k < parts else 0
                                num = num * n_class + cidx
                            books[classbook].write(bw, num)
                    for k in range(classwords):
                        if pc >= parts:
                            break
                        for vi in range(len(vecs)):
                            if skip[vi]:
                                continue
                            book_no = class_books[cls_of[vi][pc]][pas]
                            if book_no < 0:
                                continue
                            for e in ents[vi][pc][res_books.index(book_no)]:
                                books[book_no].write(bw, int(e))
                        pc += 1
            if rtype == 2:
                de = qv[0].reshape(-1, len(chans))
                for k, c_ in enumerate(chans):
                    quant[c_] = de[:, k].copy()
            else:
                for k, c_ in enumerate(chans):
                    quant[c_] = qv[k]
        if cfg.coupled:
            m, a = quant[0], quant[1]
            nm = np.where(m > 0, np.where(a > 0, m, m + a), np.where(a > 0, m, m - a))
            na = np.where(m > 0, np.where(a > 0, m - a, m), np.where(a > 0, m + a, m))
            quant = [nm, na]
        for c_ in range(ch):
            if silent[c_]:
                continue
            recon[start:start + n, c_] += imdct(quant[c_] * curves[c_]) * win
        packets.append(bw.bytes())
        granules.append(max(0, centre[bi] - centre[0]))
    total = min(produced, n_samples) if cfg.trim_end else produced
    granules[-1] = total
    expected = recon[pad + centre[0]:pad + centre[0] + total]
    per_page = None
    if cfg.start_offset:
        # the first audio page holds the first three audio packets and states fewer samples than they produce
        per_page = [1, 1, 1, 3] + [4] * len(packets)
        made = centre[2] - centre[0]
        granules[5] = made - cfg.start_offset
        for k in range(6, len(granules)):
            granules[k] -= cfg.start_offset
        expected = expected[cfg.start_offset:]
        total -= cfg.start_offset
    data = ogg_pages(packets, granules, per_page=per_page)
    return Encoded(data=data, expected=expected, blocks=[bs[w] for w in flags], total=total, packets=packets, granules=granules)


def repeat_stream(enc: Encoded, times: int) -> bytes:
    """The stream's audio packets `times` times over (a packet depends on nothing but the block before it, so any
    sequence of packets is a stream), with granule positions that follow the block sizes."""
    audio = enc.packets[3:] * times
    sizes = enc.blocks * times
    granules, c = [0, 0, 0], 0
    for i, n in enumerate(sizes):
        if i:
            c += sizes[i - 1] // 4 + n // 4
        granules.append(c)
    return ogg_pages(enc.packets[:3] + audio, granules)
