"""Test helper: a small MPEG Layer III *encoder* in numpy, written from the forward definitions of the format
(polyphase analysis filter, MDCT with the four window shapes, alias butterflies, mid/side, the power-law quantiser,
scale factors, Huffman books, side information, bit reservoir), so that the native decoder can be checked against
signals whose content is known: PCM -> this encoder -> liboasr_hip.so's decoder -> PCM.

It makes no attempt at audio quality (no psychoacoustic model; its rate loop only coarsens a granule until it fits
its share of the frame): the caller says how coarse the quantiser is, and the encoder reports the signal-to-noise ratio of the quantised spectrum, which — the filter bank being
orthogonal to within its stop-band rejection — is what a correct decoder must reproduce in the time domain.

Band-edge tables and the synthesis window are read out of the decoder's source so that they exist once; the Huffman
code books come from tools/mp3_tables_gen.py (which checks them for completeness and prefix-freeness).
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import mp3_tables_gen as books  # noqa: E402

_SRC = (ROOT / "omnilingual-asr_amd" / "csrc" / "mp3.cpp").read_text()


def _table(name):
    body = _SRC[_SRC.index(name):]
    body = body[body.index("=") + 1: body.index(";")]
    return [int(v) for v in re.findall(r"-?\d+", body)]


_RATES = [44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000]
LONG_EDGES = np.array(_table("kLongEdges[9][23]")).reshape(9, 23)
SHORT_EDGES = np.array(_table("kShortEdges[9][14]")).reshape(9, 14)
PRETAB = _table("kPretab[22]")
_base = np.array(_table("kWindowBase[257]"), dtype=np.float64) / 65536.0
_h = np.concatenate([_base, _base[255:0:-1]])                       # h[n], n = 0..511, h[512 - n] = h[n]
WINDOW_D = _h * np.where((np.arange(512) >> 6) & 1, -1.0, 1.0)     # the standard's synthesis window
_CI = np.array([-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037])
CS, CA = 1 / np.sqrt(1 + _CI ** 2), _CI / np.sqrt(1 + _CI ** 2)

LSF_PARTS = np.array(_table("kLsfPartitions[6][3][4]")).reshape(6, 3, 4).tolist()   # [rule][long | short | mixed]
LINBITS = {16: 1, 17: 2, 18: 3, 19: 4, 20: 6, 21: 8, 22: 10, 23: 13, 24: 4, 25: 5, 26: 6, 27: 7, 28: 8, 29: 9, 30: 11, 31: 13}
SLEN = [(0, 0), (0, 1), (0, 2), (0, 3), (3, 0), (1, 1), (1, 2), (1, 3), (2, 1), (2, 2), (2, 3), (3, 1), (3, 2), (3, 3), (4, 2), (4, 3)]


def _book(table):
    n = 16 if 16 <= table <= 23 else 24 if table >= 24 else table
    side, codes, lens = books.BOOKS[n]
    return side, codes, lens, LINBITS.get(table, 0)


def table_max(table):
    if table == 0:
        return 0
    side, _, _, lb = _book(table)
    return side - 1 + ((1 << lb) - 1 if lb else 0)


VALID_TABLES = [t for t in range(1, 32) if t not in (4, 14)]


class BitWriter:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value, bits):
        if bits:
            assert 0 <= value < (1 << bits), (value, bits)
            self.v = (self.v << bits) | value
            self.n += bits

    def to_bytes(self, total_bits=None):
        n = self.n if total_bits is None else total_bits
        assert n >= self.n
        pad = (-n) % 8
        return ((self.v << (n - self.n + pad))).to_bytes((n + pad) // 8, "big")


def analysis(x):
    """Polyphase analysis: x [n] -> subband samples [n // 32, 32]; slot t sees x up to sample 32 t + 31."""
    slots = len(x) // 32
    xp = np.concatenate([np.zeros(480), x[: slots * 32]])
    idx = 32 * np.arange(slots)[:, None] + 31 - np.arange(512)[None, :] + 480
    z = xp[idx] * (WINDOW_D / 32.0)
    y = z.reshape(slots, 8, 64).sum(axis=1)
    k, i = np.arange(32)[:, None], np.arange(64)[None, :]
    return y @ np.cos((2 * k + 1) * (i - 16) * np.pi / 64).T


def _long_window(bt):
    i = np.arange(36)
    w = np.sin(np.pi / 36 * (i + 0.5))
    if bt == 1:
        w = np.where(i < 18, w, np.where(i < 24, 1.0, np.where(i < 30, np.sin(np.pi / 12 * (i - 18 + 0.5)), 0.0)))
    if bt == 3:
        w = np.where(i < 6, 0.0, np.where(i < 12, np.sin(np.pi / 12 * (i - 6 + 0.5)), np.where(i < 18, 1.0, w)))
    return w


_COS36 = np.cos(np.pi / 72 * (2 * np.arange(36)[:, None] + 19) * (2 * np.arange(18)[None, :] + 1))
_COS12 = np.cos(np.pi / 24 * (2 * np.arange(12)[:, None] + 7) * (2 * np.arange(6)[None, :] + 1))
_WIN12 = np.sin(np.pi / 12 * (np.arange(12) + 0.5))


def hybrid(sub, block_types, mixed=False):
    """Subband samples [granules * 18, 32] -> spectra [granules, 576] in the order the inverse transform reads them
    (long: line 18 sb + k; short: line 18 sb + 3 k + window), alias butterflies applied between long subbands.
    `mixed`: every window-switching granule keeps the normal long window in its two lowest subbands."""
    g_n = len(block_types)
    s = sub[: g_n * 18].reshape(g_n, 18, 32).copy()
    s[:, 1::2, 1::2] *= -1.0                                        # frequency inversion
    prev = np.zeros((18, 32))
    out = np.zeros((g_n, 576))
    for g, bt in enumerate(block_types):
        z = np.concatenate([prev, s[g]], axis=0)                    # [36, 32]
        prev = s[g]
        if bt != 2 or mixed:
            win = np.repeat(_long_window(bt if bt != 2 else 0)[:, None], 32, axis=1)
            if mixed:
                win[:, :2] = _long_window(0)[:, None]
            xr = ((z * win).T @ _COS36) / 9.0                        # [32, 18]
            xr = xr.reshape(576)
            for sb in range(1, 32 if bt != 2 else 2):
                lo = xr[18 * sb - 1 - np.arange(8)].copy()
                hi = xr[18 * sb + np.arange(8)].copy()
                xr[18 * sb - 1 - np.arange(8)] = lo * CS + hi * CA
                xr[18 * sb + np.arange(8)] = hi * CS - lo * CA
            out[g] = xr
        if bt == 2:
            for w in range(3):
                seg = z[6 + 6 * w: 18 + 6 * w] * _WIN12[:, None]     # [12, 32]
                xs = (seg.T @ _COS12) / 3.0                          # [32, 6]
                out[g].reshape(32, 6, 3)[2 if mixed else 0:, :, w] = xs[2 if mixed else 0:]
    return out


def band_widths(sr_idx, bt, mixed=False):
    if bt != 2:
        return list(np.diff(LONG_EDGES[sr_idx])), 22
    w, n_long, first = [], 0, 0
    if mixed:                                                       # long bands over the lowest 36 lines
        n_long = int(np.searchsorted(LONG_EDGES[sr_idx], 36))
        assert LONG_EDGES[sr_idx][n_long] == 36 and SHORT_EDGES[sr_idx][3] == 12
        w, first = list(np.diff(LONG_EDGES[sr_idx][: n_long + 1])), 3
    for b in range(first, 13):
        w += [int(SHORT_EDGES[sr_idx][b + 1] - SHORT_EDGES[sr_idx][b])] * 3
    return w, n_long


def to_flat(xr, sr_idx, bt, mixed=False):
    """Inverse-transform order -> transmission order (short granules: [band][window][line])."""
    if bt != 2:
        return xr.copy()
    q = xr.reshape(192, 3)                                          # [short line, window]
    se = SHORT_EDGES[sr_idx]
    out = [xr[:36]] if mixed else []
    for b in range(3 if mixed else 0, 13):
        for w in range(3):
            out.append(q[se[b]: se[b + 1], w])
    return np.concatenate(out)


def _huff_pairs(bw, vals, table):
    side, codes, lens, lb = _book(table)
    for x, y in zip(vals[0::2], vals[1::2]):
        ax, ay = min(abs(x), 15) if lb else abs(x), min(abs(y), 15) if lb else abs(y)
        i = ax * side + ay
        bw.put(codes[i], lens[i])
        if lb and ax == 15:
            bw.put(abs(x) - 15, lb)
        if x:
            bw.put(1 if x < 0 else 0, 1)
        if lb and ay == 15:
            bw.put(abs(y) - 15, lb)
        if y:
            bw.put(1 if y < 0 else 0, 1)


def encode_granule(bw, isq, scf, gi, hdr, rng, opts):
    """Writes scale factors + spectrum of one granule-channel into `bw`; returns the side-information fields."""
    lsf, sr_idx, bt = hdr["lsf"], hdr["sr_idx"], gi["block_type"]
    start = bw.n
    # ---- scale factors ----
    if not lsf:
        s1, s2 = SLEN[gi["scalefac_compress"]]
        if bt == 2:
            n1 = 17 if gi.get("mixed") else 18                       # mixed: 8 long bands + short bands 3..5
            for i in range(n1):
                bw.put(int(scf[i]), s1)
            for i in range(n1, n1 + 18):
                bw.put(int(scf[i]), s2)
        else:
            edges = [0, 6, 11, 16, 21]
            for k in range(4):
                if gi["granule"] == 1 and gi["scfsi"][k]:
                    continue
                for i in range(edges[k], edges[k + 1]):
                    bw.put(int(scf[i]), s1 if k < 2 else s2)
    else:
        parts = gi.get("parts") or ([6, 5, 5, 5] if bt != 2 else [6, 9, 9, 9] if gi.get("mixed") else [9, 9, 9, 9])
        i = 0
        for k in range(4):
            for _ in range(parts[k]):
                bw.put(int(scf[i]), gi["slen"][k])
                i += 1
    # ---- spectrum ----
    nz = np.nonzero(isq)[0]
    last = int(nz[-1]) + 1 if len(nz) else 0
    big = np.nonzero(np.abs(isq) > 1)[0]
    big_end = (int(big[-1]) + 2) // 2 * 2 if len(big) else 0
    if opts.get("all_big"):
        big_end = (last + 1) // 2 * 2
    count1_end = big_end + (-(-(max(last, big_end) - big_end) // 4)) * 4
    while count1_end > 576:                                          # quadruples must fit: grow the pair region instead
        big_end += 2
        count1_end = big_end + (-(-(max(last, big_end) - big_end) // 4)) * 4
    gi["big_values"] = big_end // 2
    le = LONG_EDGES[sr_idx]
    if bt == 2:
        edges = [72 if sr_idx == 8 else 36, 576, 576]
    elif gi["switching"]:
        edges = [int(le[8]), 576, 576]
    else:
        r0, r1 = gi["region0"], gi["region1"]
        edges = [int(le[r0 + 1]), int(le[min(r0 + r1 + 2, 22)]), 576]
    lo = 0
    gi["tables"] = []
    for r in range(3 if not gi["switching"] else 2):
        hi = min(edges[r], big_end)
        vals = isq[lo:max(hi, lo)]
        m = int(np.abs(vals).max()) if len(vals) else 0
        fits = [t for t in VALID_TABLES if table_max(t) >= m]
        if m == 0 and rng.random() < 0.5:
            t = 0
        else:
            want = opts.get("tables")
            pick = [t for t in fits if want is None or t in want] or fits
            t = int(pick[rng.integers(len(pick))])
        gi["tables"].append(t)
        if t and len(vals):
            _huff_pairs(bw, [int(v) for v in vals], t)
        lo = max(hi, lo)
    gi["count1_b"] = int(rng.integers(2))
    for q in range(big_end, count1_end, 4):
        v = [int(t) for t in isq[q:q + 4]] + [0] * 4
        v = v[:4]
        idx = (abs(v[0]) << 3) | (abs(v[1]) << 2) | (abs(v[2]) << 1) | abs(v[3])
        if gi["count1_b"]:
            bw.put(15 - idx, 4)
        else:
            _, codes, lens = books.BOOKS[32]
            bw.put(codes[idx], lens[idx])
        for t in v:
            if t:
                bw.put(1 if t < 0 else 0, 1)
    gi["part23"] = bw.n - start


def _side_info(hdr, mdb, granules, scfsi):
    bw = BitWriter()
    ch = hdr["channels"]
    if hdr["lsf"]:
        bw.put(mdb, 8)
        bw.put(0, 1 if ch == 1 else 2)
    else:
        bw.put(mdb, 9)
        bw.put(0, 5 if ch == 1 else 3)
        for c in range(ch):
            for k in range(4):
                bw.put(scfsi[c][k], 1)
    for g in granules:
        for gi in g:
            bw.put(gi["part23"], 12)
            bw.put(gi["big_values"], 9)
            bw.put(gi["global_gain"], 8)
            bw.put(gi["scalefac_compress"], 9 if hdr["lsf"] else 4)
            bw.put(1 if gi["switching"] else 0, 1)
            if gi["switching"]:
                bw.put(gi["block_type"], 2)
                bw.put(1 if gi.get("mixed") else 0, 1)
                bw.put(gi["tables"][0], 5)
                bw.put(gi["tables"][1], 5)
                for w in range(3):
                    bw.put(gi["sub_gain"][w], 3)
            else:
                for t in gi["tables"]:
                    bw.put(t, 5)
                bw.put(gi["region0"], 4)
                bw.put(gi["region1"], 3)
            if not hdr["lsf"]:
                bw.put(gi["preflag"], 1)
            bw.put(gi["scf_scale"], 1)
            bw.put(gi["count1_b"], 1)
    out = bw.to_bytes()
    assert len(out) == hdr["side_bytes"], (len(out), hdr["side_bytes"])
    return out


def _crc16(data):
    crc = 0xFFFF
    for byte in data:
        crc ^= byte << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x8005 if crc & 0x8000 else crc << 1) & 0xFFFF
    return crc


def intensity_gains(pos, lsf, scale_sel=0):
    """(left, right) factors of an intensity-coded band, as the format defines them."""
    if not lsf:
        if pos == 6:
            return 1.0, 0.0
        r = np.tan(pos * np.pi / 12)
        return r / (1 + r), 1 / (1 + r)
    k = 2.0 ** (-0.25 * (((pos + 1) >> 1) << scale_sel))
    return (k, 1.0) if pos & 1 else (1.0, k)


def encode(pcm, rate, *, seed=0, ms=False, crc=False, block_types=None, peak_is=1000, scalefactors=False,
           reservoir=False, tables=None, all_big=False, mixed=False, intensity=None, kbps=None):
    """pcm float64 [n, channels] -> (stream bytes, info).  `peak_is`: the largest quantised magnitude a granule aims
    for (int, or a list to draw from per granule).  info: 'snr_spectrum_db' (signal power over quantisation-error
    power of everything written), 'granules', 'frames', 'tables_used'."""
    rng = np.random.default_rng(seed)
    pcm = np.atleast_2d(np.asarray(pcm, dtype=np.float64).T).T
    ch = pcm.shape[1]
    sr_idx = _RATES.index(rate)
    lsf = sr_idx >= 3
    n_gr = 1 if lsf else 2
    spf = 576 * n_gr
    frames = pcm.shape[0] // spf
    g_total = frames * n_gr
    block_types = list(block_types) if block_types is not None else [0] * g_total
    block_types = (block_types * (g_total // len(block_types) + 1))[:g_total]
    rates_kbps = ([0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160] if lsf
                  else [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320])
    kbps = (160 if lsf else 320) if kbps is None else kbps
    br_idx = rates_kbps.index(kbps)
    ver_bits = 3 if sr_idx < 3 else 2 if sr_idx < 6 else 0
    hdr = {"lsf": lsf, "sr_idx": sr_idx, "channels": ch,
           "side_bytes": (9 if ch == 1 else 17) if lsf else (17 if ch == 1 else 32)}
    frame_bytes = (72000 if lsf else 144000) * kbps // rate
    head_len = 4 + (2 if crc else 0) + hdr["side_bytes"]
    capacity = frame_bytes - head_len

    spectra = [hybrid(analysis(pcm[:, c]), block_types, mixed) for c in range(ch)]
    sig = err = 0.0
    area = bytearray()                       # main data of all frames, laid out as the slots of the frames
    wp = 0                                   # where the next frame's main data begins
    out = bytearray()
    used_tables = set()
    prev_scf = [None, None]
    for f in range(frames):
        granules, scfsi = [], [[0] * 4 for _ in range(ch)]
        bw = BitWriter()
        for g in range(n_gr):
            gidx = f * n_gr + g
            bt = block_types[gidx]
            row = []
            mix = mixed and bt != 0
            widths, n_long = band_widths(sr_idx, bt, mix)
            flat = [to_flat(spectra[c][gidx], sr_idx, bt, mix) for c in range(ch)]
            joint_from = 576                                         # first line that is intensity coded
            first_joint_band = None
            if intensity is not None:
                starts = np.concatenate([[0], np.cumsum(widths)])
                first_joint_band = next(b for b in range(len(widths)) if starts[b] >= intensity["bound"]
                                        and (b < n_long or (b - n_long) % 3 == 0))
                joint_from = int(starts[first_joint_band])
                flat[1][joint_from:] = 0.0
            if ms:
                m = (flat[0][:joint_from] + flat[1][:joint_from]) / np.sqrt(2)
                d = (flat[0][:joint_from] - flat[1][:joint_from]) / np.sqrt(2)
                flat[0][:joint_from], flat[1][:joint_from] = m, d
            for c in range(ch):
                xr = flat[c]
                gi = {"granule": g, "block_type": bt, "switching": bt != 0, "sub_gain": [0, 0, 0], "preflag": 0,
                      "scf_scale": 0, "scalefac_compress": 0, "region0": 0, "region1": 0, "scfsi": scfsi[c],
                      "mixed": mix}
                scf = np.zeros(40, dtype=np.int64)
                if intensity is not None and c == 1:                  # the right channel's scale factors are positions
                    n_sent = 21 if bt != 2 else (35 if (mix and not lsf) else 33 if mix else 36)
                    scf[first_joint_band:n_sent] = intensity["pos"]
                    if not lsf:
                        gi["scalefac_compress"] = 15                 # 4 and 3 bits
                    else:
                        gi["slen"] = [3, 3, 3, 0]
                        gi["scalefac_compress"] = (3 * 36 + 3 * 6 + 3) * 2 + intensity.get("scale_sel", 0)
                        gi["parts"] = [7, 7, 7, 0] if bt != 2 else [6, 15, 12, 0] if mix else [12, 12, 12, 0]
                elif scalefactors:
                    gi["scf_scale"] = int(rng.integers(2))
                    if bt == 2:
                        gi["sub_gain"] = [int(v) for v in rng.integers(0, 4, 3)]
                    if not lsf:
                        gi["scalefac_compress"] = int(rng.integers(16))
                        s1, s2 = SLEN[gi["scalefac_compress"]]
                        gi["preflag"] = int(rng.integers(2)) if bt != 2 else 0
                        if bt == 2:
                            n1 = 17 if mix else 18
                            scf[:n1] = rng.integers(0, 1 << s1, n1)
                            scf[n1:n1 + 18] = rng.integers(0, 1 << s2, 18)
                        else:
                            scf[:11] = rng.integers(0, 1 << s1, 11)
                            scf[11:21] = rng.integers(0, 1 << s2, 10)
                            if g == 1 and prev_scf[c] is not None and block_types[gidx - 1] != 2:
                                edges = [0, 6, 11, 16, 21]
                                for k in range(4):
                                    if rng.random() < 0.5:
                                        scfsi[c][k] = 1
                                        scf[edges[k]:edges[k + 1]] = prev_scf[c][edges[k]:edges[k + 1]]
                    else:
                        # the second standard's three ways of stating the scale-factor widths (the third implies preflag)
                        rule = int(rng.integers(3))
                        if rule == 0:
                            sl = [int(rng.integers(5)), int(rng.integers(5)), int(rng.integers(4)), int(rng.integers(4))]
                            gi["scalefac_compress"] = ((sl[0] * 5 + sl[1]) << 4) | (sl[2] << 2) | sl[3]
                        elif rule == 1:
                            sl = [int(rng.integers(5)), int(rng.integers(5)), int(rng.integers(4)), 0]
                            gi["scalefac_compress"] = 400 + (((sl[0] * 5 + sl[1]) << 2) | sl[2])
                        else:
                            sl = [int(rng.integers(4)), int(rng.integers(3)), 0, 0]
                            gi["scalefac_compress"] = 500 + sl[0] * 3 + sl[1]
                            gi["preflag"] = 1
                        gi["slen"] = sl
                        parts = LSF_PARTS[rule][0 if bt != 2 else 2 if mix else 1]
                        gi["parts"] = parts
                        i = 0
                        for k in range(4):
                            scf[i:i + parts[k]] = rng.integers(0, 1 << sl[k], parts[k])
                            i += parts[k]
                elif lsf:
                    gi["slen"] = [0, 0, 0, 0]
                if g == 0:
                    prev_scf[c] = scf.copy()
                if bt == 0:
                    gi["region0"], gi["region1"] = int(rng.integers(16)), int(rng.integers(8))
                # per-band exponent (quarter powers of two) on top of the global gain
                mult = 4 if gi["scf_scale"] else 2
                band_e = []
                for b, w in enumerate(widths):
                    if intensity is not None and c == 1 and b >= first_joint_band:
                        e = 0                                        # (a position, not a gain; the lines are zero)
                    elif b < n_long:
                        e = -mult * (int(scf[b]) + (PRETAB[b] if gi["preflag"] else 0))
                    else:
                        e = -8 * gi["sub_gain"][(b - n_long) % 3] - mult * int(scf[b])
                    band_e += [e] * int(w)
                band_e = np.array(band_e, dtype=np.float64)
                target = peak_is if np.isscalar(peak_is) else int(peak_is[rng.integers(len(peak_is))])
                peak = np.abs(xr / 2.0 ** (band_e / 4)).max()
                budget = min(4095, capacity * 8 // (n_gr * ch))
                while True:                                          # the rate loop: coarsen until the granule fits
                    gg = 210 if peak == 0 else int(np.ceil(4 * np.log2(peak / target ** (4 / 3)))) + 210
                    gg = int(np.clip(gg, 0, 255))
                    gi["global_gain"] = gg
                    step = 2.0 ** ((gg - 210 + band_e) / 4)
                    isq = np.minimum(np.floor((np.abs(xr) / step) ** 0.75 + 0.5), 8206).astype(np.int64)
                    isq = np.where(xr < 0, -isq, isq)
                    trial = BitWriter()
                    encode_granule(trial, isq, scf, gi, hdr, rng, {"tables": tables, "all_big": all_big})
                    if trial.n <= budget:
                        break
                    assert target > 1, "granule does not fit at the coarsest step"
                    target = max(1, int(target * 0.7))
                xq = np.sign(isq) * np.abs(isq).astype(np.float64) ** (4 / 3) * step
                sig += float((xr ** 2).sum())
                err += float(((xr - xq) ** 2).sum())
                bw.put(trial.v, trial.n)
                used_tables.update(t for t in gi["tables"] if t)
                row.append(gi)
            granules.append(row)
        main = bw.to_bytes()
        slot = f * capacity
        if not reservoir:
            wp = slot
        mdb_max = 255 if lsf else 511
        if slot - wp > mdb_max:
            wp = slot - mdb_max
        assert wp + len(main) <= slot + capacity, "frame does not fit: quantise more coarsely"
        if len(area) < slot + capacity:
            area.extend(bytes(slot + capacity - len(area)))
        area[wp:wp + len(main)] = main
        mdb = slot - wp
        wp += len(main)
        if reservoir and rng.random() < 0.3:                         # some stuffing between frames' data
            wp = min(wp + int(rng.integers(0, 40)), slot + capacity)
        side = _side_info(hdr, mdb, granules, scfsi)
        joint = ms or intensity is not None
        mode = 3 if ch == 1 else (1 if joint else 0)
        h = bytes([0xFF, 0xE0 | (ver_bits << 3) | (1 << 1) | (0 if crc else 1),
                   (br_idx << 4) | ((sr_idx % 3) << 2),
                   (mode << 6) | (((2 if ms else 0) | (1 if intensity is not None else 0)) << 4)])
        out += h
        if crc:
            out += _crc16(h[2:4] + side).to_bytes(2, "big")
        out += side
        out += b"\0" * capacity                                      # filled below, once later frames have borrowed
    # the main-data area goes into the frames' slots
    for f in range(frames):
        at = f * frame_bytes + head_len
        out[at:at + capacity] = area[f * capacity:(f + 1) * capacity]
    info = {"snr_spectrum_db": 10 * np.log10(sig / err) if err > 0 else np.inf, "frames": frames,
            "granules": g_total, "tables_used": sorted(used_tables), "frame_bytes": frame_bytes}
    return bytes(out), info
