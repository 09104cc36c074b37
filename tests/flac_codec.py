"""Test-side FLAC reader and writer, written from the published format (RFC 9639) independently of csrc/flac.hip.
libFLAC is not in this image; these let the product's encoder be decoded by a second implementation, and the product's
decoder be fed the parts of the format its own encoder never emits (LPC subframes, mid/side, left/side, side/right
stereo, Rice2 parameters, escape partitions, wasted bits, variable block sizes coded as 8 / 16-bit fields, 24-bit audio).
Never imported by the product.  Pure Python: keep inputs to a few tens of thousands of samples."""
import hashlib
import struct

import numpy as np


def crc8(b):
    c = 0
    for x in b:
        c ^= x
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(b):
    c = 0
    for x in b:
        c ^= x << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


class Bits:
    def __init__(self, data, pos=0):
        self.d, self.pos = data, pos * 8

    def get(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.pos >> 3] >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v

    def sget(self, n):
        v = self.get(n)
        return v - (1 << n) if n and v >> (n - 1) else v

    def unary(self):
        q = 0
        while not self.get(1):
            q += 1
        return q


def _residual(br, n, order):
    method = br.get(2)
    assert method in (0, 1)
    pb, esc = (4, 15) if method == 0 else (5, 31)
    po = br.get(4)
    out = []
    for p in range(1 << po):
        cnt = (n >> po) - (order if p == 0 else 0)
        k = br.get(pb)
        if k == esc:
            raw = br.get(5)
            out += [br.sget(raw) for _ in range(cnt)]
        else:
            for _ in range(cnt):
                u = (br.unary() << k) | br.get(k)
                out.append((u >> 1) ^ -(u & 1))
    return out


_FIXED = [[], [1], [2, -1], [3, -3, 1], [4, -6, 4, -1]]


def _subframe(br, n, bps):
    assert br.get(1) == 0
    t = br.get(6)
    wasted = 0
    if br.get(1):
        wasted = br.unary() + 1
    bps -= wasted
    if t == 0:
        x = [br.sget(bps)] * n
    elif t == 1:
        x = [br.sget(bps) for _ in range(n)]
    elif 8 <= t <= 12:
        order = t - 8
        x = [br.sget(bps) for _ in range(order)]
        for r in _residual(br, n, order):
            x.append(r + sum(c * x[-1 - j] for j, c in enumerate(_FIXED[order])))
    elif t >= 32:
        order = (t & 31) + 1
        x = [br.sget(bps) for _ in range(order)]
        prec = br.get(4) + 1
        shift = br.sget(5)
        coef = [br.sget(prec) for _ in range(order)]
        for r in _residual(br, n, order):
            x.append(r + (sum(c * x[-1 - j] for j, c in enumerate(coef)) >> shift))
    else:
        raise ValueError("reserved subframe type")
    return [v << wasted for v in x]


def read(data):
    """bytes -> (int array (frames, channels), sample rate, bits per sample).  Checks CRC-8, CRC-16 and the MD5."""
    assert data[:4] == b"fLaC"
    o, last, info = 4, False, None
    while not last:
        last, typ = bool(data[o] & 0x80), data[o] & 0x7F
        ln = int.from_bytes(data[o + 1:o + 4], "big")
        if info is None:
            assert typ == 0 and ln == 34
            s = data[o + 4:o + 38]
            packed = int.from_bytes(s[10:18], "big")
            info = dict(min_block=(s[0] << 8) | s[1], max_block=(s[2] << 8) | s[3], sr=packed >> 44,
                        ch=((packed >> 41) & 7) + 1, bps=((packed >> 36) & 31) + 1, total=packed & 0xFFFFFFFFF,
                        md5=bytes(s[18:34]), min_frame=int.from_bytes(s[4:7], "big"), max_frame=int.from_bytes(s[7:10], "big"))
        o += 4 + ln
    chans = [[] for _ in range(info["ch"])]
    sizes = []
    while o < len(data):
        br = Bits(data, o)
        assert br.get(14) == 0x3FFE and br.get(1) == 0
        br.get(1)
        bs_code, sr_code, ch_code, ss_code = br.get(4), br.get(4), br.get(4), br.get(3)
        assert br.get(1) == 0
        b0 = br.get(8)
        ones = 0
        while b0 & (0x80 >> ones):
            ones += 1
        for _ in range(max(0, ones - 1)):
            assert br.get(8) & 0xC0 == 0x80
        assert bs_code != 0
        bs = 192 if bs_code == 1 else 576 << (bs_code - 2) if bs_code <= 5 else 256 << (bs_code - 8) if bs_code >= 8 else None
        if bs_code == 6:
            bs = br.get(8) + 1
        elif bs_code == 7:
            bs = br.get(16) + 1
        if sr_code == 12:
            br.get(8)
        elif sr_code in (13, 14):
            br.get(16)
        hdr = br.pos >> 3
        assert br.get(8) == crc8(data[o:hdr]), "CRC-8"
        bps = info["bps"] if ss_code == 0 else [0, 8, 12, None, 16, 20, 24, 32][ss_code]
        assert bps == info["bps"]
        nch = ch_code + 1 if ch_code < 8 else 2
        sub = []
        for c in range(nch):
            side = (ch_code == 8 and c == 1) or (ch_code == 9 and c == 0) or (ch_code == 10 and c == 1)
            sub.append(_subframe(br, bs, bps + (1 if side else 0)))
        if br.pos & 7:
            assert br.get(8 - (br.pos & 7)) == 0
        body = br.pos >> 3
        assert br.get(16) == crc16(data[o:body]), "CRC-16"
        if ch_code == 8:
            sub[1] = [l - s for l, s in zip(sub[0], sub[1])]
        elif ch_code == 9:
            sub[0] = [s + r for s, r in zip(sub[0], sub[1])]
        elif ch_code == 10:
            l, r = [], []
            for m, s in zip(sub[0], sub[1]):
                m = (m << 1) | (s & 1)
                l.append((m + s) >> 1)
                r.append((m - s) >> 1)
            sub = [l, r]
        for c in range(nch):
            chans[c] += sub[c]
        sizes.append((br.pos >> 3) - o)
        o = br.pos >> 3
    a = np.array(chans, dtype=np.int64).T.reshape(-1, info["ch"])
    assert info["total"] in (0, a.shape[0])
    if sizes:
        assert info["min_frame"] in (0, min(sizes)) and info["max_frame"] in (0, max(sizes))
    if any(info["md5"]):
        nb = (info["bps"] + 7) // 8
        raw = b"".join(int(v).to_bytes(nb, "little", signed=True) for v in a.ravel())
        assert hashlib.md5(raw).digest() == info["md5"], "MD5"
    return a, info["sr"], info["bps"]


# ---- writer: the features csrc/flac.hip's encoder never uses ----------------------------------------------------
class _W:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        self.bits += [(v >> (n - 1 - i)) & 1 for i in range(n)]

    def unary(self, q):
        self.bits += [0] * q + [1]

    def bytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


def _put_residual(w, res, n, order, method, po, escape_first):
    w.put(method, 2)
    w.put(po, 4)
    pb, esc = (4, 15) if method == 0 else (5, 31)
    i = 0
    for p in range(1 << po):
        cnt = (n >> po) - (order if p == 0 else 0)
        part = res[i:i + cnt]
        i += cnt
        if escape_first and p == 0:
            raw = max([1] + [int(abs(v)).bit_length() + 1 for v in part])
            w.put(esc, pb)
            w.put(raw, 5)
            for v in part:
                w.put(v & ((1 << raw) - 1), raw)
            continue
        mean = (sum(abs(v) for v in part) / max(1, len(part))) if part else 0
        k = min(esc - 1, max(0, int(mean).bit_length()))
        w.put(k, pb)
        for v in part:
            u = (v << 1) if v >= 0 else ((-v) << 1) - 1
            w.unary(u >> k)
            w.put(u & ((1 << k) - 1), k)


def _put_subframe(w, x, bps, kind, wasted=0, method=0, po=2, escape_first=False):
    n = len(x)
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in x)
        x = [v >> wasted for v in x]
    b = bps - wasted
    m = (1 << b) - 1

    def head(t):
        w.put(0, 1)
        w.put(t, 6)
        if wasted:
            w.put(1, 1)
            w.unary(wasted - 1)
        else:
            w.put(0, 1)
    if kind == "lpc":
        order, prec, shift = 3, 12, 9
        coef = [1230, -510, 60]                 # ~ (2.4, -1.0, 0.12) * 2^9: a smooth-signal predictor
        head(32 | (order - 1))
        for v in x[:order]:
            w.put(v & m, b)
        w.put(prec - 1, 4)
        w.put(shift & 31, 5)
        for c in coef:
            w.put(c & ((1 << prec) - 1), prec)
        res = [x[i] - (sum(c * x[i - 1 - j] for j, c in enumerate(coef)) >> shift) for i in range(order, n)]
        _put_residual(w, res, n, order, method, po, escape_first)
    elif kind.startswith("fixed"):
        order = int(kind[5:])
        head(8 | order)
        for v in x[:order]:
            w.put(v & m, b)
        res = [x[i] - sum(c * x[i - 1 - j] for j, c in enumerate(_FIXED[order])) for i in range(order, n)]
        _put_residual(w, res, n, order, method, po, escape_first)
    else:
        head(1)
        for v in x:
            w.put(v & m, b)


def write(samples, sr, bps=16, block=1152, stereo_modes=("mid",), kinds=("lpc",), wasted=0, method=0, escape_first=False,
          variable_sizes=False):
    """(frames, channels) ints -> FLAC bytes using LPC / FIXED / VERBATIM subframes (cycled per frame from `kinds`), stereo
    decorrelation modes cycled from `stereo_modes` ("indep", "left", "right", "mid"), Rice (`method` 0) or Rice2 (1)."""
    a = np.asarray(samples, dtype=np.int64).reshape(len(samples), -1)
    nch = a.shape[1]
    frames = []
    pos, fno = 0, 0
    while pos < a.shape[0]:
        n = min(block - (37 * (fno % 3) if variable_sizes else 0), a.shape[0] - pos)
        blk = a[pos:pos + n]
        mode = stereo_modes[fno % len(stereo_modes)] if nch == 2 else "indep"
        kind = kinds[fno % len(kinds)]
        ch_code = {"indep": nch - 1, "left": 8, "right": 9, "mid": 10}[mode]
        std = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12}
        bs_code = std.get(n, 6 if n <= 256 else 7)
        ss_code = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6}.get(bps, 0)
        h = _W()
        h.put(0x3FFE, 14)
        h.put(0, 2)
        h.put(bs_code, 4)
        h.put(0, 4)
        h.put(ch_code, 4)
        h.put(ss_code, 3)
        h.put(0, 1)
        hb = bytearray(h.bytes())
        if fno < 0x80:
            hb.append(fno)
        else:
            hb += bytes([0xC0 | (fno >> 6), 0x80 | (fno & 0x3F)])
        if bs_code == 6:
            hb.append(n - 1)
        elif bs_code == 7:
            hb += struct.pack(">H", n - 1)
        hb.append(crc8(hb))
        w = _W()
        po = 2 if n % 4 == 0 and (n >> 2) > 4 else 0
        cols = [list(map(int, blk[:, c])) for c in range(nch)]
        if mode == "left":
            subs = [(cols[0], bps), ([l - r for l, r in zip(*cols)], bps + 1)]
        elif mode == "right":
            subs = [([l - r for l, r in zip(*cols)], bps + 1), (cols[1], bps)]
        elif mode == "mid":
            subs = [([(l + r) >> 1 for l, r in zip(*cols)], bps), ([l - r for l, r in zip(*cols)], bps + 1)]
        else:
            subs = [(c, bps) for c in cols]
        for x, b in subs:
            usable = wasted if all(v % (1 << wasted) == 0 for v in x) else 0
            _put_subframe(w, x, b, kind if n > 8 else "verbatim", usable, method, po, escape_first)
        body = bytes(hb) + w.bytes()
        frames.append(body + struct.pack(">H", crc16(body)))
        pos += n
        fno += 1
    nb = (bps + 7) // 8
    md5 = hashlib.md5(b"".join(int(v).to_bytes(nb, "little", signed=True) for v in a.ravel())).digest()
    packed = (sr << 44) | ((nch - 1) << 41) | ((bps - 1) << 36) | a.shape[0]
    si = struct.pack(">HH", 16 if variable_sizes else block, block) + (0).to_bytes(3, "big") * 2 + packed.to_bytes(8, "big") + md5
    # a PADDING block between STREAMINFO and the audio: readers must skip metadata they do not use
    return b"fLaC" + bytes([0x00, 0, 0, 34]) + si + bytes([0x81, 0, 0, 6]) + bytes(6) + b"".join(frames)
