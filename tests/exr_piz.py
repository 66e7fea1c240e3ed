"""A PIZ block writer in Python, test infrastructure for the EXR reader of csrc/host/exr.cpp: it follows the published
OpenEXR format (bitmap of the word values in use, wavelet transform of each channel plane, canonical Huffman code with
a repeat symbol) and shares no code with the reader. No OpenEXR library is importable here, so the pair is only
checked against each other: parity with the library itself is unpinned."""
import heapq
import struct

import numpy as np


def _enc14(a, b):
    a = a - 0x10000 if a & 0x8000 else a
    b = b - 0x10000 if b & 0x8000 else b
    return ((a + b) >> 1) & 0xFFFF, (a - b) & 0xFFFF


def _enc16(a, b):
    ao = (a + 0x8000) & 0xFFFF
    m, d = (ao + b) >> 1, ao - b
    if d < 0:
        m = (m + 0x8000) & 0xFFFF
    return m, d & 0xFFFF


def wavelet_encode(w, base, nx, ox, ny, oy, max_value):
    """In place on the list `w`: the plane starts at `base`, neighbours are ox / oy apart."""
    enc = _enc14 if max_value < (1 << 14) else _enc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        y = 0
        while y + p2 <= ny:
            x = 0
            while x + p2 <= nx:
                i00 = base + y * oy + x * ox
                i01, i10 = i00 + ox * p, i00 + oy * p
                i11 = i10 + ox * p
                t00, t01 = enc(w[i00], w[i01])
                t10, t11 = enc(w[i10], w[i11])
                w[i00], w[i10] = enc(t00, t10)
                w[i01], w[i11] = enc(t01, t11)
                x += p2
            if nx & p:
                i00 = base + y * oy + x * ox
                i10 = i00 + oy * p
                w[i00], w[i10] = enc(w[i00], w[i10])
            y += p2
        if ny & p:
            x = 0
            while x + p2 <= nx:
                i00 = base + y * oy + x * ox
                i01 = i00 + ox * p
                w[i00], w[i01] = enc(w[i00], w[i01])
                x += p2
        p, p2 = p2, p2 * 2


class _Bits:
    def __init__(self):
        self.parts = []
        self.n = 0

    def put(self, nbits, value):
        if nbits:
            self.parts.append(format(value, "0%db" % nbits))
            self.n += nbits

    def bytes(self):
        s = "".join(self.parts)
        s += "0" * (-len(s) % 8)
        return int(s, 2).to_bytes(len(s) // 8, "big") if s else b""


def _code_lengths(freq):
    """Huffman code lengths of the symbols with a count, by repeated merging of the two rarest groups."""
    heap = [(f, s, (s,)) for s, f in sorted(freq.items())]
    heapq.heapify(heap)
    length = dict.fromkeys(freq, 0)
    if len(heap) == 1:
        length[heap[0][1]] = 1
    while len(heap) > 1:
        fa, sa, ga = heapq.heappop(heap)
        fb, sb, gb = heapq.heappop(heap)
        for s in ga + gb:
            length[s] += 1
        heapq.heappush(heap, (fa + fb, min(sa, sb), ga + gb))
    return length


def huffman_encode(words, stats=None):
    """words: sequence of ints < 65536 -> the Huffman layer's bytes (20-byte header, table of lengths, codes)."""
    if len(words) == 0:
        return b""
    freq = {}
    for v in words:
        freq[v] = freq.get(v, 0) + 1
    lo, hi = min(freq), max(freq) + 1
    freq[hi] = 1                                   # the repeat symbol
    length = _code_lengths(freq)
    assert max(length.values()) <= 58
    # canonical codes: the longest start at 0, a shorter length goes on at half of where the longer one ended
    count = [0] * 59
    for l in length.values():
        count[l] += 1
    first, c = [0] * 59, 0
    for l in range(58, 0, -1):
        first[l] = c
        c = (c + count[l]) >> 1
    code = {}
    for s in sorted(length):
        code[s] = first[length[s]]
        first[length[s]] += 1
    table = _Bits()
    s = lo
    while s <= hi:
        l = length.get(s, 0)
        if l == 0:
            run = 1
            while s + run <= hi and run < 261 and length.get(s + run, 0) == 0:
                run += 1
            if run >= 6:
                table.put(6, 63)
                table.put(8, run - 6)
            elif run >= 2:
                table.put(6, 59 + run - 2)
            else:
                table.put(6, 0)
            s += run
        else:
            table.put(6, l)
            s += 1
    data = _Bits()
    i, n, repeats = 0, len(words), 0
    while i < n:
        v = words[i]
        run = 0
        while i + 1 + run < n and words[i + 1 + run] == v and run < 255:
            run += 1
        lv = length[v]
        if lv + length[hi] + 8 < lv * run:
            data.put(lv, code[v])
            data.put(length[hi], code[hi])
            data.put(8, run)
            repeats += 1
        else:
            for _ in range(run + 1):
                data.put(lv, code[v])
        i += run + 1
    if stats is not None:
        stats["longest_code"] = max(stats.get("longest_code", 0), max(length.values()))
        stats["repeats"] = stats.get("repeats", 0) + repeats
    tb = table.bytes()
    return struct.pack("<IIIII", lo, hi, len(tb), data.n, 0) + tb + data.bytes()


def piz_block(channels, stats=None):
    """channels: [(uint16 array (rows, cols * words_per_sample), words_per_sample)], in file order; HALF samples are
    one word, FLOAT samples two (low word first). Returns the PIZ block, or the plain bytes (row by row, channel by
    channel) when those are not longer."""
    rows_of_channels = [c[0] for c in channels]
    rows = rows_of_channels[0].shape[0]
    raw = b"".join(ch[y].astype("<u2").tobytes() for y in range(rows) for ch in rows_of_channels)
    words = np.concatenate([ch.reshape(-1) for ch in rows_of_channels]).astype(np.uint16)
    used = np.zeros(65536, bool)
    used[words] = True
    used[0] = False                                # 0 is always taken as in use, its bit stays clear
    bitmap = np.packbits(used, bitorder="little")
    filled = np.nonzero(bitmap)[0]
    lo, hi = (int(filled[0]), int(filled[-1])) if len(filled) else (8191, 0)
    used[0] = True
    dense = np.cumsum(used) - 1                    # word value -> its number among the values in use
    max_value = int(dense[-1])
    w = [int(v) for v in dense[words]]
    base = 0
    for ch, per in channels:
        cols_words = ch.shape[1]
        for j in range(per):
            wavelet_encode(w, base + j, cols_words // per, per, rows, cols_words, max_value)
        base += rows * cols_words
    coded = huffman_encode(w, stats)
    out = struct.pack("<HH", lo, hi) + (bitmap[lo:hi + 1].tobytes() if lo <= hi else b"") + struct.pack("<i", len(coded)) + coded
    if stats is not None:
        stats["max_value"] = max(stats.get("max_value", 0), max_value)
        stats["coded_blocks"] = stats.get("coded_blocks", 0) + (len(out) < len(raw))
    return out if len(out) < len(raw) else raw
