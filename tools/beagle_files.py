"""Large BGZF Beagle files for the ingest measurements (bench.py extra.paths, tools/bench_reader.py) in seconds instead
of minutes: nothing is formatted or compressed per line.  The genotype-likelihood section of every line is drawn from a
pool of lines of a simulated low-depth matrix (tests/synth.make_beagle's model: Poisson depth 2, error 0.01 -- few
distinct likelihoods per site; the text deflates ~10 : 1 like the reference's bundled 2x files), each pool line
deflated ONCE (zlib level 6, its own window).  A BGZF member is then assembled from deflate blocks: per line a stored
block with the site name and alleles, followed by the pool line's blocks (closed with Z_FULL_FLUSH, or Z_FINISH for the
last line of the member); the member's CRC-32 is combined from the pieces' CRCs (the zero-advance operator of the CRC
for the fixed line length, as zlib's crc32_combine does).  Members hold whole lines, up to 60 000 bytes of text -- one
line per member at n = 2000, like the 64 KiB members bgzip / ANGSD write."""
import struct
import zlib

import numpy as np


def bgzf_member_raw(payload, crc, isize):
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(payload) + 8 - 1) +
            payload + struct.pack("<II", crc & 0xFFFFFFFF, isize))


def bgzf_member(chunk, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    return bgzf_member_raw(co.compress(chunk) + co.flush(), zlib.crc32(chunk), len(chunk))


def lowdepth_pool(n, pool, seed=1, depth=2.0):
    """(text rows as uint8 (pool, 27 n), values float32 (pool, 2 n)): '\\td.dddddd' per likelihood, three per individual."""
    rng = np.random.default_rng(seed)
    p = np.clip(rng.beta(0.8, 0.8, size=(pool, 1)) + rng.normal(0.0, 0.08, size=(pool, n)), 0.01, 0.99)
    geno = rng.binomial(2, p)
    d = rng.poisson(depth, size=(pool, n))
    e = 0.01
    alt = rng.binomial(d, np.array([e, 0.5, 1.0 - e])[geno])
    ref = d - alt
    l0, l1, l2 = (1 - e) ** ref * e ** alt, 0.5 ** d, (1 - e) ** alt * e ** ref
    tot = l0 + l1 + l2
    a = np.rint(l0 / tot * 1e6).astype(np.int64)
    b = np.rint(l1 / tot * 1e6).astype(np.int64)
    c = np.maximum(0, 1_000_000 - a - b)
    v = np.stack([a, b, c], axis=2).reshape(pool, 3 * n)
    txt = np.empty((pool, 3 * n, 9), dtype=np.uint8)
    txt[:, :, 0] = 9
    txt[:, :, 1] = 48 + v // 1_000_000
    txt[:, :, 2] = 46
    r = v % 1_000_000
    for k in range(6):
        txt[:, :, 3 + k] = 48 + (r // 10 ** (5 - k)) % 10
    vals = (v.reshape(pool, n, 3)[:, :, :2].reshape(pool, 2 * n) / 1e6).astype(np.float32)
    return txt.reshape(pool, 27 * n), vals


def crc_zero_advance_tables(nbytes):
    """T with crc32(x + y) == advance(crc32(x)) ^ crc32(y) for len(y) == nbytes, advance(c) = T[0][c & 255] ^ T[1][(c >> 8)
    & 255] ^ T[2][(c >> 16) & 255] ^ T[3][c >> 24]: the CRC register pushed through nbytes zero bytes (GF(2)-linear)."""
    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[k]) for k in range(32)]
    op = [0xEDB88320] + [1 << (k - 1) for k in range(1, 32)]      # one zero bit (reflected CRC-32)
    op = square(square(square(op)))                                # eight zero bits = one zero byte
    total = None                                                   # the operator for nbytes zero bytes, by binary powers
    k = nbytes
    while k:
        if k & 1:
            total = op if total is None else [times(op, total[i]) for i in range(32)]
        k >>= 1
        if k:
            op = square(op)
    if total is None:
        total = [1 << i for i in range(32)]
    return [[times(total, b << (8 * j)) for b in range(256)] for j in range(4)]


def write_lowdepth_bgzf(path, n, m, pool=512, seed=1, member=60000):
    """Writes the file; returns (text bytes, pool values (pool, 2n) float32, pool index of every line)."""
    txt, vals = lowdepth_pool(n, min(pool, m), seed)
    rows = [t.tobytes() + b"\n" for t in txt]
    row_len = len(rows[0])
    assert row_len + 32 <= member <= 65280, "one line must fit a BGZF member"
    mid, fin = [], []
    for r in rows:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        mid.append(co.compress(r) + co.flush(zlib.Z_FULL_FLUSH))   # byte-aligned, not final: more blocks may follow
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        fin.append(co.compress(r) + co.flush())
    crc_row = [zlib.crc32(r) for r in rows]
    T = crc_zero_advance_tables(row_len)
    pick = np.random.default_rng(seed + 1).integers(0, len(rows), size=m)
    head = ("marker\tallele1\tallele2\t" + "\t".join("I%d\tI%d\tI%d" % (i, i, i) for i in range(n)) + "\n").encode()
    total = len(head)
    with open(path, "wb") as fh:
        for i in range(0, len(head), member):
            fh.write(bgzf_member(head[i:i + member]))
        out = []
        s = 0
        while s < m:
            parts, crc, size = [], 0, 0
            while s < m:
                name = b"chr7_%d\tA\tC" % (s + 1)
                if size and size + len(name) + row_len > member:
                    break
                k = pick[s]
                last = s + 1 == m or size + len(name) + row_len + 24 + row_len > member
                parts.append(b"\x00" + struct.pack("<HH", len(name), len(name) ^ 0xFFFF) + name)      # stored block, not final
                parts.append(fin[k] if last else mid[k])
                crc = zlib.crc32(name, crc)
                crc = T[0][crc & 255] ^ T[1][(crc >> 8) & 255] ^ T[2][(crc >> 16) & 255] ^ T[3][crc >> 24] ^ crc_row[k]
                size += len(name) + row_len
                s += 1
                if last:
                    break
            out.append(bgzf_member_raw(b"".join(parts), crc, size))
            total += size
            if len(out) >= 4096:
                fh.write(b"".join(out))
                out = []
        fh.write(b"".join(out))
        fh.write(bgzf_member(b""))
    return total, vals, pick
