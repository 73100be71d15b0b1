"""BGZF inflate on the device (csrc/inflate.hip, RFC 1951 from the specification) against zlib: every block type (stored,
fixed, dynamic), every compression level, text and binary data, empty and maximal blocks, long codes, corrupt streams."""
import ctypes
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    return co.compress(data) + co.flush()


def device_inflate(streams, sizes):
    from wgsassign_amd import _lib, device
    ctx = device.get_context()
    comp = b"".join(streams)
    n = len(streams)
    in_len = np.array([len(s) for s in streams], dtype=np.uint32)
    in_off = np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.uint64)]).astype(np.uint64) if n else np.zeros(0, np.uint64)
    isize = np.array(sizes, dtype=np.uint32)
    out_off = np.concatenate([[0], np.cumsum(isize[:-1], dtype=np.uint64)]).astype(np.uint64) if n else np.zeros(0, np.uint64)
    total = int(isize.sum())
    out = np.zeros(max(total, 1), dtype=np.uint8)
    status = np.full(max(n, 1), 9, dtype=np.uint8)
    cbuf = np.frombuffer(comp + b"\0", dtype=np.uint8).copy()
    ms = ctypes.c_float()
    _lib.check(_lib.load().wgs_debug_inflate(ctx.handle, cbuf.ctypes.data, len(comp), in_off.ctypes.data, in_len.ctypes.data,
                                             out_off.ctypes.data, isize.ctypes.data, n, out.ctypes.data, total, status.ctypes.data,
                                             ctypes.byref(ms)))
    outs = [out[int(o):int(o) + int(s)].tobytes() for o, s in zip(out_off, isize)]
    return outs, status[:n], ms.value


def test_every_block_type_level_and_kind_of_data():
    rng = np.random.default_rng(0)
    beagle = ("\t".join("%.6f" % v for v in rng.random(9000)) + "\n").encode()[:65000]
    datas = [b"", b"a", b"abcabcabcabc" * 5000, bytes(rng.integers(0, 256, size=65536, dtype=np.uint8)),
             beagle, bytes(rng.integers(0, 4, size=65536, dtype=np.uint8)), b"\0" * 65536,
             bytes(rng.integers(0, 256, size=3, dtype=np.uint8)) * 20000,
             ("0.333333\t" * 7000).encode(), bytes(np.arange(65536, dtype=np.uint32).astype(np.uint8))]
    # a skewed distribution with > 200 distinct symbols gives codes longer than 11 bits (the walk beyond the table)
    p = 1.0 / np.arange(1, 257) ** 2.2
    datas.append(bytes(rng.choice(256, size=65536, p=p / p.sum()).astype(np.uint8)))
    streams, sizes, want = [], [], []
    for d in datas:
        for level in (0, 1, 4, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                streams.append(deflate(d[:65536], level, strat))
                sizes.append(len(d[:65536]))
                want.append(d[:65536])
    outs, status, ms = device_inflate(streams, sizes)
    bad = [i for i, (o, w, s) in enumerate(zip(outs, want, status)) if s != 0 or o != w]
    assert not bad, (bad[:10], len(streams))


def test_damaged_last_member_does_not_read_on():
    """The LAST stream of a launch made of input that never ends by itself -- empty stored blocks without the final bit,
    end-of-block codes of fixed blocks without the final bit, a dynamic header cut off in its trees: the lane must stop within a
    few bytes of its stream's end (status != 0) instead of decoding on through whatever follows the chunk in device memory;
    the good streams beside it are untouched.  (Before round 4 only `pos <= want` bounded the loop.)"""
    rng = np.random.default_rng(3)
    text = ("\t".join("%.6f" % v for v in rng.random(3000)) + "\n").encode()
    good = deflate(text)
    empty_stored = bytes([0x00, 0x00, 0x00, 0xFF, 0xFF]) * 4000          # 20 000 bytes of non-final empty stored blocks
    # fixed blocks holding only the end-of-block code, never final: 10 bits each (BFINAL = 0, BTYPE = 01, the 7-bit code 0000000)
    eob_fixed = np.packbits(np.tile(np.array([0, 1, 0] + [0] * 7, dtype=np.uint8), 4800), bitorder="little").tobytes()
    dyn = deflate(bytes(rng.integers(97, 123, size=20000, dtype=np.uint8)), level=9)
    assert (dyn[0] >> 1) & 3 == 2                                       # a dynamic block: cut inside its code-length section
    for tail in (empty_stored, eob_fixed, dyn[:12], empty_stored[:7]):
        streams = [good, good, tail]
        outs, status, _ = device_inflate(streams, [len(text), len(text), len(text)])
        assert status[0] == 0 and outs[0] == text and status[1] == 0 and outs[1] == text
        assert status[2] != 0
    # and the same streams with a good one behind them in the same buffer: it is not decoded into the damaged one's slot
    for tail in (empty_stored, eob_fixed):
        outs, status, _ = device_inflate([tail, good], [len(text), len(text)])
        assert status[0] != 0 and status[1] == 0 and outs[1] == text


def test_corrupt_streams_are_marked_not_followed():
    """Truncated input, a wrong output size, flipped bits, an invalid block type, garbage: status != 0, nothing written past
    the block's own slot (the neighbouring blocks of the same launch stay intact)."""
    rng = np.random.default_rng(1)
    text = ("\t".join("%.6f" % v for v in rng.random(5000)) + "\n").encode()
    good = deflate(text)
    cases = [good[:len(good) // 2], good + b"", good[:-1], bytes([good[0] | 0x06]) + good[1:], bytes(rng.integers(0, 256, size=4000, dtype=np.uint8)), b""]
    flips = []
    for k in range(40):
        b = bytearray(good)
        b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        flips.append(bytes(b))
    streams = [good] + cases + flips + [good]
    sizes = [len(text)] * len(streams)
    sizes[2] = len(text) - 5                                  # good stream, wrong size announced
    outs, status, _ = device_inflate(streams, sizes)
    assert status[0] == 0 and outs[0] == text and status[-1] == 0 and outs[-1] == text
    assert status[1] != 0 and status[2] != 0 and status[4] != 0 and status[6] != 0
    for i, s in enumerate(streams):
        try:
            ok = zlib.decompress(s, -15) == text and sizes[i] == len(text)
        except zlib.error:
            ok = False
        if ok:
            assert status[i] == 0 and outs[i] == text, i
        else:
            # a flipped bit may still give a stream the device accepts (zlib would too, or zlib only objects to what follows
            # the announced size): then the device's bytes must be what zlib makes of it
            if status[i] == 0:
                d = zlib.decompressobj(-15)
                assert d.decompress(s)[:sizes[i]] == outs[i], i
