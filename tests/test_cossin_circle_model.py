"""The algebra behind `cossin_circle` (idsp_amd/csrc/dds_dev.h), checked on the CPU against the oracle's restatement of src/cossin.rs:14-67.

The device function evaluates each output component as the high word of ONE 64-bit multiply-add on a full-circle table of 16-byte
entries {Bh_re, A_re, Bh_im, A_im} indexed by the top ten phase bits:

    component = hi32(Bh * d17 + (A << 32 | (Bh as u32))),   d17 = dphi << 17,   Bh = +-(s << 8) or +-(c << 7),   A = +-(c << 14) or +-(s << 15)

with the octant logic (swap, negations, the index reversal of odd octants) folded into which of the four forms an entry holds, and the low
addend word Bh doubling as the rounding correction (it carries into the high word exactly when the term is subtracted and the product is not
a whole multiple of 2^32).  This model rebuilds that table and the nine-instruction evaluation in numpy integers and compares it with
`oracle.spec.cossin` — through a vectorised restatement that is itself checked against the scalar spec — over random phases, +-300 around every
one of the 1024 table-entry boundaries (which include the octant edges) and exhaustively over the 2^22 phases of a few entries."""
import numpy as np

from oracle import spec

LUT = np.array(spec.cossin_table(), dtype=np.uint64)


def ref_vec(x):
    x = x.astype(np.uint32)
    ph = np.where(x & (1 << 29), ~x, x).astype(np.uint32)
    ph = ((ph << np.uint32(3)) >> np.uint32(10)).astype(np.uint32)
    lookup = LUT[ph >> 15].astype(np.int64)
    p = (ph & 0x7FFF).astype(np.int64) - (1 << 14)
    dphi = (p * 51471) >> 16
    c = (lookup & 0xFFFF) + (1 << 16)
    s = lookup >> 16
    c, s = (c << 14) - ((s * dphi) >> 7), (s << 15) + ((c * dphi) >> 8)
    o = x ^ (x >> np.uint32(1))
    sw = (o & (1 << 29)) != 0
    re, im = np.where(sw, s, c), np.where(sw, c, s)
    re = np.where(o & (1 << 30), -re, re)
    im = np.where(o & (1 << 31), -im, im)
    return (re & 0xFFFFFFFF).astype(np.uint32), (im & 0xFFFFFFFF).astype(np.uint32)


def build_table():
    """fill_cossin_circle: entry e = top ten phase bits -> [Bh_re, A_re, Bh_im, A_im]"""
    t = np.zeros((1024, 4), dtype=np.int64)
    for e in range(1024):
        x31, x30, x29 = (e >> 9) & 1, (e >> 8) & 1, (e >> 7) & 1
        raw = e & 127
        lk = int(LUT[127 - raw if x29 else raw])
        c, s = (lk & 0xFFFF) + 65536, lk >> 16
        sw, neg = x29 ^ x30, (x30 ^ x31, x31)
        for comp in range(2):
            is_sin = (comp == 1) != bool(sw)
            a, bh = (s << 15, c << 7) if is_sin else (c << 14, s << 8)
            sub = is_sin == bool(neg[comp])  # c' subtracts its term, s' adds it; negation flips that
            t[e, 2 * comp] = -bh if sub else bh
            t[e, 2 * comp + 1] = -a if neg[comp] else a
    return t


TABLE = build_table()


def circle_vec(x):
    """cossin_circle_fetch + cossin_circle_finish"""
    x = x.astype(np.uint32)
    xx = x ^ np.where(x & (1 << 29), np.uint32(0xFFFFFFFF), np.uint32(0))
    u15 = ((xx >> np.uint32(7)) & 0x7FFF).astype(np.int64)
    t2 = u15 * (2 * 51471) - 16384 * (2 * 51471)
    assert (np.abs(t2) < 2 ** 31).all()  # v_mad_u32_u24 result taken as an i32
    d17 = t2 & ~np.int64(0x1FFFF)
    e = (x >> np.uint32(22)).astype(np.int64)

    def comp(bh, a):
        total = bh * d17 + (a << 32) + (bh & 0xFFFFFFFF)  # v_mad_i64_i32 with the pair {Bh, A} as the addend; no int64 overflow: |.| < 2^63
        return ((total >> 32) & 0xFFFFFFFF).astype(np.uint32)

    return comp(TABLE[e, 0], TABLE[e, 1]), comp(TABLE[e, 2], TABLE[e, 3])


def _same(x):
    a, b = ref_vec(x), circle_vec(x)
    return bool((a[0] == b[0]).all() and (a[1] == b[1]).all())


def test_vectorised_restatement_is_the_spec():
    rng = np.random.default_rng(5)
    xs = list(rng.integers(0, 2 ** 32, 3000, dtype=np.uint64)) + [0, 1, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0x20000000, 0x1FFFFFFF, 0x3FFFFFFF, 0x40000000]
    re, im = ref_vec(np.array(xs, dtype=np.uint64).astype(np.uint32))
    for v, r, i in zip(xs, re, im):
        c, s = spec.cossin(int(np.int32(np.uint32(v))))
        assert (c & 0xFFFFFFFF) == int(r) and (s & 0xFFFFFFFF) == int(i), hex(int(v))


def test_table_properties_the_derivation_needs():
    bh = TABLE[:, [0, 2]]
    assert (bh != 0).all() and (np.abs(bh) < 2 ** 24).all() and (bh % 128 == 0).all()  # the carry argument: |Bh| < 2^24, Bh * d17 a multiple of 2^24
    assert (np.abs(TABLE[:, [1, 3]]) < 2 ** 31).all()


def test_random_phases():
    rng = np.random.default_rng(6)
    for _ in range(4):
        assert _same(rng.integers(0, 2 ** 32, 1 << 21, dtype=np.uint64).astype(np.uint32))


def test_every_entry_boundary():
    x = np.concatenate([((k << 22) + np.arange(-300, 300)) & 0xFFFFFFFF for k in range(1025)]).astype(np.uint32)
    assert _same(x)


def test_whole_entries_exhaustively():
    for e in (0, 127, 128, 640, 1023):
        assert _same(((e << 22) + np.arange(1 << 22)).astype(np.uint32)), e
