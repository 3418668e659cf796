"""The oracle's LD slice syntax and HQ picture header against the reference's OWN compiled bit I/O (oracle/_ref:
/root/reference/src/Library/src/VLC.cpp built unmodified, driven by oracle/ref_vlc_wrap.cpp exactly like
Slices.cpp:195-303 and DataUnit.cpp:236-259 drive it).  Pins: the 7-bit quantiser index, the luma length field of
intlog2(8 * bytes - 7) bits, the bounded luma codes, the interleaved u / v codes bounded by the remainder, flush /
align padding, and the UnsignedVLC / Boolean / Bytes strings of the picture header incl. the major-version-3 flags.
(What the reference does BEFORE these bits -- which coefficient goes where -- needs its Boost containers and is
pinned only by reading: see DESIGN.md section 2.)"""
import ctypes as C

import numpy as np
import pytest

from vc2lib import KERNELS, load_ref_vlc, make_params

i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def ref():
    lib = load_ref_vlc()
    if lib is None:
        pytest.skip("oracle/_ref was never built (no reference checkout)")
    lib.ref_ld_slice_write.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i32p, C.c_int, i32p, C.c_int, u8p, C.c_long]
    lib.ref_ld_slice_write.restype = C.c_long
    lib.ref_ld_slice_read.argtypes = [u8p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), i32p, i32p]
    lib.ref_ld_slice_read.restype = C.c_long
    lib.ref_hq_picture_header.argtypes = [C.c_ulong, C.c_int] + [C.c_uint] * 6 + [u8p, C.c_long]
    lib.ref_hq_picture_header.restype = C.c_long
    return lib


def intlog2(v):          # utils::intlog2, Utils.cpp:40-48
    n, v = 0, v - 1
    while v > 0:
        v >>= 1
        n += 1
    return n


def coding_order(plane, depth):
    """coefficients of one slice (a tile of the in-place transform) in subband order: LL, then per level HL, LH, HH,
    raster inside a band (WaveletTransform.cpp:428-450 as described in SURVEY.md Appendix D)"""
    out = [plane[::1 << depth, ::1 << depth].ravel()]
    for lv in range(1, depth + 1):
        s = 1 << (depth + 1 - lv)
        o = s // 2
        out += [plane[0::s, o::s].ravel(), plane[o::s, 0::s].ravel(), plane[o::s, o::s].ravel()]
    return np.ascontiguousarray(np.concatenate(out), np.int32)


def slice_bits(ref, vals):
    gross = count = 0
    for v in vals:
        nb = ref.ref_svlc_numbits(int(v))
        gross += nb
        if nb > 1:
            count = gross
    return count


@pytest.mark.parametrize("seed,ys,xs,depth,extra", [(1, 1, 1, 2, 0), (2, 2, 3, 2, 5), (3, 3, 2, 3, 1), (4, 1, 4, 1, 3)])
def test_ld_slices_against_reference_bit_io(oracle, ref, seed, ys, xs, depth, extra):
    rng = np.random.default_rng(seed)
    sh, sw = 2 << depth, 4 << depth                 # luma slice; chroma half as wide (4:2:2)
    budget = ys * xs * sh * sw * 2 + extra          # a byte per coefficient (uneven slice sizes when extra != 0)
    y = rng.integers(-40, 41, size=(ys * sh, xs * sw)).astype(np.int32)
    u = rng.integers(-20, 21, size=(ys * sh, xs * sw // 2)).astype(np.int32)
    v = rng.integers(-20, 21, size=(ys * sh, xs * sw // 2)).astype(np.int32)
    for p in (y, u, v):
        p[rng.random(p.shape) < 0.5] = 0
    qidx = rng.integers(0, 100, size=(ys, xs)).astype(np.int32)
    sb = oracle.slice_bytes(ys, xs, budget, 1)
    packed = oracle.ld_pack(y, u, v, depth, qidx, sb)
    assert packed.size == int(sb.sum())
    pos = 0
    for i in range(ys):
        for j in range(xs):
            nbytes = int(sb[i, j])
            ty = coding_order(y[i * sh:(i + 1) * sh, j * sw:(j + 1) * sw], depth)
            tu = coding_order(u[i * sh:(i + 1) * sh, j * sw // 2:(j + 1) * sw // 2], depth)
            tv = coding_order(v[i * sh:(i + 1) * sh, j * sw // 2:(j + 1) * sw // 2], depth)
            uv = np.ascontiguousarray(np.stack([tu, tv], 1).ravel(), np.int32)
            split = intlog2(8 * nbytes - 7)
            ybits = slice_bits(ref, ty)
            out = np.zeros(nbytes + 16, np.uint8)
            n = ref.ref_ld_slice_write(int(qidx[i, j]), nbytes, split, ybits, ty, ty.size, uv, uv.size, out, out.size)
            assert n == nbytes, (i, j, n)
            assert np.array_equal(out[:n], packed[pos:pos + nbytes]), (i, j)
            # the reference's reader on the oracle's bytes
            q, yb = C.c_int(), C.c_int()
            ry, ruv = np.zeros(ty.size, np.int32), np.zeros(uv.size, np.int32)
            used = ref.ref_ld_slice_read(np.ascontiguousarray(packed[pos:pos + nbytes]), nbytes, nbytes, split, ty.size, uv.size,
                                         C.byref(q), C.byref(yb), ry, ruv)
            assert used == nbytes and q.value == qidx[i, j] and yb.value == ybits
            assert np.array_equal(ry, ty) and np.array_equal(ruv, uv)
            pos += nbytes
    # and the oracle's own reader
    gy, gu, gv, gq, used = oracle.ld_unpack(packed, y.shape, u.shape, depth, sb)
    assert used == packed.size and np.array_equal(gq, qidx)
    assert np.array_equal(gy, y) and np.array_equal(gu, u) and np.array_equal(gv, v)


def test_ld_slice_truncation_reads_ones_like_the_reference(oracle, ref):
    """arbitrary bits behind a valid header, with a luma bound that cuts codes short: the reference's bounded reader
    returns '1' bits past a bound (zero coefficients) -- the oracle's unpacker must deliver the same values"""
    rng = np.random.default_rng(9)
    depth, sh, sw, nbytes = 2, 8, 16, 40
    split = intlog2(8 * nbytes - 7)
    for ybits in (0, 1, 37, 150, 8 * nbytes - 7 - split):
        bits = format(77, "07b") + format(ybits, f"0{split}b") + "".join(rng.choice(["0", "1"], size=8 * nbytes))
        blob = np.frombuffer(int(bits[:8 * nbytes], 2).to_bytes(nbytes, "big"), np.uint8).copy()
        sb = np.array([[nbytes]], np.int32)
        gy, gu, gv, gq, used = oracle.ld_unpack(blob, (sh, sw), (sh, sw // 2), depth, sb)
        q, yb = C.c_int(), C.c_int()
        ny, nuv = sh * sw, sh * sw
        ry, ruv = np.zeros(ny, np.int32), np.zeros(nuv, np.int32)
        ref.ref_ld_slice_read(blob, nbytes, nbytes, split, ny, nuv, C.byref(q), C.byref(yb), ry, ruv)
        assert q.value == gq[0, 0] == 77 and yb.value == ybits
        assert np.array_equal(ry, coding_order(gy, depth)), ybits
        assert np.array_equal(ruv[0::2], coding_order(gu, depth)) and np.array_equal(ruv[1::2], coding_order(gv, depth)), ybits


@pytest.mark.parametrize("kernel,depth,u,a,prefix,scalar,bits,major", [("LeGall", 2, 2, 4, 0, 1, 10, 2), ("DD97", 3, 1, 2, 3, 5, 10, 2),
                                                                       ("Haar0", 1, 4, 8, 0, 2, 16, 3)])
def test_hq_picture_header_against_reference_bit_io(oracle, ref, kernel, depth, u, a, prefix, scalar, bits, major):
    """picture number + transform parameters of every HQ picture of an oracle stream, against the same fields written by
    the reference's Bytes / UnsignedVLC / Boolean / align (16-bit video forces major version 3: two more flags)"""
    from synth import synth
    w, h = 128, 64
    raw = synth(w, h, "422", bits, 5, frames=2)
    p = make_params(w, h, "422", bits, kernel, depth, u, a, q=30, scalar=scalar, prefix=prefix)
    stream = oracle.encode_stream(p, raw, 2)
    ph, pw = oracle.padded_size(h, depth), oracle.padded_size(w, depth)
    sy, sx = ph // (u << depth), pw // (a << depth)
    pos, pics = 0, 0
    while pos < len(stream):
        assert stream[pos:pos + 4] == b"BBCD"
        code, nxt = stream[pos + 4], int.from_bytes(stream[pos + 5:pos + 9], "big")
        if code == 0xE8:
            want = np.zeros(64, np.uint8)
            n = ref.ref_hq_picture_header(pics, major, KERNELS[kernel], depth, sx, sy, prefix, scalar, want, want.size)
            assert bytes(want[:n]) == stream[pos + 13:pos + 13 + n], (code, pics)
            other = np.zeros(64, np.uint8)
            m = ref.ref_hq_picture_header(pics, 5 - major, KERNELS[kernel], depth, sx, sy, prefix, scalar, other, other.size)
            assert bytes(other[:m]) != stream[pos + 13:pos + 13 + m]       # the version flags do change the bits
            pics += 1
        if nxt == 0:
            break
        pos += nxt
    assert pics == 2


@pytest.mark.gpu
def test_hip_ld_slices_against_reference_bit_io(hip, oracle, ref):
    """the PRODUCT's LD slice writer and reader against the reference's compiled bit I/O, slice by slice"""
    rng = np.random.default_rng(21)
    ys, xs, depth = 3, 4, 2
    sh, sw = 2 << depth, 4 << depth
    y = rng.integers(-60, 61, size=(ys * sh, xs * sw)).astype(np.int32)
    u = rng.integers(-30, 31, size=(ys * sh, xs * sw // 2)).astype(np.int32)
    v = rng.integers(-30, 31, size=(ys * sh, xs * sw // 2)).astype(np.int32)
    for p in (y, u, v):
        p[rng.random(p.shape) < 0.4] = 0
    qidx = rng.integers(0, 100, size=(ys, xs)).astype(np.int32)
    sb = oracle.slice_bytes(ys, xs, ys * xs * sh * sw * 2 + 7, 1)
    packed = hip.ld_pack(y, u, v, depth, qidx, sb)
    pos = 0
    for i in range(ys):
        for j in range(xs):
            nbytes = int(sb[i, j])
            ty = coding_order(y[i * sh:(i + 1) * sh, j * sw:(j + 1) * sw], depth)
            tu = coding_order(u[i * sh:(i + 1) * sh, j * sw // 2:(j + 1) * sw // 2], depth)
            tv = coding_order(v[i * sh:(i + 1) * sh, j * sw // 2:(j + 1) * sw // 2], depth)
            uv = np.ascontiguousarray(np.stack([tu, tv], 1).ravel(), np.int32)
            split = intlog2(8 * nbytes - 7)
            out = np.zeros(nbytes + 16, np.uint8)
            n = ref.ref_ld_slice_write(int(qidx[i, j]), nbytes, split, slice_bits(ref, ty), ty, ty.size, uv, uv.size, out, out.size)
            assert n == nbytes and np.array_equal(out[:n], packed[pos:pos + nbytes]), (i, j)
            pos += nbytes
    gy, gu, gv, gq = hip.ld_unpack(packed, y.shape, u.shape, depth, sb)[:4]
    assert np.array_equal(gq, qidx) and np.array_equal(gy, y) and np.array_equal(gu, u) and np.array_equal(gv, v)
