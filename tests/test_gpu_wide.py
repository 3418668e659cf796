"""GPU parity tests of the wide-picture paths: the streaming level kernels (planes >= 512 samples wide) and the
16-bit coefficient store with its escape plane, against the CPU oracle through the picture-level C-ABI calls.
Every case runs with the 16-bit store and the streaming kernels (the default), with the int32 store, and with the
tile kernels; all three must give the oracle's bytes."""
import os

import numpy as np
import pytest

from synth import noise_frame, synth
from vc2lib import KERNELS, make_params

pytestmark = pytest.mark.gpu


def _ctx(env):
    """a context created under the given environment switches (read once, at vc2hip_create)"""
    from vc2hip_py import Vc2Hip
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return Vc2Hip()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def variants():
    return {"default": _ctx({}), "store32": _ctx({"VC2HIP_STORE32": "1"}), "tiles": _ctx({"VC2HIP_NO_STREAM": "1"})}


def _fmt_cp(hip, w, h, cf, bits, kernel, depth, u, a, **kw):
    import vc2hip_py
    word_bytes = kw.pop("word_bytes", 2)
    fmt = vc2hip_py.picture_format(w, h, cf, bits, word_bytes)
    return fmt, vc2hip_py.coding_params(hip.lib, fmt, kernel, depth, u, a, **kw)


def _check(variants, oracle, raw, w, h, cf, bits, kernel, depth, u, a, **kw):
    p = make_params(w, h, cf, bits, kernel, depth, u, a, **kw)
    stream = oracle.encode_stream(p, raw, 1)
    dec, n = oracle.decode_stream(p, stream, 1)
    assert n == 1
    for name, hip in variants.items():
        fmt, cp = _fmt_cp(hip, w, h, cf, bits, kernel, depth, u, a, **kw)
        payload, _ = hip.encode_picture_hq(raw, fmt, cp)
        assert payload == stream[-13 - len(payload):-13], (name, "payload")
        assert hip.decode_picture(payload, fmt, cp) == dec, (name, "decode")
    return stream


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_wide_all_kernels(variants, oracle, kernel):
    # luma 1024 / chroma 512 wide: level 0 streams for both, level 1 only for luma, level 2 uses the tile kernels
    w, h, depth = 1024, 96, 3
    raw = noise_frame(w, h, "422", 10, seed=71)
    _check(variants, oracle, raw, w, h, "422", 10, kernel, depth, 1, 2, q=7, scalar=2)


@pytest.mark.parametrize("cf,bits,h,depth,u,a", [("444", 12, 80, 2, 2, 4), ("420", 8, 118, 2, 2, 4), ("422", 10, 72, 1, 4, 8),
                                                 ("422", 10, 270, 4, 1, 2)])
def test_wide_formats_padding_slices(variants, oracle, cf, bits, h, depth, u, a):
    # heights that need padding, one and two-byte words, several slice footprints, 4:2:0 (chroma half height)
    w = 1280
    wb = 1 if bits == 8 else 2
    raw = synth(w, h, cf, bits, 72, word_bytes=wb)
    _check(variants, oracle, raw, w, h, cf, bits, "DD97", depth, u, a, q=6, scalar=3, word_bytes=wb)
    _check(variants, oracle, raw, w, h, cf, bits, "LeGall", depth, u, a, q=0, scalar=8, word_bytes=wb)


def test_wide_cbr(variants, oracle):
    w, h, depth = 1024, 64, 3
    raw = synth(w, h, "422", 10, 73)
    _check(variants, oracle, raw, w, h, "422", 10, "DD97", depth, 1, 2, mode="HQ_CBR", s=40000, scalar=1)


def test_store16_escapes_both_directions(variants, oracle):
    """16-bit samples: transform coefficients far beyond 16 bits (encode-side escapes), and a quantiser index low
    enough that quantised values pass 32767 too (decode-side escapes) while staying inside the reference's 32-bit
    code word domain (|v| <= 65534)."""
    w, h, depth, u, a = 1024, 64, 2, 2, 4
    raw = noise_frame(w, h, "422", 16, seed=74)
    hip = variants["default"]
    ph, pw = oracle.padded_size(h, depth), oracle.padded_size(w, depth)
    hit = None
    for q in (16, 12, 10, 8, 7, 6, 5, 4, 3, 2):
        fmt, cp = _fmt_cp(hip, w, h, "422", 16, "LeGall", depth, u, a, q=q, scalar=8)
        try:
            payload, _ = hip.encode_picture_hq(raw, fmt, cp)
        except Exception:   # |quantised| > 65534: outside the reference's domain
            break
        y, cu, cv, _, _ = oracle.hq_unpack(np.frombuffer(payload, np.uint8), (ph, pw), (ph, pw // 2), depth,
                                           cp.y_slices, cp.x_slices, 0, 8)
        mx = max(int(np.abs(x).max()) for x in (y, cu, cv))
        if 32767 < mx <= 65534:
            hit = q
            _check(variants, oracle, raw, w, h, "422", 16, "LeGall", depth, u, a, q=q, scalar=8)
            break
    assert hit is not None, "no quantiser index puts quantised values between 32768 and 65534"
