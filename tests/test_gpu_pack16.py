"""GPU parity tests of the round-4 slice coders k_hq_pack16 / k_hq_pack16w (csrc/vc2hip_pack16.h) on the slice geometries
they take -- 32 x 16 slices at depth 4 in 4:2:2 (BASELINE cfg 2 / 3) and 32 x 32 slices at depth 5 in 4:4:4 (cfg 4) --
against the CPU oracle, over the whole range of their domain: the table path, the general coder behind it (quotients
beyond the table, strings beyond 63 bits per eight coefficients, escapes of the 16-bit store), head coefficients beyond
16 and beyond 2^20 bits, HQ_CBR, and the error paths.  A last test runs the -DVC2HIP_ABLATE build, which counts why
wavefronts leave the table path, and asserts that these pictures really reach both sides."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from synth import noise_frame, synth
from vc2lib import make_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fmt_cp(hip, w, h, cf, bits, kernel, depth, u, a, **kw):
    import vc2hip_py
    fmt = vc2hip_py.picture_format(w, h, cf, bits, 2)
    return fmt, vc2hip_py.coding_params(hip.lib, fmt, kernel, depth, u, a, **kw)


def _check(hip, oracle, raw, w, h, cf, bits, kernel, depth, u, a, **kw):
    p = make_params(w, h, cf, bits, kernel, depth, u, a, **kw)
    stream = oracle.encode_stream(p, raw, 1)
    dec, n = oracle.decode_stream(p, stream, 1)
    fmt, cp = _fmt_cp(hip, w, h, cf, bits, kernel, depth, u, a, **kw)
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    assert payload == stream[-13 - len(payload):-13], "payload"
    assert hip.decode_picture(payload, fmt, cp) == dec, "decode"
    return payload


def _half_noise(w, h, cf, bits, seed):
    """left half the smooth generator picture, right half uniform noise: slices of both kinds in one picture"""
    a = np.frombuffer(synth(w, h, cf, bits, seed), ">u2").copy()
    b = np.frombuffer(noise_frame(w, h, cf, bits, seed + 1), ">u2")
    cw = w if cf == "444" else w // 2
    pos = 0
    for pw in (w, cw, cw):
        pa = a[pos:pos + pw * h].reshape(h, pw)
        pa[:, pw // 2:] = b[pos:pos + pw * h].reshape(h, pw)[:, pw // 2:]
        pos += pw * h
    return a.astype(">u2").tobytes()


W, H = 2048, 256   # 64 x 16 slices of 32 x 16; the smallest picture whose deepest level still fills a tile of the fast level
                   # kernels in every component (chroma 1024 / 8 = 128 columns, 256 / 8 = 32 rows) -- below that the library
                   # keeps the int32 store and these kernels are not used (the last test checks that they are)


@pytest.mark.parametrize("q", [0, 3, 9, 16, 30, 61])
def test_pack16_smooth_noise_and_mixed_pictures(hip, oracle, q):
    """q = 0 .. 9: quotients far beyond +-126 in the noisy slices (the general coder), inside it in the smooth ones;
    16: the bench's index; 61: nearly everything quantises to zero (components whose body is all zeros)"""
    for raw in (synth(W, H, "422", 10, 81), noise_frame(W, H, "422", 10, seed=82), _half_noise(W, H, "422", 10, 83)):
        _check(hip, oracle, raw, W, H, "422", 10, "DD97", 4, 1, 2, q=q, scalar=8 if q < 10 else 4)


@pytest.mark.parametrize("kernel", ["LeGall", "Haar1", "Fidelity", "Daub97"])
def test_pack16_other_wavelets_prefix_and_scalars(hip, oracle, kernel):
    raw = _half_noise(W, H, "422", 10, 84)
    _check(hip, oracle, raw, W, H, "422", 10, kernel, 4, 1, 2, q=11, scalar=8, prefix=2)
    _check(hip, oracle, raw, W, H, "422", 10, kernel, 4, 1, 2, q=20, scalar=5)


def test_pack16_sixteen_bit_samples_escape_the_store(hip, oracle):
    """16-bit noise: body and head coefficients beyond 16 bits (escapes on the encoder's side), at indices where the
    quantised values still fit the reference's 32-bit code words"""
    raw = noise_frame(W, H, "422", 16, seed=85)
    for q in (24, 40):
        _check(hip, oracle, raw, W, H, "422", 16, "DD97", 4, 1, 2, q=q, scalar=8)


def test_pack16_cbr(hip, oracle):
    for raw in (synth(W, H, "422", 10, 86), _half_noise(W, H, "422", 10, 87)):
        _check(hip, oracle, raw, W, H, "422", 10, "DD97", 4, 1, 2, mode="HQ_CBR", s=W * H // 2, scalar=2)


def test_pack16_errors_are_the_reference_s(hip, oracle):
    from vc2hip_py import Vc2HipError
    raw = noise_frame(W, H, "422", 10, seed=88)
    fmt, cp = _fmt_cp(hip, W, H, "422", 10, "DD97", 4, 1, 2, q=0, scalar=1)   # 255 bytes per component do not hold it
    with pytest.raises(Vc2HipError, match="Slice scalar is too small"):
        hip.encode_picture_hq(raw, fmt, cp)
    p = make_params(W, H, "422", 10, "DD97", 4, 1, 2, q=0, scalar=1)
    with pytest.raises(Exception, match="Slice scalar is too small"):
        oracle.encode_stream(p, raw, 1)


# ---- large slices: a wavefront per component (k_hq_pack16w)
WW, HW = 2048, 512   # 64 x 16 slices of 32 x 32 in 4:4:4 (2048 / 16 = 128 columns, 512 / 16 = 32 rows at the deepest level)


@pytest.mark.parametrize("q", [20, 28, 40])
def test_pack16w_large_slices(hip, oracle, q):
    """12-bit 4:4:4, Fidelity, five levels (BASELINE cfg 4's coding): the LL coefficients pass 2^20 (the head's exact
    integer division; below q = 20 their quotients pass 65534, the reference's own 32-bit code limit, and the library
    refuses the picture), q = 20 sends noisy components through the general coder one wavefront at a time"""
    for raw in (synth(WW, HW, "444", 12, 91), _half_noise(WW, HW, "444", 12, 92)):
        _check(hip, oracle, raw, WW, HW, "444", 12, "Fidelity", 5, 1, 1, q=q, scalar={20: 12, 28: 12, 40: 8}[q])


def test_pack16w_dd97_cbr_and_prefix(hip, oracle):
    raw = _half_noise(WW, HW, "444", 10, 93)
    _check(hip, oracle, raw, WW, HW, "444", 10, "DD97", 5, 1, 1, mode="HQ_CBR", s=WW * HW, scalar=4)
    _check(hip, oracle, raw, WW, HW, "444", 10, "DD97", 5, 1, 1, q=7, scalar=12, prefix=3)


def test_both_paths_of_the_new_coders_are_reached(oracle, tmp_path):
    """the ablation build counts, per launch, the wavefronts of k_hq_pack16 / k_hq_pack16w and why they left the table
    path (VC2HIP_P16_STATS): the pictures above must put wavefronts on BOTH sides, or these tests prove less than they say"""
    lib = os.path.join(ROOT, "vc2-reference_amd", "libvc2hip_ablate.so")
    if not os.path.exists(lib):
        pytest.skip("libvc2hip_ablate.so not built")
    code = f"""
import sys
sys.path.insert(0, {os.path.join(ROOT, 'vc2-reference_amd')!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import vc2hip_py
from synth import synth, noise_frame
hip = vc2hip_py.Vc2Hip(0)
for (w, h, cf, bits, k, d, u, a, q, sc, noise) in (({W}, {H}, "422", 10, "DD97", 4, 1, 2, 16, 4, False), ({W}, {H}, "422", 10, "DD97", 4, 1, 2, 0, 8, True),
                                               ({WW}, {HW}, "444", 12, "Fidelity", 5, 1, 1, 40, 8, False), ({WW}, {HW}, "444", 12, "Fidelity", 5, 1, 1, 20, 12, True)):
    raw = noise_frame(w, h, cf, bits, seed=5) if noise else synth(w, h, cf, bits, 5)
    fmt = vc2hip_py.picture_format(w, h, cf, bits)
    cp = vc2hip_py.coding_params(hip.lib, fmt, k, d, u, a, q=q, scalar=sc)
    hip.encode_picture_hq(raw, fmt, cp)
"""
    env = dict(os.environ, VC2HIP_LIB=lib, VC2HIP_P16_STATS="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    stats = [tuple(int(x) for x in re.findall(r"\d+", line.split("pack16:")[1])) for line in out.stderr.splitlines() if "pack16:" in line]
    assert len(stats) == 4, out.stderr[-2000:]
    (w0, _, _, _, g0), (w1, _, big1, _, g1), (w2, _, _, _, g2), (w3, _, _, _, g3) = [s[:5] for s in stats]
    assert w0 == 1024 and g0 == 0                   # smooth, q = 16: every slice on the table path
    assert w1 == 1024 and g1 > 900 and big1 > 900   # noise, q = 0: quotients beyond the table nearly everywhere
    assert w2 == 3 * 1024 and g2 == 0               # large slices, 12-bit Fidelity: heads beyond 2^20 stay on the table path
    assert w3 == 3 * 1024 and g3 > 2500


@pytest.mark.parametrize("n,flags", [(113, ""), (3, "SINGLE_PASS_VBR"), (113, "TWO_PASS_VBR")])
def test_one_pass_coder_is_the_default_from_112_pictures_on(oracle, n, flags):
    """Round 6: k_hq_pack16 in its look-back form writes every slice where it belongs in the payload -- no slots, no scan of
    the sizes, no compaction.  The library takes it by itself from 112 pictures per call on (the grid puts the same tile of
    all pictures side by side; with fewer the predecessor tile is too close), with VC2HIP_FLAG_SINGLE_PASS_VBR for any batch,
    never with VC2HIP_FLAG_TWO_PASS_VBR.  2176 x 272: 68 x 17 slices (the 16-bit store's kernels take widths that are multiples of
    128 samples, so a picture's slices always fill whole tiles of four: tools/probe/pack16_geoms.py); five different pictures in turn -- smooth, noise (the general coder: quotients
    beyond the table), half and half -- so that neighbouring pictures' slices differ in length; every payload, every length
    and every decoded picture against the oracle, and the library's launch profile says which path ran."""
    import torch
    import vc2hip_py
    hip = vc2hip_py.Vc2Hip(flags=sum(vc2hip_py.FLAGS[f] for f in flags.split(",") if f))
    w, h = 2176, 272
    kinds = [synth(w, h, "422", 10, 91), noise_frame(w, h, "422", 10, seed=92), _half_noise(w, h, "422", 10, 93),
             synth(w, h, "422", 10, 94), _half_noise(w, h, "422", 10, 95)]
    p = make_params(w, h, "422", 10, "DD97", 4, 1, 2, q=9, scalar=8)
    want = []
    for raw in kinds:
        stream = oracle.encode_stream(p, raw, 1)
        dec, _ = oracle.decode_stream(p, stream, 1)
        want.append((stream, dec))
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "DD97", 4, 1, 2, q=9, scalar=8)
    rb = hip.raw_picture_bytes(fmt)
    stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    dev = torch.device("cuda:0")
    d_raw = torch.frombuffer(bytearray(b"".join(kinds[k % 5] for k in range(n))), dtype=torch.uint8).to(dev)
    for call in range(2):   # (twice: the status words of the first call must not leak into the second)
        d_pay = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
        d_len = torch.zeros(n, dtype=torch.int64, device=dev)
        d_out = torch.zeros(n * rb, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        hip.profile_reset(); hip.profile_enable(True)
        hip.encode_batch_dev(d_raw.data_ptr(), n, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
        hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), n, fmt, cp, d_out.data_ptr())
        hip.sync()
        hip.profile_enable(False)
        seen = {k for k, v in hip.profile().items() if v[0] > 0}
        one_pass = flags == "SINGLE_PASS_VBR" or (flags == "" and n >= 112)
        assert "hq_pack" in seen and ("slice_compact" in seen) == (not one_pass), (seen, n, flags)
        lens = d_len.cpu().tolist()
        pay = d_pay.cpu().numpy()
        out = d_out.cpu().numpy()
        for k in range(n):
            stream, dec = want[k % 5]
            body = bytes(pay[k * stride:k * stride + lens[k]])
            assert body == stream[-13 - len(body):-13] and len(body) > 1000, (k, call)
            assert out[k * rb:(k + 1) * rb].tobytes() == dec, (k, call)
    hip.close()
