"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs (bit-exact: all integer work)."""
import hashlib
import json
import os

import numpy as np
import pytest

from synth import noise_frame, synth
from vc2lib import KERNELS, make_params

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_digests.json")))


def _planes(oracle, raw, w, h, cf, bits, word_bytes=2):
    cw = w if cf == "444" else w // 2
    ch = h // 2 if cf == "420" else h
    n0 = w * h * word_bytes
    n1 = cw * ch * word_bytes
    y = oracle.ingest(raw[:n0], word_bytes, bits, (h, w))
    u = oracle.ingest(raw[n0:n0 + n1], word_bytes, bits, (ch, cw))
    v = oracle.ingest(raw[n0 + n1:n0 + 2 * n1], word_bytes, bits, (ch, cw))
    return y, u, v


@pytest.mark.parametrize("kernel", list(KERNELS))
@pytest.mark.parametrize("shape,depth", [((48, 80), 3), ((44, 70), 2), ((270, 130), 1), ((64, 96), 4)])
def test_dwt_forward_matches_oracle(hip, oracle, kernel, shape, depth):
    rng = np.random.default_rng(11)
    plane = rng.integers(-512, 512, size=shape).astype(np.int32)
    want = oracle.dwt_forward(plane, KERNELS[kernel], depth)
    got = hip.dwt_forward(plane, KERNELS[kernel], depth)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kernel", list(KERNELS))
@pytest.mark.parametrize("shape,depth", [((48, 80), 3), ((44, 70), 2), ((64, 96), 4)])
def test_dwt_inverse_matches_oracle(hip, oracle, kernel, shape, depth):
    rng = np.random.default_rng(12)
    ph, pw = oracle.padded_size(shape[0], depth), oracle.padded_size(shape[1], depth)
    coef = rng.integers(-3000, 3000, size=(ph, pw)).astype(np.int32)
    want = oracle.dwt_inverse(coef, KERNELS[kernel], depth, shape)
    got = hip.dwt_inverse(coef, KERNELS[kernel], depth, shape)
    assert np.array_equal(got, want)


def test_dwt_large_plane_all_tiles(hip, oracle):
    # bigger than one tile in both directions, ragged tile edges, extreme values (12-bit range)
    rng = np.random.default_rng(13)
    plane = (rng.integers(0, 2, size=(400, 720)) * 4095 - 2048).astype(np.int32)
    for name in ("DD97", "Fidelity", "LeGall"):
        c = oracle.dwt_forward(plane, KERNELS[name], 4)
        assert np.array_equal(hip.dwt_forward(plane, KERNELS[name], 4), c), name
        assert np.array_equal(hip.dwt_inverse(c, KERNELS[name], 4, plane.shape), plane), name


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_dwt_fast_path_all_kernels(hip, oracle, kernel):
    # planes larger than the 32 x 128 register-blocked tile (ragged: 200 x 264, 136 x 392), every
    # wavelet, both directions; deeper levels fall back to the generic kernel (mixed pipeline)
    rng = np.random.default_rng(15)
    for shape, depth in (((200, 264), 3), ((136, 392), 2), ((64, 128), 1)):
        plane = rng.integers(-2048, 2048, size=shape).astype(np.int32)
        c = oracle.dwt_forward(plane, KERNELS[kernel], depth)
        assert np.array_equal(hip.dwt_forward(plane, KERNELS[kernel], depth), c), (kernel, shape)
        coef = rng.integers(-5000, 5000, size=c.shape).astype(np.int32)
        want = oracle.dwt_inverse(coef, KERNELS[kernel], depth, shape)
        assert np.array_equal(hip.dwt_inverse(coef, KERNELS[kernel], depth, shape), want), (kernel, shape)


def test_quantise_dequantise_match_oracle(hip, oracle):
    rng = np.random.default_rng(14)
    depth, ys, xs = 3, 4, 5
    coef = rng.integers(-20000, 20000, size=(ys * 16, xs * 8)).astype(np.int32)
    qidx = rng.integers(0, 60, size=(ys, xs)).astype(np.int32)
    for name in ("DD97", "Fidelity", "Haar0"):
        qm = oracle.quant_matrix(KERNELS[name], depth)
        q_want = oracle.quantise_np(coef, depth, qidx, qm)
        q_got = hip.quantise_np(coef, depth, qidx, qm)
        assert np.array_equal(q_got, q_want)
        assert np.array_equal(hip.dequantise_np(q_want, depth, qidx, qm),
                              oracle.dequantise_np(q_want, depth, qidx, qm))
        assert np.array_equal(hip.dequantise_ld(q_want, depth, qidx, qm),
                              oracle.dequantise_ld(q_want, depth, qidx, qm))


def test_quantiser_full_index_range(hip, oracle):
    # every quantiser index the reference table serves with a positive factor (0..115), large values
    rng = np.random.default_rng(16)
    depth, ys, xs = 2, 29, 4
    coef = rng.integers(-(1 << 28), 1 << 28, size=(ys * 4, xs * 8)).astype(np.int32)
    coef[::3, ::5] = rng.integers(-40, 40, size=coef[::3, ::5].shape)
    qidx = np.arange(ys * xs, dtype=np.int32).reshape(ys, xs)
    qm = np.zeros(7, np.int32)
    assert np.array_equal(hip.quantise_np(coef, depth, qidx, qm), oracle.quantise_np(coef, depth, qidx, qm))


def test_quantiser_index_limit_error(hip, oracle):
    from vc2hip_py import Vc2HipError
    coef = np.ones((16, 16), np.int32)
    qm = np.zeros(7, np.int32)
    with pytest.raises(Vc2HipError, match="quantization index exceeds maximum implemented value."):
        hip.quantise_np(coef, 2, np.full((1, 1), 125, np.int32), qm)
    hip.quantise_np(coef, 2, np.full((1, 1), 119, np.int32), qm)


def _quantised_planes(oracle, seed, lshape, cshape, depth, ys, xs, kernel, q, spread=400):
    rng = np.random.default_rng(seed)
    qm = oracle.quant_matrix(KERNELS[kernel], depth)
    qidx = np.full((ys, xs), q, np.int32)
    out = []
    for shape in (lshape, cshape, cshape):
        c = (rng.standard_normal(shape) * spread).astype(np.int32)
        c[rng.random(shape) < 0.4] = 0
        out.append(oracle.quantise_np(c, depth, qidx, qm))
    return out, qidx, qm


@pytest.mark.parametrize("prefix,scalar", [(0, 1), (2, 3)])
def test_hq_pack_vbr_matches_oracle(hip, oracle, prefix, scalar):
    depth, ys, xs = 3, 6, 5
    (y, u, v), qidx, _ = _quantised_planes(oracle, 21, (ys * 8, xs * 16), (ys * 8, xs * 8), depth, ys, xs, "DD97", 8)
    qidx[1, 2] = 33
    want = oracle.hq_pack(y, u, v, depth, qidx, prefix, scalar)
    got = hip.hq_pack(y, u, v, depth, qidx, prefix, scalar)
    assert bytes(got) == bytes(want)


def test_hq_pack_single_pass_option(oracle):
    # VC2HIP_FLAG_SINGLE_PASS_VBR: slice offsets by decoupled look-back inside the pack kernel
    import vc2hip_py
    h2 = vc2hip_py.Vc2Hip(0, flags=vc2hip_py.FLAGS["SINGLE_PASS_VBR"])
    depth, ys, xs = 3, 9, 7
    (y, u, v), qidx, _ = _quantised_planes(oracle, 26, (ys * 8, xs * 16), (ys * 8, xs * 8), depth, ys, xs, "DD97", 8)
    for prefix, scalar in ((0, 1), (3, 2)):
        assert bytes(h2.hq_pack(y, u, v, depth, qidx, prefix, scalar)) == bytes(oracle.hq_pack(y, u, v, depth, qidx, prefix, scalar))
    h2.close()


def test_hq_pack_empty_and_dense_slices(hip, oracle):
    depth, ys, xs = 2, 3, 4
    rng = np.random.default_rng(22)
    y = np.zeros((ys * 8, xs * 8), np.int32)
    u = np.zeros((ys * 8, xs * 4), np.int32)
    v = np.zeros((ys * 8, xs * 4), np.int32)
    y[8:16, 8:16] = rng.integers(-65534, 65535, size=(8, 8))   # maximum-length codes
    v[0, 0] = -1
    u[16:24, 4:8] = rng.integers(-3, 4, size=(8, 4))
    qidx = np.zeros((ys, xs), np.int32)
    want = oracle.hq_pack(y, u, v, depth, qidx, 0, 2)
    got = hip.hq_pack(y, u, v, depth, qidx, 0, 2)
    assert bytes(got) == bytes(want)


def test_hq_pack_scalar_too_small_error(hip, oracle):
    from vc2hip_py import Vc2HipError
    depth, ys, xs = 2, 1, 1
    y = np.full((16, 16), 30000, np.int32)
    u = np.zeros((16, 8), np.int32)
    with pytest.raises(Vc2HipError, match="Slice scalar is too small"):
        hip.hq_pack(y, u, u, depth, np.zeros((1, 1), np.int32), 0, 1)


def test_hq_pack_cbr_matches_oracle(hip, oracle):
    depth, ys, xs = 3, 4, 4
    (y, u, v), qidx, _ = _quantised_planes(oracle, 23, (ys * 8, xs * 16), (ys * 8, xs * 8), depth, ys, xs, "DD97", 20, 200)
    sb = oracle.slice_bytes(ys, xs, 16 * 330, 2)
    want = oracle.hq_pack(y, u, v, depth, qidx, 1, 2, cbr=sb)
    got = hip.hq_pack(y, u, v, depth, qidx, 1, 2, cbr=sb)
    assert bytes(got) == bytes(want)


@pytest.mark.parametrize("prefix,scalar", [(0, 1), (3, 2)])
def test_hq_unpack_matches_oracle(hip, oracle, prefix, scalar):
    depth, ys, xs = 3, 6, 5
    lshape, cshape = (ys * 8, xs * 16), (ys * 8, xs * 8)
    (y, u, v), qidx, _ = _quantised_planes(oracle, 24, lshape, cshape, depth, ys, xs, "LeGall", 6)
    qidx[2, 3] = 17
    payload = oracle.hq_pack(y, u, v, depth, qidx, prefix, scalar)
    y2, u2, v2, q2, used = hip.hq_unpack(payload, lshape, cshape, depth, ys, xs, prefix, scalar)
    assert np.array_equal(y2, y) and np.array_equal(u2, u) and np.array_equal(v2, v)
    assert np.array_equal(q2, qidx) and used == payload.size


def test_hq_unpack_arbitrary_bytes_matches_oracle(hip, oracle):
    # third-party / corrupt streams: any byte string parses; bits past a bound read as 1
    depth, ys, xs = 2, 2, 3
    lshape, cshape = (ys * 4, xs * 8), (ys * 4, xs * 4)
    rng = np.random.default_rng(25)
    for trial in range(5):
        blob = rng.integers(0, 256, size=4096).astype(np.uint8)
        blob[rng.random(4096) < 0.3] = 0xFF       # plenty of zero-runs
        want = oracle.hq_unpack(blob, lshape, cshape, depth, ys, xs, 0, 1)
        got = hip.hq_unpack(blob, lshape, cshape, depth, ys, xs, 0, 1)
        for a, b in zip(got[:4], want[:4]):
            assert np.array_equal(a, b)


def test_cbr_qindices_match_oracle(hip, oracle):
    w, h, depth = 256, 128, 3
    raw = synth(w, h, "422", 10, 31)
    y, u, v = _planes(oracle, raw, w, h, "422", 10)
    k = KERNELS["DD97"]
    ty, tu, tv = (oracle.dwt_forward(p, k, depth) for p in (y, u, v))
    qm = oracle.quant_matrix(k, depth)
    ys, xs = 16, 8
    sb = oracle.slice_bytes(ys, xs, 24000, 1)
    want = oracle.cbr_qindices(ty, tu, tv, depth, qm, sb, 1)
    got = hip.cbr_qindices(ty, tu, tv, depth, qm, sb, 1)
    assert np.array_equal(got, want)
    assert len(set(want.flatten().tolist())) > 1


def _fmt_cp(hip, w, h, cf, bits, kernel, depth, u, a, **kw):
    import vc2hip_py
    word_bytes = kw.pop("word_bytes", 2)
    fmt = vc2hip_py.picture_format(w, h, cf, bits, word_bytes)
    cp = vc2hip_py.coding_params(hip.lib, fmt, kernel, depth, u, a, **kw)
    return fmt, cp


def _oracle_payload(oracle, p, raw):
    """slice payload + decoded picture from the oracle's whole-stream drivers"""
    stream = oracle.encode_stream(p, raw, 1)
    dec, n = oracle.decode_stream(p, stream, 1)
    assert n == 1
    return stream, dec


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_encode_picture_constq_all_kernels(hip, oracle, kernel):
    w, h, depth = 208, 120, 3        # pads 120 -> 120 (u=1 -> 15 slices), chroma 104 wide
    raw = noise_frame(w, h, "422", 10, seed=41)
    p = make_params(w, h, "422", 10, kernel, depth, 1, 2, q=9, scalar=3)
    stream, dec = _oracle_payload(oracle, p, raw)
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, kernel, depth, 1, 2, q=9, scalar=3)
    payload, qidx = hip.encode_picture_hq(raw, fmt, cp)
    assert payload == stream[-13 - len(payload):-13]      # slices sit right before end-of-sequence
    assert hip.decode_picture(payload, fmt, cp) == dec


@pytest.mark.parametrize("cf,bits,word_bytes", [("444", 12, 2), ("420", 8, 1), ("422", 16, 2), ("420", 10, 2)])
def test_encode_decode_formats_and_padding(hip, oracle, cf, bits, word_bytes):
    w, h, depth = 174, 126, 2        # pads to 176 x 128 (chroma 87 -> 88, 63 -> 64)
    raw = synth(w, h, cf, bits, 43, word_bytes=word_bytes)
    q = 5 if bits < 16 else 30   # 16-bit: keep |quantised| <= 65534 (the reference's VLC domain)
    p = make_params(w, h, cf, bits, "DD97", depth, 2, 2, q=q, scalar=4, word_bytes=word_bytes)
    stream, dec = _oracle_payload(oracle, p, raw)
    fmt, cp = _fmt_cp(hip, w, h, cf, bits, "DD97", depth, 2, 2, q=q, scalar=4, word_bytes=word_bytes)
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    assert payload == stream[-13 - len(payload):-13]
    assert hip.decode_picture(payload, fmt, cp) == dec


def test_encode_picture_cbr(hip, oracle):
    w, h, depth = 256, 128, 3
    raw = synth(w, h, "422", 10, 44)
    p = make_params(w, h, "422", 10, "DD97", depth, 1, 2, mode="HQ_CBR", s=24000, scalar=1)
    stream, dec = _oracle_payload(oracle, p, raw)
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "DD97", depth, 1, 2, mode="HQ_CBR", s=24000, scalar=1)
    payload, qidx = hip.encode_picture_hq(raw, fmt, cp)
    assert payload == stream[-13 - len(payload):-13]
    assert hip.decode_picture(payload, fmt, cp) == dec


def test_q0_lossless_roundtrip_on_gpu(hip):
    w, h = 320, 176
    raw = noise_frame(w, h, "422", 10, seed=45)
    for kernel in KERNELS:
        fmt, cp = _fmt_cp(hip, w, h, "422", 10, kernel, 4, 1, 2, q=0, scalar=8)
        payload, _ = hip.encode_picture_hq(raw, fmt, cp)
        assert hip.decode_picture(payload, fmt, cp) == raw, kernel


def test_ld_decode_matches_oracle(hip, oracle):
    # cfg 5 shape in miniature: LD stream made by the oracle, decoded by the GPU
    w, h, depth = 256, 120, 3
    raw = synth(w, h, "422", 8, 46, word_bytes=1)
    p = make_params(w, h, "422", 8, "LeGall", depth, 1, 2, mode="LD", s=12000, word_bytes=1)
    stream = oracle.encode_stream(p, raw, 1)
    dec, n = oracle.decode_stream(p, stream, 1)
    fmt, cp = _fmt_cp(hip, w, h, "422", 8, "LeGall", depth, 1, 2, mode="LD", s=12000, word_bytes=1)
    ns = cp.y_slices * cp.x_slices
    sb = oracle.slice_bytes(cp.y_slices, cp.x_slices, 12000, 1)
    payload = stream[-13 - int(sb.sum()):-13]
    assert hip.decode_picture(payload, fmt, cp) == dec
    # fine-grained LD unpack
    ph, pw = oracle.padded_size(h, depth), oracle.padded_size(w, depth)
    want = oracle.ld_unpack(np.frombuffer(payload, np.uint8), (ph, pw), (ph, pw // 2), depth, sb)
    got = hip.ld_unpack(np.frombuffer(payload, np.uint8), (ph, pw), (ph, pw // 2), depth, sb)
    for a, b in zip(got[:4], want[:4]):
        assert np.array_equal(a, b)


def test_cfg1_reference_digest_on_gpu(hip, oracle):
    """BASELINE config 1 at full size: GPU payload + oracle stream headers == the reference's stream."""
    g = GOLD["cfg1"]
    raw = synth(1920, 1080, "422", 10, 1234)
    fmt, cp = _fmt_cp(hip, 1920, 1080, "422", 10, "LeGall", 2, 2, 4, q=12, scalar=1)
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    p = make_params(1920, 1080, "422", 10, "LeGall", 2, 2, 4, q=12)
    stream = oracle.encode_stream(p, raw, 1)
    assert hashlib.sha256(stream).hexdigest() == g["stream"]["sha256"]
    assert payload == stream[-13 - len(payload):-13]
    dec = hip.decode_picture(payload, fmt, cp)
    assert hashlib.sha256(dec).hexdigest() == g["decoded"]["sha256"]


def test_cfg2_uhd_batch_device_resident(hip, oracle):
    """BASELINE config 2 (UHD-1 4:2:2 10-bit DD97 d4 q16 S2), 2 frames through the device-resident
    batch path; digests of reference output from SURVEY Appendix B."""
    import torch
    import vc2hip_py
    g = GOLD["cfg2"]
    raw = synth(3840, 2160, "422", 10, 1234, frames=2)
    fmt, cp = _fmt_cp(hip, 3840, 2160, "422", 10, "DD97", 4, 1, 2, q=16, scalar=2)
    n = 2
    rb = hip.raw_picture_bytes(fmt)
    stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    dev = torch.device("cuda:0")
    d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    d_pay = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(n, dtype=torch.int64, device=dev)
    d_out = torch.zeros(n * rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    hip.encode_batch_dev(d_raw.data_ptr(), n, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
    hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), n, fmt, cp, d_out.data_ptr())
    hip.sync()
    lens = d_len.cpu().tolist()
    pay = d_pay.cpu().numpy()
    # rebuild the reference stream: oracle headers around the GPU payloads
    p = make_params(3840, 2160, "422", 10, "DD97", 4, 1, 2, q=16, scalar=2)
    stream = oracle.encode_stream(p, raw, n)
    assert hashlib.sha256(stream).hexdigest() == g["stream"]["sha256"]
    pos = len(stream) - 13
    for k in reversed(range(n)):
        body = bytes(pay[k * stride:k * stride + lens[k]])
        assert stream[pos - lens[k]:pos] == body, f"frame {k}"
        pos -= lens[k]
        pos = stream.rfind(b"BBCD", 0, pos)
    assert hashlib.sha256(d_out.cpu().numpy().tobytes()).hexdigest() == g["decoded"]["sha256"]


@pytest.mark.parametrize("flags,forms", [("", (16, 8)), ("PLANES8_ALWAYS", (8, 8)), ("PLANES8_NEVER", (16, 16)), ("SINGLE_PASS_VBR", (16, 8))])
def test_cfg2_timed_path_at_full_size_in_every_plane_form(oracle, flags, forms):
    """The configuration bench.py times, pinned here: cfg 2 at full size through vc2hip_encode_batch_dev /
    vc2hip_decode_batch_dev, 4 pictures per call, TWICE on a context of its own -- with default flags the second call has
    taken the look at the first and decodes through BYTE band planes (the dequantiser table and the non-temporal stores ride
    on that instantiation), with the PLANES8 flags the form is fixed from the first call.  vc2hip_band_plane_bits says which
    form each call used, the library's own launch profile which kernels ran; payloads and decoded pictures against the
    reference's digests after BOTH calls."""
    import torch
    import vc2hip_py
    g = GOLD["cfg2"]
    hip = vc2hip_py.Vc2Hip(flags=sum(vc2hip_py.FLAGS[f] for f in flags.split(",") if f))
    raw2 = synth(3840, 2160, "422", 10, 1234, frames=2)
    n = 4
    raw = raw2 * 2                      # slots 2, 3 = slots 0, 1 again
    fmt, cp = _fmt_cp(hip, 3840, 2160, "422", 10, "DD97", 4, 1, 2, q=16, scalar=2)
    rb = hip.raw_picture_bytes(fmt)
    stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    dev = torch.device("cuda:0")
    d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    p = make_params(3840, 2160, "422", 10, "DD97", 4, 1, 2, q=16, scalar=2)
    stream = oracle.encode_stream(p, raw2, 2)
    assert hashlib.sha256(stream).hexdigest() == g["stream"]["sha256"]
    for call in range(2):
        d_pay = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
        d_len = torch.zeros(n, dtype=torch.int64, device=dev)
        d_out = torch.zeros(n * rb, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        hip.profile_reset(); hip.profile_enable(True)
        hip.encode_batch_dev(d_raw.data_ptr(), n, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
        hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), n, fmt, cp, d_out.data_ptr())
        hip.sync()
        hip.profile_enable(False)
        assert hip.band_plane_bits() == forms[call], (flags, call)
        seen = {k for k, v in hip.profile().items() if v[0] > 0}
        assert {"dwt_pair_first", "dwt_pair", "hq_pack", "slice_index_tables", "hq_unpack", "idwt_pair", "idwt_level",
                "idwt_level_final"} <= seen, seen
        # round 6: from 112 pictures per call on (bench.py: 128) the slice coder writes every slice where it belongs (look-back
        # over the workgroups' byte counts) -- no slots, no scan, no compaction; VC2HIP_FLAG_SINGLE_PASS_VBR: whatever the batch.
        # (tests/test_gpu_pack16.py::test_one_pass_coder_is_the_default_from_112_pictures_on runs the default's choice)
        assert ("slice_compact" in seen) == ("SINGLE_PASS_VBR" not in flags), seen
        lens = d_len.cpu().tolist()
        pay = d_pay.cpu().numpy()
        assert lens[2:] == lens[:2]
        for half in range(2):
            pos = len(stream) - 13
            for k in reversed(range(2)):
                slot = 2 * half + k
                body = bytes(pay[slot * stride:slot * stride + lens[slot]])
                assert stream[pos - lens[slot]:pos] == body, (flags, call, slot)
                pos -= lens[slot]
                pos = stream.rfind(b"BBCD", 0, pos)
            out = d_out[2 * half * rb:2 * (half + 1) * rb].cpu().numpy().tobytes()
            assert hashlib.sha256(out).hexdigest() == g["decoded"]["sha256"], (flags, call, half)
    hip.close()


@pytest.mark.parametrize("n", [3, 5])
def test_shared_slots_hold_whole_tiles(oracle, n):
    """HD-size slices (16 x 8 luma samples, LeGall depth 2): the slice coder puts sixteen slices of a workgroup into one shared
    slot, so a picture's slots are WHOLE tiles.  1920 x 72 has 1080 slices = 67.5 tiles: rounds 3 - 5 sized the slot buffer by
    the slice count, every picture's tiles started 8 slots further into it than the allocation assumed, and the last pictures
    of a batch ran off its end -- hidden by the allocator's rounding until 136 pictures of 1080p faulted in the compaction
    (round 6).  Noise at a low index makes every slice long, so the last tile's bytes reach far into its slots; a fresh context
    (its buffers are sized by this batch alone); every slot against the oracle."""
    import torch
    import vc2hip_py
    hip = vc2hip_py.Vc2Hip()
    w, h = 1920, 72
    raw = noise_frame(w, h, "422", 10, seed=4100 + n)
    p = make_params(w, h, "422", 10, "LeGall", 2, 2, 4, q=6, scalar=2)
    stream = oracle.encode_stream(p, raw, 1)
    dec, _ = oracle.decode_stream(p, stream, 1)
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "LeGall", 2, 2, 4, q=6, scalar=2)
    assert (cp.y_slices * cp.x_slices) % 16 == 8
    rb = hip.raw_picture_bytes(fmt)
    stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    dev = torch.device("cuda:0")
    d_raw = torch.frombuffer(bytearray(raw * n), dtype=torch.uint8).to(dev)
    d_pay = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(n, dtype=torch.int64, device=dev)
    d_out = torch.zeros(n * rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    hip.encode_batch_dev(d_raw.data_ptr(), n, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
    hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), n, fmt, cp, d_out.data_ptr())
    hip.sync()
    lens = d_len.cpu().tolist()
    pay = d_pay.cpu().numpy()
    out = d_out.cpu().numpy().tobytes()
    for k in range(n):
        body = bytes(pay[k * stride:k * stride + lens[k]])
        assert body == stream[-13 - len(body):-13], k
        assert out[k * rb:(k + 1) * rb] == dec, k
    hip.close()


def test_cfg3_uhd_cbr_reference_digests(hip, oracle):
    """BASELINE config 3 (UHD-1 HQ_CBR, -s 8294400 -S 2) at full size: per-slice quantiser search + CBR
    packing on the GPU; stream and decoded picture digests of reference output."""
    g = GOLD["cfg3"]
    raw = synth(3840, 2160, "422", 10, 1234, frames=1)
    fmt, cp = _fmt_cp(hip, 3840, 2160, "422", 10, "DD97", 4, 1, 2, mode="HQ_CBR", s=8294400, scalar=2)
    payload, qidx = hip.encode_picture_hq(raw, fmt, cp)
    assert set(np.unique(qidx).tolist()) <= {17, 18, 19}          # SURVEY section 8(d), cfg 3
    # reference stream = sequence header + picture header (oracle writer, pinned by the digest) + payload
    p = make_params(3840, 2160, "422", 10, "DD97", 4, 1, 2, mode="HQ_CBR", s=8294400, scalar=2)
    import ctypes as C
    from vc2lib import Params
    hdr = np.zeros(64, np.uint8); n = C.c_size_t(); major = C.c_int()
    oracle.lib.vc2o_write_sequence_header_payload(C.byref(p), hdr.ctypes.data_as(C.c_void_p), 64, C.byref(n), C.byref(major))
    seq = bytes(hdr[:n.value])
    ph = np.zeros(64, np.uint8); m = C.c_size_t()
    oracle.lib.vc2o_write_hq_picture_header(0, 0, 4, cp.x_slices, cp.y_slices, 0, 2, major.value, ph.ctypes.data_as(C.c_void_p), 64, C.byref(m))
    pich = bytes(ph[:m.value])

    def pi(code, nxt, prev):
        return b"BBCD" + bytes([code]) + nxt.to_bytes(4, "big") + prev.to_bytes(4, "big")
    n1 = 13 + len(seq)
    n2 = 13 + len(pich) + len(payload)
    stream = pi(0x00, n1, 0) + seq + pi(0xE8, n2, n1) + pich + payload + pi(0x10, 0, n2)
    assert len(stream) == g["stream"]["bytes"]
    assert hashlib.sha256(stream).hexdigest() == g["stream"]["sha256"]
    dec = hip.decode_picture(payload, fmt, cp)
    assert hashlib.sha256(dec).hexdigest() == g["decoded"]["sha256"]


def _hq_stream(oracle, p, cp, depth, scalar, payloads):
    """Reference stream syntax (oracle header writers, pinned by the cfg 1-4 digests) around GPU payloads."""
    import ctypes as C
    hdr = np.zeros(64, np.uint8); n = C.c_size_t(); major = C.c_int()
    oracle.lib.vc2o_write_sequence_header_payload(C.byref(p), hdr.ctypes.data_as(C.c_void_p), 64, C.byref(n), C.byref(major))
    seq = bytes(hdr[:n.value])

    def pi(code, nxt, prev):
        return b"BBCD" + bytes([code]) + nxt.to_bytes(4, "big") + prev.to_bytes(4, "big")
    out = [pi(0x00, 13 + len(seq), 0), seq]
    prev = 13 + len(seq)
    for k, payload in enumerate(payloads):
        ph = np.zeros(64, np.uint8); m = C.c_size_t()
        oracle.lib.vc2o_write_hq_picture_header(k, p.kernel, depth, cp.x_slices, cp.y_slices, 0, scalar, major.value,
                                                ph.ctypes.data_as(C.c_void_p), 64, C.byref(m))
        nxt = 13 + m.value + len(payload)
        out += [pi(0xE8, nxt, prev), bytes(ph[:m.value]), payload]
        prev = nxt
    out.append(pi(0x10, 0, prev))
    return b"".join(out)


def test_cfg4_uhd2_fidelity_reference_digests(hip, oracle):
    """BASELINE config 4 (UHD-2 7680x4320 4:4:4 12-bit, Fidelity depth 5, -u 1 -a 1 -q 40 -S 8) at full
    size on the GPU; input, stream and decoded digests are those of reference output.
    The oracle only writes the 3 headers here (its own cfg 4 run is the opt-in VC2_SLOW CPU test)."""
    g = GOLD["cfg4"]
    raw = synth(7680, 4320, "444", 12, 1234, frames=1)
    assert hashlib.sha256(raw).hexdigest() == g["input"]["sha256"]
    fmt, cp = _fmt_cp(hip, 7680, 4320, "444", 12, "Fidelity", 5, 1, 1, q=40, scalar=8)
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    p = make_params(7680, 4320, "444", 12, "Fidelity", 5, 1, 1, q=40, scalar=8)
    stream = _hq_stream(oracle, p, cp, 5, 8, [payload])
    assert len(stream) == g["stream"]["bytes"]
    assert hashlib.sha256(stream).hexdigest() == g["stream"]["sha256"]
    dec = hip.decode_picture(payload, fmt, cp)
    assert hashlib.sha256(dec).hexdigest() == g["decoded"]["sha256"]


def test_cfg5_ld_1080p_decode_matches_oracle(hip, oracle):
    """BASELINE config 5 at full size: 1920x1080 4:2:2 8-bit LD stream (LeGall depth 3, -u 1 -a 2,
    -s 1036800) made by the oracle, decoded on the GPU, equal to the oracle's decode byte for byte."""
    w, h, depth, s = 1920, 1080, 3, 1036800
    raw = synth(w, h, "422", 8, 1234, word_bytes=1)
    p = make_params(w, h, "422", 8, "LeGall", depth, 1, 2, mode="LD", s=s, word_bytes=1)
    stream = oracle.encode_stream(p, raw, 1)
    dec, n = oracle.decode_stream(p, stream, 1)
    assert n == 1
    fmt, cp = _fmt_cp(hip, w, h, "422", 8, "LeGall", depth, 1, 2, mode="LD", s=s, word_bytes=1)
    sb = oracle.slice_bytes(cp.y_slices, cp.x_slices, s, 1)
    assert int(sb.sum()) == s
    payload = stream[-13 - s:-13]
    assert hip.decode_picture(payload, fmt, cp) == dec
    # and the LD encoder at the same size
    got, qidx = hip.encode_picture_hq(raw, fmt, cp)
    assert got == payload


# ---- LD encode: quantIndicesLD, quantise_transform with DC prediction, LDSliceIO out
LD_CASES = [  # w, h, cf, bits, word_bytes, kernel, depth, u, a, bytes
    (256, 120, "422", 8, 1, "LeGall", 3, 1, 2, 12000),
    (128, 64, "444", 10, 2, "DD97", 2, 2, 2, 9000),
    (256, 128, "420", 8, 1, "Haar1", 3, 2, 2, 6000),
    (192, 96, "422", 12, 2, "Fidelity", 2, 4, 4, 5000),
]


@pytest.mark.parametrize("n", [3, 11])
def test_ld_batch_device_resident(hip, oracle, n):
    """LD batches whose size is no multiple of the 8 XCDs (the index search numbers its workgroups row * 8k + picture):
    every picture of the batch equals the oracle's stream payload, and the batch decodes to the oracle's pictures."""
    import torch
    w, h, depth, nbytes = 256, 120, 3, 12000
    raw = b"".join(synth(w, h, "422", 8, 100 + k, word_bytes=1) for k in range(n))
    p = make_params(w, h, "422", 8, "LeGall", depth, 1, 2, mode="LD", s=nbytes, word_bytes=1)
    fmt, cp = _fmt_cp(hip, w, h, "422", 8, "LeGall", depth, 1, 2, mode="LD", s=nbytes, word_bytes=1)
    rb = hip.raw_picture_bytes(fmt)
    assert rb * n == len(raw)
    stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    dev = torch.device("cuda:0")
    d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    d_pay = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(n, dtype=torch.int64, device=dev)
    d_out = torch.zeros(n * rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    hip.encode_batch_dev(d_raw.data_ptr(), n, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
    hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), n, fmt, cp, d_out.data_ptr())
    hip.sync()
    assert d_len.cpu().tolist() == [nbytes] * n
    pay = d_pay.cpu().numpy()
    out = d_out.cpu().numpy().tobytes()
    for k in range(n):
        one = raw[k * rb:(k + 1) * rb]
        stream = oracle.encode_stream(p, one, 1)
        assert bytes(pay[k * stride:k * stride + nbytes]) == stream[-13 - nbytes:-13], f"picture {k}"
        dec, _ = oracle.decode_stream(p, stream, 1)
        assert out[k * rb:(k + 1) * rb] == dec, f"picture {k}"


@pytest.mark.parametrize("case", LD_CASES, ids=lambda c: f"{c[5]}_{c[2]}_{c[0]}x{c[1]}")
def test_ld_encode_matches_oracle(hip, oracle, case):
    w, h, cf, bits, wb, kernel, depth, u, a, nbytes = case
    raw = synth(w, h, cf, bits, 47, word_bytes=wb)
    p = make_params(w, h, cf, bits, kernel, depth, u, a, mode="LD", s=nbytes, word_bytes=wb)
    stream = oracle.encode_stream(p, raw, 1)
    fmt, cp = _fmt_cp(hip, w, h, cf, bits, kernel, depth, u, a, mode="LD", s=nbytes, word_bytes=wb)
    payload, qidx = hip.encode_picture_hq(raw, fmt, cp)
    assert len(payload) == nbytes
    assert payload == stream[-13 - nbytes:-13]
    dec, _ = oracle.decode_stream(p, stream, 1)
    assert hip.decode_picture(payload, fmt, cp) == dec


def test_ld_fine_grained_entry_points(hip, oracle):
    """vc2hip_ld_qindices / vc2hip_quantise_ld / vc2hip_ld_pack against the oracle's restatements of
    quantIndicesLD (EncodeStream.cpp:141-245), quantise_transform (Quantisation.cpp:213-234, :358-367)
    and LDSliceIO (Slices.cpp:195-244)."""
    rng = np.random.default_rng(48)
    depth, ys, xs = 3, 6, 5
    lshape, cshape = (ys * 16, xs * 32), (ys * 16, xs * 16)
    planes = [np.rint(rng.normal(0, 300, s)).astype(np.int32) for s in (lshape, cshape, cshape)]
    tr = [oracle.dwt_forward(pl, KERNELS["LeGall"], depth) for pl in planes]
    qm = oracle.quant_matrix(KERNELS["LeGall"], depth)
    sb = oracle.slice_bytes(ys, xs, ys * xs * 90 + 7, 1)
    q_want = oracle.ld_qindices(tr[0], tr[1], tr[2], depth, qm, sb)
    q_got = hip.ld_qindices(tr[0], tr[1], tr[2], depth, qm, sb)
    assert np.array_equal(q_got, q_want)
    assert q_want.min() < q_want.max()
    qp_want = [oracle.quantise_ld(t, depth, q_want, qm) for t in tr]
    qp_got = [hip.quantise_ld(t, depth, q_want, qm) for t in tr]
    for a, b in zip(qp_got, qp_want):
        assert np.array_equal(a, b)
    want = oracle.ld_pack(qp_want[0], qp_want[1], qp_want[2], depth, q_want, sb)
    got = hip.ld_pack(qp_want[0], qp_want[1], qp_want[2], depth, q_want, sb)
    assert np.array_equal(got, want)


def test_ld_encode_errors(hip, oracle):
    # smallest budget the syntax allows (4 bytes per slice): every slice ends at a high index, no throw
    w, h = 128, 64
    raw = noise_frame(w, h, "422", 10, seed=49)
    p = make_params(w, h, "422", 10, "LeGall", 2, 1, 2, mode="LD", s=16 * 16 * 4, word_bytes=2)
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "LeGall", 2, 1, 2, mode="LD", s=16 * 16 * 4, word_bytes=2)
    payload, qidx = hip.encode_picture_hq(raw, fmt, cp)
    assert payload == oracle.encode_stream(p, raw, 1)[-13 - 1024:-13]
    assert qidx.min() > 40
    # LD pack with a luma slice that leaves no room for chroma
    depth, ys, xs = 2, 2, 2
    rng = np.random.default_rng(50)
    y = rng.integers(-200, 200, (16, 32)).astype(np.int32)
    u = rng.integers(-200, 200, (16, 16)).astype(np.int32)
    sb = np.full((ys, xs), 40, np.int32)
    q = np.zeros((ys, xs), np.int32)
    with pytest.raises(Exception, match="Too many bytes for the U and V slices"):
        hip.ld_pack(y, u, u, depth, q, sb)
    with pytest.raises(Exception):
        oracle.ld_pack(y, u, u, depth, q, sb)


def test_multi_stream_batches_are_identical(oracle):
    """vc2hip_set_streams: a batch cut over 3 streams / workspaces gives the same payloads, lengths and pictures
    (5 pictures -> sub-batches of 2, 2, 1), and device-side errors of a lane surface at sync."""
    import torch
    import vc2hip_py
    hip = vc2hip_py.Vc2Hip(0)
    w, h, n = 256, 128, 5
    raw = synth(w, h, "422", 10, 57, frames=n)
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "DD97", 3, 1, 2, q=11, scalar=2)
    rb = hip.raw_picture_bytes(fmt)
    stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    dev = torch.device("cuda:0")
    d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)

    def run():
        d_pay = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
        d_len = torch.zeros(n, dtype=torch.int64, device=dev)
        d_out = torch.zeros(n * rb, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        hip.profile_reset(); hip.profile_enable(True)
        hip.encode_batch_dev(d_raw.data_ptr(), n, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
        hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), n, fmt, cp, d_out.data_ptr())
        hip.sync()
        hip.profile_enable(False)
        return d_pay.cpu().numpy(), d_len.cpu().tolist(), d_out.cpu().numpy().tobytes(), hip.profile()

    pay1, len1, out1, prof1 = run()
    hip.set_streams(3)
    pay3, len3, out3, prof3 = run()
    assert len1 == len3 and out1 == out3
    for k in range(n):
        assert np.array_equal(pay1[k * stride:k * stride + len1[k]], pay3[k * stride:k * stride + len3[k]])
    assert prof1["hq_pack"][0] == 1 and prof3["hq_pack"][0] == 3          # one launch per lane
    p = make_params(w, h, "422", 10, "DD97", 3, 1, 2, q=11, scalar=2)
    assert out1 == oracle.decode_stream(p, oracle.encode_stream(p, raw, n), n)[0]
    # an error inside one lane's sub-batch is reported by the parent's sync
    bad = noise_frame(w, h, "422", 10, seed=58, full_scale=True)
    d_bad = torch.frombuffer(bytearray(raw[:4 * rb] + bad), dtype=torch.uint8).to(dev)
    fmt2, cp2 = _fmt_cp(hip, w, h, "422", 10, "DD97", 3, 1, 2, q=0, scalar=1)
    d_pay = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    hip.encode_batch_dev(d_bad.data_ptr(), n, fmt2, cp2, d_pay.data_ptr(), stride, d_len.data_ptr())
    with pytest.raises(Exception, match="Slice scalar is too small"):
        hip.sync()
    hip.set_streams(1)


def test_multi_stream_ld_batches(oracle):
    """LD through three streams: the lanes' row-walking index searches (each a single launch whose workgroups wait on
    one another) run side by side; payloads and pictures equal the oracle's."""
    import torch
    import vc2hip_py
    hip = vc2hip_py.Vc2Hip(0)
    hip.set_streams(3)
    w, h, depth, nbytes, n = 256, 120, 3, 12000, 7
    raw = b"".join(synth(w, h, "422", 8, 300 + k, word_bytes=1) for k in range(n))
    p = make_params(w, h, "422", 8, "LeGall", depth, 1, 2, mode="LD", s=nbytes, word_bytes=1)
    fmt, cp = _fmt_cp(hip, w, h, "422", 8, "LeGall", depth, 1, 2, mode="LD", s=nbytes, word_bytes=1)
    rb = hip.raw_picture_bytes(fmt)
    stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    dev = torch.device("cuda:0")
    d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    d_pay = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(n, dtype=torch.int64, device=dev)
    d_out = torch.zeros(n * rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):  # repeated: the flags of a previous batch must not leak into the next
        hip.encode_batch_dev(d_raw.data_ptr(), n, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
        hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), n, fmt, cp, d_out.data_ptr())
    hip.sync()
    pay = d_pay.cpu().numpy()
    out = d_out.cpu().numpy().tobytes()
    for k in range(n):
        stream = oracle.encode_stream(p, raw[k * rb:(k + 1) * rb], 1)
        assert bytes(pay[k * stride:k * stride + nbytes]) == stream[-13 - nbytes:-13], f"picture {k}"
        assert out[k * rb:(k + 1) * rb] == oracle.decode_stream(p, stream, 1)[0], f"picture {k}"
    hip.set_streams(1)


@pytest.mark.parametrize("scalar", [11, 30, 45])
def test_large_slice_scalars(hip, oracle, scalar):
    """Slices that may exceed 8191 bytes use 32 KiB index chunks, beyond 32767 bytes a serial walk: any slice size
    scalar the stream syntax allows is decoded and coded (the slice coder keeps four slice images in LDS up to scalar 40,
    fewer beyond, see test_gpu_wide.py for the largest)."""
    import ctypes as C
    w, h, depth, prefix = 256, 128, 3, 2
    raw = noise_frame(w, h, "422", 10, seed=59)
    p = make_params(w, h, "422", 10, "LeGall", depth, 2, 2, q=6, scalar=scalar, prefix=prefix)
    stream = oracle.encode_stream(p, raw, 1)
    dec, _ = oracle.decode_stream(p, stream, 1)
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "LeGall", depth, 2, 2, q=6, scalar=scalar, prefix=prefix)
    ph = np.zeros(64, np.uint8); m = C.c_size_t()
    oracle.lib.vc2o_write_hq_picture_header(0, p.kernel, depth, cp.x_slices, cp.y_slices, prefix, scalar, 2,
                                            ph.ctypes.data_as(C.c_void_p), 64, C.byref(m))
    second = stream.index(b"BBCD", 13)
    assert stream[second + 13:second + 13 + m.value] == bytes(ph[:m.value])
    payload = stream[second + 13 + m.value:-13]
    assert hip.decode_picture(payload, fmt, cp) == dec
    got, _ = hip.encode_picture_hq(raw, fmt, cp)   # beyond scalar 40: fewer slice images per workgroup (then in global memory)
    assert got == payload


@pytest.mark.parametrize("u,a,cf", [(1, 1, "444"), (1, 2, "422"), (2, 2, "422"), (2, 2, "420"), (2, 4, "444"), (4, 4, "422")])
@pytest.mark.parametrize("mode", ["HQ_ConstQ", "HQ_CBR"])
def test_pack_lane_widths(hip, oracle, u, a, cf, mode):
    """The slice coder gives a slice 16, 32 or 64 lanes depending on its size (4 / 2 / 1 slices per wavefront):
    slices of 64 ... 2048 luma coefficients, VBR and CBR, against the oracle's stream."""
    w, h, depth = 256, 128, 3
    raw = synth(w, h, cf, 10, 60)
    kw = dict(q=7, scalar=8, prefix=1) if mode == "HQ_ConstQ" else dict(mode="HQ_CBR", s=w * h // 2, scalar=8, prefix=1)
    p = make_params(w, h, cf, 10, "DD97", depth, u, a, **kw)
    stream = oracle.encode_stream(p, raw, 1)
    fmt, cp = _fmt_cp(hip, w, h, cf, 10, "DD97", depth, u, a, **kw)
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    assert stream.endswith(payload + stream[-13:])
    assert hip.decode_picture(payload, fmt, cp) == oracle.decode_stream(p, stream, 1)[0]


def test_random_geometries_against_oracle():
    """tools/fuzz_geometry.py: 80 random combinations of picture size (with padding), chroma format, bit depth, kernel,
    depth, slice size and mode; every encode payload and decoded picture equals the oracle's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_geometry.py"), "11", "80"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "80 cases, 0 bad" in out.stdout, out.stdout[-2000:]


# ---- round-3 additions: the top of the quantiser table, and the LD search's hand-over failure path
def test_quantiser_indices_116_to_119_fine_grained(hip, oracle):
    """Indices 116..119: the reference's quant_factor table overflows `int` there (Quantisation.cpp:42-58 through
    static_cast<int>): negative factors.  quantise / dequantise must wrap exactly like the oracle."""
    rng = np.random.default_rng(316)
    depth, ys, xs = 2, 2, 2
    coef = rng.integers(-(1 << 28), 1 << 28, size=(ys * 8, xs * 16)).astype(np.int32)
    coef[::2, ::3] = rng.integers(-2000, 2000, size=coef[::2, ::3].shape)
    qidx = np.array([[116, 117], [118, 119]], np.int32)
    qm = np.zeros(7, np.int32)
    want = oracle.quantise_np(coef, depth, qidx, qm)
    assert np.array_equal(hip.quantise_np(coef, depth, qidx, qm), want)
    small = rng.integers(-3, 4, size=coef.shape).astype(np.int32)
    assert np.array_equal(hip.dequantise_np(small, depth, qidx, qm), oracle.dequantise_np(small, depth, qidx, qm))


@pytest.mark.parametrize("kernel", ["Haar1", "DD97"])
@pytest.mark.parametrize("q", [116, 117, 118, 119])
def test_constq_indices_116_to_119_pictures(hip, oracle, kernel, q):
    """HQ_ConstQ -q 116..119 through the picture path (the float-reciprocal quantiser of k_hq_pack, the fused dequantiser
    of the inverse transform): payload and decoded picture against the oracle.  12-bit noise: coefficients large enough
    that some bands (matrix entry > 0: adjusted index below 116) still quantise to non-zero values."""
    w, h, depth = 256, 128, 3
    raw = noise_frame(w, h, "422", 12, seed=1160 + q)
    p = make_params(w, h, "422", 12, kernel, depth, 1, 2, q=q, scalar=2)
    stream, dec = _oracle_payload(oracle, p, raw)
    fmt, cp = _fmt_cp(hip, w, h, "422", 12, kernel, depth, 1, 2, q=q, scalar=2)
    payload, qidx = hip.encode_picture_hq(raw, fmt, cp)
    assert payload == stream[-13 - len(payload):-13]
    assert hip.decode_picture(payload, fmt, cp) == dec
    # and through the device-resident batch path (16-bit store, streaming / tile kernels)
    import torch
    dev = torch.device("cuda:0")
    rb = hip.raw_picture_bytes(fmt)
    stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    d_raw = torch.frombuffer(bytearray(raw + raw), dtype=torch.uint8).to(dev)
    d_pay = torch.zeros(2 * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(2, dtype=torch.int64, device=dev)
    d_out = torch.zeros(2 * rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    hip.encode_batch_dev(d_raw.data_ptr(), 2, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
    hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), 2, fmt, cp, d_out.data_ptr())
    hip.sync()
    assert d_len.cpu().tolist() == [len(payload)] * 2
    assert d_pay.cpu().numpy()[stride:stride + len(payload)].tobytes() == payload
    assert d_out.cpu().numpy()[rb:].tobytes() == dec


def test_ld_handoff_failure_path(tmp_path):
    """k_ld_search_rows hands reconstructed LL samples from one workgroup to another inside one launch (bounded wait, poison
    flag, VC2_DEVERR_HANDOFF, per-diagonal fallback from then on: DESIGN.md 'LD index search').  The failure path cannot
    be reached by a healthy GPU, so the -DVC2HIP_ABLATE library (never loaded by the product path) lets one row of
    slices of picture 0 never come: the batch must fail with the documented message, write nothing silently wrong, and
    the SAME batch submitted again must come out bit-exact through the fallback.  Own process: the fallback is a
    process-wide state."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, "..", "vc2-reference_amd", "libvc2hip_ablate.so")
    if not os.path.exists(lib):
        pytest.fail("libvc2hip_ablate.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    script = tmp_path / "handoff.py"
    script.write_text(f'''
import os, sys
sys.path.insert(0, {os.path.join(here, "..", "vc2-reference_amd")!r}); sys.path.insert(0, {here!r})
import vc2hip_py
from synth import synth
from vc2lib import load_oracle, make_params
w, h, depth, nbytes = 256, 120, 3, 12000
raw = synth(w, h, "422", 8, 77, word_bytes=1)
hip = vc2hip_py.Vc2Hip(0)
fmt = vc2hip_py.picture_format(w, h, "422", 8, 1)
cp = vc2hip_py.coding_params(hip.lib, fmt, "LeGall", depth, 1, 2, mode="LD", s=nbytes)
try:
    hip.encode_picture_hq(raw, fmt, cp)
    print("NOERROR")
except vc2hip_py.Vc2HipError as e:
    print("ERROR:", e)
os.environ.pop("VC2HIP_DEBUG_LD_DEAD_ROW")          # (read at every launch in the ablation build)
payload, _ = hip.encode_picture_hq(raw, fmt, cp)   # the same batch again: the per-diagonal launches
oracle = load_oracle()
p = make_params(w, h, "422", 8, "LeGall", depth, 1, 2, mode="LD", s=nbytes, word_bytes=1)
stream = oracle.encode_stream(p, raw, 1)
print("SECOND", "OK" if payload == stream[-13 - len(payload):-13] else "DIFFERS")
''')
    env = dict(os.environ, VC2HIP_LIB=os.path.abspath(lib), VC2HIP_DEBUG_LD_DEAD_ROW="3")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ERROR: LD index search: a hand-over between workgroups timed out" in out.stdout, out.stdout + out.stderr
    assert "submit the batch again" in out.stdout
    assert "SECOND OK" in out.stdout, out.stdout
