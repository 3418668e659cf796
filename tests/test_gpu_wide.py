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


def _ctx(*flags):
    """a context with the given vc2hip_create_with_flags switches (round 5: the release library reads no environment)"""
    from vc2hip_py import FLAGS, Vc2Hip
    return Vc2Hip(flags=sum(FLAGS[f] for f in flags))


@pytest.fixture(scope="module")
def variants():
    """default = everything on (two-level kernels, streaming kernels, 16-bit store, band planes, record heads); the others
    each take ONE of them away, so that every alternative path runs the same cases"""
    return {"default": _ctx(), "store32": _ctx("STORE32"), "tiles": _ctx("NO_STREAM"), "records": _ctx("NO_BANDPLANES"),
            "levels": _ctx("NO_PAIR"), "bytes": _ctx("PLANES8_ALWAYS"), "words": _ctx("PLANES8_NEVER"), "onepass": _ctx("SINGLE_PASS_VBR")}


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


@pytest.mark.parametrize("q", [116, 118, 119])
def test_byte_planes_high_and_mixed_indices(variants, oracle, q):
    """the byte planes' dequantiser table at the top of the quantiser table.  With q = 116 ... 119 and matrix entries
    of 0 the adjusted indices reach 116 ... 119, where quant_factor wraps in 32 bits (Quantisation.cpp:40-58 as the reference
    computes it); 16-bit noise keeps some quantised values non-zero, so the table's entries for those factors are read.
    Every variant (the `bytes` one forces the byte planes) against the oracle; slices with DIFFERENT indices side by side on
    the byte planes are test_wide_cbr's `bytes` variant."""
    w, h, depth = 1024, 64, 2
    raw = noise_frame(w, h, "422", 16, seed=300 + q)
    _check(variants, oracle, raw, w, h, "422", 16, "LeGall", depth, 2, 4, q=q, scalar=8)


@pytest.mark.parametrize("cf", ["422", "444"])
def test_deep_level_shapes_with_escapes(variants, oracle, cf):
    """The deep levels of the UHD slice geometry (32 x 16 slices, depth 4: blocks of 2 x 1 and 1 x 1 coefficients with LL
    at the deepest level, 4 x 2 and 2 x 2 above it -- the compile-time shapes of the inverse tile kernel's slice
    gather) on 16-bit noise: LL and the coarse bands pass 32767 after quantisation, so the record heads hold escapes
    exactly where that gather reads them; the lowest index the reference's code words allow."""
    w, h, depth, u, a = 2048, 256, 4, 1, 2
    raw = noise_frame(w, h, cf, 16, seed=91)
    hip = variants["default"]
    hit = None
    for q in (40, 36, 32, 28, 24, 20, 16, 12):
        fmt, cp = _fmt_cp(hip, w, h, cf, 16, "DD97", depth, u, a, q=q, scalar=8)
        try:
            payload, _ = hip.encode_picture_hq(raw, fmt, cp)
        except Exception:   # |quantised| > 65534: outside the reference's domain
            break
        cw = w if cf == "444" else w // 2
        y, cu, cv, _, _ = oracle.hq_unpack(np.frombuffer(payload, np.uint8), (h, w), (h, cw), depth, cp.y_slices, cp.x_slices, 0, 8)
        mx = max(int(np.abs(x).max()) for x in (y, cu, cv))
        if 32767 < mx <= 65534:
            hit = q
    assert hit is not None, "no quantiser index puts quantised values between 32768 and 65534"
    _check(variants, oracle, raw, w, h, cf, 16, "DD97", depth, u, a, q=hit, scalar=8)


def test_pipelined_picture_calls(variants, oracle):
    """vc2hip_encode_picture_begin / _end and the decode pair (two pictures in flight, pinned staging): the bytes of the
    synchronous calls, in order; a third _begin before an _end is refused"""
    hip = variants["default"]
    w, h, depth, n = 1024, 64, 3, 5
    raw = synth(w, h, "422", 10, 75, frames=n)
    rb = len(raw) // n
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "DD97", depth, 1, 2, q=5, scalar=2)
    raws = [raw[k * rb:(k + 1) * rb] for k in range(n)]
    pays = hip.encode_pictures_pipelined(raws, fmt, cp)
    assert pays == [hip.encode_picture_hq(r, fmt, cp)[0] for r in raws]
    assert hip.decode_pictures_pipelined(pays, fmt, cp) == [hip.decode_picture(p, fmt, cp) for p in pays]


@pytest.mark.parametrize("scalar", [60, 120, 200])
def test_long_slices_large_scalar(variants, oracle, scalar):
    """slice size scalars beyond what four LDS slice images hold (> 40): two, one wavefront per workgroup, then the
    images in global memory; VBR and CBR.  The reference takes any scalar (Slices.cpp:97-119)."""
    hip = variants["default"]
    w, h, depth = 256, 64, 2
    raw = noise_frame(w, h, "422", 10, seed=81)
    p = make_params(w, h, "422", 10, "LeGall", depth, 2, 4, q=0, scalar=scalar)
    stream = oracle.encode_stream(p, raw, 1)
    dec, _ = oracle.decode_stream(p, stream, 1)
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "LeGall", depth, 2, 4, q=0, scalar=scalar)
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    assert payload == stream[-13 - len(payload):-13]
    assert hip.decode_picture(payload, fmt, cp) == dec == raw
    raw = synth(w, h, "422", 10, 83)
    s = cp.y_slices * cp.x_slices * (4 + 3 * scalar * 2)   # CBR: two units per component and slice
    p = make_params(w, h, "422", 10, "LeGall", depth, 2, 4, mode="HQ_CBR", s=s, scalar=scalar)
    stream = oracle.encode_stream(p, raw, 1)
    dec, _ = oracle.decode_stream(p, stream, 1)
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "LeGall", depth, 2, 4, mode="HQ_CBR", s=s, scalar=scalar)
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    assert payload == stream[-13 - len(payload):-13]
    assert hip.decode_picture(payload, fmt, cp) == dec


@pytest.mark.parametrize("kernel,mode", [("DD97", "HQ_ConstQ"), ("Fidelity", "HQ_ConstQ"), ("LeGall", "HQ_CBR")])
def test_one_slice_per_picture(variants, oracle, kernel, mode):
    """the largest slice the reference admits: the whole picture (WaveletTransform.cpp:116-136).  No LDS tile holds it:
    the transform runs on whole planes in HBM, the CBR search reads the slice from the store in every trial, the slice
    coder keeps its image in global memory (components of ~50 KB need a scalar beyond 200)."""
    hip = variants["default"]
    w, h, depth = 512, 256, 2
    raw = synth(w, h, "444", 10, 82)
    u, a = h >> depth, w >> depth          # one slice (4:4:4: with subsampled chroma the reference wants two across)
    kw = dict(q=20, scalar=1000) if mode == "HQ_ConstQ" else dict(mode="HQ_CBR", s=300000, scalar=1000)
    p = make_params(w, h, "444", 10, kernel, depth, u, a, **kw)
    stream = oracle.encode_stream(p, raw, 1)
    dec, _ = oracle.decode_stream(p, stream, 1)
    fmt, cp = _fmt_cp(hip, w, h, "444", 10, kernel, depth, u, a, **kw)
    assert cp.y_slices == 1 and cp.x_slices == 1
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    assert payload == stream[-13 - len(payload):-13]
    assert hip.decode_picture(payload, fmt, cp) == dec


@pytest.mark.parametrize("amp,budget,scalar", [(40, 30000, 1), (600, 9000, 1), (600, 60000, 3), (30000, 3000, 1), (30000, 200000, 2),
                                               (200000, 40000, 1), (5, 1400, 1)])
def test_cbr_search_register_kernel_and_hand_back(variants, oracle, amp, budget, scalar):
    """HQ_CBR quantiser indices over coefficient planes of every magnitude and budgets from starved to generous: the
    register search inside its domain, and the slices it hands to the general kernel (coefficients beyond 16 bits, trial
    indices above 79, length-byte overflow) -- all against the oracle's search (EncodeStream.cpp:73-125)."""
    depth, ys, xs = 3, 8, 16
    rng = np.random.default_rng(amp + budget)
    ph, pw = 64, 512
    def plane(h, w):
        p = (rng.laplace(0, amp / 3.0, size=(h, w))).astype(np.int64)
        p[rng.random((h, w)) < 0.5] = 0
        return np.clip(p, -amp * 4, amp * 4).astype(np.int32)
    ty, tu, tv = plane(ph, pw), plane(ph, pw // 2), plane(ph, pw // 2)
    qm = oracle.quant_matrix(KERNELS["DD97"], depth)
    sb = oracle.slice_bytes(ys, xs, budget, scalar)
    try:
        want = oracle.cbr_qindices(ty, tu, tv, depth, qm, sb, scalar)
    except Exception:               # the search leaves the quantiser table: an error on both sides
        for name, hip in variants.items():
            with pytest.raises(Exception):
                hip.cbr_qindices(ty, tu, tv, depth, qm, sb, scalar)
        return
    for name, hip in variants.items():
        try:
            got = hip.cbr_qindices(ty, tu, tv, depth, qm, sb, scalar)
        except Exception as e:      # a device error flag (index / length byte out of range): the general kernel raises the same
            general = _ctx("CBR_GENERAL")
            with pytest.raises(type(e)):
                general.cbr_qindices(ty, tu, tv, depth, qm, sb, scalar)
            general.close()
            continue
        assert np.array_equal(got, want), name


def test_random_wide_geometries_against_oracle():
    """tools/fuzz_geometry.py in its wide mode: 120 random combinations with planes of 512 ... 2560 samples across, so
    that streaming and tile transform levels, band planes and slice records, 16-bit and escaped coefficients mix."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_geometry.py"), "31", "120", "wide"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "120 cases, 0 bad" in out.stdout, out.stdout[-2000:]


def test_cbr_decode_offsets_from_budgets_and_fallback(variants, oracle):
    """Decoding with HQ_CBR coding parameters claims the slice offsets from the byte budgets and verifies them against the
    length bytes; a stream that does not fill the budgets (here: a constant-quantiser stream, and a CBR stream of another
    budget) falls back to the general slice index and decodes to the same picture."""
    hip = variants["default"]
    w, h, depth = 1024, 64, 3
    raw = synth(w, h, "422", 10, 75)
    # a conforming CBR stream
    p = make_params(w, h, "422", 10, "DD97", depth, 1, 2, mode="HQ_CBR", s=30000, scalar=1)
    stream = oracle.encode_stream(p, raw, 1)
    dec, _ = oracle.decode_stream(p, stream, 1)
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "DD97", depth, 1, 2, mode="HQ_CBR", s=30000, scalar=1)
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    assert payload == stream[-13 - len(payload):-13]
    assert hip.decode_picture(payload, fmt, cp) == dec
    # the same payload under CBR parameters of another budget: the claim fails, the index finds the slices
    _, cp_other = _fmt_cp(hip, w, h, "422", 10, "DD97", depth, 1, 2, mode="HQ_CBR", s=31000, scalar=1)
    assert hip.decode_picture(payload, fmt, cp_other) == dec
    # a constant-quantiser payload under CBR parameters
    pq = make_params(w, h, "422", 10, "DD97", depth, 1, 2, q=9, scalar=1)
    sq = oracle.encode_stream(pq, raw, 1)
    dq, _ = oracle.decode_stream(pq, sq, 1)
    _, cpq = _fmt_cp(hip, w, h, "422", 10, "DD97", depth, 1, 2, q=9, scalar=1)
    payq, _ = hip.encode_picture_hq(raw, fmt, cpq)
    assert hip.decode_picture(payq, fmt, cp) == dq
    assert hip.decode_picture(payq, fmt, cpq) == dq


def test_hostile_lengths_stay_inside_their_slot(variants, oracle):
    """Device-resident decode with a per-picture length beyond the payload stride (hostile or uninitialised): the kernels
    clamp it to the slot, the call reports a stream error, the neighbouring pictures of the batch are untouched by it."""
    import torch
    hip = variants["default"]
    w, h, depth, n = 1024, 64, 3, 3
    raw = b"".join(synth(w, h, "422", 10, 400 + k) for k in range(n))
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "DD97", depth, 1, 2, q=8, scalar=1)
    rb = hip.raw_picture_bytes(fmt)
    stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    dev = torch.device("cuda:0")
    d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    d_pay = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(n, dtype=torch.int64, device=dev)
    d_out = torch.zeros(n * rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()   # (torch fills on its stream, the library works on its own)
    hip.encode_batch_dev(d_raw.data_ptr(), n, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
    hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), n, fmt, cp, d_out.data_ptr())
    hip.sync()
    good = d_out.cpu().numpy().tobytes()
    lens = d_len.clone()
    for bad_len in (stride + 1, 1 << 40):
        d_len2 = lens.clone()
        d_len2[1] = bad_len
        d_out.zero_()
        torch.cuda.synchronize()
        hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len2.data_ptr(), n, fmt, cp, d_out.data_ptr())
        with pytest.raises(Exception):
            hip.sync()
        out = d_out.cpu().numpy().tobytes()
        assert out[:rb] == good[:rb] and out[2 * rb:] == good[2 * rb:]
    # the context is usable afterwards
    hip.decode_batch_dev(d_pay.data_ptr(), stride, lens.data_ptr(), n, fmt, cp, d_out.data_ptr())
    hip.sync()
    assert d_out.cpu().numpy().tobytes() == good


def test_payload_slot_beyond_the_index_chain_table(variants, oracle):
    """A payload slot of more than 256 MiB has more chunk groups than the per-picture chain table of the slice index holds
    (1024 groups of 16 chunks of 16 KiB): the index then walks the length bytes serially -- same offsets, same picture."""
    import torch
    hip = variants["default"]
    w, h, depth = 1024, 64, 3
    raw = synth(w, h, "422", 10, 77)
    p = make_params(w, h, "422", 10, "LeGall", depth, 1, 2, q=6, scalar=1)
    stream = oracle.encode_stream(p, raw, 1)
    dec, _ = oracle.decode_stream(p, stream, 1)
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "LeGall", depth, 1, 2, q=6, scalar=1)
    rb = hip.raw_picture_bytes(fmt)
    stride = 300 * 1024 * 1024
    dev = torch.device("cuda:0")
    d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    d_pay = torch.zeros(stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(1, dtype=torch.int64, device=dev)
    d_out = torch.zeros(rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()   # (torch fills on its stream, the library works on its own)
    hip.encode_batch_dev(d_raw.data_ptr(), 1, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
    hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), 1, fmt, cp, d_out.data_ptr())
    hip.sync()
    n = int(d_len.cpu()[0])
    assert d_pay[:n].cpu().numpy().tobytes() == stream[-13 - n:-13]
    assert d_out.cpu().numpy().tobytes() == dec


@pytest.mark.parametrize("kernel", ["DD97", "LeGall", "DD137", "Haar1", "Daub97", "Fidelity"])
def test_streaming_levels_with_odd_pair_counts_and_narrow_planes(variants, oracle, kernel):
    """4:4:4, 768 x 88, depth 3, slices 8 x 64: the three levels are 768 x 88 (44 row pairs), 384 x 44 (22: remainder 2 of
    the ring of four) and 192 x 22 (11: remainder 3) -- the tail instantiations of the streaming kernels, and planes
    narrower than a wavefront of 8-sample chunks (48 and 24 of 64 lanes at work).  Fidelity (rings of eight, whole blocks
    only) takes the tile kernels below the first level."""
    w, h, depth = 768, 88, 3
    raw = noise_frame(w, h, "444", 10, seed=91)
    _check(variants, oracle, raw, w, h, "444", 10, kernel, depth, 1, 8, q=9, scalar=8)
    raw = synth(w, h, "444", 10, 92)
    _check(variants, oracle, raw, w, h, "444", 10, kernel, depth, 1, 8, q=0, scalar=8)


def test_ld_decode_of_slices_beyond_any_lds_tile(hip, oracle):
    """LD pictures whose slices do not fit an LDS tile of the level kernels (the reference's only limit is sliceSizeIsValid,
    WaveletTransform.cpp:116-136): the decoder takes the whole-plane transform that HQ pictures of such slices take, with the
    DC-predicted LL reconstruction put into the plane's LL band.  512 x 512 4:4:4, four slices of 256 x 256."""
    from test_gpu_parity import _fmt_cp
    w, h, depth, nbytes = 512, 512, 2, 90000
    raw = synth(w, h, "444", 8, 321, word_bytes=1)
    p = make_params(w, h, "444", 8, "LeGall", depth, 64, 64, mode="LD", s=nbytes, word_bytes=1)
    stream = oracle.encode_stream(p, raw, 1)
    dec, n = oracle.decode_stream(p, stream, 1)
    assert n == 1
    fmt, cp = _fmt_cp(hip, w, h, "444", 8, "LeGall", depth, 64, 64, mode="LD", s=nbytes, word_bytes=1)
    assert cp.y_slices == 2 and cp.x_slices == 2
    payload = stream[-13 - nbytes:-13]
    assert hip.decode_picture(payload, fmt, cp) == dec


def test_ld_encode_of_slices_beyond_lds(hip, oracle):
    """... and the LD encoder for such slices: the index search and the DC-predicted quantiser read the coefficients from
    the store (k_ld_quantise_diag<GLOBAL>), the transform runs on whole planes.  Payload against the oracle's stream."""
    from test_gpu_parity import _fmt_cp
    w, h, depth, nbytes = 512, 512, 2, 90000
    raw = synth(w, h, "444", 8, 322, word_bytes=1)
    p = make_params(w, h, "444", 8, "LeGall", depth, 64, 64, mode="LD", s=nbytes, word_bytes=1)
    stream = oracle.encode_stream(p, raw, 1)
    fmt, cp = _fmt_cp(hip, w, h, "444", 8, "LeGall", depth, 64, 64, mode="LD", s=nbytes, word_bytes=1)
    got, qidx = hip.encode_picture_hq(raw, fmt, cp)
    assert got == stream[-13 - nbytes:-13]
    # a slice size between the LDS tile of the level kernels and the LD coder's LDS budget: 128 x 256 (98 K coefficients)
    w, h, nbytes = 512, 256, 50000
    raw = synth(w, h, "444", 10, 323)
    p = make_params(w, h, "444", 10, "DD97", 2, 32, 64, mode="LD", s=nbytes)
    stream = oracle.encode_stream(p, raw, 1)
    fmt, cp = _fmt_cp(hip, w, h, "444", 10, "DD97", 2, 32, 64, mode="LD", s=nbytes)
    got, _ = hip.encode_picture_hq(raw, fmt, cp)
    assert got == stream[-13 - nbytes:-13]
    assert hip.decode_picture(got, fmt, cp) == oracle.decode_stream(p, stream, 1)[0]


@pytest.mark.parametrize("variant", ["default", "tiles"])
def test_separate_luma_and_chroma_bit_depths(variants, oracle, variant):
    """EncodeStream -l 10 -c 8 (pictureio::bitDepth(lumaDepth, chromaDepth), EncodeStream.cpp:322): the chroma words carry their
    own depth on the encoder's input.  Expected payload: the oracle's fine-grained functions plane by plane (ingest with
    the component's depth, transform, quantise, HQ slice coding); the picture is wide enough for the streaming kernels."""
    import vc2hip_py
    hip = variants[variant]
    w, h, depth, kernel, q, scalar = 1024, 64, 2, "DD97", 16, 2
    rng = np.random.default_rng(77)
    ywords = (rng.integers(0, 1024, size=(h, w)).astype(np.uint16) << 6).astype(">u2").tobytes()
    cwords = [(rng.integers(0, 256, size=(h, w // 2)).astype(np.uint16) << 8).astype(">u2").tobytes() for _ in range(2)]
    raw = ywords + cwords[0] + cwords[1]
    fmt = vc2hip_py.picture_format(w, h, "422", 10, 2, chroma_bits=8)
    cp = vc2hip_py.coding_params(hip.lib, fmt, kernel, depth, 2, 4, q=q, scalar=scalar)
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    k = KERNELS[kernel]
    y = oracle.dwt_forward(oracle.ingest(ywords, 2, 10, (h, w)), k, depth)
    u = oracle.dwt_forward(oracle.ingest(cwords[0], 2, 8, (h, w // 2)), k, depth)
    v = oracle.dwt_forward(oracle.ingest(cwords[1], 2, 8, (h, w // 2)), k, depth)
    qm = oracle.quant_matrix(k, depth)
    qidx = np.full((cp.y_slices, cp.x_slices), q, np.int32)
    want = oracle.hq_pack(oracle.quantise_np(y, depth, qidx, qm), oracle.quantise_np(u, depth, qidx, qm), oracle.quantise_np(v, depth, qidx, qm),
                          depth, qidx, 0, scalar)
    assert payload == bytes(want)


@pytest.mark.parametrize("h,cf", [(24, "420"), (200, "422"), (400, "422"), (512, "444"), (600, "422")])
def test_ld_decode_dc_prediction_by_one_wavefront(hip, oracle, h, cf):
    """The DC-predicted LL band (Quantisation.cpp:191-234) of planes of up to 256 rows is reconstructed by one wavefront
    whose lanes own 1, 2, 3 or 4 consecutive rows (ld_ll_wave<R>); taller planes take the anti-diagonal sweep.  Depth 1:
    LL planes of 12 / 6, 100, 200, 256 and 300 rows (a partial last lane, every R, the limit, the fallback); noise, so that
    the residuals are large and of both signs."""
    from test_gpu_parity import _fmt_cp
    w, depth = 96, 1
    nbytes = w * h * 3 // 4
    raw = noise_frame(w, h, cf, 8, 1000 + h, word_bytes=1)
    p = make_params(w, h, cf, 8, "Haar0", depth, 2, 4, mode="LD", s=nbytes, word_bytes=1)
    stream = oracle.encode_stream(p, raw, 1)
    dec, n = oracle.decode_stream(p, stream, 1)
    assert n == 1
    fmt, cp = _fmt_cp(hip, w, h, cf, 8, "Haar0", depth, 2, 4, mode="LD", s=nbytes, word_bytes=1)
    payload = stream[-13 - nbytes:-13]
    assert hip.decode_picture(payload, fmt, cp) == dec


@pytest.mark.parametrize("w,h,n_slices", [(512, 256, 32768), (640, 256, 40960), (1024, 512, 131072)])
def test_many_small_slices(variants, oracle, w, h, n_slices):
    """Slices of 2 x 2 samples (depth 1, -u 1 -a 1, 4:4:4): 32768 slices (the size scan's two coalesced variants end here),
    40960 and 131072 (its any-size variant; the slice index with hundreds of groups per picture)."""
    raw = synth(w, h, "444", 8, 9000 + w, word_bytes=1)
    hip = variants["default"]
    fmt, cp = _fmt_cp(hip, w, h, "444", 8, "Haar0", 1, 1, 1, q=3, word_bytes=1)
    assert cp.y_slices * cp.x_slices == n_slices
    _check({"default": hip}, oracle, raw, w, h, "444", 8, "Haar0", 1, 1, 1, q=3, word_bytes=1)


def test_cbr_search_trial_with_a_code_beyond_32_bits(hip, oracle):
    """A picture found by tools/fuzz_geometry.py (seed 1101, wide): 16-bit noise, slices of 2 x 2 samples, HQ_CBR.  With ~28
    bytes per slice every slice ends at an index of 12 or more and no quantised value exceeds 22214 -- but the trials of
    quantIndicesCBR (EncodeStream.cpp:73-125) on the way there (15, then 7) quantise LL coefficients of up to 94456 with
    factor 1, beyond 65534, whose code has more than 32 bits.  The reference only MEASURES it there
    (SignedVLC::numOfBits in luma_slice_bits, Slices.cpp:51-70): the trial does not fit and the search goes up again.  The
    GPU's search counted such a code as one bit (and raised the 32-bit-domain error): too small an index, a failure in the
    slice coder.  (With the fuzz case's own 37 bytes per slice the reference's final indices code values beyond 65534:
    outside its domain, refused here with the documented error.)"""
    from test_gpu_parity import _fmt_cp
    import vc2hip_py
    w, h = 768, 4
    raw = open(os.path.join(os.path.dirname(__file__), "golden", "cbr_oversize_trial_768x4_444_16bit.raw"), "rb").read()
    kw = dict(mode="HQ_CBR", s=26000, scalar=2, prefix=0)
    p = make_params(w, h, "444", 16, "Haar1", 1, 1, 1, word_bytes=2, **kw)
    stream = oracle.encode_stream(p, raw, 1)
    dec, n = oracle.decode_stream(p, stream, 1)
    assert n == 1
    fmt, cp = _fmt_cp(hip, w, h, "444", 16, "Haar1", 1, 1, 1, **kw)   # (the register search hands such slices to the general kernel)
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    assert stream[:-13].endswith(payload)
    assert hip.decode_picture(payload, fmt, cp) == dec
    kw["s"] = 31418
    fmt, cp = _fmt_cp(hip, w, h, "444", 16, "Haar1", 1, 1, 1, **kw)
    with pytest.raises(vc2hip_py.Vc2HipError, match="exceeds 65534"):
        hip.encode_picture_hq(raw, fmt, cp)


@pytest.mark.parametrize("what", ["luma length beyond the slice", "huge quantiser index in the top row", "both"])
def test_ld_corrupt_slice_headers_like_the_reference(hip, oracle, what):
    """Corrupt LD slice headers (tools/fuzz_decode.py).  A luma length field beyond its slice: the reference reads that many
    bits from the stream (LDSliceIO, Slices.cpp:246-303), so every later slice starts late -- flag, serial walk, second pass
    in the decoder.  A quantiser index near 127: dequantised LL samples around 2^31, sums of three wrap, and the top row's
    prediction (the left neighbour) must not go through the three-term formula."""
    from test_gpu_parity import _fmt_cp
    w, h, depth, nbytes = 256, 16, 1, 4000
    raw = synth(w, h, "422", 10, 4242)
    kw = dict(mode="LD", s=nbytes)
    p = make_params(w, h, "422", 10, "Haar1", depth, 2, 4, **kw)
    stream = oracle.encode_stream(p, raw, 1)
    fmt, cp = _fmt_cp(hip, w, h, "422", 10, "Haar1", depth, 2, 4, **kw)
    sb = oracle.slice_bytes(cp.y_slices, cp.x_slices, nbytes, 1).ravel()
    off = np.concatenate([[0], np.cumsum(sb)])
    pay = bytearray(stream[-13 - nbytes:-13])
    if what != "huge quantiser index in the top row":
        for sl in (5, 40):   # luma length 255: beyond the 8 * 31 - 7 - 8 bits a slice has
            pay[off[sl]] |= 1
            pay[off[sl] + 1] = 0xFF
    if what != "luma length beyond the slice":
        for sl in (9, 17, 18, 70):
            pay[off[sl]] = (115 << 1) | (pay[off[sl]] & 1)
    pay = bytes(pay)
    head, tail = stream[:-13 - nbytes], stream[-13:]
    want, n = oracle.decode_stream(p, head + pay + tail, 1)
    assert n == 1
    assert hip.decode_picture(pay + tail, fmt, cp) == want   # (the reference's reader sees what follows the data unit too)


def test_two_level_kernels_are_the_path_taken(variants, oracle):
    """Round 5: a UHD-like geometry (planes from 256 samples wide at level 2, 32 x 16 slices, DD97 depth 4, 4:2:2) must go through
    k_fwd_pair / k_inv_pair -- seen in the library's own launch profile, so that the parity tests of this geometry are tests
    of those kernels -- and through the one-level kernels in the `levels` variant; both bit-exact with the oracle."""
    w, h = 2048, 256   # (chroma planes of level 3: 128 samples wide, the narrowest that keeps the 16-bit store, which the pairs need)
    raw = synth(w, h, "422", 10, 77)
    p = make_params(w, h, "422", 10, "DD97", 4, 1, 2, q=16, scalar=2)
    stream = oracle.encode_stream(p, raw, 1)
    dec, _ = oracle.decode_stream(p, stream, 1)
    seen = {}
    for name in ("default", "levels"):
        hip = variants[name]
        fmt, cp = _fmt_cp(hip, w, h, "422", 10, "DD97", 4, 1, 2, q=16, scalar=2)
        hip.profile_reset(); hip.profile_enable(True)
        payload, _ = hip.encode_picture_hq(raw, fmt, cp)
        out = hip.decode_picture(payload, fmt, cp)
        hip.sync()
        hip.profile_enable(False)
        seen[name] = {k for k, v in hip.profile().items() if v[0] > 0}
        assert payload == stream[-13 - len(payload):-13] and out == dec, name
    assert {"dwt_pair_first", "dwt_pair", "idwt_pair"} <= seen["default"], seen["default"]
    assert "dwt_level_first" not in seen["default"]
    assert not any("pair" in k for k in seen["levels"]), seen["levels"]
    assert {"dwt_level_first", "dwt_level", "idwt_level", "idwt_level_final"} <= seen["levels"]


def test_band_plane_form_follows_the_batches_and_never_changes_a_result(oracle):
    """Round 5: the decoder chooses byte or 16-bit band planes from the batch before (payload bits per sample, escape count).
    A fresh context decodes coarse pictures (-> bytes from its second call: the first look is waited for), then fine noise
    (many bits per sample -> back to 16-bit planes), then coarse ones again; every picture must be the oracle's, whatever
    form its call happened to use (the forced forms are the `bytes` / `words` variants of every other case of this file)."""
    hip = _ctx()
    w, h = 2048, 128
    cases = []
    for seed, q, gen in ((5, 24, synth), (6, 24, synth), (7, 24, synth), (8, 0, noise_frame), (9, 0, noise_frame), (10, 0, noise_frame),
                         (11, 20, synth), (12, 20, synth), (13, 20, synth)):
        raw = gen(w, h, "422", 10, seed)
        p = make_params(w, h, "422", 10, "DD97", 3, 1, 2, q=q, scalar=2)
        stream = oracle.encode_stream(p, raw, 1)
        dec, _ = oracle.decode_stream(p, stream, 1)
        fmt, cp = _fmt_cp(hip, w, h, "422", 10, "DD97", 3, 1, 2, q=q, scalar=2)
        payload, _ = hip.encode_picture_hq(raw, fmt, cp)
        assert payload == stream[-13 - len(payload):-13], (seed, "payload")
        cases.append((seed, payload, fmt, cp, dec))
    for seed, payload, fmt, cp, dec in cases:
        assert hip.decode_picture(payload, fmt, cp) == dec, (seed, "decode")
    # and without a synchronisation between the calls (the look at the batch before is then usually not there yet)
    pays = [c[1] for c in cases[:3]] * 3
    assert hip.decode_pictures_pipelined(pays, cases[0][2], cases[0][3]) == [c[4] for c in cases[:3]] * 3
