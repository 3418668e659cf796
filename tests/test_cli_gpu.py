"""GPU tests of the drop-in tools: EncodeStream / DecodeStream (vc2-reference_amd/bin) against whole
streams and decoded files of the oracle, and against the reference's own digests at full size."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from synth import noise_frame, synth
from vc2lib import KERNELS, make_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vc2-reference_amd", "bin")
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_digests.json")))


@pytest.fixture(scope="module")
def tools():
    if not os.path.exists(os.path.join(BIN, "EncodeStream")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "vc2-reference_amd", "host")], stdout=subprocess.DEVNULL)
    return BIN


def run(tool, *args):
    out = subprocess.run([os.path.join(BIN, tool)] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    return out


def enc_args(w, h, cf, bits, kernel, depth, u, a, mode="HQ_ConstQ", q=None, s=None, scalar=1, prefix=0, n=2):
    args = ["-m", mode, "-k", kernel, "-d", depth, "-u", u, "-a", a, "-f", {"444": "4:4:4", "422": "4:2:2", "420": "4:2:0"}[cf],
            "-x", w, "-y", h, "-l", bits, "-n", n]
    if mode != "LD":
        args += ["-S", scalar, "-P", prefix]
    if q is not None:
        args += ["-q", q]
    if s is not None:
        args += ["-s", s]
    return args


@pytest.mark.parametrize("kernel", ["DD97", "Fidelity", "Haar0"])
def test_stream_roundtrip_matches_oracle(tools, oracle, tmp_path, kernel):
    w, h, frames = 208, 120, 3
    raw = synth(w, h, "422", 10, 51, frames=frames)
    p = make_params(w, h, "422", 10, kernel, 3, 1, 2, q=9, scalar=3, prefix=1)
    want_stream = oracle.encode_stream(p, raw, frames)
    want_dec, n = oracle.decode_stream(p, want_stream, frames)
    (tmp_path / "in.raw").write_bytes(raw)
    run("EncodeStream", *enc_args(w, h, "422", 10, kernel, 3, 1, 2, q=9, scalar=3, prefix=1), tmp_path / "in.raw", tmp_path / "out.vc2")
    assert (tmp_path / "out.vc2").read_bytes() == want_stream
    run("DecodeStream", tmp_path / "out.vc2", tmp_path / "dec.raw")
    assert (tmp_path / "dec.raw").read_bytes() == want_dec


def test_cbr_stream_and_8bit(tools, oracle, tmp_path):
    w, h = 256, 128
    raw = synth(w, h, "420", 8, 52, frames=2, word_bytes=1)
    p = make_params(w, h, "420", 8, "LeGall", 3, 2, 2, mode="HQ_CBR", s=20000, scalar=2, word_bytes=1)
    want = oracle.encode_stream(p, raw, 2)
    (tmp_path / "in.raw").write_bytes(raw)
    run("EncodeStream", *enc_args(w, h, "420", 8, "LeGall", 3, 2, 2, mode="HQ_CBR", s=20000, scalar=2, n=1), tmp_path / "in.raw", tmp_path / "o.vc2")
    assert (tmp_path / "o.vc2").read_bytes() == want
    run("DecodeStream", tmp_path / "o.vc2", tmp_path / "d.raw")
    assert (tmp_path / "d.raw").read_bytes() == oracle.decode_stream(p, want, 2)[0]


def _planes_be4(data, shapes):
    out, pos = [], 0
    for s in shapes:
        n = s[0] * s[1]
        out.append(np.frombuffer(data[pos:pos + 4 * n], ">i4").astype(np.int32).reshape(s))
        pos += 4 * n
    assert pos == len(data)
    return out


def test_diagnostic_outputs_match_oracle(tools, oracle, tmp_path):
    # -o Transform / Quantised / Indices / Packaged (EncodeParams.cpp:251-296), DecodeStream -o Quantised / Transform / Indices
    w, h, depth, kernel = 128, 64, 3, "DD97"
    raw = synth(w, h, "422", 10, 53)
    (tmp_path / "in.raw").write_bytes(raw)
    k = KERNELS[kernel]
    y = oracle.ingest(raw[:w * h * 2], 2, 10, (h, w))
    u = oracle.ingest(raw[w * h * 2:w * h * 3], 2, 10, (h, w // 2))
    v = oracle.ingest(raw[w * h * 3:], 2, 10, (h, w // 2))
    t = [oracle.dwt_forward(pl, k, depth) for pl in (y, u, v)]
    base = enc_args(w, h, "422", 10, kernel, depth, 1, 2, mode="HQ_CBR", s=6000)
    run("EncodeStream", *base, "-o", "Transform", tmp_path / "in.raw", tmp_path / "t.bin")
    got = _planes_be4((tmp_path / "t.bin").read_bytes(), [a.shape for a in t])
    assert all(np.array_equal(a, b) for a, b in zip(got, t))
    qm = oracle.quant_matrix(k, depth)
    sb = oracle.slice_bytes(8, 8, 6000, 1)
    qi = oracle.cbr_qindices(t[0], t[1], t[2], depth, qm, sb, 1)
    run("EncodeStream", *base, "-o", "Indices", tmp_path / "in.raw", tmp_path / "i.bin")
    assert np.array_equal(np.frombuffer((tmp_path / "i.bin").read_bytes(), np.uint8).reshape(8, 8), qi)
    q = [oracle.quantise_np(a, depth, qi, qm) for a in t]
    run("EncodeStream", *base, "-o", "Quantised", tmp_path / "in.raw", tmp_path / "q.bin")
    got = _planes_be4((tmp_path / "q.bin").read_bytes(), [a.shape for a in q])
    assert all(np.array_equal(a, b) for a, b in zip(got, q))
    run("EncodeStream", *base, "-o", "Packaged", tmp_path / "in.raw", tmp_path / "p.bin")
    assert (tmp_path / "p.bin").read_bytes() == bytes(oracle.hq_pack(q[0], q[1], q[2], depth, qi, 0, 1, cbr=sb))
    run("EncodeStream", *base, tmp_path / "in.raw", tmp_path / "s.vc2")
    run("DecodeStream", "-o", "Quantised", tmp_path / "s.vc2", tmp_path / "dq.bin")
    got = _planes_be4((tmp_path / "dq.bin").read_bytes(), [a.shape for a in q])
    assert all(np.array_equal(a, b) for a, b in zip(got, q))
    run("DecodeStream", "-o", "Indices", tmp_path / "s.vc2", tmp_path / "di.bin")
    assert np.array_equal(np.frombuffer((tmp_path / "di.bin").read_bytes(), np.uint8).reshape(8, 8), qi)
    run("DecodeStream", "-o", "Transform", tmp_path / "s.vc2", tmp_path / "dt.bin")
    want = [oracle.dequantise_np(a, depth, qi, qm) for a in q]
    got = _planes_be4((tmp_path / "dt.bin").read_bytes(), [a.shape for a in want])
    assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_decodestream_ld_stream(tools, oracle, tmp_path):
    w, h = 256, 120
    raw = synth(w, h, "422", 8, 54, frames=2, word_bytes=1)
    p = make_params(w, h, "422", 8, "LeGall", 3, 1, 2, mode="LD", s=12000, word_bytes=1)
    stream = oracle.encode_stream(p, raw, 2)
    (tmp_path / "ld.vc2").write_bytes(stream)
    run("DecodeStream", tmp_path / "ld.vc2", tmp_path / "ld.raw")
    assert (tmp_path / "ld.raw").read_bytes() == oracle.decode_stream(p, stream, 2)[0]


def test_error_reporting_like_the_reference(tools, tmp_path):
    # "Error: <what()>" on standard output, exit status 1 (EncodeStream.cpp:782-785)
    w, h = 64, 64
    (tmp_path / "in.raw").write_bytes(noise_frame(w, h, "444", 12, 5, full_scale=True))
    out = subprocess.run([os.path.join(BIN, "EncodeStream")] + [str(a) for a in enc_args(w, h, "444", 12, "Fidelity", 2, 4, 4, q=0, scalar=1)]
                         + [str(tmp_path / "in.raw"), str(tmp_path / "o.vc2")], capture_output=True, text=True)
    assert out.returncode == 1
    assert "Error: Slice scalar is too small, consider using a larger slice scalar." in out.stdout


def test_cfg1_full_size_against_reference_digests(tools, tmp_path):
    """BASELINE config 1 through the tools only: byte-identical to the reference's stream and decode."""
    g = GOLD["cfg1"]
    (tmp_path / "in.raw").write_bytes(synth(1920, 1080, "422", 10, 1234))
    run("EncodeStream", *enc_args(1920, 1080, "422", 10, "LeGall", 2, 2, 4, q=12), tmp_path / "in.raw", tmp_path / "o.vc2")
    s = (tmp_path / "o.vc2").read_bytes()
    assert len(s) == g["stream"]["bytes"] and hashlib.sha256(s).hexdigest() == g["stream"]["sha256"]
    run("DecodeStream", tmp_path / "o.vc2", tmp_path / "d.raw")
    assert hashlib.sha256((tmp_path / "d.raw").read_bytes()).hexdigest() == g["decoded"]["sha256"]


# ---- SURVEY 8(f)3/4 through the tools: LD encode, interlace, picture fragments
def _roundtrip(tools, oracle, tmp_path, w, h, cf, bits, kernel, depth, u, a, frames, extra, n=2, **kw):
    raw = synth(w, h, cf, bits, 55, frames=frames, word_bytes=n)
    p = make_params(w, h, cf, bits, kernel, depth, u, a, word_bytes=n, **kw)
    want = oracle.encode_stream(p, raw, frames)
    (tmp_path / "in.raw").write_bytes(raw)
    cli = {k: v for k, v in kw.items() if k in ("mode", "q", "s", "scalar", "prefix")}
    run("EncodeStream", *enc_args(w, h, cf, bits, kernel, depth, u, a, n=n, **cli), *extra, tmp_path / "in.raw", tmp_path / "o.vc2")
    got = (tmp_path / "o.vc2").read_bytes()
    assert got == want
    run("DecodeStream", tmp_path / "o.vc2", tmp_path / "d.raw")
    dec = (tmp_path / "d.raw").read_bytes()
    assert dec == oracle.decode_stream(p, want, frames)[0]
    return raw, dec


def test_ld_encode_stream(tools, oracle, tmp_path):
    _roundtrip(tools, oracle, tmp_path, 256, 120, "422", 8, "LeGall", 3, 1, 2, 2, [], n=1, mode="LD", s=12000)


@pytest.mark.parametrize("bff", [False, True])
def test_interlaced_streams(tools, oracle, tmp_path, bff):
    flags = ["-i", "-b"] if bff else ["-i"]
    raw, dec = _roundtrip(tools, oracle, tmp_path, 128, 96, "420", 8, "Haar0", 2, 2, 2, 3, flags, n=1, q=0, scalar=4,
                          interlaced=True, bottom_field_first=bff)
    assert dec == raw          # index 0 + integer lifting: lossless, so the fields went back to their rows
    _roundtrip(tools, oracle, tmp_path, 128, 96, "422", 10, "DD97", 2, 1, 2, 2, flags, mode="HQ_CBR", s=9000, scalar=1,
               interlaced=True, bottom_field_first=bff)


@pytest.mark.parametrize("mode,kw", [("HQ_CBR", dict(s=9000, scalar=2, prefix=1)), ("LD", dict(s=7000))])
@pytest.mark.parametrize("flen", [1, 700, 60000])
def test_fragmented_streams(tools, oracle, tmp_path, mode, kw, flen):
    _roundtrip(tools, oracle, tmp_path, 128, 64, "422", 10, "LeGall", 2, 2, 2, 2, ["-F", flen], mode=mode, fragment_length=flen, **kw)


def test_interlaced_fragmented_cbr(tools, oracle, tmp_path):
    _roundtrip(tools, oracle, tmp_path, 128, 64, "444", 12, "Fidelity", 2, 1, 1, 2, ["-i", "-F", 400], mode="HQ_CBR", s=16000,
               scalar=2, interlaced=True, fragment_length=400)


def test_ld_diagnostic_outputs(tools, oracle, tmp_path):
    # LD mode through -o Indices / Quantised / Packaged: quantIndicesLD, quantise_transform (DC-predicted), LDSliceIO
    w, h, depth, kernel, s = 128, 64, 2, "LeGall", 5000
    raw = synth(w, h, "422", 8, 56, word_bytes=1)
    (tmp_path / "in.raw").write_bytes(raw)
    k = KERNELS[kernel]
    y = oracle.ingest(raw[:w * h], 1, 8, (h, w))
    u = oracle.ingest(raw[w * h:w * h * 3 // 2], 1, 8, (h, w // 2))
    v = oracle.ingest(raw[w * h * 3 // 2:], 1, 8, (h, w // 2))
    t = [oracle.dwt_forward(pl, k, depth) for pl in (y, u, v)]
    qm = oracle.quant_matrix(k, depth)
    sb = oracle.slice_bytes(8, 16, s, 1)
    qi = oracle.ld_qindices(t[0], t[1], t[2], depth, qm, sb)
    base = enc_args(w, h, "422", 8, kernel, depth, 2, 2, mode="LD", s=s, n=1)
    run("EncodeStream", *base, "-o", "Indices", tmp_path / "in.raw", tmp_path / "i.bin")
    assert np.array_equal(np.frombuffer((tmp_path / "i.bin").read_bytes(), np.uint8).reshape(8, 16), qi)
    q = [oracle.quantise_ld(a, depth, qi, qm) for a in t]
    run("EncodeStream", *base, "-o", "Quantised", tmp_path / "in.raw", tmp_path / "q.bin")
    got = _planes_be4((tmp_path / "q.bin").read_bytes(), [a.shape for a in q])
    assert all(np.array_equal(a, b) for a, b in zip(got, q))
    run("EncodeStream", *base, "-o", "Packaged", tmp_path / "in.raw", tmp_path / "p.bin")
    assert (tmp_path / "p.bin").read_bytes() == bytes(oracle.ld_pack(q[0], q[1], q[2], depth, qi, sb))


def test_decodestream_resynchronises_and_skips_other_units(tools, oracle, tmp_path):
    """Bytes before the first parse-info prefix are skipped (dataunitio::synchronise, DataUnit.cpp:1086-1109); auxiliary
    and padding data units between the pictures are stepped over (DecodeStream.cpp:281-290)."""
    w, h = 128, 64
    raw = synth(w, h, "422", 10, 57, frames=2)
    p = make_params(w, h, "422", 10, "LeGall", 2, 2, 2, q=9, scalar=2)
    stream = oracle.encode_stream(p, raw, 2)
    want = oracle.decode_stream(p, stream, 2)[0]
    # cut the stream into its data units and splice an auxiliary (0x20) and a padding (0x30) unit between the pictures
    pos, units = 0, []
    while pos < len(stream):
        nxt = int.from_bytes(stream[pos + 5:pos + 9], "big") or 13
        units.append(bytearray(stream[pos:pos + nxt]))
        pos += nxt
    aux = bytearray(b"BBCD\x20" + (13 + 5).to_bytes(4, "big") + bytes(4) + b"hello")
    pad = bytearray(b"BBCD\x30" + (13 + 32).to_bytes(4, "big") + bytes(4) + bytes(32))
    units = units[:2] + [aux, pad] + units[2:]
    prev = 0
    for u in units:  # re-chain the previous-parse-offset fields
        u[9:13] = prev.to_bytes(4, "big")
        prev = int.from_bytes(u[5:9], "big")
    spliced = b"\x00garbage BBC" + b"".join(bytes(u) for u in units)
    (tmp_path / "s.vc2").write_bytes(spliced)
    run("DecodeStream", tmp_path / "s.vc2", tmp_path / "d.raw")
    assert (tmp_path / "d.raw").read_bytes() == want


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_several_workers_keep_the_stream_order(tools, oracle, tmp_path, devices):
    """N > 1 per-GPU workers (here: several workers on the one GPU of the box, --devices 0,0,..): picture k goes to
    worker k mod N, two pictures in flight per worker; parse offsets and picture numbers chain through the ordered
    writer, so the stream and the decoded file must be byte for byte those of one worker and of the oracle."""
    w, h, frames = 256, 128, 11
    raw = synth(w, h, "422", 10, 61, frames=frames)
    p = make_params(w, h, "422", 10, "DD97", 3, 1, 2, q=7, scalar=2)
    want_stream = oracle.encode_stream(p, raw, frames)
    want_dec, n = oracle.decode_stream(p, want_stream, frames)
    (tmp_path / "in.raw").write_bytes(raw)
    args = enc_args(w, h, "422", 10, "DD97", 3, 1, 2, q=7, scalar=2)
    run("EncodeStream", *args, tmp_path / "in.raw", tmp_path / "one.vc2")
    run("EncodeStream", *args, "--devices", devices, tmp_path / "in.raw", tmp_path / "many.vc2")
    assert (tmp_path / "one.vc2").read_bytes() == want_stream
    assert (tmp_path / "many.vc2").read_bytes() == want_stream
    run("DecodeStream", "--devices", devices, tmp_path / "many.vc2", tmp_path / "dec.raw")
    assert (tmp_path / "dec.raw").read_bytes() == want_dec


def test_more_pictures_than_the_output_mapping_was_sized_for(tools, oracle, tmp_path):
    """DecodeStream sizes its mapped output from a pre-scan that stops at a zero next-parse-offset, while
    the main loop parses on behind a sequence header that carries one.  [picture][sequence header, next = 0][picture]
    [picture]: the pre-scan counts one picture, the loop skips the first (no sequence header yet) and decodes two -- the
    second has no place in the mapping and must go out through pwrite, not past the end of the map."""
    w, h, frames = 256, 128, 2
    raw = synth(w, h, "422", 10, 71, frames=frames)
    p = make_params(w, h, "422", 10, "DD97", 3, 1, 2, q=7, scalar=2)
    stream = oracle.encode_stream(p, raw, frames)
    want, n = oracle.decode_stream(p, stream, frames)
    units, pos = [], 0
    while pos < len(stream):
        nxt = int.from_bytes(stream[pos + 5:pos + 9], "big") or 13
        units.append(bytearray(stream[pos:pos + nxt]))
        pos += nxt
    assert len(units) == 4 and units[0][4] == 0x00 and units[1][4] == 0xE8   # sequence header, 2 pictures, end of sequence
    seq = bytearray(units[0])
    seq[5:9] = (0).to_bytes(4, "big")   # next parse offset 0: "the rest follows"
    crafted = bytes(units[1]) + bytes(seq) + bytes(units[1]) + bytes(units[2]) + bytes(units[3])
    (tmp_path / "c.vc2").write_bytes(crafted)
    out = subprocess.run([os.path.join(BIN, "DecodeStream"), str(tmp_path / "c.vc2"), str(tmp_path / "d.raw")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert (tmp_path / "d.raw").read_bytes() == want


def _gpu_count():
    import torch
    return torch.cuda.device_count()


def test_workers_on_distinct_devices(tools, oracle, tmp_path):
    """--devices 0,1: one worker, one library context and one pinned staging pool PER DEVICE (vc2hip_host_alloc on the
    worker's own device); the stream and the decoded file must be those of one device and of the oracle."""
    if _gpu_count() < 2:   # (asked inside the test: nothing touches torch or the GPU when the module is collected)
        pytest.skip("needs two GPUs (the gpurun boxes have one): runs on the driver's multi-GPU node")
    w, h, frames = 512, 256, 9
    raw = synth(w, h, "422", 10, 67, frames=frames)
    p = make_params(w, h, "422", 10, "DD97", 4, 1, 2, q=16, scalar=2)
    want_stream = oracle.encode_stream(p, raw, frames)
    want_dec, n = oracle.decode_stream(p, want_stream, frames)
    (tmp_path / "in.raw").write_bytes(raw)
    args = enc_args(w, h, "422", 10, "DD97", 4, 1, 2, q=16, scalar=2)
    devices = ",".join(str(d) for d in range(min(_gpu_count(), 8)))
    run("EncodeStream", *args, "--devices", devices, tmp_path / "in.raw", tmp_path / "many.vc2")
    assert (tmp_path / "many.vc2").read_bytes() == want_stream
    run("DecodeStream", "--devices", devices, tmp_path / "many.vc2", tmp_path / "dec.raw")
    assert (tmp_path / "dec.raw").read_bytes() == want_dec
    run("DecodeStream", "--devices", "1", tmp_path / "many.vc2", tmp_path / "dec1.raw")   # a context on a device other than 0
    assert (tmp_path / "dec1.raw").read_bytes() == want_dec


def test_several_workers_interlaced_cbr(tools, oracle, tmp_path):
    w, h, frames = 256, 128, 5
    raw = synth(w, h, "422", 10, 62, frames=frames)
    p = make_params(w, h, "422", 10, "LeGall", 2, 2, 4, mode="HQ_CBR", s=20000, scalar=1, interlaced=True)
    want_stream = oracle.encode_stream(p, raw, frames)
    want_dec, n = oracle.decode_stream(p, want_stream, frames)
    (tmp_path / "in.raw").write_bytes(raw)
    run("EncodeStream", *enc_args(w, h, "422", 10, "LeGall", 2, 2, 4, mode="HQ_CBR", s=20000, scalar=1), "-i", "--devices", "0,0",
        tmp_path / "in.raw", tmp_path / "out.vc2")
    assert (tmp_path / "out.vc2").read_bytes() == want_stream
    run("DecodeStream", "--devices", "0,0,0", tmp_path / "out.vc2", tmp_path / "dec.raw")
    assert (tmp_path / "dec.raw").read_bytes() == want_dec


def test_slice_surface_encode_body(tools):
    """the encoder / decoder picture body written with the reference's Slices / sliceio / split_into_blocks vocabulary
    (vc2-reference_amd/host/slicetest.cpp after EncodeStream.cpp:482-647, DecodeStream.cpp:451-613)"""
    out = run("slicetest")
    assert "slicetest ok" in out.stdout


def test_separate_luma_and_chroma_depths_through_the_tool(tools, oracle, tmp_path):
    # EncodeStream -l 10 -c 8 (EncodeStream.cpp:266-267, 322): the chroma planes of the input carry their own bit depth; the
    # sequence header still signals the luma depth (EncodeStream.cpp:447), so the stream's payload is compared, not a round trip.
    w, h, depth, kernel, q = 128, 64, 2, "LeGall", 20
    rng = np.random.default_rng(91)
    ywords = (rng.integers(0, 1024, size=(h, w)).astype(np.uint16) << 6).astype(">u2").tobytes()
    cwords = [(rng.integers(0, 256, size=(h, w // 2)).astype(np.uint16) << 8).astype(">u2").tobytes() for _ in range(2)]
    (tmp_path / "in.raw").write_bytes(ywords + cwords[0] + cwords[1])
    k = KERNELS[kernel]
    t = [oracle.dwt_forward(oracle.ingest(ywords, 2, 10, (h, w)), k, depth)] + \
        [oracle.dwt_forward(oracle.ingest(c, 2, 8, (h, w // 2)), k, depth) for c in cwords]
    base = enc_args(w, h, "422", 10, kernel, depth, 2, 4, q=q, scalar=2) + ["-c", 8]
    run("EncodeStream", *base, "-o", "Transform", tmp_path / "in.raw", tmp_path / "t.bin")
    got = _planes_be4((tmp_path / "t.bin").read_bytes(), [a.shape for a in t])
    assert all(np.array_equal(a, b) for a, b in zip(got, t))
    qm = oracle.quant_matrix(k, depth)
    qi = np.full((h // (2 * 4), w // (4 * 4)), q, np.int32)     # -u 2 -a 4 at depth 2: slices 8 high, 16 wide
    qs = [oracle.quantise_np(a, depth, qi, qm) for a in t]
    want = bytes(oracle.hq_pack(qs[0], qs[1], qs[2], depth, qi, 0, 2))
    run("EncodeStream", *base, tmp_path / "in.raw", tmp_path / "s.vc2")
    stream = (tmp_path / "s.vc2").read_bytes()
    assert stream[-13 - len(want):-13] == want


def test_slices_that_run_past_their_data_unit(tools, oracle, tmp_path):
    # A corrupt stream: the last length byte of a picture's only slice claims two bytes more than the data unit holds.  The
    # reference parses slices straight from the input stream (DecodeStream.cpp:513), so it reads on into the next parse info
    # and then resynchronises; the tool does the same (the decoder sees one slice's worth of bytes behind the unit).
    w, h, frames = 16, 8, 2
    raw = synth(w, h, "444", 8, 77, frames=frames, word_bytes=1)
    p = make_params(w, h, "444", 8, "Haar0", 1, 4, 8, q=0, word_bytes=1)
    stream = bytearray(oracle.encode_stream(p, raw, frames))
    # first picture: parse info (13) + sequence header, then parse info + picture header + payload [q][ly][Y..][lu][U..][lv][V..]
    at = 13
    while stream[at:at + 4] != b"BBCD" or stream[at + 4] != 0xE8:
        at += 1
    nxt = int.from_bytes(stream[at + 5:at + 9], "big")
    pay_end = at + nxt
    # find the payload start: the slice fills the unit's tail exactly, so walk back from the candidates
    for start in range(at + 13, pay_end):
        ly = stream[start + 1]
        lu_at = start + 2 + ly
        if lu_at >= pay_end: continue
        lv_at = lu_at + 1 + stream[lu_at]
        if lv_at < pay_end and lv_at + 1 + stream[lv_at] == pay_end and stream[start] == 0:
            break
    else:
        raise AssertionError("payload not found")
    stream[lv_at] += 2
    stream = bytes(stream)
    want, n = oracle.decode_stream(p, stream, frames)
    assert n == frames
    (tmp_path / "in.vc2").write_bytes(stream)
    run("DecodeStream", tmp_path / "in.vc2", tmp_path / "dec.raw")
    assert (tmp_path / "dec.raw").read_bytes() == want
