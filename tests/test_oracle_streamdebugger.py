"""Pins the oracle's HQ stream syntax on the reference's own stream parser: tests/golden/streamdebugger.json
holds what /root/reference/tools/vc2streamdebugger reported for five oracle streams (whole pictures v2 and v3,
HQ fragments, interlaced field pictures, a base-video-format match).  Here the oracle rebuilds the streams
(digest must match the record) and every fact the reference tool read -- header fields, data-unit chain,
fragment headers, per-slice quantiser index and component lengths -- is checked against the coding
parameters and against the oracle's own reading of the slices."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

from vc2lib import KERNELS

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "streamdebugger.json")))
_spec = importlib.util.spec_from_file_location("mkfix", os.path.join(HERE, "golden", "make_streamdebugger_fixtures.py"))
mkfix = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mkfix)


@pytest.mark.parametrize("name", sorted(mkfix.CASES))
def test_reference_parser_reads_the_oracle_stream_as_coded(oracle, name):
    case, rec = mkfix.CASES[name], GOLD[name]
    kw = case["kw"]
    stream = mkfix.build(case)
    assert len(stream) == rec["stream_bytes"]
    assert hashlib.sha256(stream).hexdigest() == rec["stream_sha256"]
    units = rec["units"]
    assert not any(u.get("problems") for u in units)

    interlaced = kw.get("interlaced", False)
    pics = case["frames"] * (2 if interlaced else 1)
    fragmented = kw.get("fragment_length", 0) > 0
    # data-unit chain: offsets are where the units are, prev = the previous next
    pos, prev = 0, 0
    for u in units:
        assert u["offset"] == pos and stream[pos:pos + 4] == b"BBCD"
        assert int(u["fields"]["parse_code"], 16) == stream[pos + 4]
        nxt = int(u["fields"]["next_parse_offset"], 16)
        assert int(u["fields"]["prev_parse_offset"], 16) == prev
        prev = nxt
        pos += nxt if nxt else 13
    assert pos == len(stream)
    codes = [int(u["fields"]["parse_code"], 16) for u in units]
    assert codes[0] == 0x00 and codes[-1] == 0x10
    assert set(codes[1:-1]) == ({0xEC} if fragmented else {0xE8})

    # sequence header as the reference reads it
    sh = units[0]["fields"]
    want_major = 3 if (fragmented or case["bits"] > 12) else 2
    assert int(sh["Major Version"]) == want_major and int(sh["Profile"]) == 3
    assert int(sh["Picture Coding Mode"]) == (1 if interlaced and "Source Sampling" in sh else 0)
    if "Frame Width" in sh:
        assert (int(sh["Frame Width"]), int(sh["Frame Height"])) == (case["w"], case["h"])
    if name == "hq_1080i50_base_format":   # DataUnit.cpp:630: 1080i50 is base video format 12, level 3, nothing custom
        assert int(sh["Base Video Format"]) == 12 and int(sh["Level"]) == 3 and "Frame Width" not in sh

    # transform parameters + slices of every picture
    ph = case["h"] // (2 if interlaced else 1)
    xs, ys = case["w"] // (case["a"] << case["depth"]), ph // (case["u"] << case["depth"])
    headers = [u for u in units[1:-1] if not fragmented or int(u["fields"]["Slices"]) == 0]
    assert [int(u["fields"]["Picture Number"]) for u in headers] == list(range(pics))
    for u in headers:
        f = u["fields"]
        assert int(f["Wavelet"]) == KERNELS[case["kernel"]] and int(f["Depth"].split()[0]) == case["depth"]
        assert (int(f["Slices X"]), int(f["Slices Y"])) == (xs, ys)
        assert (int(f["Prefix Bytes"]), int(f["Slice Size Scalar"])) == (kw.get("prefix", 0), kw.get("scalar", 1))
    per_picture = {}
    for u in units[1:-1]:
        f = u["fields"]
        k = int(f["Picture Number"])
        if fragmented and int(f["Slices"]):
            have = len(per_picture.get(k, []))
            assert (int(f["Slice Y Offset"]) * xs + int(f["Slice X Offset"])) == have    # DataUnit.cpp:311-321
            assert int(f["Slices"]) == len(u["slices"])
            assert int(f["Fragment Length"]) <= max(kw["fragment_length"], int(f["Fragment Length"]) // len(u["slices"]))
        per_picture.setdefault(k, []).extend(u["slices"])
    assert sorted(per_picture) == list(range(pics))
    scalar = kw.get("scalar", 1)
    budget = None
    if kw.get("mode") == "HQ_CBR":
        budget = oracle.slice_bytes(ys, xs, kw["s"] // (2 if interlaced else 1), scalar).ravel()
    for k, sl in per_picture.items():
        assert len(sl) == xs * ys
        q = np.array([s[0] for s in sl])
        if kw.get("mode", "HQ_ConstQ") == "HQ_ConstQ":
            assert (q == kw["q"]).all()                      # quantIndicesConstQ, EncodeStream.cpp:128-138
        else:
            sizes = np.array([4 + sum(s[1:]) for s in sl])       # the tool prints lengths in bytes (length byte x scalar)
            assert np.array_equal(sizes, budget)             # HQ CBR: every slice fills its budget, Slices.cpp:305-382
            assert q.min() >= 0 and q.max() < 64
