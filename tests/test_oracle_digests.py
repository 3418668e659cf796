"""Pins oracle/ against bbc/vc2-reference OUTPUT: the SHA-256 digests of reference
streams and decoded files recorded in SURVEY.md Appendix B (tests/golden/reference_digests.json).
A byte-exact match of a multi-megabyte stream covers ingest, padding, the DWT, the quantiser,
the quantisation matrix, HQ slice packing (VBR and CBR), the CBR search and stream syntax."""
import hashlib
import json
import os

import pytest

from synth import synth
from vc2lib import make_params

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_digests.json")))


def _run(oracle, name, raw=None):
    g = GOLD[name]
    pr = g["params"]
    p = make_params(pr["width"], pr["height"], pr["cf"], pr["bits"], pr["kernel"], pr["depth"],
                    pr["u"], pr["a"], mode=pr["mode"], q=pr.get("q", 0), s=pr.get("s", 0),
                    scalar=pr["scalar"])
    if raw is None:
        raw = synth(pr["width"], pr["height"], pr["cf"], pr["bits"], 1234, frames=g["frames"])
    stream = oracle.encode_stream(p, raw, g["frames"])
    assert len(stream) == g["stream"]["bytes"]
    assert hashlib.sha256(stream).hexdigest() == g["stream"]["sha256"]
    dec, n = oracle.decode_stream(p, stream, g["frames"])
    assert n == g["frames"] and len(dec) == g["decoded"]["bytes"]
    assert hashlib.sha256(dec).hexdigest() == g["decoded"]["sha256"]


def test_synth_generator_digest():
    raw = synth(1920, 1080, "422", 10, 1234)
    assert hashlib.sha256(raw).hexdigest() == GOLD["synth_1080p_422_10b"]["sha256"]


def test_cfg1_1080p_legall_constq(oracle):
    _run(oracle, "cfg1")


@pytest.mark.slow
def test_cfg2_and_cfg3_uhd_dd97(oracle):
    raw = synth(3840, 2160, "422", 10, 1234, frames=2)
    _run(oracle, "cfg2", raw)
    _run(oracle, "cfg3", raw[:len(raw) // 2])   # HQ_CBR on the first frame


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("VC2_SLOW") != "1", reason="~3 min / 4 GB: set VC2_SLOW=1")
def test_cfg4_uhd2_fidelity(oracle):
    _run(oracle, "cfg4")
