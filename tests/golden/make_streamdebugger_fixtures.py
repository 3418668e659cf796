#!/usr/bin/env python3
"""Generates tests/golden/streamdebugger.json: what the reference's own stream parser
(/root/reference/tools/vc2streamdebugger, run here as a program -- it cannot travel) reports for
streams written by oracle/ : sequence header fields, data-unit chain, fragment headers and the
per-slice quantiser index / component lengths of every HQ slice.  tests/test_oracle_streamdebugger.py
rebuilds the same streams with the oracle and checks them against this record, which pins the oracle's
HQ stream syntax -- including HQ fragments and interlaced field pictures, for which SURVEY Appendix B
holds no digest -- on the reference's reading of it.

Run in the build container only:  python tests/golden/make_streamdebugger_fixtures.py
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from synth import synth                      # noqa: E402
from vc2lib import load_oracle, make_params  # noqa: E402

TOOL = "/root/reference/tools/vc2streamdebugger"

CASES = {
    # name: (synth args, make_params args/kwargs, frames)
    "hq_constq_v2": dict(w=128, h=64, cf="422", bits=10, kernel="DD97", depth=2, u=2, a=2, frames=2, seed=71,
                         kw=dict(q=7, scalar=2, prefix=1)),
    "hq_cbr_fragments": dict(w=128, h=64, cf="422", bits=10, kernel="LeGall", depth=2, u=2, a=2, frames=1, seed=72,
                             kw=dict(mode="HQ_CBR", s=5000, scalar=2, prefix=1, fragment_length=300)),
    "hq_interlaced_bff": dict(w=128, h=64, cf="420", bits=8, kernel="Haar1", depth=2, u=2, a=2, frames=2, seed=73, wb=1,
                              kw=dict(q=5, scalar=1, interlaced=True, bottom_field_first=True)),
    "hq_1080i50_base_format": dict(w=1920, h=1080, cf="422", bits=10, kernel="LeGall", depth=2, u=27, a=32, frames=1, seed=74,
                                   kw=dict(q=30, scalar=16, interlaced=True, frame_rate=3)),
    "hq_16bit_v3": dict(w=64, h=64, cf="444", bits=16, kernel="Haar0", depth=2, u=1, a=1, frames=1, seed=75,
                        kw=dict(q=20, scalar=4)),
}


def build(case):
    wb = case.get("wb", 2)
    raw = synth(case["w"], case["h"], case["cf"], case["bits"], case["seed"], frames=case["frames"], word_bytes=wb)
    p = make_params(case["w"], case["h"], case["cf"], case["bits"], case["kernel"], case["depth"], case["u"], case["a"],
                    word_bytes=wb, **case["kw"])
    return load_oracle().encode_stream(p, raw, case["frames"])


def digest_of_report(text):
    """The facts of the tool's verbose report, as data."""
    units, cur = [], None
    for line in text.splitlines():
        m = re.match(r"0x([0-9a-f]+) : \[ PARSE INFO \]", line)
        if m:
            cur = {"offset": int(m.group(1), 16), "fields": {}, "slices": []}
            units.append(cur)
            continue
        if cur is None:
            continue
        m = re.match(r"\s+(\d+) -> \(\s*(\d+),\s*(\d+),\s*(\d+)\)", line)
        if m:
            cur["slices"].append([int(g) for g in m.groups()])
            continue
        m = re.match(r"\s+([A-Za-z_][A-Za-z0-9_ ]*?)\s*:\s+(\S+)", line)
        if m:
            cur["fields"][m.group(1).strip()] = m.group(2)
        if "Error" in line or "Warning" in line:
            cur.setdefault("problems", []).append(line.strip())
    return units


def main():
    out = {"_comment": "Output of /root/reference/tools/vc2streamdebugger -v on oracle streams (data only). "
                       "Made by tests/golden/make_streamdebugger_fixtures.py."}
    for name, case in CASES.items():
        stream = build(case)
        with tempfile.NamedTemporaryFile(suffix=".vc2") as f:
            f.write(stream)
            f.flush()
            r = subprocess.run([sys.executable, TOOL, "-v", f.name], capture_output=True, text=True, check=True)
        units = digest_of_report(r.stdout)
        out[name] = {"stream_bytes": len(stream), "stream_sha256": hashlib.sha256(stream).hexdigest(), "units": units}
        print(name, len(stream), len(units), "units", sum(len(u["slices"]) for u in units), "slices")
    json.dump(out, open(os.path.join(HERE, "streamdebugger.json"), "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
