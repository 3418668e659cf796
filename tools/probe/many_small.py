import os, sys, hashlib
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vc2hip_py
from synth import synth
from vc2lib import load_oracle, make_params
from test_gpu_parity import _fmt_cp
hip = vc2hip_py.Vc2Hip(0)
oracle = load_oracle()
w, h = 512, 256
raw = synth(w, h, "444", 8, 9000 + w, word_bytes=1)
p = make_params(w, h, "444", 8, "Haar0", 1, 1, 1, q=3, word_bytes=1)
stream = oracle.encode_stream(p, raw, 1)
fmt, cp = _fmt_cp(hip, w, h, "444", 8, "Haar0", 1, 1, 1, q=3, word_bytes=1)
for it in range(6):
    payload, _ = hip.encode_picture_hq(raw, fmt, cp)
    want = stream[-13 - len(payload):-13]
    bad = [i for i in range(len(payload)) if payload[i] != want[i]]
    print(it, len(payload), "ok" if payload == want else ("DIFF n=%d first=%d last=%d" % (len(bad), bad[0], bad[-1])))
# which stage differs: the same picture under the context flags, and (ablation build) with the pack kernel's phases switched
for name, fl in (("SINGLE_PASS_VBR", 0x80), ("GENERIC_DWT", 0x40), ("NO_STREAM", 2)):
    h2 = vc2hip_py.Vc2Hip(0, flags=fl)
    f2, c2 = _fmt_cp(h2, w, h, "444", 8, "Haar0", 1, 1, 1, q=3, word_bytes=1)
    pay, _ = h2.encode_picture_hq(raw, f2, c2)
    print(name, "ok" if pay == stream[-13 - len(pay):-13] else "DIFF")
# the pieces: transform + quantise + pack through the fine-grained calls
import numpy as np
