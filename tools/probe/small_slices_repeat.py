"""The smallest slices (2 x 2 samples and the like) coded and decoded again and again by one context: every result against the oracle
(the ablation build fails the first case; the release build must never)."""
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vc2hip_py
from synth import synth
from vc2lib import load_oracle, make_params
from test_gpu_parity import _fmt_cp
hip = vc2hip_py.Vc2Hip(0)
oracle = load_oracle()
bad = 0; n = 0
for (w, h, cf, bits, wb, k, d, u, a, q) in ((512, 256, "444", 8, 1, "Haar0", 1, 1, 1, 3), (256, 128, "422", 10, 2, "LeGall", 1, 1, 2, 0), (384, 192, "420", 8, 1, "Haar1", 1, 2, 2, 5), (512, 64, "444", 10, 2, "DD97", 2, 1, 1, 7)):
    raw = synth(w, h, cf, bits, 9000 + w, word_bytes=wb)
    p = make_params(w, h, cf, bits, k, d, u, a, q=q, word_bytes=wb)
    stream = oracle.encode_stream(p, raw, 1)
    dec, _ = oracle.decode_stream(p, stream, 1)
    fmt, cp = _fmt_cp(hip, w, h, cf, bits, k, d, u, a, q=q, word_bytes=wb)
    for it in range(int(sys.argv[1])):
        payload, _ = hip.encode_picture_hq(raw, fmt, cp)
        n += 1
        if payload != stream[-13 - len(payload):-13]: bad += 1
        if hip.decode_picture(payload, fmt, cp) != dec: bad += 1
    print(w, h, cf, k, "bad so far", bad, "of", n)
