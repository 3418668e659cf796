import os, sys
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
payload, _ = hip.encode_picture_hq(raw, fmt, cp)
want = stream[-13 - len(payload):-13]
print("len", len(payload), len(want), payload == want)
def walk(buf):
    out = []; pos = 0
    while pos + 4 <= len(buf):
        s = pos; pos += 1
        for _c in range(3): pos += 1 + buf[pos]
        out.append((s, pos))
    return out
ws = walk(want); gs = walk(payload)
print("slices", len(ws), len(gs))
bad = [i for i, (a, b) in enumerate(ws) if i >= len(gs) or payload[a:b] != want[a:b]]
print("bad slices", len(bad), bad[:40])
for i in bad[:5]:
    a, b = ws[i]; print(i, want[a:b].hex(), payload[a:b].hex())
