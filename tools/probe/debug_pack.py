"""Slice-by-slice comparison of the GPU's HQ payload with the oracle's (debugging aid for the slice coders).
python tools/probe/debug_pack.py [w h q]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vc2hip_py
from synth import synth
from vc2lib import load_oracle, make_params
w, h, q = (int(a) for a in (sys.argv[1:4] + ["3840", "2160", "16"][len(sys.argv) - 1:]))
hip = vc2hip_py.Vc2Hip(0); oracle = load_oracle()
raw = synth(w, h, "422", 10, 1234)
fmt = vc2hip_py.picture_format(w, h, "422", 10)
cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 4, 1, 2, q=q, scalar=2)
stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
dev = torch.device("cuda:0")
d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
d_pay = torch.zeros(stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
hip.encode_batch_dev(d_raw.data_ptr(), 1, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr()); hip.sync()
n = int(d_len.cpu()[0]); got = bytes(d_pay.cpu().numpy()[:n])
p = make_params(w, h, "422", 10, "DD97", 4, 1, 2, q=q, scalar=2)
stream = oracle.encode_stream(p, raw, 1)
# the picture's slices are the tail of the stream before the end-of-sequence parse info (13 bytes); walk the oracle's slices
def walk(buf, count):
    out = []; pos = 0
    for _ in range(count):
        s = pos; pos += 1
        for _c in range(3): pos += 1 + 2 * buf[pos]
        out.append(buf[s:pos])
    return out, pos
ns = (w // 32) * (h // 16)
# find where the slices start in the oracle stream: try from the end
want_tail = stream[:-13]
pu = stream.find(b"BBCD", 1)
ws = None
for start in range(pu + 13, pu + 13 + 64):
    try:
        sl, end = walk(want_tail[start:], ns)
    except IndexError:
        continue
    if start + end == len(want_tail): ws = sl; break
print("gpu bytes", n, "oracle slices bytes", sum(len(s) for s in ws) if ws else None)
gs, gend = walk(got + bytes(4096), ns)
bad = [i for i in range(ns) if gs[i] != ws[i]]
print("slices", ns, "mismatching", len(bad), bad[:10])
for i in bad[:3]:
    print("slice", i, "len gpu", len(gs[i]), "oracle", len(ws[i]))
    print(" gpu   ", gs[i][:48].hex()); print(" oracle", ws[i][:48].hex())
    k = next((j for j in range(min(len(gs[i]), len(ws[i]))) if gs[i][j] != ws[i][j]), None)
    print(" first difference at byte", k)
