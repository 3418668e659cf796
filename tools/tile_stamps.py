"""Phase timeline of the tile transform kernels (diagnostic build -DVC2HIP_STAMPS): python tools/tile_stamps.py"""
import os, struct, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = os.environ.setdefault("VC2HIP_TILE_STAMPS_FILE", "/tmp/tl.bin")
if os.path.exists(f): os.remove(f)
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vc2hip_py
from synth import synth
dev = torch.device("cuda:0"); hip = vc2hip_py.Vc2Hip(0)
W, H, B = 3840, 2160, 16
fmt = vc2hip_py.picture_format(W, H, "422", 10)
cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 4, 1, 2, q=16, scalar=2)
rb = hip.raw_picture_bytes(fmt); stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
raw = synth(W, H, "422", 10, 1234, frames=2) * (B // 2)
d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
d_out = torch.zeros(B * rb, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for _ in range(2):
    hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
    hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, d_out.data_ptr())
    hip.sync()
data = open(f, "rb").read()
pos = 0; launches = []
while pos < len(data):
    hdr = struct.unpack_from("8i", data, pos); pos += 32
    n = hdr[2] * hdr[3] * hdr[4]
    a = np.frombuffer(data, dtype=np.uint64, count=n * 16, offset=pos).reshape(n, 16).astype(np.int64); pos += n * 128
    launches.append((hdr, a))
for hdr, a in launches[len(launches) // 2:]:
    inv, w, gx, gy, gz, lds = hdr[:6]
    last = 5 if inv else 3
    t = a[a[:, 0] > 0]
    base = t[:, 0].min()
    names = ["qtab", "issue+land loads", "small gather", "lifting", "-", "write"] if inv else ["stage", "lifting", "write"]
    print(f"{'inv' if inv else 'fwd'} plane width {w}: grid {gx}x{gy}x{gz}, lds {lds}: {len(t)} working WGs, span {(t[:, last].max() - base) / 100:.1f} us, WG life median {np.median(t[:, last] - t[:, 0]) / 100:.2f} us")
    for i in range(last):
        if inv and i == 3: d = (t[:, 4] - t[:, 3]) / 100.0
        elif inv and i == 4: continue
        else: d = (t[:, i + 1] - t[:, i]) / 100.0
        print(f"   {names[i]:18s} median {np.median(d):6.2f} us  p90 {np.percentile(d, 90):6.2f}")
    st = (t[:, 0] - base) / 100.0; en = (t[:, last] - base) / 100.0
    print("   resident WGs:", [int(((st <= x) & (en > x)).sum()) for x in np.linspace(0, en.max(), 10)[1:-1]], " last start", round(float(st.max()), 1))
