"""Phase timeline of k_hq_pack (diagnostic build -DVC2HIP_STAMPS; workgroups of picture 0)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = os.environ.setdefault("VC2HIP_PACK_STAMPS_FILE", "/tmp/pk.bin")
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
torch.cuda.synchronize()
for _ in range(2):
    hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
    hip.sync()
a = np.fromfile(f, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
t = a[a[:, 0] > 0]
print(f"{len(t)} workgroups of picture 0; life median {np.median(t[:, 5] - t[:, 0]) / 100:.2f} us")
order = [0, 1, 6, 7, 2, 3, 4, 5]   # stamps in program order (6, 7: inside the luma round)
names = ["set-up (tables, zero, barrier)", "luma: load, quantise, codes", "luma: scan, lengths", "luma: code writes", "chroma round",
         "(end of the slice)", "copy-out"]
for i, nm in enumerate(names):
    d = (t[:, order[i + 1]] - t[:, order[i]]) / 100.0
    print(f"   {nm:32s} median {np.median(d):6.2f} us  p90 {np.percentile(d, 90):6.2f}")
