"""Timing experiment (not a benchmark): hq_pack with phases disabled through VC2HIP_DEBUG_PACK
(1 no code writes, 2 no copy-out)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vc2hip_py
from synth import synth
hip = vc2hip_py.Vc2Hip(0)
fmt = vc2hip_py.picture_format(3840, 2160, "422", 10)
cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 4, 1, 2, q=16, scalar=2)
B = 16
rb = hip.raw_picture_bytes(fmt); stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
dev = torch.device("cuda:0")
raw = synth(3840, 2160, "422", 10, 1234)
d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev).repeat(B)
d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for it in range(2):
    hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
try: hip.sync()
except Exception: pass
hip.profile_reset(); hip.profile_enable(True)
for it in range(5):
    hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
try: hip.sync()
except Exception: pass
print(os.environ.get("VC2HIP_DEBUG_PACK", "0"), {k: round(v[1] / 5, 4) for k, v in hip.profile().items() if "pack" in k or "compact" in k})
