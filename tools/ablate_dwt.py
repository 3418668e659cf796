"""Timing experiment (not a benchmark): per-kernel times of the cfg-2 encode with phases of the
first DWT level disabled through VC2HIP_DEBUG_SKIP (1 no loads, 2 no lifting, 4 no stores)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vc2hip_py
import numpy as np
hip = vc2hip_py.Vc2Hip(0)
fmt = vc2hip_py.picture_format(3840, 2160, "422", 10)
cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 4, 1, 2, q=16, scalar=2)
B = 16
rb = hip.raw_picture_bytes(fmt); stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
dev = torch.device("cuda:0")
d_raw = torch.randint(0, 255, (B * rb,), dtype=torch.uint8, device=dev)
d_raw[::2] &= 0x3F   # keep 10-bit MSB-justified words plausible
d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for it in range(2):
    hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
try: hip.sync()
except Exception as e: pass
hip.profile_reset(); hip.profile_enable(True)
for it in range(5):
    hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
try: hip.sync()
except Exception as e: pass
print(os.environ.get("VC2HIP_DEBUG_SKIP", "0"), {k: round(v[1] / 5, 4) for k, v in hip.profile().items() if k.startswith("dwt")})
