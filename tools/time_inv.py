"""Timing experiment (not a benchmark): the inverse transform kernels of the cfg-2 decode, 32 pictures (experiment builds whose
results are wrong on purpose: nothing is checked)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vc2hip_py
from synth import synth
hip = vc2hip_py.Vc2Hip(0)
fmt = vc2hip_py.picture_format(3840, 2160, "422", 10)
cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 4, 1, 2, q=16, scalar=2)
B = 32
rb = hip.raw_picture_bytes(fmt); stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
dev = torch.device("cuda:0")
raw = synth(3840, 2160, "422", 10, 1234, frames=1)
d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev).repeat(B)
d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
d_out = torch.zeros(B * rb, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr()); hip.sync()
def run(n):
    for _ in range(n):
        hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, d_out.data_ptr())
    try: hip.sync()
    except Exception: pass
run(2)
hip.profile_reset(); hip.profile_enable(True)
run(5)
print({k: round(v[1] / 5, 4) for k, v in hip.profile().items() if k.startswith("idwt") or k == "hq_unpack"})
