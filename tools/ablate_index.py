"""Timing experiment (not a benchmark): slice-index kernels of the cfg-2 decode with phases of the
tables kernel cut short through VC2HIP_DEBUG_INDEX (1 stage only, 2 + walk, 3 + resolve)."""
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
host = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
d_raw = host.repeat(B)
d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
d_out = torch.zeros(B * rb, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
dbg = os.environ.pop("VC2HIP_DEBUG_INDEX", "0")
hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr()); hip.sync()
os.environ["VC2HIP_DEBUG_INDEX"] = dbg
import ctypes; ctypes.CDLL(None).setenv(b"VC2HIP_DEBUG_INDEX", dbg.encode(), 1)
for it in range(2):
    hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, d_out.data_ptr())
try: hip.sync()
except Exception: pass
hip.profile_reset(); hip.profile_enable(True)
for it in range(5):
    hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, d_out.data_ptr())
try: hip.sync()
except Exception: pass
print(dbg, {k: round(v[1] / 5, 4) for k, v in hip.profile().items() if k.startswith("slice_index")})
