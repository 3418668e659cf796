import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vc2hip_py
from synth import synth
pad = int(os.environ.get("PAD", "0"))
w,h=3840,2160
dev = torch.device("cuda:0")
hip = vc2hip_py.Vc2Hip(0)
fmt = vc2hip_py.picture_format(w, h, "422", 10)
cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 4, 1, 2, q=16, scalar=2)
B=32
rb = hip.raw_picture_bytes(fmt); stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
raw = synth(w, h, "422", 10, 1234)
one = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
big = torch.empty(B*rb + pad + 4096, dtype=torch.uint8, device=dev)
d_raw = big[pad:pad+B*rb]; d_raw.view(B, rb)[:] = one
d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
d_out = torch.zeros(B * rb, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
def step():
    hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
    hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, d_out.data_ptr())
for _ in range(2): step()
hip.sync(); hip.profile_reset(); hip.profile_enable(True)
for _ in range(5): step()
hip.sync(); hip.profile_enable(False)
prof = {k: round(v[1] / 5, 3) for k, v in hip.profile().items()}
print("pad", pad, "raw%2M", hex(d_raw.data_ptr() % (1<<21)), "out%2M", hex(d_out.data_ptr() % (1<<21)), prof["dwt_level_first"], prof["idwt_level_final"], prof["hq_unpack"], prof["hq_pack"])
