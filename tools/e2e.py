"""PCIe-inclusive timing (reported beside, never as `value`): UHD cfg-2 pictures that start and end
in host memory.  (a) the synchronous host-buffer entry points the tools use, pageable memory; (b) pinned staging
buffers, copies and kernels of a batch issued back to back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, vc2hip_py
from synth import synth
W, H = 3840, 2160
hip = vc2hip_py.Vc2Hip(0)
fmt = vc2hip_py.picture_format(W, H, "422", 10)
cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 4, 1, 2, q=16, scalar=2)
raw = synth(W, H, "422", 10, 1234, frames=1)
rb = len(raw)
# (a) host entry points
payload, _ = hip.encode_picture_hq(raw, fmt, cp)
hip.decode_picture(payload, fmt, cp)
N = 10
t0 = time.perf_counter()
for _ in range(N): payload, _ = hip.encode_picture_hq(raw, fmt, cp)
te = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N): out = hip.decode_picture(payload, fmt, cp)
td = (time.perf_counter() - t0) / N
assert out == hip.decode_picture(payload, fmt, cp)
print(f"host entry points (pageable): encode {te * 1e3:.2f} ms/picture = {W * H / te / 1e9:.2f} Gpx/s, decode {td * 1e3:.2f} ms = {W * H / td / 1e9:.2f} Gpx/s")
# (b) pinned staging, batch of B
B = 8
dev = torch.device("cuda:0")
stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
h_raw = torch.frombuffer(bytearray(raw * B), dtype=torch.uint8).pin_memory()
h_pay = torch.empty(B * stride, dtype=torch.uint8).pin_memory()
h_out = torch.empty(B * rb, dtype=torch.uint8).pin_memory()
d_raw = torch.empty(B * rb, dtype=torch.uint8, device=dev)
d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev)
d_len = torch.zeros(B, dtype=torch.int64, device=dev)
d_out = torch.empty(B * rb, dtype=torch.uint8, device=dev)
coded = len(payload)
def enc():
    d_raw.copy_(h_raw, non_blocking=True)
    torch.cuda.synchronize()
    hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
    hip.sync()
    for k in range(B):  # only the coded bytes travel back
        h_pay[k * stride:k * stride + coded].copy_(d_pay[k * stride:k * stride + coded], non_blocking=True)
    torch.cuda.synchronize()
def dec():
    for k in range(B):
        d_pay[k * stride:k * stride + coded].copy_(h_pay[k * stride:k * stride + coded], non_blocking=True)
    torch.cuda.synchronize()
    hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, d_out.data_ptr())
    hip.sync()
    h_out.copy_(d_out, non_blocking=True)
    torch.cuda.synchronize()
enc(); dec()
t0 = time.perf_counter()
for _ in range(5): enc()
te = (time.perf_counter() - t0) / 5 / B
t0 = time.perf_counter()
for _ in range(5): dec()
td = (time.perf_counter() - t0) / 5 / B
assert bytes(h_out[:rb].numpy()) == out
print(f"pinned staging, batch {B}, copies not overlapped with kernels: encode {te * 1e3:.2f} ms/picture = {W * H / te / 1e9:.2f} Gpx/s "
      f"({rb / te / 1e9:.1f} GB/s in), decode {td * 1e3:.2f} ms = {W * H / td / 1e9:.2f} Gpx/s ({rb / td / 1e9:.1f} GB/s out)")
