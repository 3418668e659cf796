"""Which allocation's placement decides the mode of the last inverse level?  One process: three output buffers and three contexts
(each with its own coefficient store), cfg 2, 128 pictures; the kernel's time for every (context, buffer) pair."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vc2hip_py
from synth import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
ctxs = [vc2hip_py.Vc2Hip(0) for _ in range(3)]
hip = ctxs[0]
fmt = vc2hip_py.picture_format(3840, 2160, "422", 10)
cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 4, 1, 2, q=16, scalar=2)
rb = hip.raw_picture_bytes(fmt); stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
raw = synth(3840, 2160, "422", 10, 1234, frames=1)
d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev).repeat(B)
d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr()); hip.sync()
outs = [torch.zeros(B * rb, dtype=torch.uint8, device=dev) for _ in range(3)]
for rep in range(2):
    for ci, c in enumerate(ctxs):
        for oi, o in enumerate(outs):
            for _ in range(3): c.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, o.data_ptr())
            c.sync(); c.profile_reset(); c.profile_enable(True)
            for _ in range(3): c.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, o.data_ptr())
            c.sync(); c.profile_enable(False)
            pr = c.profile()
            print(f"rep {rep} context {ci} out {oi} (at {o.data_ptr():#x}): idwt_level_final {pr['idwt_level_final'][1] / 3:.3f}  hq_unpack {pr['hq_unpack'][1] / 3:.3f} ms")
