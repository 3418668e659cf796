"""Several library contexts alive at once (each with its own stream): is a context fast or slow for life?"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vc2hip_py
from synth import synth
w,h=3840,2160
dev = torch.device("cuda:0")
B=int(os.environ.get("B","16"))
raw = synth(w, h, "422", 10, 1234)
one = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
N=int(os.environ.get("NCTX","5"))
ctxs=[]
for i in range(N):
    hip = vc2hip_py.Vc2Hip(0)
    fmt = vc2hip_py.picture_format(w, h, "422", 10)
    cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 4, 1, 2, q=16, scalar=2)
    rb = hip.raw_picture_bytes(fmt); stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    d_raw = one.repeat(B)
    d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
    ctxs.append((hip, fmt, cp, d_raw, d_pay, d_len, stride))
torch.cuda.synchronize()
for rnd in range(3):
    out=[]
    for (hip, fmt, cp, d_raw, d_pay, d_len, stride) in ctxs:
        def step(): hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
        for _ in range(2): step()
        hip.sync(); hip.profile_reset(); hip.profile_enable(True)
        for _ in range(5): step()
        hip.sync(); hip.profile_enable(False)
        prof = {k: round(v[1] / 5, 3) for k, v in hip.profile().items()}
        out.append((prof["dwt_level_first"], prof["dwt_level"]))
    print("round", rnd, out)
