"""What the adaptive choice of byte band planes sees per BASELINE configuration (ablation build, VC2HIP_PLANES8_DEBUG=1):
a few decode batches with a sync between them, so that every batch's statistics arrive before the next call."""
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["VC2HIP_PLANES8_DEBUG"] = "1"
import torch, vc2hip_py
from synth import synth
CFG = {
    "cfg1": dict(w=1920, h=1080, cf="422", bits=10, k="LeGall", d=2, u=2, a=4, B=8, kw=dict(q=12)),
    "cfg2": dict(w=3840, h=2160, cf="422", bits=10, k="DD97", d=4, u=1, a=2, B=4, kw=dict(q=16, scalar=2)),
    "cfg3": dict(w=3840, h=2160, cf="422", bits=10, k="DD97", d=4, u=1, a=2, B=4, kw=dict(mode="HQ_CBR", s=8294400, scalar=2)),
    "cfg4": dict(w=7680, h=4320, cf="444", bits=12, k="Fidelity", d=5, u=1, a=1, B=2, kw=dict(q=40, scalar=8)),
}
dev = torch.device("cuda:0")
hip = vc2hip_py.Vc2Hip(0)
for name, c in CFG.items():
    print("==", name, flush=True)
    fmt = vc2hip_py.picture_format(c["w"], c["h"], c["cf"], c["bits"], 2)
    cp = vc2hip_py.coding_params(hip.lib, fmt, c["k"], c["d"], c["u"], c["a"], **c["kw"])
    B = c["B"]
    rb = hip.raw_picture_bytes(fmt); stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    raw = synth(c["w"], c["h"], c["cf"], c["bits"], 1234, frames=1)
    d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev).repeat(B)
    d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
    d_out = torch.zeros(B * rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr()); hip.sync()
    for it in range(4):
        hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, d_out.data_ptr()); hip.sync()
