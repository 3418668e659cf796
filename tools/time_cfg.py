"""Timing of the other BASELINE configurations through the device-resident batch path (not the bench metric)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vc2hip_py
from synth import synth
CFG = {
    "cfg1": dict(w=1920, h=1080, cf="422", bits=10, k="LeGall", d=2, u=2, a=4, B=32, kw=dict(q=12)),
    "cfg2": dict(w=3840, h=2160, cf="422", bits=10, k="DD97", d=4, u=1, a=2, B=16, kw=dict(q=16, scalar=2)),
    "cfg3": dict(w=3840, h=2160, cf="422", bits=10, k="DD97", d=4, u=1, a=2, B=16, kw=dict(mode="HQ_CBR", s=8294400, scalar=2)),
    "cfg4": dict(w=7680, h=4320, cf="444", bits=12, k="Fidelity", d=5, u=1, a=1, B=4, kw=dict(q=40, scalar=8)),
    "cfg5": dict(w=1920, h=1080, cf="422", bits=8, k="LeGall", d=3, u=1, a=2, B=16, wb=1, kw=dict(mode="LD", s=1036800)),
}
dev = torch.device("cuda:0")
# TIME_CFG_FLAGS=two_pass_vbr,planes8_never ...: context flags by their names in vc2hip_py.FLAGS (A/B of two correct paths in ONE build)
_flags = 0
for _n in filter(None, os.environ.get("TIME_CFG_FLAGS", "").split(",")): _flags |= vc2hip_py.FLAGS[_n.strip().upper()]
hip = vc2hip_py.Vc2Hip(0, flags=_flags)
for name in sys.argv[1:] or list(CFG):
    batch = None
    if "@" in name: name, batch = name.split("@")[0], int(name.split("@")[1])   # cfg2@32: another batch size
    c = dict(CFG[name])
    if batch: c["B"] = batch
    wb = c.get("wb", 2)
    fmt = vc2hip_py.picture_format(c["w"], c["h"], c["cf"], c["bits"], wb)
    cp = vc2hip_py.coding_params(hip.lib, fmt, c["k"], c["d"], c["u"], c["a"], **c["kw"])
    B = c["B"]
    rb = hip.raw_picture_bytes(fmt); stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    tight = int(os.environ.get("TIME_CFG_STRIDE", "0"))   # (experiments: the decoder reads from tighter payload slots)
    raw = synth(c["w"], c["h"], c["cf"], c["bits"], 1234, frames=1, word_bytes=wb)
    one = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    d_raw = one.repeat(B)
    d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
    d_out = torch.zeros(B * rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    d_pay2 = torch.zeros(B * tight, dtype=torch.uint8, device=dev) if tight else None
    def step():
        hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
        if tight:
            hip.sync()
            d_pay2.view(B, tight).copy_(d_pay.view(B, stride)[:, :tight]); torch.cuda.synchronize()
            hip.decode_batch_dev(d_pay2.data_ptr(), tight, d_len.data_ptr(), B, fmt, cp, d_out.data_ptr())
        else:
            hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, d_out.data_ptr())
    def sync():   # TIME_CFG_IGNORE_ERRORS=1: pricing builds code wrong streams on purpose; their decoder's complaints are not the point
        try: hip.sync()
        except vc2hip_py.Vc2HipError:
            if not os.environ.get("TIME_CFG_IGNORE_ERRORS"): raise
    for _ in range(2): step()
    sync()
    hip.profile_reset(); hip.profile_enable(True)
    N = 5
    t0 = time.perf_counter()
    for _ in range(N): step()
    sync()
    dt = (time.perf_counter() - t0) / N
    hip.profile_enable(False)
    prof = {k: round(v[1] / N, 3) for k, v in hip.profile().items()}
    print(name, f"{dt * 1e3:.2f} ms/step of {B} pictures  {c['w'] * c['h'] * B / dt / 1e9:.2f} Gpx/s", prof)
