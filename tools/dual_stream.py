"""Timing experiment (not a benchmark): does splitting the batch over two contexts (two streams, two workspaces)
let the GPU overlap the bandwidth-bound and the instruction-bound kernels of the two halves?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vc2hip_py
from synth import synth
W, H = 3840, 2160
B = 16
raw = synth(W, H, "422", 10, 1234, frames=1)
dev = torch.device("cuda:0")
def setup(nctx):
    ctxs = [vc2hip_py.Vc2Hip(0) for _ in range(nctx)]
    fmt = vc2hip_py.picture_format(W, H, "422", 10)
    cp = vc2hip_py.coding_params(ctxs[0].lib, fmt, "DD97", 4, 1, 2, q=16, scalar=2)
    rb = ctxs[0].raw_picture_bytes(fmt); stride = (ctxs[0].max_payload_bytes(fmt, cp) + 255) // 256 * 256
    one = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    d_raw = one.repeat(B)
    d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
    d_out = torch.zeros(B * rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    per = B // nctx
    def step(order):
        if order == "enc_dec_per_ctx":
            for k, c in enumerate(ctxs):
                o = k * per
                c.encode_batch_dev(d_raw.data_ptr() + o * rb, per, fmt, cp, d_pay.data_ptr() + o * stride, stride, d_len.data_ptr() + o * 8)
                c.decode_batch_dev(d_pay.data_ptr() + o * stride, stride, d_len.data_ptr() + o * 8, per, fmt, cp, d_out.data_ptr() + o * rb)
        else:
            for k, c in enumerate(ctxs):
                o = k * per
                c.encode_batch_dev(d_raw.data_ptr() + o * rb, per, fmt, cp, d_pay.data_ptr() + o * stride, stride, d_len.data_ptr() + o * 8)
            for k, c in enumerate(ctxs):
                o = k * per
                c.decode_batch_dev(d_pay.data_ptr() + o * stride, stride, d_len.data_ptr() + o * 8, per, fmt, cp, d_out.data_ptr() + o * rb)
    return ctxs, step, d_out, rb
for nctx in (1, 2, 4):
    for order in ("enc_then_dec", "enc_dec_per_ctx"):
        ctxs, step, d_out, rb = setup(nctx)
        for _ in range(3): step(order)
        for c in ctxs: c.sync()
        t0 = time.perf_counter()
        N = 20
        for _ in range(N): step(order)
        for c in ctxs: c.sync()
        dt = (time.perf_counter() - t0) / N
        ok = bool((d_out[:rb] == d_out[(B - 1) * rb:]).all())
        print(nctx, order, f"{dt * 1e3:.3f} ms/step  {W * H * B / dt / 1e9:.1f} Gpx/s  same={ok}")
        del ctxs
