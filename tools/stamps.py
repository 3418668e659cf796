"""Per-wavefront timeline of the streaming transform kernels (diagnostic build only: -DVC2HIP_STAMPS, see tools/README.md).
  VC2HIP_STAMPS_FILE=/tmp/st.bin python tools/stamps.py   -> runs one cfg-2 batch and prints, per launch: wavefront durations,
  resident wavefronts over time, dispatch spread."""
import os, struct, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = os.environ.get("VC2HIP_STAMPS_FILE", "/tmp/st.bin")
if len(sys.argv) < 2 or sys.argv[1] != "read":
    if os.path.exists(f): os.remove(f)
    os.environ["VC2HIP_STAMPS_FILE"] = f
    sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch, vc2hip_py
    from synth import synth
    dev = torch.device("cuda:0"); hip = vc2hip_py.Vc2Hip(0)
    W, H, B = 3840, 2160, 16
    fmt = vc2hip_py.picture_format(W, H, "422", 10)
    cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 4, 1, 2, q=16, scalar=2)
    rb = hip.raw_picture_bytes(fmt); stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256
    raw = synth(W, H, "422", 10, 1234, frames=B)
    d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int64, device=dev)
    d_out = torch.zeros(B * rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(2):
        hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
        hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, d_out.data_ptr())
        hip.sync()
data = open(f, "rb").read()
pos = 0; launches = []
while pos < len(data):
    hdr = struct.unpack_from("8i", data, pos); pos += 32
    n = hdr[2] * hdr[3] * hdr[4]
    a = np.frombuffer(data, dtype=np.uint64, count=n * 4, offset=pos).reshape(n, 4); pos += n * 32
    launches.append((hdr, a))
half = len(launches) // 2
for hdr, a in launches[half:]:   # the second step (warm)
    inv, edge, gx, gy, gz, lds = hdr[:6]
    t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64); work = a[:, 3] > 0
    base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0   # us (100 MHz counter)
    dur = (t1 - t0)[work]
    span = t1.max()
    print(f"{'inv' if inv else 'fwd'} edge={edge} grid {gx}x{gy}x{gz} lds {lds}: {work.sum()} working wavefronts of {len(a)}, kernel span {span:.1f} us")
    print(f"   wave duration us: min {dur.min():.1f} p10 {np.percentile(dur,10):.1f} median {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} max {dur.max():.1f}; sum/span = {dur.sum()/span:.0f} resident on average")
    print(f"   last start {t0.max():.1f} us; empty-block starts: first {t0[~work].min() if (~work).any() else 0:.1f} last {t0[~work].max() if (~work).any() else 0:.1f}")
    # resident working wavefronts at 20 points of the span
    pts = np.linspace(0, span, 21)[1:-1]
    res = [(int(((t0[work] <= x) & (t1[work] > x)).sum())) for x in pts]
    print("   resident:", res)
    hw = a[:, 2]
    cu = ((hw >> 8) & 0xF).astype(int); se = ((hw >> 13) & 0x7).astype(int); xcc = (hw >> 32).astype(int) & 0xF
    # waves per (xcc, se, cu)
    key = xcc * 1000 + se * 16 + cu
    uniq, cnt = np.unique(key[work], return_counts=True)
    print(f"   CUs used {len(uniq)}, working wavefronts per CU min {cnt.min()} max {cnt.max()}")
    # duration by start-time quartile
    order = np.argsort(t0[work]); q = len(order) // 4
    print("   median duration by start quartile:", [round(float(np.median(dur[order[i*q:(i+1)*q]])), 1) for i in range(4)])
