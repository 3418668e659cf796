"""Random pictures through the slice coders of csrc/vc2hip_pack16.h (k_hq_pack16: 32 x 16 slices, depth 4, 4:2:2;
k_hq_pack16w: 32 x 32 slices, depth 5, 4:4:4) against the oracle: wavelets, bit depths, quantiser indices 0 .. 70, slice
size scalars, prefixes, HQ_ConstQ / HQ_CBR, and contents that put slices on either side of the table path (smooth, noise,
half and half, sparse spikes, all zero).  Both sides must give the same bytes, or both must refuse the picture.
   python tools/fuzz_pack16.py <seed> <cases>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vc2hip_py
from vc2lib import load_oracle, make_params, KERNELS

seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
# FUZZ_FLAGS=single_pass_vbr,...: context flags by their names in vc2hip_py.FLAGS (round 6: the one-pass form of the coder)
_flags = 0
for _n in filter(None, os.environ.get("FUZZ_FLAGS", "").split(",")): _flags |= vc2hip_py.FLAGS[_n.strip().upper()]
hip = vc2hip_py.Vc2Hip(0, flags=_flags); oracle = load_oracle()
bad = refused = 0
for case in range(cases):
    wide = rng.random() < 0.35
    cf, depth, u, a = ("444", 5, 1, 1) if wide else ("422", 4, 1, 2)
    w, h = (2048, 512) if wide else (2048, 256)
    bits = int(rng.choice([8, 10, 12, 16]))
    kernel = str(rng.choice(list(KERNELS)))
    q = int(rng.integers(0, 71))
    scalar = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16]))
    prefix = int(rng.choice([0, 0, 0, 1, 5]))
    kind = int(rng.integers(0, 5))
    cw = w if cf == "444" else w // 2
    planes = []
    for pw in (w, cw, cw):
        yy, xx = np.mgrid[0:h, 0:pw]
        smooth = (0.5 + 0.4 * np.sin(xx / pw * 7 + case) * np.cos(yy / h * 5)) * (2 ** bits - 1)
        noise = rng.integers(0, 2 ** bits, size=(h, pw))
        if kind == 0: v = smooth + rng.normal(0, 2 ** bits * 0.005, size=(h, pw))
        elif kind == 1: v = noise
        elif kind == 2: v = np.where(xx < pw // 2, smooth, noise)
        elif kind == 3:
            v = np.full((h, pw), 2 ** (bits - 1), float); idx = rng.integers(0, h * pw, size=200); v.reshape(-1)[idx] = rng.integers(0, 2 ** bits, size=200)
        else: v = np.full((h, pw), int(rng.integers(0, 2 ** bits)), float)
        planes.append((np.clip(np.rint(v), 0, 2 ** bits - 1).astype(np.uint16) << (16 - bits)).astype(">u2").tobytes())
    raw = b"".join(planes)
    kw = dict(q=q, scalar=scalar, prefix=prefix)
    if rng.random() < 0.3: kw = dict(mode="HQ_CBR", s=int(w * h * rng.uniform(0.2, 2.0)) // 16 * 16, scalar=scalar, prefix=prefix)
    p = make_params(w, h, cf, bits, kernel, depth, u, a, **kw)
    fmt = vc2hip_py.picture_format(w, h, cf, bits)
    want = err_o = got = err_g = None
    try: stream = oracle.encode_stream(p, raw, 1); want = stream
    except Exception as e: err_o = str(e)
    try:
        cp = vc2hip_py.coding_params(hip.lib, fmt, kernel, depth, u, a, **kw)
        got, _ = hip.encode_picture_hq(raw, fmt, cp)
    except Exception as e: err_g = str(e)
    if err_o or err_g:
        refused += 1
        if not (err_o and err_g) and "65534" not in (err_g or ""):   # (codes beyond 32 bits: the oracle wraps like the reference, the library refuses: DESIGN 8)
            bad += 1; print("ONE SIDE REFUSES", case, wide, bits, kernel, kw, kind, "| oracle:", err_o, "| gpu:", err_g)
        continue
    if got != want[-13 - len(got):-13]:
        bad += 1; print("PAYLOAD", case, wide, bits, kernel, kw, kind)
        continue
    dec, _ = oracle.decode_stream(p, want, 1)
    if hip.decode_picture(got, fmt, cp) != dec:
        bad += 1; print("DECODE", case, wide, bits, kernel, kw, kind)
print(f"fuzz_pack16 seed {seed}: {cases} cases, {refused} refused by both, {bad} mismatches")
