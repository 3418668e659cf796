"""Random-geometry fuzz of the fused picture path against the oracle (GPU box): picture sizes (with padding),
chroma formats, bit depths, kernels, depths, slice sizes, modes.  Prints the failing cases.

  python tools/fuzz_geometry.py <seed> <cases> [wide]

wide: planes of 512 ... 2560 samples across and a few slice rows, so that levels go through the streaming transform
kernels and the decoder's band planes (the default sizes stay below them), mixed with tile-kernel levels underneath.
tall: such planes with 3 ... 33 slice rows (round 5: the segments of the streaming and the two-level kernels).
FUZZ_BATCH=1 (round 6): every case through vc2hip_encode_batch_dev / vc2hip_decode_batch_dev with 2 ... 8 pictures on a context of
its own."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vc2hip_py
from vc2lib import load_oracle, make_params, KERNELS
from synth import synth, noise_frame
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
wide = len(sys.argv) > 3 and sys.argv[3] in ("wide", "tall", "pair")
pairm = len(sys.argv) > 3 and sys.argv[3] == "pair"   # geometries the two-level kernels take: short filters, depth 2 - 4, every plane from 192 samples wide at the pair's level
tall = len(sys.argv) > 3 and sys.argv[3] in ("tall", "pair")   # wide planes with MANY slice rows: the streaming / pair kernels' segments (top, middle, bottom walks)
rnd = random.Random(seed)
# FUZZ_FLAGS=planes8_always,no_pair ...: context flags by their names in vc2hip_py.FLAGS (round 5: the byte band planes are
# chosen from the batch before, so a fuzz run that wants them in every case forces them)
_flags = 0
for _n in filter(None, os.environ.get("FUZZ_FLAGS", "").split(",")): _flags |= vc2hip_py.FLAGS[_n.strip().upper()]
hip = vc2hip_py.Vc2Hip(0, flags=_flags)
hip.profile_enable(True)   # (only to count, at the end, which transform kernels the cases went through)
oracle = load_oracle()
bad = 0
done = 0
skipped_ld = 0
while done < count:
    depth = rnd.choice([1, 2, 3, 4])
    cf = rnd.choice(["444", "422", "420"])
    bits, wb = rnd.choice([(8, 1), (10, 2), (12, 2), (16, 2)])
    kernel = rnd.choice(list(KERNELS))
    unit = 1 << depth
    # slice size in units (chroma needs >= 1 unit)
    u = rnd.choice([1, 2, 3, 4]) * (2 if cf == "420" else 1)
    a = rnd.choice([1, 2, 3, 4, 6]) * (1 if cf == "444" else 2)
    ys, xs = rnd.choice([1, 2, 3, 5]), rnd.choice([1, 2, 3, 7])
    if wide:
        a = rnd.choice([1, 2, 4, 8]) * (1 if cf == "444" else 2)
        xs = max(1, rnd.choice([512, 640, 768, 1024, 1536, 2560]) // (a * unit))
        ys = rnd.choice([1, 2, 3])
        u = rnd.choice([1, 2, 4]) * (2 if cf == "420" else 1)
        if tall:
            xs = max(1, rnd.choice([256, 512, 640, 1024]) // (a * unit))
            ys = rnd.choice([3, 5, 6, 7, 9, 12, 17, 24, 33])
        if pairm:
            depth = rnd.choice([2, 3, 4]); unit = 1 << depth
            kernel = rnd.choice(["DD97", "LeGall", "DD137", "Haar0", "Haar1"])
            a = rnd.choice([1, 2, 4]) * (1 if cf == "444" else 2)
            u = rnd.choice([1, 2]) * (2 if cf == "420" else 1)
            xs = max(1, rnd.choice([1536, 2048, 3072]) // (a * unit))
            ys = rnd.choice([4, 6, 7, 9, 12, 17, 24, 40])
    ph, pw = ys * u * unit, xs * a * unit
    # unpadded size: up to one unit less than padded (keeps chroma consistent: even crops)
    h = ph - rnd.choice([0, 0, 2, unit - 2 if unit > 2 else 0])
    w = pw - rnd.choice([0, 0, 2, unit - 2 if unit > 2 else 0])
    if h < 2 or w < 2 or (cf != "444" and w % 2) or (cf == "420" and h % 2):
        continue
    # the decoder pads chroma from the padded luma size: keep both paddings consistent
    cw = w if cf == "444" else w // 2
    ch = h // 2 if cf == "420" else h
    def pad(v): return (v + unit - 1) // unit * unit
    if pad(cw) != (pw if cf == "444" else pw // 2) or pad(ch) != (ph // 2 if cf == "420" else ph):
        continue
    mode = rnd.choice(["HQ_ConstQ", "HQ_ConstQ", "HQ_CBR", "LD"])
    scalar = rnd.choice([1, 2, 4, 8, 16])
    prefix = rnd.choice([0, 0, 1, 3])
    q = rnd.choice([0, 5, 12, 20, 33])
    ns = ys * xs
    sbytes = ns * rnd.choice([40, 90, 200]) + rnd.randrange(0, ns)
    kw = dict(q=q, scalar=scalar, prefix=prefix) if mode == "HQ_ConstQ" else (
        dict(mode="HQ_CBR", s=sbytes, scalar=rnd.choice([1, 2]), prefix=prefix) if mode == "HQ_CBR" else dict(mode="LD", s=sbytes))
    raw = (noise_frame if rnd.random() < 0.3 else synth)(w, h, cf, bits, rnd.randrange(1 << 30), word_bytes=wb)
    p = make_params(w, h, cf, bits, kernel, depth, u, a, word_bytes=wb, **kw)
    desc = f"{w}x{h} {cf} {bits}b {kernel} d{depth} u{u} a{a} {mode} {kw}"
    try:
        stream = oracle.encode_stream(p, raw, 1)
    except Exception as e:
        continue  # the reference itself rejects the case (scalar too small, index overflow, ...)
    done += 1
    if os.environ.get("FUZZ_VERBOSE"):
        print("case", done, desc, flush=True)
    try:
        dec, _ = oracle.decode_stream(p, stream, 1)
        fmt = vc2hip_py.picture_format(w, h, cf, bits, wb)
        cp = vc2hip_py.coding_params(hip.lib, fmt, kernel, depth, u, a, **kw)
        if os.environ.get("FUZZ_BATCH"):
            # round 6: the device-resident BATCH path on a context of its own (its workspaces are sized by this case alone, so a
            # buffer that is a few slots short runs off its end instead of into the slack of an earlier, larger case): n copies
            # of the picture, every slot against the oracle
            import torch
            nb = rnd.choice([2, 3, 5, 8])
            hb = vc2hip_py.Vc2Hip(0, flags=_flags)
            rb = hb.raw_picture_bytes(fmt)
            assert rb == len(raw)
            stride = (hb.max_payload_bytes(fmt, cp) + 255) // 256 * 256
            dev = torch.device("cuda:0")
            rpad = (-rb) % 16   # (pictures of a batch are packed back to back; the buffers must be 16-byte aligned as a whole)
            d_raw = torch.frombuffer(bytearray(raw * nb + bytes(rpad)), dtype=torch.uint8).to(dev)
            d_pay = torch.zeros(nb * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(nb, dtype=torch.int64, device=dev)
            d_out = torch.zeros(nb * rb + rpad, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            hb.encode_batch_dev(d_raw.data_ptr(), nb, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
            hb.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), nb, fmt, cp, d_out.data_ptr())
            hb.sync()
            lens = d_len.cpu().tolist(); pay = d_pay.cpu().numpy(); outs = d_out.cpu().numpy().tobytes()
            ok_e = all(stream[:-13].endswith(bytes(pay[k * stride:k * stride + lens[k]])) and lens[k] > 0 for k in range(nb))
            ok_d = all(outs[k * rb:(k + 1) * rb] == dec for k in range(nb))
            hb.close()
            del d_raw, d_pay, d_len, d_out
        else:
            payload, _ = hip.encode_picture_hq(raw, fmt, cp)
            ok_e = stream[:-13].endswith(payload)
            out = hip.decode_picture(payload, fmt, cp)
            ok_d = out == dec
        if not (ok_e and ok_d):
            bad += 1
            print("MISMATCH", "enc" if not ok_e else "", "dec" if not ok_d else "", desc)
    except Exception as e:
        if "exceeds 65534" in str(e):
            continue  # outside the reference's own 32-bit code domain (undefined behaviour there): refused here
        if mode == "LD" and "slice too large for the LD encode kernels" in str(e):
            skipped_ld += 1
            continue  # the one documented limit of the LD encoder: LL blocks / slice bytes beyond LDS -- a clean error
        bad += 1
        print("EXCEPTION", desc, str(e)[:120])
print(f"seed {seed}: {done} cases, {bad} bad, {skipped_ld} refused by the LD encoder's documented limit")
try:
    hip.sync()
    print("transform launches:", {k: v[0] for k, v in sorted(hip.profile().items()) if "dwt" in k})
except Exception:
    pass
