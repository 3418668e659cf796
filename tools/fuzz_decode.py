"""Mutation fuzz of the picture decoder (GPU box): valid payloads made by the oracle, a few bytes overwritten at random
(anywhere: slice headers, length bytes, coefficient data), decoded by the GPU's picture path and by the oracle (HQ_ConstQ,
HQ_CBR -- the decoder's byte-budget short cut and its fall-back -- and LD).  Both must
either refuse the payload or return the same picture.

  python tools/fuzz_decode.py <seed> <cases>        (FUZZ_BIG=1: pictures of 1000 - 7000 slices, payloads of many index chunks)"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vc2hip_py
from vc2lib import load_oracle, make_params, KERNELS
from synth import synth, noise_frame
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rnd = random.Random(seed)
# FUZZ_FLAGS=planes8_always,no_pair ...: context flags by their names in vc2hip_py.FLAGS (round 5: the byte band planes are
# chosen from the batch before, so a fuzz run that wants them in every case forces them)
_flags = 0
for _n in filter(None, os.environ.get("FUZZ_FLAGS", "").split(",")): _flags |= vc2hip_py.FLAGS[_n.strip().upper()]
hip = vc2hip_py.Vc2Hip(0, flags=_flags)
oracle = load_oracle()
bad = both_err = same = 0
for case in range(count):
    depth = rnd.choice([1, 2, 3])
    cf = rnd.choice(["444", "422", "420"])
    kernel = rnd.choice(list(KERNELS))
    unit = 1 << depth
    u = rnd.choice([1, 2]) * (2 if cf == "420" else 1)
    a = rnd.choice([1, 2, 4]) * (1 if cf == "444" else 2)
    wide = rnd.random() < 0.3
    ys, xs = rnd.choice([1, 2, 4]), (rnd.choice([32, 64]) if wide else rnd.choice([1, 3, 8]))
    if os.environ.get("FUZZ_BIG"):   # payloads of many index chunks (round 6: the recorded walks of the slice index): 1000 - 7000 slices
        ys, xs = rnd.choice([16, 33, 56]), rnd.choice([64, 120])
    h, w = ys * u * unit, xs * a * unit
    scalar, prefix, q = rnd.choice([1, 2, 4]), rnd.choice([0, 0, 2]), rnd.choice([0, 8, 20])
    raw = (noise_frame if rnd.random() < 0.3 else synth)(w, h, cf, 10, rnd.randrange(1 << 30))
    mode = rnd.choice(["HQ_ConstQ", "HQ_ConstQ", "HQ_CBR", "LD"])
    ns = ys * xs
    sbytes = ns * rnd.choice([30, 60, 150]) + rnd.randrange(0, ns)
    kw = dict(q=q, scalar=scalar, prefix=prefix) if mode == "HQ_ConstQ" else (
        dict(mode="HQ_CBR", s=sbytes, scalar=rnd.choice([1, 2]), prefix=prefix) if mode == "HQ_CBR" else dict(mode="LD", s=sbytes))
    p = make_params(w, h, cf, 10, kernel, depth, u, a, **kw)
    try:
        stream = oracle.encode_stream(p, raw, 1)
    except Exception:
        continue
    fmt = vc2hip_py.picture_format(w, h, cf, 10, 2)
    cp = vc2hip_py.coding_params(hip.lib, fmt, kernel, depth, u, a, **kw)
    try:
        payload0, _ = hip.encode_picture_hq(raw, fmt, cp)
    except Exception as e:
        print("ENCODE", f"{w}x{h} {cf} {kernel} d{depth} u{u} a{a} {kw}", str(e)[:80]); bad += 1
        continue
    if not stream[:-13].endswith(payload0):
        print("ENCODE MISMATCH", f"{w}x{h} {cf} {kernel} d{depth} u{u} a{a} {kw}"); bad += 1
        continue
    head = stream[:len(stream) - 13 - len(payload0)]
    for m in range(4):
        pay = bytearray(payload0)
        for _ in range(rnd.choice([1, 1, 2, 5]) * (8 if os.environ.get("FUZZ_BIG") else 1)):
            pay[rnd.randrange(len(pay))] = rnd.choice([0, 0xFF, rnd.randrange(256)])
        pay = bytes(pay)
        try:
            got = hip.decode_picture(pay + (stream[-13:] if os.environ.get("FUZZ_TAIL") else b""), fmt, cp); gerr = None   # FUZZ_TAIL=1: with the bytes that follow the data unit in the stream
        except Exception as e:
            got, gerr = None, str(e)
        try:   # the oracle decodes whole streams: the stream's own headers around the mutated payload (same length)
            want, nfr = oracle.decode_stream(p, head + pay + stream[-13:], 1); werr = None if nfr == 1 else "no picture"
        except Exception as e:
            want, werr = None, str(e)
        if gerr and werr: both_err += 1
        elif gerr or werr or got != want:
            bad += 1
            if os.environ.get("FUZZ_DUMP") and bad == 1:
                import pickle
                pickle.dump(dict(w=w, h=h, cf=cf, kernel=kernel, depth=depth, u=u, a=a, kw=kw, pay=pay, pay0=payload0, head=head, tail=stream[-13:],
                                 got=got, want=want), open(os.environ["FUZZ_DUMP"], "wb"))
            print("MISMATCH", f"{w}x{h} {cf} {kernel} d{depth} u{u} a{a} {kw} mutation {m}: hip {'err ' + gerr[:60] if gerr else 'ok'} / oracle {'err ' + werr[:60] if werr else 'ok'}")
        else: same += 1
print(f"seed {seed}: {same} same pictures, {both_err} refused by both, {bad} bad")
