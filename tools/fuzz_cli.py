"""Random command lines through the tools (GPU box): EncodeStream's stream against the oracle's, DecodeStream's decoded file
against the oracle's -- picture sizes with padding, chroma formats, bit depths and word sizes, wavelets, depths, slice sizes,
modes (HQ_ConstQ / HQ_CBR / LD), interlace (-i, -b), fragments (-F), several frames, one or two workers.

  python tools/fuzz_cli.py <seed> <cases>"""
import os, sys, random, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from vc2lib import load_oracle, make_params, KERNELS
from synth import synth, noise_frame
BIN = os.path.join(ROOT, "vc2-reference_amd", "bin")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rnd = random.Random(seed)
oracle = load_oracle()
tmp = tempfile.mkdtemp(prefix="vc2fuzz", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
bad = done = 0
while done < count:
    depth = rnd.choice([1, 2, 3])
    cf = rnd.choice(["444", "422", "420"])
    bits, wb = rnd.choice([(8, 1), (10, 2), (12, 2), (16, 2)])
    kernel = rnd.choice(list(KERNELS))
    unit = 1 << depth
    interlaced = rnd.random() < 0.3
    u = rnd.choice([1, 2, 4]) * (2 if cf == "420" else 1)
    a = rnd.choice([1, 2, 4]) * (1 if cf == "444" else 2)
    ys, xs = rnd.choice([1, 2, 3]), rnd.choice([1, 2, 5, 16])
    ph, pw = ys * u * unit, xs * a * unit            # padded picture (field) size
    hh = ph - rnd.choice([0, 0, 2]) if ph > 4 else ph
    w = pw - rnd.choice([0, 0, 2]) if pw > 4 else pw
    if cf != "444" and w % 2: continue
    if cf == "420" and hh % 2: continue
    cw, ch = (w if cf == "444" else w // 2), (hh // 2 if cf == "420" else hh)
    def pad(v): return (v + unit - 1) // unit * unit
    if pad(w) != pw or pad(hh) != ph or pad(cw) != (pw if cf == "444" else pw // 2) or pad(ch) != (ph // 2 if cf == "420" else ph): continue
    h = 2 * hh if interlaced else hh               # frame height
    frames = rnd.choice([1, 2, 3])
    mode = rnd.choice(["HQ_ConstQ", "HQ_ConstQ", "HQ_CBR", "LD"])
    scalar, prefix, q = rnd.choice([1, 2, 4]), rnd.choice([0, 0, 1, 3]), rnd.choice([0, 6, 15, 30])
    ns = ys * xs
    sbytes = ns * rnd.choice([40, 90, 200]) + rnd.randrange(0, ns)
    flen = rnd.choice([0, 0, 1, 100, 5000]) if mode != "HQ_ConstQ" else 0
    bff = interlaced and rnd.random() < 0.5
    kw = dict(q=q, scalar=scalar, prefix=prefix) if mode == "HQ_ConstQ" else (
        dict(mode="HQ_CBR", s=sbytes, scalar=rnd.choice([1, 2]), prefix=prefix) if mode == "HQ_CBR" else dict(mode="LD", s=sbytes))
    raw = b"".join((noise_frame if rnd.random() < 0.2 else synth)(w, h, cf, bits, rnd.randrange(1 << 30), word_bytes=wb) for _ in range(frames))
    p = make_params(w, h, cf, bits, kernel, depth, u, a, word_bytes=wb, interlaced=interlaced, bottom_field_first=bff, fragment_length=flen, **kw)
    try:
        want = oracle.encode_stream(p, raw, frames)
        wdec, n = oracle.decode_stream(p, want, frames)
    except Exception:
        continue   # the reference itself rejects the case
    done += 1
    args = ["-m", mode, "-k", kernel, "-d", depth, "-u", u, "-a", a, "-f", {"444": "4:4:4", "422": "4:2:2", "420": "4:2:0"}[cf],
            "-x", w, "-y", h, "-l", bits, "-n", wb]
    if mode != "LD": args += ["-S", kw["scalar"], "-P", prefix]
    if mode == "HQ_ConstQ": args += ["-q", q]
    else: args += ["-s", sbytes]
    if interlaced: args += ["-i"] + (["-b"] if bff else [])
    if flen: args += ["-F", flen]
    if rnd.random() < 0.3: args += ["--devices", "0,0"]
    desc = " ".join(str(x) for x in args) + f" ({frames} frames)"
    open(os.path.join(tmp, "in.raw"), "wb").write(raw)
    def run(tool, *a2):
        return subprocess.run([os.path.join(BIN, tool)] + [str(x) for x in a2], capture_output=True, text=True, timeout=120)
    r = run("EncodeStream", *args, os.path.join(tmp, "in.raw"), os.path.join(tmp, "o.vc2"))
    if r.returncode != 0:
        if "exceeds 65534" in r.stderr + r.stdout: continue   # outside the reference's own 32-bit code domain: refused here
        bad += 1; print("ENCODE FAILED", desc, "rc", r.returncode, "err:", r.stderr.strip()[-160:], "out:", r.stdout.strip()[-160:]); continue
    got = open(os.path.join(tmp, "o.vc2"), "rb").read()
    if got != want:
        bad += 1; print("STREAM DIFFERS", desc); continue
    r = run("DecodeStream", os.path.join(tmp, "o.vc2"), os.path.join(tmp, "d.raw"))
    if r.returncode != 0:
        bad += 1; print("DECODE FAILED", desc, r.stderr.strip()[-120:]); continue
    if open(os.path.join(tmp, "d.raw"), "rb").read() != wdec:
        # (interlaced LD streams: the reference's decoder halves the picture's byte budget a second time, DecodeStream.cpp:331,
        # and parses every slice with half its size; where a luma length then exceeds its slice its reader runs on -- the
        # decoder follows it since the end of round 3, LdUnpackParams)
        bad += 1; print("DECODED FILE DIFFERS", desc)
for f in os.listdir(tmp): os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
print(f"seed {seed}: {done} command lines, {bad} bad")
