"""tests/test_gpu_pack16.py::test_pack16w_dd97_cbr_and_prefix under the context flags (which path is at fault when it fails)"""
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vc2hip_py
from vc2lib import load_oracle, make_params
import test_gpu_pack16 as T
oracle = load_oracle()
raw = T._half_noise(T.WW, T.HW, "444", 10, 93)
kw = dict(mode="HQ_CBR", s=T.WW * T.HW, scalar=4)
p = make_params(T.WW, T.HW, "444", 10, "DD97", 5, 1, 1, **kw)
stream = oracle.encode_stream(p, raw, 1)
for name, fl in (("default", 0), ("NO_PAIR", 4), ("CBR_GENERAL", 0x100), ("NO_STREAM", 2), ("STORE32", 1)):
    hip = vc2hip_py.Vc2Hip(0, flags=fl)
    fmt = vc2hip_py.picture_format(T.WW, T.HW, "444", 10)
    cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 5, 1, 1, **kw)
    try:
        payload, _ = hip.encode_picture_hq(raw, fmt, cp)
        print(name, "ok" if stream[:-13].endswith(payload) else "MISMATCH")
    except Exception as e:
        print(name, "ERROR", str(e)[:80])
