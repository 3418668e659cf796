"""Which picture sizes run through the 16-bit store / k_hq_pack16 (the library's launch profile), and their slice counts mod 4:
how tests/test_gpu_pack16.py::test_one_pass_coder_is_the_default_from_112_pictures_on found a geometry whose last tile is ragged."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vc2hip_py
from synth import synth
hip = vc2hip_py.Vc2Hip(0, flags=vc2hip_py.FLAGS["SINGLE_PASS_VBR"])
for w, h in [(2080, 272), (2112, 272), (2080, 256), (2048, 272), (2048, 288), (2112, 288), (2080, 288), (2144, 256), (2176, 272), (2048, 304), (2304, 272), (2560, 272), (2048, 320)]:
    fmt = vc2hip_py.picture_format(w, h, "422", 10, 2)
    cp = vc2hip_py.coding_params(hip.lib, fmt, "DD97", 4, 1, 2, q=9, scalar=8)
    raw = synth(w, h, "422", 10, 7)
    hip.profile_reset(); hip.profile_enable(True)
    hip.encode_picture_hq(raw, fmt, cp)
    hip.profile_enable(False)
    seen = sorted(k for k, v in hip.profile().items() if v[0] > 0)
    print(w, h, "slices", cp.y_slices * cp.x_slices, "mod 4 =", (cp.y_slices * cp.x_slices) % 4, seen)
