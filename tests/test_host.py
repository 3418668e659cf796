"""CPU tests of the C++ host layer (vc2-reference_amd/host): the reference's own unit-test known answers
(hosttest.cpp) and the command-line validation of the two tools, which runs before any GPU work."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vc2-reference_amd", "bin")


@pytest.fixture(scope="module")
def tools():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "vc2-reference_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "vc2-reference_amd", "host")], stdout=subprocess.DEVNULL)
    return BIN


def test_reference_unit_test_known_answers(tools):
    out = subprocess.run([os.path.join(tools, "hosttest")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "all host tests passed" in out.stdout


BASE = ["-m", "HQ_ConstQ", "-k", "LeGall", "-d", "2", "-u", "2", "-a", "4", "-f", "4:2:2", "-x", "64", "-y", "32", "-l", "10"]


@pytest.mark.parametrize("extra,message", [
    ([], "Quantisation index must be set in HQ_ConstQ mode"),
    (["-q", "120"], "quantisation index must be in the range 0 to 119"),
    (["-q", "3", "-s", "100"], "Compressed bytes is only used in HQ_CBR and LD modes"),
    (["-q", "3", "-z", "10"], "bitDepth is incompatible with luma depth (and/or chroma depth): use one or the other"),
    (["-q", "3", "-n", "5"], "bytes must be in range 1 to 4"),
    (["-q", "3", "-S", "0"], "slice scalar must be >=1"),
    (["-q", "3", "-p", "-i"], "image can't be both interlaced and progressive: specify one or the other"),
])
def test_encodestream_argument_validation(tools, tmp_path, extra, message):
    # same checks and texts as /root/reference/src/EncodeStream/EncodeParams.cpp:148-204
    cmd = [os.path.join(tools, "EncodeStream")] + BASE + extra + [str(tmp_path / "in"), str(tmp_path / "out")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode != 0
    assert "Command line error: " + message in out.stderr


def test_encodestream_unknown_kernel_and_format(tools, tmp_path):
    for flag, val, msg in (("-k", "Bogus", "invalid wavelet kernel"), ("-f", "4:1:1", "invalid colour format")):
        args = list(BASE) + ["-q", "1"]
        args[args.index(flag) + 1] = val
        out = subprocess.run([os.path.join(tools, "EncodeStream")] + args + ["a", "b"], capture_output=True, text=True)
        assert out.returncode != 0 and msg in out.stderr


def test_tools_fail_loudly_without_gpu(tools, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    src = tmp_path / "in.raw"
    src.write_bytes(bytes(64 * 32 * 2 * 2))
    cmd = [os.path.join(tools, "EncodeStream")] + BASE + ["-q", "3", str(src), str(tmp_path / "out.vc2")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode != 0 and "no CPU fallback" in out.stdout


# ---- the reference's own sources built against host/*.h, from where they lie (build container only) ----------------
REF = "/root/reference"
HOST = os.path.join(ROOT, "vc2-reference_amd", "host")
CXX = ["g++", "-O1", "-std=c++14", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + HOST]
HOSTSRC = [os.path.join(HOST, f) for f in ("Arrays.cpp", "Picture.cpp")]
needs_reference = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "src/Library/src/Frame.cpp")),
                                     reason="reference checkout absent (GPU box): nothing of it is stored in this repository")


@needs_reference
def test_reference_frame_cpp_builds_against_host_headers_and_agrees(tmp_path):
    """src/Library/src/Frame.cpp (constructors, field accessors written with indices[Range(top, bottom, 2)][Range()]
    views as l- and r-values) is compiled unmodified against host/Frame.h, Picture.h, Arrays.h and linked with
    host/frametest.cpp; the same driver linked with host/Frame.cpp must print the same 90 lines."""
    outs = []
    for name, frame_cpp in (("ours", os.path.join(HOST, "Frame.cpp")), ("ref", os.path.join(REF, "src/Library/src/Frame.cpp"))):
        exe = str(tmp_path / ("frametest_" + name))
        subprocess.check_call(CXX + ["-o", exe, os.path.join(HOST, "frametest.cpp"), frame_cpp] + HOSTSRC)
        outs.append(subprocess.run([exe], capture_output=True, text=True, check=True).stdout)
    assert outs[0].count("\n") == 90 and outs[0] == outs[1]


@needs_reference
def test_reference_quant_indices_constq_builds_against_host_headers(tmp_path):
    """The body of quantIndicesConstQ is cut out of /root/reference/src/EncodeStream/EncodeStream.cpp at test time (into
    the pytest temp directory, never into the repository) and compiled against host/*.h: extents[a][b] construction,
    data(), num_elements(), return by value."""
    import re
    text = open(os.path.join(REF, "src/EncodeStream/EncodeStream.cpp")).read()
    m = re.search(r"^const Array2D quantIndicesConstQ\(.*?^}\n", text, re.S | re.M)
    assert m, "quantIndicesConstQ not found in the reference"
    (tmp_path / "region.inc").write_text(m.group(0))
    (tmp_path / "tu.cpp").write_text(
        '#include <algorithm>\n#include <cstdio>\n#include "Picture.h"\n#include "region.inc"\n'
        "int main() {\n"
        "  Array1D m(extents[7]);\n"
        "  const Array2D q = quantIndicesConstQ(Picture(), 3, 5, m, 21);\n"
        "  bool ok = q.shape()[0] == 3 && q.shape()[1] == 5;\n"
        "  for (int v = 0; v < 3; ++v) for (int h = 0; h < 5; ++h) ok = ok && q[v][h] == 21;\n"
        '  std::puts(ok ? "ok" : "bad");\n  return ok ? 0 : 1;\n}\n')
    exe = str(tmp_path / "constq")
    subprocess.check_call(CXX + ["-I" + str(tmp_path), "-o", exe, str(tmp_path / "tu.cpp")] + HOSTSRC)
    assert subprocess.run([exe], capture_output=True, text=True).stdout.strip() == "ok"


def _reference_quant_tool(tmp_path):
    """quant_factor, quant, quant_offset and scale (/root/reference/src/Library/src/Quantisation.cpp:40-95: four functions
    over the standard library only) cut into the pytest temp directory at test time and compiled as they stand; the
    driver around them prints `what q v result` lines for every index 0..119 and a set of values."""
    import re
    text = open(os.path.join(REF, "src/Library/src/Quantisation.cpp")).read()
    m = re.search(r"^const int quant_factor\(int q\) \{.*?^const int scale\(int value, int q\) \{.*?^}\n", text, re.S | re.M)
    assert m, "quant_factor .. scale not found in the reference"
    (tmp_path / "quant_region.inc").write_text(m.group(0))
    (tmp_path / "quant_tu.cpp").write_text(
        '#include <cstdio>\n#include <cstdlib>\n#include <stdexcept>\n#include "quant_region.inc"\n'
        "int main() {\n"
        "  static const int vals[] = {0, 1, 2, 3, 5, 12, 100, 511, 512, 1023, 4095, 32767, 65534, 100000, 1000000};\n"
        "  for (int q = 0; q < 120; ++q) {\n"
        '    std::printf("factor %d 0 %d\\noffset %d 0 %d\\n", q, quant_factor(q), q, quant_offset(q));\n'
        "    for (int v : vals) for (int s = -1; s <= 1; s += 2) {\n"
        '      std::printf("quant %d %d %d\\n", q, s * v, quant(s * v, q));\n'
        "      if ((long long)v * (unsigned)quant_factor(q) < (1ll << 31) - (1ll << 30))\n"   # scale() inside int arithmetic (the reference's own domain)
        '        std::printf("scale %d %d %d\\n", q, s * v, scale(s * v, q));\n'
        "    }\n  }\n"
        '  try { quant_factor(120); std::puts("nothrow"); } catch (const std::logic_error &e) { std::printf("throws %s\\n", e.what()); }\n'
        "  return 0;\n}\n")
    exe = str(tmp_path / "quanttool")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-w", "-I" + str(tmp_path), "-o", exe, str(tmp_path / "quant_tu.cpp")])
    return subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines()


@needs_reference
def test_reference_quant_scale_functions_agree_with_the_oracle(tmp_path):
    """moves quant_offset / scale (the dequantiser) from "digest only" to "reference code" in DESIGN
    section 2's table: the oracle's vc2o_quant_factor / vc2o_quant / vc2o_scale against the reference's own functions
    compiled from where they lie, all 120 indices, both signs, values up to the 32-bit code limit and beyond."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from vc2lib import load_oracle
    oracle = load_oracle()
    lines = _reference_quant_tool(tmp_path)
    assert lines[-1] == "throws quantization index exceeds maximum implemented value."
    n = 0
    for line in lines[:-1]:
        what, q, v, want = line.split()
        q, v, want = int(q), int(v), int(want)
        if what == "factor":
            assert oracle.quant_factor(q) == want, line
        elif what == "offset":   # quant_offset: what scale() adds to a non-zero product -- seen through scale(1, q)
            f = oracle.quant_factor(q)
            if f > 0 and want > 0 and f + want + 2 < (1 << 31):   # (indices 116..119: the factor's int is negative; beyond 2^31 the int sum wraps: both outside the domain of scale())
                assert oracle.scale(1, q) == (f + want + 2) // 4, line
        elif what == "quant":
            assert oracle.quant(v, q) == want, line
        else:
            assert oracle.scale(v, q) == want, line
        n += 1
    assert n > 120 * 40
