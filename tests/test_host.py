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
