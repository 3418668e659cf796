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
