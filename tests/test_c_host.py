"""The C ABI from a plain C host: include/nst_hip.h is valid C99 and C++17, examples/host_c/nst_min.c builds against
libnst_hip.so with nothing but gcc and the HIP runtime, refuses to run without a GPU, and on an MI355X optimises a
small job end to end without Python or torch in the process."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

ROCM = "/opt/rocm"
LIBDIR = os.path.join(ROOT, "artstyletransfer_amd")
SRC = os.path.join(ROOT, "examples", "host_c", "nst_min.c")


def _build(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libnst_hip.so")):
        import __graft_entry__ as g
        g.build()
    exe = str(tmp_path / "nst_min")
    cmd = ["gcc", "-std=gnu99", "-O2", "-Wall", "-Werror", f"-I{ROOT}/include", f"-I{ROCM}/include", "-D__HIP_PLATFORM_AMD__", SRC,
           f"-L{LIBDIR}", "-lnst_hip", f"-L{ROCM}/lib", "-lamdhip64", "-lm", f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{ROCM}/lib",
           "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
@pytest.mark.parametrize("compiler,std", [("gcc", "-std=c99"), ("g++", "-std=c++17")])
def test_header_is_plain_c_and_cxx(tmp_path, compiler, std):
    src = tmp_path / ("t.c" if compiler == "gcc" else "t.cpp")
    src.write_text('#include "nst_hip.h"\nint main(void) { return nst_version() < 0; }\n')
    subprocess.run([compiler, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", f"-I{ROOT}/include", str(src)],
                   check=True, capture_output=True, text=True)


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_host_builds_and_refuses_without_a_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no GPU" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("optimizer,steps", [("lbfgs", 6), ("adam", 8)])
def test_c_host_optimises_a_job(tmp_path, optimizer, steps):
    exe = _build(tmp_path)
    r = subprocess.run([exe, optimizer, str(steps)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("step ")]
    assert len(lines) == steps and "DECREASED" in r.stdout
