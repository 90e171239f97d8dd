"""include/bjx.h is a C header: a plain C99 program (tests/abi_smoke.c) compiles against it with -pedantic -Werror
(CPU test) and, on the GPU box, links libbjx_hip.so, runs bjx_chain and checks the numbers against libm."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "abi_smoke.c")
LIB = os.path.join(ROOT, "bijectors.jl_amd", "libbjx_hip.so")
CFLAGS = ["-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include")]


def test_header_and_consumer_compile_as_c99(tmp_path):
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    subprocess.check_call([gcc, *CFLAGS, "-c", SRC, "-o", str(tmp_path / "abi_smoke.o")])
    # the header alone, included first, with nothing else in scope
    hdr = tmp_path / "only_header.c"
    hdr.write_text('#include "bjx.h"\nint main(void) { return BJX_VERSION == 100 ? 0 : 1; }\n')
    subprocess.check_call([gcc, *CFLAGS, str(hdr), "-o", str(tmp_path / "only_header")])
    subprocess.check_call([str(tmp_path / "only_header")])


@pytest.mark.gpu
def test_c_program_runs_the_chain_through_the_abi(tmp_path):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    exe = str(tmp_path / "abi_smoke")
    rocm_lib = "/opt/rocm/lib"
    subprocess.check_call(["gcc", *CFLAGS, SRC, "-o", exe, LIB, f"-L{rocm_lib}", "-lamdhip64", "-lm",
                           f"-Wl,-rpath,{os.path.dirname(LIB)}", f"-Wl,-rpath,{rocm_lib}"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke ok" in out.stdout
