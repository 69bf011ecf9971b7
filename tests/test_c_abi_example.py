"""The C-ABI is usable from plain C: examples/c_abi_smoke.c is compiled with gcc (no hipcc, no Python, no torch in the
consumer) against include/midiemo.h and the in-tree library; on a GPU box it is also run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "midi-emotion_amd", "midiemo")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build(out):
    if not os.path.exists(os.path.join(LIBDIR, "libmidiemo_hip.so")):
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    cmd = ["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "examples", "c_abi_smoke.c"),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROCM, "include"), "-L" + LIBDIR,
           "-L" + os.path.join(ROCM, "lib"), "-lmidiemo_hip", "-lamdhip64", "-lm",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_consumer_compiles_against_the_header(tmp_path):
    exe = build(str(tmp_path / "c_abi_smoke"))
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_c_consumer_runs(tmp_path):
    exe = build(str(tmp_path / "c_abi_smoke"))
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout
