"""Builds libmidiemo_hip.so (gfx950 only) in-tree with hipcc.  No torch headers,
no hipify, no multi-arch: the library is a plain C-ABI shared object."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
LIB = os.path.join(HERE, "libmidiemo_hip.so")
SOURCES = ["me_gemm.hip", "me_elem.hip", "me_attn.hip", "me_attn64.hip", "me_decode.hip", "me_decode_token.hip"]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs.  Without it hipcc parks the accumulators of loops whose
# MFMAs sit under a wave-uniform branch in AGPRs and copies all of them (v_accvgpr_read/write) around every
# iteration: 128 of ~250 VALU instructions per step in rga_bwd_kv, 96 of ~230 in rga_bwd_e.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-value",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    common = [os.path.join(CSRC, "me_common.h"), os.path.join(CSRC, "me_attn_common.h"), os.path.join(CSRC, "me_decode_common.h"), os.path.join(INCLUDE, "midiemo.h")]
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + common):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode()))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
