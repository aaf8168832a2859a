"""Builds libwxengine.so (HIP, gfx950) in-tree with hipcc.  `python miles-credit_amd/build.py [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "wxengine", "libwxengine.so")
SOURCES = [os.path.join(CSRC, "wx_engine.hip")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
    os.path.join(os.path.dirname(HERE), "include", "wxengine.h")]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fgpu-rdc" if False else "-fno-gpu-rdc",
           "-Wno-unused-result", "-o", OUT] + SOURCES
    if verbose:
        print("[wxengine] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
