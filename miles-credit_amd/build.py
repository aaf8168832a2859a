"""Builds libwxengine.so (HIP, gfx950) in-tree with hipcc.  `python miles-credit_amd/build.py [--force]`.

The library carries a hash of the sources it was built from (`wx_version()` ends in "wxsrc:<16 hex>"): the .so is git-ignored and
travels prebuilt to the GPU box, so staleness is decided by comparing that hash with the sources on disk -- here (rebuild or not)
and again in wxengine.engine.load_library (refuse to run a library that does not match its sources)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "wxengine", "libwxengine.so")
SOURCES = [os.path.join(CSRC, "wx_engine.hip")]
MARK = b"wxsrc:"


def deps():
    return SOURCES + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(os.path.dirname(HERE), "include", "wxengine.h")]


def source_hash() -> str:
    h = hashlib.sha256()
    for path in deps():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def built_hash(path: str = OUT):
    """The source hash embedded in an existing library (read from its bytes: no dlopen needed), or None."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        blob = f.read()
    i = blob.find(MARK)
    if i < 0:
        return None
    return blob[i + len(MARK):i + len(MARK) + 16].decode("ascii", "replace")


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    return built_hash() != source_hash()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
           "-Wno-unused-result", f'-DWX_SOURCE_HASH="{source_hash()}"', "-o", OUT] + SOURCES
    if verbose:
        print("[wxengine] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT, built_hash())
