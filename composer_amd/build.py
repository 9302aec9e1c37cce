"""Builds composer_amd/lib/libcomposer_hip.so from composer_amd/csrc/*.hip with hipcc for gfx950.

In-tree build (the .so travels to the GPU box with the repo snapshot).  Object files are cached under
composer_amd/csrc/_obj and rebuilt when a source or header is newer.
    python -m composer_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcomposer_hip.so")
ROOT = os.path.dirname(HERE)
SOURCES = ["elementwise.hip", "gemm.hip", "attention.hip", "model.hip", "decode.hip"]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-inline-asm", "-I" + os.path.join(ROOT, "include")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newest_header():
    t = 0.0
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _compile(src, force):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    sp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(sp)
            and os.path.getmtime(obj) >= _newest_header()):
        return obj, False
    cmd = [_hipcc()] + FLAGS + ["-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-4000:]))
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(5, os.cpu_count() or 1)) as ex:
        res = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + \
              ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
