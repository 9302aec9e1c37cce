#!/usr/bin/env python3
"""Builds a VARIANT of libcomposer_hip.so for same-box A/B timing: one or more sources recompiled with extra -D flags, linked
with the cached objects of the others into composer_amd/lib/<name>.so.  Run the arms in ONE gpurun call (boxes differ by
up to ~10 %):   COMPOSER_HIP_LIB=composer_amd/lib/<name>.so python tools/kbench.py gemm
    python tools/ab_build.py <name> <source.hip>[,<source2.hip>...] -DFOO [-DBAR ...]
The experiments build (round-3 attention forwards, first-generation decode kernels, *_DIAG measurement ladders): with
-DCOMPOSER_EXPERIMENTS a source that has a lab copy (composer_amd/csrc/experiments/<name>_lab.hip: the translation unit as it
stood at the end of round 5, ladders and dead kernels included) is compiled from that copy -- the shipped files hold shipped
kernels only:
    python tools/ab_build.py experiments attention.hip,decode.hip,gemm.hip -DCOMPOSER_EXPERIMENTS
"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composer_amd import build as B

def main():
    name, srcs, extra = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    B.build(verbose=False)
    built = {}
    for src in srcs:
        built[src] = os.path.join(B.OBJ, "ab_%s_%s" % (name, src.replace(".hip", ".o")))
        path = os.path.join(B.CSRC, src)
        lab = os.path.join(B.CSRC, "experiments", src.replace(".hip", "_lab.hip"))
        if "-DCOMPOSER_EXPERIMENTS" in extra and os.path.exists(lab):
            path = lab
        subprocess.run([B._hipcc()] + B.flags_for(src) + extra + ["-c", path, "-o", built[src]], check=True)
    objs = [built.get(s, os.path.join(B.OBJ, s.replace(".hip", ".o"))) for s in B.SOURCES] + [os.path.join(B.OBJ, "buildkey.o")]
    out = os.path.join(B.LIBDIR, name + ".so")
    subprocess.run([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs +
                   ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    print("built", out)

if __name__ == "__main__":
    main()
