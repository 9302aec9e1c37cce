#!/usr/bin/env python3
"""Per-kernel hash of the device code in a built library (or two: prints the kernels that differ).  Used to check that a
source reorganisation left the shipped kernels instruction-for-instruction unchanged.
    python tools/isa_hash.py composer_amd/lib/libcomposer_hip.so [other.so]
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels(lib):
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            part = os.path.join(d, "b%d.bin" % i)
            open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(d, "b%d.co" % i)
            r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", co], capture_output=True, text=True).stdout
            cur, body = None, []
            for line in dis.split("\n"):
                m = re.match(r"^<(.+)>:$", line.strip()) if line and not line.startswith(" ") and not line.startswith("\t") else None
                if m:
                    if cur:
                        out[cur] = hashlib.md5("\n".join(body).encode()).hexdigest()[:12] + " %d" % len(body)
                    cur, body = m.group(1), []
                elif cur is not None:
                    body.append(re.sub(r"//.*$", "", line).strip())
            if cur:
                out[cur] = hashlib.md5("\n".join(body).encode()).hexdigest()[:12] + " %d" % len(body)
    return out


if __name__ == "__main__":
    a = kernels(sys.argv[1])
    if len(sys.argv) == 2:
        for k in sorted(a):
            print(a[k], k)
    else:
        b = kernels(sys.argv[2])
        same = sum(1 for k in a if b.get(k) == a[k])
        print("%d kernels in A, %d in B, %d identical" % (len(a), len(b), same))
        for k in sorted(set(a) | set(b)):
            if a.get(k) != b.get(k):
                print("  %-14s %-14s %s" % (a.get(k, "-"), b.get(k, "-"), k[:150]))
