"""Builds composer_amd/lib/libcomposer_hip.so from composer_amd/csrc/*.hip with hipcc for gfx950.

In-tree build (the .so travels to the GPU box with the repo snapshot).  Object files are cached under
composer_amd/csrc/_obj (untracked, key file beside each object so both always come from the same build) keyed by the SHA-256 of (source, every header, compiler flags, hipcc version): an object is rebuilt
when that key changes, never by file time.  composer_amd/lib/BUILD_INFO.json records, per source, the key and whether this
call compiled it or reused the cached object, plus the key the linked library was built from -- `verify()` (used by
__graft_entry__.build and the ABI test) fails if the library on disk does not match the sources in the tree.
    python -m composer_amd.build [--force] [--asan]
--asan: host-side AddressSanitizer build (device code unchanged) into lib/libcomposer_hip_asan.so; CPU container only.
"""
import hashlib
import json
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
# per-source additions.  attention.hip: without SLP vectorisation -- hipcc packs neighbouring f32 multiplies / adds of the softmax
# into v_pk_*_f32, which cost more issue time beside MFMAs than the scalar pairs they replace (MI355X_MICROARCH.md, "packed f32
# VALU ... an anti-lever beside MFMAs"); whole-step A/B: C2 26.47 -> 26.32 ms, B=32 7.43 -> 7.40, C4 56.04 -> 55.68.
SOURCE_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}


def flags_for(src, flags=None):
    return list(FLAGS if flags is None else flags) + SOURCE_FLAGS.get(src, [])


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _hipcc_version():
    try:
        return subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout.strip().split("\n")[0]
    except Exception:
        return "unknown"


def _headers():
    out = []
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith(".h"):
                out.append(os.path.join(d, f))
    return out


def source_key(src, flags):
    h = hashlib.sha256()
    h.update(("\0".join(flags) + "\0" + _hipcc_version()).encode())
    for path in [os.path.join(CSRC, src)] + _headers():
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()


def _compile(src, force, flags, tag=""):
    obj = os.path.join(OBJ, tag + src.replace(".hip", ".o"))
    keyfile = obj + ".key"
    key = source_key(src, flags)
    if not force and os.path.exists(obj) and os.path.exists(keyfile) and open(keyfile).read().strip() == key:
        return obj, False, key
    sp = os.path.join(CSRC, src)
    cmd = [_hipcc()] + flags + ["-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-4000:]))
    open(keyfile, "w").write(key)
    return obj, True, key


def _lib_key(keys):
    return hashlib.sha256("".join(keys).encode()).hexdigest()


def _file_sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _key_object(lib_key, tag):
    """A one-function object carrying the key the library is linked from: cmp_build_key() reads it back out of the .so itself,
    so verify() checks the LIBRARY, not only the JSON beside it."""
    src = os.path.join(OBJ, tag + "buildkey.cpp")
    obj = os.path.join(OBJ, tag + "buildkey.o")
    open(src, "w").write('extern "C" const char* cmp_build_key(void) { return "%s"; }\n' % lib_key)
    r = subprocess.run(["g++", "-O1", "-fPIC", "-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed for buildkey.cpp:\n" + r.stderr[-2000:])
    return obj


def build(force=False, verbose=True, asan=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    flags = FLAGS + (["-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan", "-g", "-fno-omit-frame-pointer"] if asan else [])
    tag = "asan_" if asan else ""
    lib = os.path.join(LIBDIR, "libcomposer_hip_asan.so") if asan else LIB
    info_path = os.path.join(LIBDIR, "BUILD_INFO_asan.json" if asan else "BUILD_INFO.json")
    with ThreadPoolExecutor(max_workers=min(5, os.cpu_count() or 1)) as ex:
        res = list(ex.map(lambda s: _compile(s, force, flags_for(s, flags), tag), SOURCES))
    objs = [o for o, _, _ in res]
    lib_key = _lib_key([k for _, _, k in res])
    old = {}
    try:
        old = json.load(open(info_path))
    except Exception:
        pass
    relink = any(c for _, c, _ in res) or not os.path.exists(lib) or old.get("library_key") != lib_key \
        or old.get("library_sha256") != _file_sha256(lib)
    if relink:
        objs = objs + [_key_object(lib_key, tag)]
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs + \
              ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"] + (["-fsanitize=address", "-shared-libsan"] if asan else [])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
    info = {"library": os.path.basename(lib), "library_key": lib_key, "library_sha256": _file_sha256(lib), "hipcc": _hipcc_version(), "arch": ARCH,
            "sources": {s: {"key": k, "compiled_by_this_call": bool(c)} for s, (_, c, k) in zip(SOURCES, res)},
            "relinked_by_this_call": bool(relink)}
    json.dump(info, open(info_path, "w"), indent=1)
    if verbose:
        for s, (_, c, _) in zip(SOURCES, res):
            print("  %-18s %s" % (s, "compiled" if c else "cached object (source key unchanged)"))
        print(("built " if relink else "up to date: ") + lib)
    return lib


def verify():
    """True when lib/libcomposer_hip.so was linked from exactly the sources + headers + flags now in the tree."""
    try:
        info = json.load(open(os.path.join(LIBDIR, "BUILD_INFO.json")))
    except Exception:
        return False
    if not os.path.exists(LIB):
        return False
    want = _lib_key([source_key(s, flags_for(s)) for s in SOURCES])
    if info.get("library_key") != want or info.get("library_sha256") != _file_sha256(LIB):
        return False
    # ... and the key linked INTO the library (a replaced or older .so next to a fresh BUILD_INFO.json fails here)
    import ctypes
    try:
        fn = ctypes.CDLL(LIB).cmp_build_key
    except (OSError, AttributeError):
        return False
    fn.restype = ctypes.c_char_p
    return fn().decode() == want


if __name__ == "__main__":
    build(force="--force" in sys.argv, asan="--asan" in sys.argv)
