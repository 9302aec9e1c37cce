"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU and exports
every symbol include/composer_hip.h declares; the product path refuses to run without a device."""
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from composer_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "composer_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cmp_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from composer_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), "library does not export " + s
        assert s in _lib.SIGNATURES, "ctypes binding missing for " + s
    assert set(_lib.SIGNATURES) == set(syms)


def test_loads_without_gpu_and_reports_version(lib):
    assert lib.cmp_version() == 1
    assert lib.cmp_device_count() >= 0


def test_product_path_fails_loudly_without_gpu(lib):
    """No CPU fallback: constructing the model without a HIP device must raise, not degrade."""
    import torch
    if lib.cmp_device_count() > 0:
        pytest.skip("a GPU is present")
    from composer_amd import _lib
    from composer_amd.transformer import Transformer
    with pytest.raises(_lib.HipLibraryError):
        Transformer(390, 64, 32, 1, 4)


def test_library_on_disk_was_built_from_the_sources_in_the_tree():
    """The .so ships prebuilt to the GPU box: its BUILD_INFO.json key must equal the key of the current sources, headers,
    flags and compiler (composer_amd/build.py), so a stale library cannot pass for the tree's code."""
    from composer_amd import build as b
    assert b.verify(), "run `python -m composer_amd.build`: lib/libcomposer_hip.so is older than the sources"


def test_product_never_imports_the_oracle():
    """... and neither do the measurement scripts under tools/: the checker is used from tests/ (tests/extra holds the fuzzers),
    __graft_entry__.smoke() and bench.py's cpu_baseline() only."""
    for top in ("composer_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                    assert "transformer_oracle" not in src, f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    body = bench[bench.index("def cpu_baseline("):]
    body = body[:body.index("\ndef ", 10)]
    rest = bench.replace(body, "")
    assert not re.search(r"^\s*(from|import)\s+oracle\b", rest, flags=re.M)      # bench.py: inside cpu_baseline() only


@pytest.mark.skipif(os.environ.get("COMPOSER_TEST_ASAN") != "1", reason="set COMPOSER_TEST_ASAN=1 (builds the host-ASan library, ~1 min)")
def test_host_asan_build_runs_the_error_paths_clean():
    """`python -m composer_amd.build --asan` (host-side AddressSanitizer, device code unchanged) + tools/asan_probe.py."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-m", "composer_amd.build", "--asan"], cwd=root, check=True)
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "--print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "asan_probe.py")], capture_output=True, text=True,
                       env=dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "asan probe done" in r.stdout and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]


def test_the_library_refuses_a_second_gpu_runtime_in_the_process():
    """Library load order: libcomposer_hip.so loaded BEFORE torch binds to /opt/rocm's runtime, a later `import
    torch` maps torch's own copies beside it; the binding notices at its next check and names both paths and the fix.  The other
    order (torch first: this test session, bench.py, the CLI under a launcher) shares one runtime."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from composer_amd import _lib\n"
            "_lib.load(); assert all(len(v) == 1 for v in _lib.mapped_runtime_libraries().values())\n"
            "import torch\n"
            "try:\n"
            "    _lib.require_gpu()\n"
            "except _lib.HipLibraryError as e:\n"
            "    assert 'two copies' in str(e) and 'torch/lib' in str(e) and 'import torch' in str(e); print('REFUSED')\n" % root)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "REFUSED" in p.stdout, (p.stdout[-800:], p.stderr[-800:])
    from composer_amd import _lib
    assert all(len(v) == 1 for v in _lib.mapped_runtime_libraries().values()), _lib.mapped_runtime_libraries()
