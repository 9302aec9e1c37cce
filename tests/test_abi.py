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


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "composer_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "transformer_oracle" not in src, f
