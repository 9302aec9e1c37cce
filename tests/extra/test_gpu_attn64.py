"""-m gpu: the 64-rows-per-wave LDS-DMA attention kernels (bf16, D = 64; attention.hip "fwd64" ...) at EVERY size.  The size
heuristic sends only chip-filling launches to them; COMPOSER_ATTN64=force makes the small and ragged shapes of the kernel-level
parity tests (float64 restatement of transformer.py:331-371 with the exact -1e4 mask, bit-identical dropout masks) take them
too, and the full-length BASELINE shapes are run both ways."""
import os
import sys
import pytest
import torch

# NOT part of the default suite: these kernels exist only in a library built with -DCOMPOSER_EXPERIMENTS
#   python tools/ab_build.py experiments attention.hip -DCOMPOSER_EXPERIMENTS
#   COMPOSER_HIP_LIB=composer_amd/lib/experiments.so python -m pytest tests/extra/test_gpu_attn64.py -m gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_kernels as K

pytestmark = pytest.mark.gpu
lib = K.lib


@pytest.fixture(autouse=True, params=["force", "pipe", "dense"])
def force_attn64(request):
    """force: 64 rows per wave ("fwd64"); pipe: 32 rows per wave, software-pipelined across 32-key units ("fwd32p");
    dense: 32 rows per wave, round-2 tile body on the LDS-DMA ring at four waves per SIMD ("fwd32d")"""
    old = os.environ.get("COMPOSER_ATTN64")
    os.environ["COMPOSER_ATTN64"] = request.param
    yield
    if old is None:
        os.environ.pop("COMPOSER_ATTN64", None)
    else:
        os.environ["COMPOSER_ATTN64"] = old


@pytest.mark.parametrize("B,T,H", [(2, 33, 2), (2, 200, 2), (1, 256, 3), (1, 257, 1), (1, 333, 2), (3, 520, 1), (1, 777, 2), (2, 1024, 2)])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_attention_fwd_bwd_forced(lib, B, T, H, p):
    K.test_attention_fwd_bwd(lib, K.BF16, B, T, H, 64, p)


@pytest.mark.parametrize("T,p", [(1024, 0.0), (1024, 0.1), (2048, 0.1)])
def test_attention_full_length_forced(lib, T, p):
    K.test_attention_full_length_many_groups(lib, K.BF16, T, p)


def test_attention_forced_rescale_forced(lib):
    K.test_attention_forced_rescale(lib)


def test_forced_and_default_kernels_agree_bitwise_on_lse_within_tolerance_on_o(lib):
    """Same inputs through the round-2 kernel (COMPOSER_ATTN64=off) and the new one: lse within 1e-5, o within bf16 rounding."""
    B, T, H, D = 4, 1024, 8, 64
    E = H * D
    g = torch.Generator().manual_seed(5)
    qkv = K.dev(torch.randn(B * T, 3 * E, generator=g), K.BF16)
    outs = []
    for mode in ("off", "force"):
        os.environ["COMPOSER_ATTN64"] = mode
        o = torch.zeros(B * T, E, device="cuda", dtype=torch.bfloat16)
        lse = torch.zeros(B * H * T, device="cuda")
        K.ck(lib, lib.cmp_k_attn_fwd(K.stream(), K.P(qkv), K.P(o), K.P(lse), B, T, H, D, 1, K.BF16, 0.1, 3, 4))
        torch.cuda.synchronize()
        outs.append((o.float().cpu(), lse.cpu()))
    assert (outs[0][1] - outs[1][1]).abs().max() < 1e-4
    assert (outs[0][0] - outs[1][0]).abs().max() < 3e-2
