#!/usr/bin/env python3
"""Reference point only (not used by the product): torch's fused scaled_dot_product_attention (the ROCm flash-attention
kernels shipped with torch) on the model's attention shape, causal, bf16, forward and backward, dropout 0 and 0.1."""
import torch, time
import torch.nn.functional as F
B, H, T, D = 128, 8, 1024, 64
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
q, k, v = (torch.randn(B, H, T, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
fl = 2.0 * B * H * T * T * D
for p in (0.0, 0.1):
    for backend in ("flash", "efficient", "default"):
        try:
            from torch.nn.attention import sdpa_kernel, SDPBackend
            ctx = {"flash": sdpa_kernel(SDPBackend.FLASH_ATTENTION), "efficient": sdpa_kernel(SDPBackend.EFFICIENT_ATTENTION)}.get(backend)
            def fwd():
                return F.scaled_dot_product_attention(q, k, v, dropout_p=p, is_causal=True)
            if ctx is not None:
                with ctx:
                    us_f = t(fwd)
                    o = fwd(); do = torch.randn_like(o)
                    us_fb = t(lambda: fwd().backward(do))
            else:
                us_f = t(fwd)
                o = fwd(); do = torch.randn_like(o)
                us_fb = t(lambda: fwd().backward(do))
            print("sdpa %-9s p=%.1f  fwd %8.1f us (%6.1f TF/s causal-half)   fwd+bwd %8.1f us -> bwd ~%8.1f us" % (backend, p, us_f, fl / us_f / 1e6, us_fb, us_fb - us_f))
        except Exception as e:
            print("sdpa %-9s p=%.1f  unavailable: %s" % (backend, p, str(e)[:80]))
