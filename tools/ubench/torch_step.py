#!/usr/bin/env python3
"""Reference point only (not used by the product): the same train step written in plain PyTorch (bf16 autocast, fused
SDPA, torch.optim.Adam(fused)) on the same GPU -- i.e. what the vendor-library path gives for this model and batch."""
import sys, time, math, torch
import torch.nn as nn, torch.nn.functional as F
V, E, H, L, T = 390, 512, 8, 6, 1024
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64

class Block(nn.Module):
    def __init__(s):
        super().__init__()
        s.ln1, s.ln2 = nn.LayerNorm(E, eps=1e-5), nn.LayerNorm(E, eps=1e-5)
        s.c_attn, s.c_proj = nn.Linear(E, 3 * E), nn.Linear(E, E)
        s.c_fc, s.c_proj2 = nn.Linear(E, 4 * E), nn.Linear(4 * E, E)
    def forward(s, x):
        u = s.ln1(x)
        q, k, v = s.c_attn(u).view(x.shape[0], T, 3, H, E // H).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v, dropout_p=0.1, is_causal=True).transpose(1, 2).reshape(x.shape[0], T, E)
        r = u + F.dropout(s.c_proj(a), 0.1)                     # the reference's LN1 output replaces the residual stream
        return r + F.dropout(s.c_proj2(F.gelu(s.c_fc(s.ln2(r)), approximate="tanh")), 0.1)

class Model(nn.Module):
    def __init__(s):
        super().__init__()
        s.wte, s.wpe = nn.Embedding(V, E), nn.Embedding(T, E)
        s.blocks = nn.ModuleList(Block() for _ in range(L))
        s.lnf = nn.LayerNorm(E, eps=1e-5)
    def forward(s, x, y):
        h = F.dropout(s.wte(x) + s.wpe(torch.arange(T, device=x.device)), 0.1)
        for b in s.blocks: h = b(h)
        z = s.lnf(h) @ s.wte.weight.t()
        return F.cross_entropy(z.float().view(-1, V), y.view(-1))

m = Model().cuda()
opt = torch.optim.Adam(m.parameters(), lr=1e-3, eps=1e-7, fused=True)
x = torch.randint(0, V, (B, T), device="cuda"); y = torch.randint(0, V, (B, T), device="cuda")
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = m(x, y)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 8
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("torch eager bf16 (autocast + fused SDPA + fused Adam), B=%d: %.1f ms/step, %.2f M tokens/s" % (B, dt * 1e3, B * T / dt / 1e6))
