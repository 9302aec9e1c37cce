#!/usr/bin/env python3
"""HBM-side traffic per launch of the GEMM kernels from the two PMC passes of tools/profile_round.sh.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane streaming
reads (MI355X_MICROARCH.md, HBM section): doubled here.  WRITE_SIZE is exact for 16-B-per-lane stores and f32 atomics.
    python tools/make_traffic_json.py profiles/r1_05_pmc_FETCH_SIZE_summary.txt profiles/r1_05_pmc_WRITE_SIZE_summary.txt profiles/hbm_traffic.json
"""
import json, re, sys

def parse(path):
    out, name = {}, None
    for ln in open(path):
        if not ln.startswith(" "):
            name = ln.strip()
        else:
            m = re.search(r"n=\s*(\d+) avg=([0-9.e+]+)", ln)
            out[name] = (int(m.group(1)), float(m.group(2)))
    return out

fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
KIND = {"false, 1>": "fwd c_attn (bias)", "false, 2>": "fwd c_fc (bias, gelu, aux)", "false, 3>": "fwd c_proj x2 (bias, dropout, residual)",
        "true, 4>": "dgrad mlp (gelu')", "true, 1>": "dgrad c_fc / attn c_proj", "true, 3>": "dgrad c_attn (+residual)",
        "false, 0>": "wgrad x4 (split-K atomics)"}
rows = {}
for k, (n, f) in fetch.items():
    w = write[k][1]
    label = next((v for kk, v in KIND.items() if kk in k), k)
    rows[label] = {"kernel": k, "launches_sampled": n, "fetch_bytes": 2.0 * f * 1024, "write_bytes": w * 1024,
                   "traffic_bytes": 2.0 * f * 1024 + w * 1024}
fwd = [rows["fwd c_attn (bias)"], rows["fwd c_proj x2 (bias, dropout, residual)"], rows["fwd c_proj x2 (bias, dropout, residual)"],
       rows["fwd c_fc (bias, gelu, aux)"]]
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `KB_B=128 python tools/kbench.py gemm`; "
                 "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B)",
       "tokens_per_launch": 131072, "per_kernel": rows,
       "class0_forward_gemm_mean_bytes_per_launch": sum(r["traffic_bytes"] for r in fwd) / 4}
json.dump(doc, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["traffic_bytes"] / 1e6, 1) for k, v in rows.items()}), doc["class0_forward_gemm_mean_bytes_per_launch"] / 1e6)
