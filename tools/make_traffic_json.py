#!/usr/bin/env python3
"""HBM-side traffic per launch of the forward GEMMs from the two PMC passes of tools/profile_round.sh (which run
`tools/kbench.py gemmfwd`: c_attn, attn c_proj, c_fc, mlp c_proj, the model's layouts and epilogues).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane streaming
reads (MI355X_MICROARCH.md, HBM section): doubled here.  WRITE_SIZE is exact for 16-B-per-lane stores.
    python tools/make_traffic_json.py profiles/<tag>_pmc_FETCH_SIZE_summary.txt profiles/<tag>_pmc_WRITE_SIZE_summary.txt profiles/hbm_traffic.json
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
KIND = {", 1>": "fwd c_attn (bias)", ", 2>": "fwd c_fc (bias, gelu, aux)", ", 3>": "fwd attn c_proj + mlp c_proj (bias, dropout, residual)"}
rows, tot, n = {}, 0.0, 0
for k, (cnt, f) in fetch.items():
    w = write[k][1]
    label = next((v for kk, v in KIND.items() if kk in k), k)
    rows[label] = {"kernel": k, "launches_sampled": cnt, "fetch_bytes": 2.0 * f * 1024, "write_bytes": w * 1024,
                   "traffic_bytes": 2.0 * f * 1024 + w * 1024}
    tot += cnt * (2.0 * f * 1024 + w * 1024)
    n += cnt
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `KB_B=128 python tools/kbench.py gemmfwd`; "
                 "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B)",
       "tokens_per_launch": 131072, "per_kernel": rows,
       "class0_forward_gemm_mean_bytes_per_launch": tot / n}
json.dump(doc, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["traffic_bytes"] / 1e6, 1) for k, v in rows.items()}), doc["class0_forward_gemm_mean_bytes_per_launch"] / 1e6)
