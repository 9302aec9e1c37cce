#!/usr/bin/env python3
"""HBM-side traffic per launch of every kernel class of the TRAIN STEP, from PMC passes over bench.py itself
(tools/profile_round4.sh):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --no-extras --no-cpu-baseline --no-decode --steps 3 --warmup 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python3 bench.py ... (same command; the two counters do not fit one pass)
Every dispatch is assigned the class bench.py's `classes` table uses (0 forward GEMM, 1 dgrad GEMM, 2 wgrad GEMM, 3/4/5 attention
forward / dQ / dK-dV, 6 LayerNorm forward, 7 Adam, 8 LayerNorm backward).  Forward and input-gradient GEMMs run the same kernel
template, so the class comes from WHERE in the step the dispatch sits: dispatches are walked in order, the forward pass opens at
embed_fwd_kernel, the backward pass at softmax_xent*_kernel, the step ends at adam_kernel.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request of a 16-byte-per-lane streaming read at 64
bytes (MI355X_MICROARCH.md, HBM section): doubled here.  WRITE_SIZE is exact for 16-byte-per-lane stores and f32 atomics.

    python tools/make_traffic_json.py <run key, e.g. c2_tokens131072> <FETCH pass dir> <WRITE pass dir> profiles/hbm_traffic.json
merges one run into the JSON (other runs in the file are kept).
"""
import collections
import csv
import glob
import json
import sys


def classify(name, phase):
    if "gemm_wgrad_group" in name:
        return 2                                        # the grouped weight gradients of a block (contraction over tokens)
    if "gemm_bf16" in name or "gemm_f32" in name:
        if phase == "fwd":
            return 0
        if phase == "bwd":
            a_km = name.split("<", 1)[1].split(",")[0].strip() if "<" in name else "true"
            return 2 if a_km == "false" else 1         # first template argument A_KM = !ta: A stored [K,M] = contraction over tokens
        return None
    for key, cls in (("attn_fwd", 3), ("attn_dq_", 4), ("attn_dkv_", 5), ("layernorm_fwd_kernel", 6), ("adam_kernel", 7),
                     ("layernorm_bwd_kernel", 8)):
        if key in name:
            return cls
    return None


def read_pass(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r.get("Dispatch_Id", len(rows))), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort(key=lambda t: t[0])
    phase, out, steps = None, collections.defaultdict(list), 0
    per_kernel = collections.defaultdict(list)
    for _, name, val in rows:
        if "embed_fwd_kernel" in name:
            phase = "fwd"
            steps += 1
        elif "softmax_xent" in name:
            phase = "bwd"
        cls = classify(name, phase)
        if cls is not None and (phase is not None or cls == 7):
            out[cls].append(val)
        per_kernel[name[:100]].append(val)
        if "adam_kernel" in name:
            phase = None
    return out, per_kernel, steps


def main():
    key, fdir, wdir, path = sys.argv[1:5]
    fetch, fk, steps = read_pass(fdir, "FETCH_SIZE")
    write, wk, _ = read_pass(wdir, "WRITE_SIZE")
    classes = {}
    for cls in sorted(set(fetch) | set(write)):
        f = 2.0 * 1024.0 * sum(fetch[cls]) / max(1, len(fetch[cls]))
        w = 1024.0 * sum(write[cls]) / max(1, len(write[cls]))
        classes[str(cls)] = {"launches_sampled": len(fetch[cls]), "fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
                             "traffic_bytes_per_launch": f + w}
    kernels = {}
    for name in sorted(set(fk) | set(wk)):
        f = 2.0 * 1024.0 * sum(fk[name]) / max(1, len(fk[name]))
        w = 1024.0 * sum(wk[name]) / max(1, len(wk[name]))
        if f + w >= 1e6:
            kernels[name] = {"launches_sampled": len(fk[name]), "fetch_bytes": f, "write_bytes": w}
    try:
        doc = json.load(open(path))
        if "runs" not in doc:
            doc = {}
    except Exception:
        doc = {}
    doc["source"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --no-extras "
                     "--no-cpu-baseline --no-decode` itself: bytes at the L2's memory side per launch INSIDE the train step; FETCH_SIZE "
                     "doubled (gfx950 tallies 128-byte requests at 64 bytes); classes as in bench.py's `classes` table")
    # the library the passes ran on: bench.py prints a run's traffic only while the loaded library still has this key
    import ctypes, os
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "composer_amd", "lib", "libcomposer_hip.so"))
    lib.cmp_build_key.restype = ctypes.c_char_p
    doc.setdefault("runs", {})[key] = {"steps_sampled": steps, "build_key": lib.cmp_build_key().decode(), "classes": classes, "kernels": kernels}
    json.dump(doc, open(path, "w"), indent=1)
    print(key, {c: round(v["traffic_bytes_per_launch"] / 1e6, 1) for c, v in classes.items()})


if __name__ == "__main__":
    main()
