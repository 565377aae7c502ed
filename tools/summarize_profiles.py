#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/profile_round.sh into small, committable files:
<tag>_kernel_stats.csv (per-kernel time), <tag>_pmc_summary.json (HBM bytes + MFMA busy per launch
for every nrl:: kernel) and profiles/pmc_in_proj_fwd.json (what bench.py reports as `traffic`)."""
import collections
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
engine = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
# what the counters were taken on: the git head (handed in by the caller: the GPU box has no .git) and the build id of
# the library (content hash of the kernel sources) -- bench.py compares the latter with the library it runs
git_head = sys.argv[4] if len(sys.argv) > 4 else None
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from newsreclib_amd import _build
    build_id = _build.library_build_id()
except Exception:
    build_id = None


def find(sub, pattern):
    hits = glob.glob(os.path.join(out_dir, f"{tag}_{sub}", "**", pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("void ", "").replace("nrl::", "")
    cut = name.find(">(")
    if cut > 0:
        name = name[: cut + 1]
    else:
        cut = name.find("(")
        if cut > 0:
            name = name[:cut]
    return name


summary = {}
stats = find("trace", "*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs']) / 1e6:.3f}",
                        f"{float(r['AverageNs']) / 1e3:.1f}", r["Percentage"]])

per = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "mfma"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if "nrl::" not in r["Kernel_Name"]:
            continue
        per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, ctrs in per.items():
    e = {c: sum(v) / len(v) for c, v in ctrs.items()}
    e["launches_sampled"] = max(len(v) for v in ctrs.values())
    if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
        # MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
        # reports exactly 1/2 of the bytes of a wide coalesced read -> doubled; WRITE_SIZE as is.
        e["hbm_read_bytes_corrected"] = 2.0 * e.get("FETCH_SIZE", 0.0) * 1024.0
        e["hbm_write_bytes"] = e.get("WRITE_SIZE", 0.0) * 1024.0
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"] > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines
        e["mfma_busy_frac"] = (e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (e["SQ_BUSY_CYCLES"] / 32.0)
    summary[k] = e
# counter traffic of one whole step: sum over kernels of (bytes per launch x launches per step); the PMC passes
# ran `steps_profiled` steps (env NRL_PROFILE_STEPS, default 23 = 3 warm-up + 10 timed + 10 instrumented)
steps_profiled = float(os.environ.get("NRL_PROFILE_STEPS", "23"))
step_bytes = 0.0
for k, e in summary.items():
    if "hbm_bytes_per_launch" in e:
        n_launch = max(len(per[k].get("FETCH_SIZE", [])), len(per[k].get("WRITE_SIZE", [])))
        e["launches_per_step"] = round(n_launch / steps_profiled, 2)
        step_bytes += e["hbm_bytes_per_launch"] * n_launch / steps_profiled
summary["_step"] = {"hbm_bytes_per_step": step_bytes, "steps_profiled": steps_profiled, "git_head": git_head, "build_id": build_id}
json.dump(summary, open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k, e in summary.items():
    if "news_fused_fwd_kernel" in k and "true" in k and "hbm_bytes_per_launch" in e:
        json.dump({"kernel": k, "hbm_bytes_per_launch": round(e["hbm_bytes_per_launch"]),
                   "hbm_bytes_per_step": round(step_bytes), "mfma_busy_frac": e.get("mfma_busy_frac"),
                   "raw_FETCH_SIZE_KiB": e.get("FETCH_SIZE"), "raw_WRITE_SIZE_KiB": e.get("WRITE_SIZE"),
                   "correction": "read = 2 x FETCH_SIZE x 1024 (gfx950 half-count), write = WRITE_SIZE x 1024",
                   "source": f"{tag}_pmc_summary.json", "git_head": git_head, "build_id": build_id},
                  open(os.path.join(out_dir, f"pmc_news_fused_fwd_{engine}.json"), "w"), indent=1)
    if "KCGather" in k and "hbm_bytes_per_launch" in e:
        json.dump({"kernel": k, "hbm_bytes_per_launch": round(e["hbm_bytes_per_launch"]),
                   "raw_FETCH_SIZE_KiB": e.get("FETCH_SIZE"), "raw_WRITE_SIZE_KiB": e.get("WRITE_SIZE"),
                   "correction": "read = 2 x FETCH_SIZE x 1024 (gfx950 half-count), write = WRITE_SIZE x 1024",
                   "source": f"{tag}_pmc_summary.json", "git_head": git_head, "build_id": build_id},
                  open(os.path.join(out_dir, f"pmc_in_proj_fwd_{engine}.json"), "w"), indent=1)
print(json.dumps({k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in e.items()}
                  for k, e in summary.items() if "gemm" in k or "fused" in k or k == "_step"}, indent=1)[:4000])
