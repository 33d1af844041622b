"""Round 5 (a copy of exp/r04_summarise.py that also knows lif_pair_kernel).  Summaries of exp/r05_profiles.sh (gpurun_out/prof5/) -> gpurun_out/prof5/summary/ (copied into profiles/ by hand):
kernel-stats CSVs, per-kernel PMC means, the memory-side traffic of the reservoir kernel per launch
((2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes: gfx950 FETCH_SIZE counts half of the fetched bytes,
MI355X_MICROARCH.md HBM section) keyed `<cfg>_B<clips>_<kernel>` as bench.py looks them up, and the
instruction-mix table.

    python exp/r05_summarise.py gpurun_out/prof5
"""
import csv
import glob
import json
import os
import shutil
import sys

SRC = sys.argv[1]
DST = os.path.join(SRC, "summary")
os.makedirs(DST, exist_ok=True)
LIF = ("lif_dense_kernel", "lif_ring_kernel", "lif_pair_kernel", "lif_kernel<")
PRODUCT = ("gammatone_spikes_kernel", "gammatone_kernel", "spec_to_spikes_kernel", "mel_spikes_kernel", "mel_power_kernel", "power_to_db_kernel") + LIF


def newest(pattern):
    hits = glob.glob(os.path.join(SRC, pattern))
    return max(hits, key=os.path.getmtime) if hits else None


def last_json_line(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def short(name):
    for k in PRODUCT:
        if k in name:
            return k
    return None


for src, dst in (("bench_driver.json", "r05_bench_driver_first.json"), ("bench_default.json", "r05_bench_default.json"),
                 ("bench_serial.json", "r05_bench_serial.json")):
    line = last_json_line(os.path.join(SRC, src)) if os.path.exists(os.path.join(SRC, src)) else None
    if line:
        json.dump(line, open(os.path.join(DST, dst), "w"), indent=1)
for d, dst in (("stats_driver", "r05_kernel_stats_driver.csv"), ("stats", "r05_kernel_stats.csv"),
               ("stats_serial", "r05_kernel_stats_serial.csv")):
    f = newest(f"{d}/*/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(DST, dst))

traffic = {"_how": "exp/r05_profiles.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum in "
                   "separate passes over `python3 bench.py --config <cfg> --batch <B> --kernel <k> --stage reservoir --steps 2 "
                   "--warmup 1 --streams 1`; per-launch mean over the reservoir-kernel dispatches; bytes = (2*FETCH_SIZE + "
                   "WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts half of the fetched bytes; Infinity-Cache hits are included: it "
                   "is the L2's memory-side traffic)"}
rows_out = []
for d in sorted(glob.glob(os.path.join(SRC, "pmc_*_fetch"))):
    stem = os.path.basename(d)[len("pmc_"):-len("_fetch")]           # cfg5_B512_auto
    vals, names, dur = {}, set(), []
    for part in ("fetch", "write", "l2"):
        f = newest(f"pmc_{stem}_{part}/*/*_counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in LIF):
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                names.add(r["Kernel_Name"])
                if part == "fetch":
                    dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        continue
    mean = {k: sum(v) / len(v) for k, v in vals.items()}
    line = last_json_line(os.path.join(SRC, f"pmc_{stem}_fetch.json"))
    kernel = line["config"]["reservoir_kernel"] if line else stem.split("_")[-1]
    cfg, bsz = stem.split("_")[0], stem.split("_")[1]
    key = f"{cfg}_{bsz}_{kernel}"
    nbytes = int(round((2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024))
    traffic[key] = nbytes
    traffic[key + "_detail"] = {
        "kernel": sorted(names)[0], "launches": len(vals["FETCH_SIZE"]), "fetch_kb": mean["FETCH_SIZE"],
        "write_kb": mean["WRITE_SIZE"], "kernel_ms_in_the_counter_pass": sum(dur) / len(dur) if dur else None,
        "memory_side_gbs": (nbytes / (sum(dur) / len(dur) * 1e-3) / 1e9) if dur else None,
        "l2_hit_rate": (mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])) if "TCC_HIT_sum" in mean else None,
        "algorithmic_bytes": (line["roofline"]["bytes_per_clip"] * line["config"]["clips_per_gpu"]) if line and "roofline" in line else None,
    }
    det = traffic[key + "_detail"]
    if det["algorithmic_bytes"]:
        det["traffic_over_algorithmic"] = nbytes / det["algorithmic_bytes"]
json.dump(traffic, open(os.path.join(DST, "lif_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))

# instruction mix
with open(os.path.join(DST, "r05_sq_counters.csv"), "w", newline="") as out:
    wr = csv.writer(out)
    wr.writerow(["config", "kernel", "full_name", "counter", "mean_per_launch", "launches"])
    for cfg in ("cfg2", "cfg4"):
        acc = {}
        for d in sorted(glob.glob(os.path.join(SRC, f"sq_{cfg}_*"))):
            if not os.path.isdir(d):
                continue
            f = newest(os.path.join(os.path.basename(d), "*", "*_counter_collection.csv"))
            if not f:
                continue
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    acc.setdefault((k, r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, full, cn), v in sorted(acc.items()):
            wr.writerow([cfg, k, full, cn, sum(v) / len(v), len(v)])
print(open(os.path.join(DST, "r05_sq_counters.csv")).read()[:3000])
for f in ("r05_kernel_stats_driver.csv", "r05_kernel_stats.csv", "r05_kernel_stats_serial.csv"):
    p = os.path.join(DST, f)
    if os.path.exists(p):
        print("==", f)
        for r in csv.DictReader(open(p)):
            if short(r["Name"]):
                print(r["Name"][:80], r["Calls"], "avg_ns", r["AverageNs"], "pct", r["Percentage"])
