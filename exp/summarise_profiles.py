"""Copy the summaries of exp/collect_profiles.sh (gpurun_out/prof/) into profiles/ as r<NN>_* files and
derive profiles/lif_traffic.json (HBM bytes per LIF launch) from the two size counters.

    python exp/summarise_profiles.py [round_number]
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
PRODUCT = ("gammatone_kernel", "spec_to_spikes_kernel", "lif_dense_kernel", "lif_kernel")


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern))
    if not hits:
        raise SystemExit(f"no file for {pattern}")
    return max(hits, key=os.path.getmtime)      # gpurun merges into the local copy: take the newest run


def last_json_line(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def main():
    rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    pre = f"r{rnd:02d}_"
    for src, dst in (("bench_default.json", "bench_default.json"), ("bench_serial.json", "bench_serial.json")):
        line = last_json_line(os.path.join(SRC, src))
        json.dump(line, open(os.path.join(DST, pre + dst), "w"), indent=1)
    shutil.copy(one("stats/*/*_kernel_stats.csv"), os.path.join(DST, pre + "kernel_stats.csv"))
    shutil.copy(one("stats_serial/*/*_kernel_stats.csv"), os.path.join(DST, pre + "kernel_stats_serial.csv"))

    # PMC passes: keep the rows of the product kernels, one file
    rows, header = [], None
    per_kernel = {}
    for name in ("fetch", "write", "l2"):
        with open(one(f"pmc_{name}/*/*_counter_collection.csv")) as f:
            rd = csv.DictReader(f)
            header = header or rd.fieldnames
            for r in rd:
                if any(k in r["Kernel_Name"] for k in PRODUCT):
                    rows.append(r)
                    key = "lif" if "lif_" in r["Kernel_Name"] else None
                    if key:
                        per_kernel.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
            "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    keep = [k for k in keep if k in header]
    with open(os.path.join(DST, pre + "pmc_counters.csv"), "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=keep, extrasaction="ignore")
        wr.writeheader()
        wr.writerows(rows)

    mean = {k: sum(v) / len(v) for k, v in per_kernel.items()}
    bench = last_json_line(os.path.join(SRC, "pmc_fetch.json"))
    key = f"{bench['config']['workload'].split(':')[0]}_B{bench['config']['clips_per_gpu']}"
    fetch_kb, write_kb = mean["FETCH_SIZE"], mean["WRITE_SIZE"]
    traffic = {
        key: int(round((2 * fetch_kb + write_kb) * 1024)),
        "_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py "
                "--steps 4 --warmup 1 --no-cpu-baseline --streams 1` (exp/collect_profiles.sh); per-launch mean "
                "over the LIF kernel dispatches; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE "
                "counts half of the fetched bytes, MI355X_MICROARCH.md HBM section; checked against the known "
                "minimum: 13.1 MB raster + 8 XCD copies of the 4.1 MB dense weight table = 45.9 MB; WRITE_SIZE "
                "= 2000 KB = 256 x 8000 B of features exactly)",
        key + "_fetch_kb": fetch_kb, key + "_write_kb": write_kb,
    }
    if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean:
        traffic[key + "_l2_hit_rate"] = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
    # _keep_large: the large-configuration entries come from exp/big_traffic.sh (profiles/r01_big_traffic.json)
    old_path = os.path.join(DST, "lif_traffic.json")
    if os.path.exists(old_path):
        for k, v in json.load(open(old_path)).items():
            if k.startswith(("cfg4_", "cfg5_", "_how_large")):
                traffic.setdefault(k, v)
    json.dump(traffic, open(os.path.join(DST, "lif_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
    for f in ("kernel_stats.csv", "kernel_stats_serial.csv"):
        print("==", f)
        with open(os.path.join(DST, pre + f)) as fh:
            for r in csv.DictReader(fh):
                if any(k in r["Name"] for k in PRODUCT):
                    print(r["Name"][:70], r["Calls"], "avg_ns", r["AverageNs"], "pct", r["Percentage"])


if __name__ == "__main__":
    main()
