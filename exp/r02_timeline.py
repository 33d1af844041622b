"""Timeline of the driver's command from a rocprofv3 kernel trace: when do the kernels of the 20 timed steps start and
end, how many filterbank launches run at a time, where are the gaps?  usage: r02_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        kind = "gt" if "gammatone" in name else "sp" if "spec_to_spikes" in name else "lif" if "lif_" in name else None
        if kind:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, name[:40]))
rows.sort()
gt = [r for r in rows if r[2] == "gt"]
lif = [r for r in rows if r[2] == "lif"]
# the timed region = the last 20 filterbank launches before the 5 idle-GPU reference launches at the end; find it as
# the longest run of filterbank launches whose successive starts are < 3 ms apart
t_end = max(r[1] for r in rows)
print("filterbank launches:", len(gt), "reservoir launches:", len(lif))
starts = [g[0] for g in gt]
runs, cur = [], [0]
for i in range(1, len(gt)):
    if starts[i] - gt[i - 1][1] < 1_000_000:        # next starts within 1 ms of the previous end
        cur.append(i)
    else:
        runs.append(cur); cur = [i]
runs.append(cur)
for run in runs:
    if len(run) < 15:
        continue
    a, b = gt[run[0]][0], max(gt[i][1] for i in run)
    lifs = [l for l in lif if a <= l[0] <= b + 3_000_000]
    end = max([b] + [l[1] for l in lifs])
    print(f"\nrun of {len(run)} filterbank launches: {(b - a) / 1e6:.3f} ms from first start to last end; "
          f"with the reservoir launches behind it {(end - a) / 1e6:.3f} ms")
    for k, i in enumerate(run):
        g = gt[i]
        print(f"  gt {k:2d}: start {(g[0] - a) / 1e6:7.3f}  end {(g[1] - a) / 1e6:7.3f}  dur {(g[1] - g[0]) / 1e6:6.3f}")
    for k, l in enumerate(lifs):
        print(f"  lif {k:2d}: start {(l[0] - a) / 1e6:7.3f}  end {(l[1] - a) / 1e6:7.3f}  dur {(l[1] - l[0]) / 1e6:6.3f}  {l[3]}")
