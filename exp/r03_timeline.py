"""Timeline of the driver's command from a rocprofv3 kernel trace (round 3: one front-end launch per step): when do
the front-end and reservoir launches of the timed steps start and end, how many run at a time, where is the chip idle?
usage: r03_timeline.py <kernel_trace.csv> [n_steps]"""
import csv, sys
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        kind = ("fe" if "gammatone" in name else "sp" if "spec_to_spikes" in name else "lif" if "lif_" in name else None)
        if kind:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
rows.sort()
fe = [r for r in rows if r[2] == "fe"]
lif = [r for r in rows if r[2] == "lif"]
print("front-end launches:", len(fe), "reservoir launches:", len(lif))
# runs of front-end launches: a new run starts after >= 1 ms without any front end running
runs, cur, busy_until = [], [], 0
for i, g in enumerate(fe):
    if cur and g[0] - busy_until > 1_000_000:
        runs.append(cur); cur = []
    cur.append(i); busy_until = max(busy_until, g[1])
runs.append(cur)
for run in runs:
    if len(run) != n_steps:
        continue
    a = fe[run[0]][0]
    b = max(fe[i][1] for i in run)
    lifs = [l for l in lif if a <= l[0] <= b + 5_000_000][:n_steps]
    end = max([b] + [l[1] for l in lifs])
    print(f"\nrun of {len(run)} front ends: first start -> last front-end end {(b - a) / 1e6:.3f} ms; -> last reservoir end {(end - a) / 1e6:.3f} ms "
          f"= {(end - a) / 1e6 / n_steps:.4f} ms/step")
    ev = sorted([(fe[i][0], +1, "fe") for i in run] + [(fe[i][1], -1, "fe") for i in run] +
                [(l[0], +1, "lif") for l in lifs] + [(l[1], -1, "lif") for l in lifs])
    # time-weighted concurrency
    cnt = {"fe": 0, "lif": 0}; last = a; acc = {}
    for t, d, k in ev:
        key = (cnt["fe"], cnt["lif"])
        acc[key] = acc.get(key, 0) + (t - last)
        cnt[k] += d; last = t
    print("  time with (front ends, reservoirs) in flight:")
    for key in sorted(acc):
        if acc[key] > 20_000:
            print(f"    {key}: {acc[key] / 1e6:.3f} ms")
    for k, i in enumerate(run):
        g = fe[i]
        print(f"  fe  {k:2d}: start {(g[0] - a) / 1e6:7.3f}  end {(g[1] - a) / 1e6:7.3f}  dur {(g[1] - g[0]) / 1e6:6.3f}")
    for k, l in enumerate(lifs):
        print(f"  lif {k:2d}: start {(l[0] - a) / 1e6:7.3f}  end {(l[1] - a) / 1e6:7.3f}  dur {(l[1] - l[0]) / 1e6:6.3f}")
