"""Timeline of the timed 20-step burst of `bench.py --steps 20` from a rocprofv3 kernel trace.

usage: python exp/r04_timeline.py <kernel_trace.csv> [n_steps]
Kernels are grouped into busy periods (a new period starts when nothing is in flight for > 50 us); the last
period holding exactly n_steps reservoir launches and n_steps front ends is the timed region of the headline pass."""
import csv
import sys


def kind(n):
    if "gammatone_spikes_kernel" in n:
        return "fe"
    if "lif_" in n:
        return "lif"
    return None


def main(path, n_steps=20):
    rows = []
    for r in csv.DictReader(open(path)):
        k = kind(r["Kernel_Name"])
        if k:
            wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else 0
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, wg))
    rows.sort()
    periods, cur, busy_until = [], [], None
    for r in rows:
        if busy_until is not None and r[0] - busy_until > 50_000:
            periods.append(cur); cur = []
        cur.append(r)
        busy_until = r[1] if busy_until is None else max(busy_until, r[1])
    periods.append(cur)
    full = [p for p in periods if sum(r[2] == "fe" for r in p) == n_steps and sum(r[2] == "lif" for r in p) == n_steps]
    print(f"{len(periods)} busy periods, {len(full)} with {n_steps} front ends + {n_steps} reservoir launches; the last one:")
    p = full[-1]
    a = p[0][0]; end = max(r[1] for r in p)
    print(f"first front end start -> last reservoir end {(end - a) / 1e6:.3f} ms = {(end - a) / 1e6 / n_steps:.4f} ms/step")
    ev = sorted([(r[0], +1, r[2]) for r in p] + [(r[1], -1, r[2]) for r in p])
    cnt = {"fe": 0, "lif": 0}; last = a; acc = {}
    for t, d, k in ev:
        key = (cnt["fe"], cnt["lif"])
        acc[key] = acc.get(key, 0) + (t - last)
        cnt[k] += d; last = t
    print("time with (front ends, reservoir launches) in flight:")
    for key in sorted(acc):
        if acc[key] > 20_000:
            print(f"  {key}: {acc[key] / 1e6:.3f} ms")
    idle = sum(v for k, v in acc.items() if k == (0, 0))
    print(f"time with nothing in flight inside the burst: {idle / 1e6:.3f} ms")
    fes = [r for r in p if r[2] == "fe"]; lifs = [r for r in p if r[2] == "lif"]
    print(f"front-end duration in the burst: mean {sum(r[1] - r[0] for r in fes) / len(fes) / 1e6:.3f} ms, "
          f"reservoir {sum(r[1] - r[0] for r in lifs) / len(lifs) / 1e6:.3f} ms")
    lif_end = sorted(r[1] for r in lifs)
    print("completion times of the steps (ms): " + " ".join(f"{(t - a) / 1e6:.2f}" for t in lif_end))
    d = [(lif_end[i] - lif_end[i - 1]) / 1e6 for i in range(1, len(lif_end))]
    print(f"first completion {(lif_end[0] - a) / 1e6:.3f} ms; mean spacing after it {sum(d) / len(d):.4f} ms")
    for r in p:
        print(f"  {r[2]:3s} wgs {r[3]:5d}  start {(r[0] - a) / 1e6:7.3f}  end {(r[1] - a) / 1e6:7.3f}  dur {(r[1] - r[0]) / 1e6:6.3f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20)
