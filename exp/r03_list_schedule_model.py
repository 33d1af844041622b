"""Round 3: a reservoir launch as a list schedule.  Input: profiles/r03_clip_cycles_<cfg>_B<n>.npy = [input spike count, measured cycles]
per clip (written by exp/r03_ring_phases.py from the diagnostic build).  Workgroups are handed to the first free slot in launch order;
prints the makespan of several orders against work / slots.  CPU only.

    python3 exp/r03_list_schedule_model.py
"""
import heapq, os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def makespan(order, dur, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for c in order:
        heapq.heappush(h, heapq.heappop(h) + dur[c])
    return max(h)


def ffd_order(est, slots):
    """First-fit-decreasing packing of the estimates into `slots` bins (smallest capacity that needs no more bins, by bisection),
    turned into a launch order: clips sorted by the start time the packing gives them."""
    idx = np.argsort(-est, kind="stable")
    lo = max(est.max(), est.sum() / slots)
    hi = 2 * lo

    def pack(cap):
        bins, loads = [], []
        for j in idx:
            for b in range(len(bins)):
                if loads[b] + est[j] <= cap:
                    bins[b].append(j)
                    loads[b] += est[j]
                    break
            else:
                bins.append([j])
                loads.append(est[j])
                if len(bins) > slots:
                    return None
        return bins

    best = None
    for _ in range(30):
        mid = (lo + hi) / 2
        b = pack(mid)
        if b is None:
            lo = mid
        else:
            best, hi = b, mid
    starts = []
    for b in best:
        t = 0.0
        for j in b:
            starts.append((t, -est[j], j))
            t += est[j]
    starts.sort()
    return np.array([j for _, _, j in starts])


if __name__ == "__main__":
    for name, slots in (("r03_clip_cycles_cfg4_B1024.npy", 512), ("r03_clip_cycles_cfg5_B512.npy", 256)):
        keys, dur = np.load(os.path.join(ROOT, "profiles", name))
        n = len(dur)
        print(f"{name}: {n} clips on {slots} slots; work / slots {dur.sum() / slots / 1e6:.2f} Mcycles, longest clip {dur.max() / 1e6:.2f}")
        print(f"  batch order                         {makespan(range(n), dur, slots) / 1e6:6.2f}")
        print(f"  most input spikes first (shipped)   {makespan(np.argsort(-keys, kind='stable'), dur, slots) / 1e6:6.2f}")
        print(f"  longest clip first (true cycles)    {makespan(np.argsort(-dur, kind='stable'), dur, slots) / 1e6:6.2f}")
        print(f"  first-fit-decreasing on the counts  {makespan(ffd_order(keys.copy(), slots), dur, slots) / 1e6:6.2f}")
        print(f"  first-fit-decreasing on true cycles {makespan(ffd_order(dur.copy(), slots), dur, slots) / 1e6:6.2f}")
