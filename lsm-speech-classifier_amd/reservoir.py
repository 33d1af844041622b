"""Reservoir construction (host side, one-off per run) — SPEC.md §2.

Replaces what the reference obtains from ``SimulationParams(...)`` / ``SNN(simulation_params=...)``
(/root/reference/extract_lsm_features.py:164-175,185-188).  The
third-party package that implements those two types (snn_reservoir_py==2.0.0) is not available,
so every clause here is this build's frozen specification (SPEC.md tags each clause
P = pinned by a reference call site, I = inferred, B = build choice).

The builder is plain NumPy and fully determined by ``(params, n_channels, seed)``: the same
arrays feed the HIP kernels (through ``snn.SNN``) and, in the tests, the CPU oracle.
"""
from __future__ import annotations

import numpy as np

# SPEC.md §2.5 — input coupling constants (B).
W_IN_SCALE = 0.05          # w_in = W_IN_SCALE * membrane_threshold (0.1 at theta = 2.0)
# SPEC.md §4 — burst feature: an inter-spike interval <= this many steps counts as a burst (B).
BURST_ISI_MAX = 5

DEFAULT_SEED = 42           # /root/reference/extract_lsm_features.py:30 seeds NumPy with 42 (I)


class SimulationParams:
    """Parameter bag with the field names the reference passes and mutates
    (/root/reference/extract_lsm_features.py:164-175 construct, :185-186 mutate, :50,55-56 read)."""

    def __init__(self, num_neurons, mean_weight=0.0, num_output_neurons=None,
                 membrane_threshold=2.0, leak_coefficient=0.01, refractory_period=2,
                 small_world_graph_p=0.1, small_world_graph_k=None, input_spike_times=None,
                 leak_variance_divisor=None, weight_variance=10.0, seed=DEFAULT_SEED):
        self.num_neurons = int(num_neurons)
        self.mean_weight = float(mean_weight)
        self.weight_variance = float(weight_variance)
        self.num_output_neurons = int(num_output_neurons if num_output_neurons is not None
                                      else num_neurons)
        self.membrane_threshold = float(membrane_threshold)
        self.leak_coefficient = float(leak_coefficient)
        self.refractory_period = int(refractory_period)
        self.small_world_graph_p = float(small_world_graph_p)
        self.small_world_graph_k = int(small_world_graph_k if small_world_graph_k is not None
                                       else int(0.10 * num_neurons * 2))
        self.input_spike_times = input_spike_times
        self.leak_variance_divisor = leak_variance_divisor
        self.seed = int(seed)


def input_fanout(num_neurons: int, n_channels: int) -> int:
    """Targets per input channel: floor(N/C + 1/2), at least 1 (SPEC.md §2.5)."""
    return max(1, (2 * num_neurons + n_channels) // (2 * n_channels))


def small_world_edges(n: int, k: int, p: float, rs: np.random.RandomState) -> np.ndarray:
    """Watts-Strogatz graph as a symmetric boolean adjacency matrix (SPEC.md §2.1).

    Ring lattice with k//2 neighbours on each side; every lattice edge (u, u+j) is visited in
    the order j = 1..k//2 (outer), u = 0..n-1 (inner) and rewired with probability p to a
    uniformly drawn node w that is neither u nor already adjacent to u.
    """
    half = k // 2
    if half < 1 or k >= n:
        raise ValueError(f"small_world_graph_k={k} must satisfy 2 <= k < num_neurons={n}")
    adj = np.zeros((n, n), dtype=bool)
    nodes = np.arange(n)
    for j in range(1, half + 1):
        v = (nodes + j) % n
        adj[nodes, v] = True
        adj[v, nodes] = True
    decide = rs.random_sample((half, n)) < p          # one block of draws, order (j, u)
    deg = adj.sum(axis=1)
    for j in range(1, half + 1):
        for u in np.nonzero(decide[j - 1])[0]:
            v = (u + j) % n
            if deg[u] >= n - 1:
                continue                               # u is saturated: keep the edge
            w = int(rs.randint(0, n))
            while w == u or adj[u, w]:
                w = int(rs.randint(0, n))
            adj[u, v] = adj[v, u] = False
            adj[u, w] = adj[w, u] = True
            deg[v] -= 1
            deg[w] += 1
    return adj


class Reservoir:
    """Immutable wiring of one reservoir; every array is C-contiguous NumPy on the host."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def nnz(self) -> int:
        return int(self.csr_ptr[-1])

    def csr_bytes(self) -> int:
        """|W| as SURVEY.md §8(d) counts it: fp32 value + int32 index per synapse."""
        return self.nnz * 8


def build_reservoir(params: SimulationParams, n_channels: int, seed: int | None = None) -> Reservoir:
    """Build the wiring once (SPEC.md §2).  RNG draw order: graph -> weights -> leaks ->
    input map -> output set, all from one ``RandomState(seed)`` stream."""
    n = params.num_neurons
    k = params.small_world_graph_k
    seed = params.seed if seed is None else int(seed)
    rs = np.random.RandomState(seed)

    adj = small_world_edges(n, k, params.small_world_graph_p, rs)
    # CSR by postsynaptic row i, presynaptic j ascending (np.nonzero is row-major ascending).
    post, pre = np.nonzero(adj)
    csr_ptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(post, minlength=n), out=csr_ptr[1:])
    nnz = int(csr_ptr[-1])

    mean_w = params.mean_weight
    sd_w = abs(mean_w) / params.weight_variance if params.weight_variance else 0.0
    csr_w = rs.normal(mean_w, sd_w, size=nnz).astype(np.float32)

    lam = params.leak_coefficient
    if params.leak_variance_divisor:
        leak = np.clip(rs.normal(lam, lam / params.leak_variance_divisor, size=n), 0.0, 1.0)
        leak = leak.astype(np.float32)
    else:
        leak = np.full(n, lam, dtype=np.float32)

    d_in = input_fanout(n, n_channels)
    in_tgt = np.empty((n_channels, d_in), dtype=np.int32)
    for c in range(n_channels):
        in_tgt[c] = np.sort(rs.choice(n, size=d_in, replace=False))

    n_out = params.num_output_neurons
    if not 1 <= n_out <= n:
        raise ValueError(f"num_output_neurons={n_out} must be in [1, {n}]")
    out_idx = np.sort(rs.choice(n, size=n_out, replace=False)).astype(np.int32)

    # CSC view (by presynaptic j, postsynaptic i ascending) for the event-driven device kernel:
    # the same synapses and weights, reordered.
    order = np.lexsort((post, pre))
    csc_post = post[order].astype(np.int32)
    csc_w = csr_w[order]
    csc_ptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(pre, minlength=n), out=csc_ptr[1:])

    # Input map by postsynaptic neuron (channels ascending) for the gather-form oracle.
    flat_c = np.repeat(np.arange(n_channels, dtype=np.int32), d_in)
    flat_i = in_tgt.reshape(-1)
    o2 = np.lexsort((flat_c, flat_i))
    in_chan = flat_c[o2].astype(np.int32)
    in_ptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(flat_i, minlength=n), out=in_ptr[1:])

    return Reservoir(
        num_neurons=n, n_channels=int(n_channels), seed=seed,
        theta=np.float32(params.membrane_threshold),
        refractory_period=int(params.refractory_period),
        w_in=np.float32(W_IN_SCALE * params.membrane_threshold),
        burst_isi_max=BURST_ISI_MAX,
        csr_ptr=csr_ptr, csr_pre=pre.astype(np.int32), csr_w=csr_w,
        csc_ptr=csc_ptr, csc_post=csc_post, csc_w=np.ascontiguousarray(csc_w),
        leak=leak,
        in_fanout=d_in, in_tgt=in_tgt, in_ptr=in_ptr, in_chan=in_chan,
        out_idx=out_idx,
    )
