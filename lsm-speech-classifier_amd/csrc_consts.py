"""Constants of the HIP sources that host-side Python needs to know (kept in step by tests/test_pair_layout.py, which reads
the header)."""
PAIR_DUMP_BYTES = 256       # csrc/lif_pair.h: LDS bytes in front of the accumulators of the pair-block ring kernel
