"""CPU oracle for the LSM hot path — TEST INFRASTRUCTURE ONLY.

Two independent restatements of the same arithmetic: ``ref_numpy`` (literal NumPy/SciPy, slow)
and ``cport`` (plain C via ctypes, fast; also the timed CPU baseline).  Nothing under the
product package or the root scripts may import this package.
"""
