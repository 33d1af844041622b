"""File 1 on disk: the reference's schema and a bit-packed variant.

Reference schema (/root/reference/create_dataset.py:168-176, read back at
/root/reference/extract_lsm_features.py:62-66): ``X_spikes`` uint8 (n, C, T) of 0/1 and ``y_labels``
int32 (n,), written with ``np.savez_compressed``.

Packed schema, 8x fewer raster bytes before compression and over PCIe: ``X_spikes_packed`` uint8
(n, C, ceil(T/8)) with time step 8q+k in bit k of byte q, ``time_steps`` int32 scalar T, ``y_labels``
as above.  ``load`` accepts either and always hands back the reference's arrays, so every consumer
of File 1 keeps working; ``load_packed`` keeps the rasters packed for callers that upload them as
they are and unpack on the GPU (``SNN.run_batch(..., packed_time_steps=T)``).

This module is file-format code (like ``np.load`` itself); the per-clip arithmetic of the pipeline
is nowhere in here.
"""
from __future__ import annotations

import numpy as np

PACKED_KEY = "X_spikes_packed"
DENSE_KEY = "X_spikes"


def pack_host(x: np.ndarray) -> np.ndarray:
    """(..., T) -> (..., ceil(T/8)) uint8, little bit order; any non-zero byte is a spike."""
    return np.packbits(np.asarray(x) != 0, axis=-1, bitorder="little")


def unpack_host(p: np.ndarray, time_steps: int) -> np.ndarray:
    p = np.asarray(p, dtype=np.uint8)
    if p.shape[-1] != (time_steps + 7) // 8:
        raise ValueError(f"packed rasters with last dimension {p.shape[-1]} cannot hold {time_steps} steps")
    return np.unpackbits(p, axis=-1, count=time_steps, bitorder="little")


def save(filename, X_spikes=None, y_labels=None, *, packed=None, time_steps=None) -> None:
    """Write File 1.  Give ``X_spikes`` (n, C, T) for the reference schema, or ``packed`` +
    ``time_steps`` for the packed one."""
    y = np.asarray(y_labels, dtype=np.int32)
    if packed is not None:
        if time_steps is None:
            raise ValueError("the packed schema needs time_steps")
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        if packed.ndim != 3 or packed.shape[-1] != (int(time_steps) + 7) // 8 or len(packed) != len(y):
            raise ValueError("packed rasters must be (n, C, ceil(time_steps/8)) with one label per clip")
        np.savez_compressed(filename, **{PACKED_KEY: packed}, time_steps=np.int32(time_steps), y_labels=y)
    else:
        X = np.ascontiguousarray(X_spikes, dtype=np.uint8)
        if X.ndim != 3 or len(X) != len(y):
            raise ValueError("X_spikes must be (n, C, T) with one label per clip")
        np.savez_compressed(filename, **{DENSE_KEY: X}, y_labels=y)


def load_packed(filename):
    """-> (packed (n, C, ceil(T/8)) uint8, T, y_labels); packs a reference-schema file on the host."""
    with np.load(filename) as data:
        y = data["y_labels"]
        if PACKED_KEY in data.files:
            return data[PACKED_KEY], int(data["time_steps"]), y
        X = data[DENSE_KEY]
        return pack_host(X), int(X.shape[-1]), y


def load(filename):
    """-> (X_spikes (n, C, T) uint8 of 0/1, y_labels), whichever schema the file uses."""
    with np.load(filename) as data:
        y = data["y_labels"]
        if PACKED_KEY in data.files:
            return unpack_host(data[PACKED_KEY], int(data["time_steps"])), y
        return data[DENSE_KEY], y
