"""CPU: the tiled-format cache file.  The committed golden files are read with the format restatement
(tests/cachefmt.py) and must hold exactly the oracle's sorted tile payload; the restatement's writer must reproduce
them byte for byte.  (The GPU suite requires pem_tiled_save to write the same bytes and pem_tiled_load to read them.)"""
import os

import numpy as np
import pytest

import cachefmt
from matgen import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["tiny_9x9", "rand_50"])
def test_golden_cache_holds_the_oracle_payload(oracle, name):
    rows, cols, I, J, V, _ = cases()[name]
    o = oracle.Tiled(rows, cols, I, J, V)
    z = cachefmt.read_cache(os.path.join(GOLD, name + ".pemtile"))
    assert (z["rows"], z["cols"], z["nnz"], z["ntiles"], z["key"]) == (o.rows, o.cols, o.nnz, o.ntiles, (0, 0, 0))
    for arr in ("tile_keys", "tile_nnz_ptr", "rowcolidx", "vals"):
        assert np.array_equal(z[arr], getattr(o, arr)), arr
    again = cachefmt.cache_bytes(o.rows, o.cols, o.tile_keys, o.tile_nnz_ptr, o.rowcolidx, o.vals)
    assert again == open(os.path.join(GOLD, name + ".pemtile"), "rb").read()


def test_hash_is_sensitive_to_every_byte_and_to_length():
    base = bytes(range(64)) + b"tail!"
    h = cachefmt.hash64(base, cachefmt.SEED_PAYLOAD)
    assert h == cachefmt.hash64(base, cachefmt.SEED_PAYLOAD)
    assert h != cachefmt.hash64(base, cachefmt.SEED_HEADER)
    assert h != cachefmt.hash64(base + b"\0", cachefmt.SEED_PAYLOAD)      # zero padding of the tail is not invisible
    for i in range(len(base)):
        b = bytearray(base)
        b[i] ^= 1
        assert cachefmt.hash64(bytes(b), cachefmt.SEED_PAYLOAD) != h, i


def test_header_is_128_bytes_and_arrays_are_64_byte_aligned():
    raw = cachefmt.cache_bytes(20, 20, np.array([0, 1], np.int64), np.array([0, 1, 3], np.int32), np.array([0, 1, 17], np.uint8),
                               np.array([1.0, 2.0, 3.0]))
    assert len(raw) == 128 + 64 + 64 + 64 + 64
    assert raw[:8] == b"PEMTILE1"
