"""Writes tests/golden/{tiny_9x9,rand_50}.pemtile: the tiled-format cache files of two matgen cases, produced from the
ORACLE's conversion through the format restatement (tests/cachefmt.py).  The GPU suite requires pem_tiled_save to
write exactly these bytes; the CPU suite reads the file back.  Run from the repo root: python tests/golden/make_cache_golden.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import cachefmt                      # noqa: E402
from matgen import cases             # noqa: E402
import __graft_entry__ as g          # noqa: E402

oracle = g.load_oracle()
for name in ("tiny_9x9", "rand_50"):
    rows, cols, I, J, V, _ = cases()[name]
    o = oracle.Tiled(rows, cols, I, J, V)
    with open(os.path.join(HERE, name + ".pemtile"), "wb") as f:
        f.write(cachefmt.cache_bytes(o.rows, o.cols, o.tile_keys, o.tile_nnz_ptr, o.rowcolidx, o.vals))
    print(f"wrote {name}.pemtile:", o.nnz, "entries in", o.ntiles, "tiles")
