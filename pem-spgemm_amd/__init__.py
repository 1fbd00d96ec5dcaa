"""pem-spgemm_amd -- host-side Python view of libpemspgemm_hip.so (ctypes over the C ABI).

The product is the HIP library behind include/pem_spgemm.h plus the C++ `pemspgemm` CLI
(host/); this module is the thin binding the tests, bench.py and the multi-GPU driver use.
There is NO CPU fallback: if the shared library is missing or no GPU is visible, every
compute entry point raises.  The directory name carries a hyphen (it mirrors the reference
repo's name), so import it through `__graft_entry__.load_package()`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpemspgemm_hip.so")

PEM_OK = 0
STATUS_NAMES = {0: "PEM_OK", -1: "PEM_E_INVALID", -2: "PEM_E_DUPLICATE", -3: "PEM_E_NOMEM", -4: "PEM_E_OVERFLOW",
                -5: "PEM_E_HIP", -6: "PEM_E_STATE", -7: "PEM_E_NODEVICE", -8: "PEM_E_IO", -9: "PEM_E_STALE"}

# enum pem_tiled_array / pem_cplan_array (include/pem_spgemm.h)
T_ARRAYS = dict(tile_keys=(0, np.int64), tile_nnz_ptr=(1, np.int32), masks=(2, np.uint16), rowptr=(3, np.uint8),
                rowcolidx=(4, np.uint8), vals=(5, np.float64), masks_t=(6, np.uint16), tile_rowptr=(7, np.int32),
                tile_colidx=(8, np.int32), tile_colptr=(9, np.int32), tile_rowidx=(10, np.int32), tile_offsets=(11, np.int32))
C_ARRAYS = dict(c_tile_rowptr=(0, np.int32), c_tile_rowidx=(1, np.int32), c_tile_colidx=(2, np.int32), pairs_offset=(3, np.int32),
                pairs_a=(4, np.int32), pairs_b=(5, np.int32), c_mask=(6, np.uint32), c_tile_nnz_ptr=(7, np.int32),
                c_rowptr=(8, np.uint8), c_rowcolidx=(9, np.uint8), c_vals=(10, np.float64))

# every symbol include/pem_spgemm.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "pem_last_error", "pem_version", "pem_ctx_create", "pem_ctx_create_on_stream", "pem_ctx_destroy", "pem_ctx_synchronize",
    "pem_tiled_from_coo", "pem_tiled_from_coo_device", "pem_tiled_from_csr", "pem_tiled_destroy", "pem_tiled_get_info",
    "pem_tiled_get_array", "pem_flop_count", "pem_cplan_create", "pem_cplan_destroy", "pem_spgemm_step1", "pem_spgemm_step2",
    "pem_spgemm_step3", "pem_spgemm", "pem_cplan_get_info", "pem_cplan_get_array", "pem_c_export_csr", "pem_c_export_csr_device",
    "pem_c_export_coo", "pem_split_tile_rows", "pem_tile_row_weights", "pem_get_timings", "pem_set_kernel_profiling", "pem_reset_kernel_stats",
    "pem_kernel_stats_count", "pem_kernel_stats_get", "pem_tiled_save", "pem_tiled_load",
    "pem_tiled_from_coo_f32", "pem_tiled_from_coo_device_f32", "pem_tiled_from_csr_f32", "pem_c_export_csr_f32",
    "pem_c_export_csr_device_f32", "pem_c_export_coo_f32", "pem_set_graph_replay",
    "pem_ctx_reserve", "pem_ctx_trim", "pem_ctx_memory_stats", "pem_cplan_set_option", "pem_cplan_get_option", "pem_debug_scan_i32",
    "pem_debug_refused_launch",
]

# enum pem_option (include/pem_spgemm.h): kernel variants / test hooks of a plan
OPTIONS = dict(prune=0, step1_global_sort=1, wide=2, warm=3, s3_band=4, s1_force_key64=5, s1_xlcap=6, export_rows=7, s1_serial=8, s3_decode=9, s1_xl_global=10, s3_epw=11, s3_idx64=12, s3_mark=13, s3_xcd=14, s1_segments=15, s2_transposed=16)


class PemError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status


class TiledInfo(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("nnz", C.c_int64), ("tile_rows", C.c_int32), ("tile_cols", C.c_int32),
                ("ntiles", C.c_int64), ("conv_ms", C.c_double), ("conv_tile_kernel_ms", C.c_double), ("value_bytes", C.c_int32),
                ("reserved", C.c_int32)]


class CPlanInfo(C.Structure):
    _fields_ = [("tile_row_begin", C.c_int32), ("tile_row_end", C.c_int32), ("row_begin", C.c_int32), ("row_end", C.c_int32),
                ("ntiles_c", C.c_int64), ("npairs", C.c_int64), ("nnz_c", C.c_int64), ("npairs_all", C.c_int64)]


class CacheKey(C.Structure):
    """pem_cache_key: what a cached tiling was made from (size + mtime of the source file, transpose flag)."""
    _fields_ = [("source_size", C.c_uint64), ("source_mtime_ns", C.c_int64), ("transpose", C.c_uint32), ("reserved", C.c_uint32)]

    @classmethod
    def of_file(cls, path, transpose=False):
        st = os.stat(path)
        return cls(st.st_size, st.st_mtime_ns, int(bool(transpose)), 0)


class MemoryStats(C.Structure):
    _fields_ = [("slab_bytes", C.c_int64), ("in_use_bytes", C.c_int64), ("peak_in_use_bytes", C.c_int64), ("largest_free_bytes", C.c_int64),
                ("driver_allocs", C.c_int64), ("block_allocs", C.c_int64)]


class Timings(C.Structure):
    _fields_ = [("step1_ms", C.c_double), ("step2_ms", C.c_double), ("step3_ms", C.c_double), ("spgemm_wall_ms", C.c_double),
                ("export_ms", C.c_double)]


def build(verbose=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], stdout=out)
    return LIB_PATH


_lib = None


def lib():
    """Load libpemspgemm_hip.so; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the product path has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.pem_last_error.restype = C.c_char_p
        L.pem_version.restype = C.c_char_p
        _lib = L
    return _lib


def _check(status):
    if status != PEM_OK:
        raise PemError(status, lib().pem_last_error().decode(errors="replace"))


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Context:
    """pem_ctx: one per GPU rank (replaces the stream/pool set-up of spgemm.cu:730-758, 808-817)."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        _check(lib().pem_ctx_create_on_stream(int(device), C.c_void_p(stream or 0), C.byref(self._h)))
        self.device = device

    def synchronize(self):
        _check(lib().pem_ctx_synchronize(self._h))

    def timings(self):
        t = Timings()
        _check(lib().pem_get_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Timings._fields_}

    def reserve(self, nbytes):
        """size the context's device memory arena ahead of time (one driver allocation; the reference sizes its rmm pools
        once, spgemm.cu:808-817)"""
        _check(lib().pem_ctx_reserve(self._h, C.c_int64(int(nbytes))))

    def trim(self):
        """return wholly free slabs of the arena to the driver"""
        _check(lib().pem_ctx_trim(self._h))

    def memory_stats(self):
        m = MemoryStats()
        _check(lib().pem_ctx_memory_stats(self._h, C.byref(m)))
        return {k: getattr(m, k) for k, _ in MemoryStats._fields_}

    def debug_scan(self, values, regime=0, in_place=True, stall_ticket=-1):
        """test hook: the device exclusive scan on a host array -> (prefix sums incl. the closing total, total)"""
        v = np.ascontiguousarray(values, dtype=np.int32)
        out = np.zeros(len(v) + 1, dtype=np.int32)
        total = C.c_int64()
        _check(lib().pem_debug_scan_i32(self._h, _p(v, C.c_int32), C.c_int64(len(v)), int(regime), int(bool(in_place)), int(stall_ticket),
                                        _p(out, C.c_int32), C.byref(total)))
        return out, total.value

    def debug_refused_launch(self):
        """test hook: a kernel launch the runtime refuses; the error surfaces at the next synchronising call"""
        _check(lib().pem_debug_refused_launch(self._h))

    def set_graph_replay(self, on):
        """repeat passes of CPlan.spgemm() replayed as one hipGraph (no per-step timings for those passes)"""
        _check(lib().pem_set_graph_replay(self._h, int(bool(on))))

    def set_kernel_profiling(self, on):
        _check(lib().pem_set_kernel_profiling(self._h, int(bool(on))))

    def reset_kernel_stats(self):
        _check(lib().pem_reset_kernel_stats(self._h))

    def kernel_stats(self):
        n = C.c_int()
        _check(lib().pem_kernel_stats_count(self._h, C.byref(n)))
        out = {}
        for i in range(n.value):
            name = C.create_string_buffer(128)
            calls, ms = C.c_int64(), C.c_double()
            _check(lib().pem_kernel_stats_get(self._h, i, name, 128, C.byref(calls), C.byref(ms)))
            out[name.value.decode()] = dict(calls=calls.value, total_ms=ms.value)
        return out

    def close(self):
        if self._h:
            lib().pem_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Tiled:
    """pem_tiled: a matrix in 16x16 tiled-CSR form on the device (spgemm.cu:832-1066)."""

    def __init__(self, ctx, handle):
        self.ctx, self._h = ctx, handle
        info = TiledInfo()
        _check(lib().pem_tiled_get_info(self._h, C.byref(info)))
        for k, _ in TiledInfo._fields_:
            setattr(self, k, getattr(info, k))

    @property
    def dtype(self):
        return np.dtype(np.float32 if self.value_bytes == 4 else np.float64)

    @classmethod
    def from_coo(cls, ctx, rows, cols, I, J, V, transpose=False, dtype=np.float64):
        """dtype float64 (the reference's ValueType) or float32 (values are rounded to float here)."""
        f32 = np.dtype(dtype) == np.float32
        I = np.ascontiguousarray(I, dtype=np.int32)
        J = np.ascontiguousarray(J, dtype=np.int32)
        V = np.ascontiguousarray(V, dtype=np.float32 if f32 else np.float64)
        if not (len(I) == len(J) == len(V)):
            raise ValueError("I, J, V lengths differ")
        h = C.c_void_p()
        fn = lib().pem_tiled_from_coo_f32 if f32 else lib().pem_tiled_from_coo
        _check(fn(ctx._h, int(rows), int(cols), C.c_int64(len(I)), _p(I, C.c_int32), _p(J, C.c_int32),
                  _p(V, C.c_float if f32 else C.c_double), int(bool(transpose)), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_coo_device(cls, ctx, rows, cols, nnz, dI, dJ, dV, transpose=False, dtype=np.float64):
        """dI/dJ/dV: device pointers (ints), e.g. torch tensors' data_ptr(); dV points at `dtype` values."""
        h = C.c_void_p()
        fn = lib().pem_tiled_from_coo_device_f32 if np.dtype(dtype) == np.float32 else lib().pem_tiled_from_coo_device
        _check(fn(ctx._h, int(rows), int(cols), C.c_int64(nnz), C.c_void_p(dI), C.c_void_p(dJ), C.c_void_p(dV), int(bool(transpose)),
                  C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_csr(cls, ctx, rows, cols, rowptr, colidx, V, dtype=np.float64):
        f32 = np.dtype(dtype) == np.float32
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        V = np.ascontiguousarray(V, dtype=np.float32 if f32 else np.float64)
        h = C.c_void_p()
        fn = lib().pem_tiled_from_csr_f32 if f32 else lib().pem_tiled_from_csr
        _check(fn(ctx._h, int(rows), int(cols), _p(rowptr, C.c_int32), _p(colidx, C.c_int32), _p(V, C.c_float if f32 else C.c_double),
                  C.byref(h)))
        return cls(ctx, h)

    def save(self, path, key=None):
        """Write the sorted tile payload to a cache file."""
        _check(lib().pem_tiled_save(self.ctx._h, self._h, os.fsencode(path), C.byref(key) if key is not None else None))

    @classmethod
    def load(cls, ctx, path, expect=None):
        """Rebuild a tiling from a cache file: upload, device-side validation, derived arrays -- no parse, no sort."""
        h = C.c_void_p()
        _check(lib().pem_tiled_load(ctx._h, os.fsencode(path), C.byref(expect) if expect is not None else None, C.byref(h)))
        return cls(ctx, h)

    def _count(self, name):
        T, nnz = self.ntiles, self.nnz
        return dict(tile_keys=T, tile_nnz_ptr=T + 1, masks=16 * T, rowptr=16 * T, rowcolidx=nnz, vals=nnz, masks_t=16 * T,
                    tile_rowptr=self.tile_rows + 1, tile_colidx=T, tile_colptr=self.tile_cols + 1, tile_rowidx=T, tile_offsets=T)[name]

    def array(self, name):
        which, dt = T_ARRAYS[name]
        if name == "vals":
            dt = self.dtype
        out = np.zeros(self._count(name), dtype=dt)
        _check(lib().pem_tiled_get_array(self.ctx._h, self._h, which, out.ctypes.data_as(C.c_void_p), C.c_int64(out.nbytes)))
        return out

    def close(self):
        if self._h:
            lib().pem_tiled_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def flop_count(ctx, A, B):
    f = C.c_uint64()
    _check(lib().pem_flop_count(ctx._h, A._h, B._h, C.byref(f)))
    return f.value


def split_tile_rows(ctx, A, B, nparts):
    b = np.zeros(nparts + 1, dtype=np.int32)
    _check(lib().pem_split_tile_rows(ctx._h, A._h, B._h, int(nparts), _p(b, C.c_int32)))
    return b


def tile_row_weights(ctx, A, B):
    """the per-tile-row weights pem_split_tile_rows balances (float64, one per tile row of A)"""
    w = np.zeros(A.tile_rows, dtype=np.float64)
    _check(lib().pem_tile_row_weights(ctx._h, A._h, B._h, _p(w, C.c_double)))
    return w


class CPlan:
    """pem_cplan: C = A*B over tile rows [tile_row_begin, tile_row_end) of A (steps 1-3)."""

    def __init__(self, ctx, A, B, tile_row_begin=0, tile_row_end=-1):
        self.ctx, self.A, self.B = ctx, A, B
        self._h = C.c_void_p()
        _check(lib().pem_cplan_create(ctx._h, A._h, B._h, int(tile_row_begin), int(tile_row_end), C.byref(self._h)))

    def set_option(self, name, value):
        """pem_cplan_set_option: kernel variants / test hooks (OPTIONS); the next pass is a full one"""
        _check(lib().pem_cplan_set_option(self._h, OPTIONS[name], C.c_int64(int(value))))

    def get_option(self, name):
        v = C.c_int64()
        _check(lib().pem_cplan_get_option(self._h, OPTIONS[name], C.byref(v)))
        return v.value

    def step1(self):
        _check(lib().pem_spgemm_step1(self.ctx._h, self._h))

    def step2(self):
        _check(lib().pem_spgemm_step2(self.ctx._h, self._h))

    def step3(self):
        _check(lib().pem_spgemm_step3(self.ctx._h, self._h))

    def spgemm(self):
        _check(lib().pem_spgemm(self.ctx._h, self._h))

    def info(self):
        i = CPlanInfo()
        _check(lib().pem_cplan_get_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in CPlanInfo._fields_}

    def array(self, name):
        which, dt = C_ARRAYS[name]
        i = self.info()
        TC, P, NZ, mt = i["ntiles_c"], i["npairs"], i["nnz_c"], i["tile_row_end"] - i["tile_row_begin"]
        cnt = dict(c_tile_rowptr=mt + 1, c_tile_rowidx=TC, c_tile_colidx=TC, pairs_offset=TC + 1, pairs_a=P, pairs_b=P, c_mask=8 * TC,
                   c_tile_nnz_ptr=TC + 1, c_rowptr=16 * TC, c_rowcolidx=NZ, c_vals=NZ)[name]
        if name == "c_vals":
            dt = self.A.dtype
        out = np.zeros(cnt, dtype=dt)
        _check(lib().pem_cplan_get_array(self.ctx._h, self._h, which, out.ctypes.data_as(C.c_void_p), C.c_int64(out.nbytes)))
        return out

    def export_csr(self):
        i = self.info()
        nrows, nz = i["row_end"] - i["row_begin"], i["nnz_c"]
        f32 = self.A.value_bytes == 4
        rp, ci, v = np.zeros(nrows + 1, np.int32), np.zeros(nz, np.int32), np.zeros(nz, self.A.dtype)
        n = C.c_int64()
        fn = lib().pem_c_export_csr_f32 if f32 else lib().pem_c_export_csr
        _check(fn(self.ctx._h, self._h, C.byref(n), _p(rp, C.c_int32), _p(ci, C.c_int32), _p(v, C.c_float if f32 else C.c_double)))
        return rp, ci, v

    def export_csr_device(self, d_rowptr, d_colidx, d_vals):
        """d_vals points at values of the plan's type (float64, or float32 for fp32 tilings)."""
        fn = lib().pem_c_export_csr_device_f32 if self.A.value_bytes == 4 else lib().pem_c_export_csr_device
        _check(fn(self.ctx._h, self._h, C.c_void_p(d_rowptr), C.c_void_p(d_colidx), C.c_void_p(d_vals)))

    def export_coo(self):
        nz = self.info()["nnz_c"]
        f32 = self.A.value_bytes == 4
        r, c, v = np.zeros(nz, np.int32), np.zeros(nz, np.int32), np.zeros(nz, self.A.dtype)
        n = C.c_int64()
        fn = lib().pem_c_export_coo_f32 if f32 else lib().pem_c_export_coo
        _check(fn(self.ctx._h, self._h, C.byref(n), _p(r, C.c_int32), _p(c, C.c_int32), _p(v, C.c_float if f32 else C.c_double)))
        return r, c, v

    def close(self):
        if self._h:
            lib().pem_cplan_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
