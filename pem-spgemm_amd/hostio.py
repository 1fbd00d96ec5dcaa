"""ctypes view of libpemhost.so (include/pem_host.h): the pemspgemm tool's Matrix-Market
reader and result/CSV writers.  Host-only; no GPU needed."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpemhost.so")
MGPU_LIB_PATH = os.path.join(_HERE, "libpemmgpu.so")   # include/pem_mgpu.h: contexts of one process over several GPUs + RCCL gather
CLI_PATH = os.path.join(_HERE, "pemspgemm")

HOST_SYMBOLS = ["pem_mm_read", "pem_coo_free", "pem_host_last_error", "pem_write_result_files", "pem_csv_append", "pem_write_mtx_csr",
                "pem_standin_generate", "pem_standin_names"]


class _Coo(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("nnz", C.c_int64), ("I", C.POINTER(C.c_int32)), ("J", C.POINTER(C.c_int32)),
                ("V", C.POINTER(C.c_double)), ("symmetric", C.c_int32), ("field", C.c_int32)]


class CsvRecord(C.Structure):
    _fields_ = [("matrix", C.c_char_p), ("flop", C.c_uint64), ("c_nnz", C.c_int64), ("compression_ratio", C.c_double),
                ("a_conversion_kernel_ms", C.c_double), ("b_conversion_kernel_ms", C.c_double), ("total_conversion_ms", C.c_double),
                ("step1_ms", C.c_double), ("step2_ms", C.c_double), ("step3_ms", C.c_double), ("spgemm_ms", C.c_double),
                ("kernel_ms", C.c_double), ("malloc_ms", C.c_double), ("gflops", C.c_double)]


MGPU_SYMBOLS = ["pem_mgpu_create", "pem_mgpu_destroy", "pem_mgpu_size", "pem_mgpu_ctx", "pem_mgpu_slice_offsets", "pem_mgpu_rebase_rowptr",
                "pem_mgpu_gather_csr", "pem_mgpu_recut_bounds", "pem_mgpu_spgemm_gather_chunked"]

_lib = None
_mgpu = None


def mgpu_lib():
    """libpemmgpu.so (links RCCL); the C++ tool's `--gpus N` path.  Loads without a GPU; creating contexts needs one."""
    global _mgpu
    if _mgpu is None:
        if not os.path.exists(MGPU_LIB_PATH):
            raise ImportError(f"{MGPU_LIB_PATH} is missing: run __graft_entry__.build()")
        C.CDLL(os.path.join(_HERE, "libpemspgemm_hip.so"), mode=C.RTLD_GLOBAL)
        _mgpu = C.CDLL(MGPU_LIB_PATH)
    return _mgpu


def mgpu_recut_bounds(weights, bounds, ms, fixed_ms=0.0):
    """pem_mgpu_recut_bounds: row-block boundaries re-cut from measured per-rank pass times (host arithmetic of the C++ `--gpus N` path)"""
    w = np.ascontiguousarray(weights, np.float64)
    b = np.ascontiguousarray(bounds, np.int32)
    t = np.ascontiguousarray(ms, np.float64)
    out = np.zeros(len(b), np.int32)
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    st = mgpu_lib().pem_mgpu_recut_bounds(len(b) - 1, len(w), w.ctypes.data_as(pd), b.ctypes.data_as(pi), t.ctypes.data_as(pd), C.c_double(fixed_ms),
                                          out.ctypes.data_as(pi))
    if st != 0:
        raise ValueError(f"pem_mgpu_recut_bounds: status {st}")
    return out


def mgpu_chunked_pass(devices, rows, cols, I, J, V, nchunks, transpose_b=False):
    """One product through libpemmgpu.so's chunked, overlapped pass (pem_mgpu_spgemm_gather_chunked) on the listed devices:
    A = B = the COO given (B transposed if asked), every rank's row block in `nchunks` chunks.  Returns (rowptr, colidx, vals,
    pass_ms, tail_ms) of the assembled C."""
    L = mgpu_lib()
    H = C.CDLL(os.path.join(_HERE, "libpemspgemm_hip.so"), mode=C.RTLD_GLOBAL)
    L.pem_mgpu_ctx.restype = C.c_void_p
    n = len(devices)
    m = C.c_void_p()
    devs = (C.c_int * n)(*devices)
    if L.pem_mgpu_create(n, devs, C.byref(m)) != 0:
        raise RuntimeError("pem_mgpu_create failed")
    I = np.ascontiguousarray(I, np.int32)
    J = np.ascontiguousarray(J, np.int32)
    V = np.ascontiguousarray(V, np.float64)
    pi, pd = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    As, Bs, plans = [], [], []
    try:
        for g in range(n):
            ctx = C.c_void_p(L.pem_mgpu_ctx(m, g))
            a = C.c_void_p()
            assert H.pem_tiled_from_coo(ctx, rows, cols, C.c_int64(len(I)), I.ctypes.data_as(pi), J.ctypes.data_as(pi), V.ctypes.data_as(pd), 0, C.byref(a)) == 0
            As.append(a)
            if transpose_b:
                b = C.c_void_p()
                assert H.pem_tiled_from_coo(ctx, rows, cols, C.c_int64(len(I)), I.ctypes.data_as(pi), J.ctypes.data_as(pi), V.ctypes.data_as(pd), 1, C.byref(b)) == 0
                Bs.append(b)
            else:
                Bs.append(a)
        S = n * nchunks
        bounds = (C.c_int32 * (S + 1))()
        assert H.pem_split_tile_rows(C.c_void_p(L.pem_mgpu_ctx(m, 0)), As[0], Bs[0], S, bounds) == 0
        for s in range(S):
            p = C.c_void_p()
            assert H.pem_cplan_create(C.c_void_p(L.pem_mgpu_ctx(m, s // nchunks)), As[s // nchunks], Bs[s // nchunks], bounds[s], bounds[s + 1], C.byref(p)) == 0
            plans.append(p)
        parr = (C.c_void_p * S)(*[p.value for p in plans])
        nr, nz, pass_ms, tail_ms = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        st = L.pem_mgpu_spgemm_gather_chunked(m, parr, nchunks, 0, C.byref(nr), C.byref(nz), None, None, None, C.byref(pass_ms), C.byref(tail_ms))
        if st != 0:
            H.pem_last_error.restype = C.c_char_p
            raise RuntimeError(H.pem_last_error().decode(errors="replace"))
        rp = np.zeros(nr.value + 1, np.int32)
        ci = np.zeros(nz.value, np.int32)
        v = np.zeros(nz.value, np.float64)
        st = L.pem_mgpu_spgemm_gather_chunked(m, parr, nchunks, 0, C.byref(nr), C.byref(nz), rp.ctypes.data_as(pi), ci.ctypes.data_as(pi), v.ctypes.data_as(pd),
                                              C.byref(pass_ms), C.byref(tail_ms))
        if st != 0:
            H.pem_last_error.restype = C.c_char_p
            raise RuntimeError(H.pem_last_error().decode(errors="replace"))
        return rp, ci, v, pass_ms.value, tail_ms.value
    finally:
        for s, p in enumerate(plans):
            H.pem_cplan_destroy(C.c_void_p(L.pem_mgpu_ctx(m, s // nchunks)), p)
        for g in range(len(As)):
            ctx = C.c_void_p(L.pem_mgpu_ctx(m, g))
            if Bs[g].value != As[g].value:
                H.pem_tiled_destroy(ctx, Bs[g])
            H.pem_tiled_destroy(ctx, As[g])
        L.pem_mgpu_destroy(m)


def mgpu_assemble_rowptr(slice_rowptrs, nnzs):
    """the host arithmetic of pem_mgpu_gather_csr: (row_off, nnz_off, assembled rowptr) for CSR slices given by their
    relative row pointers and entry counts"""
    n = len(slice_rowptrs)
    L = mgpu_lib()
    sl = [np.ascontiguousarray(r, np.int32) for r in slice_rowptrs]
    nrows = np.array([len(r) - 1 for r in sl], np.int64)
    nnz = np.array(nnzs, np.int64)
    row_off, nnz_off = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
    p64 = C.POINTER(C.c_int64)
    L.pem_mgpu_slice_offsets(n, nrows.ctypes.data_as(p64), nnz.ctypes.data_as(p64), row_off.ctypes.data_as(p64), nnz_off.ctypes.data_as(p64))
    out = np.zeros(int(row_off[n]) + 1, np.int32)
    ptrs = (C.POINTER(C.c_int32) * n)(*[r.ctypes.data_as(C.POINTER(C.c_int32)) for r in sl])
    L.pem_mgpu_rebase_rowptr(n, nrows.ctypes.data_as(p64), ptrs, row_off.ctypes.data_as(p64), nnz_off.ctypes.data_as(p64),
                             out.ctypes.data_as(C.POINTER(C.c_int32)))
    return row_off, nnz_off, out



def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        _lib = C.CDLL(LIB_PATH)
        _lib.pem_host_last_error.restype = C.c_char_p
    return _lib


def mm_read(path, threads=0):
    m = _Coo()
    rc = lib().pem_mm_read(path.encode(), int(threads), C.byref(m))
    if rc != 0:
        raise RuntimeError(f"pem_mm_read -> {rc}: {lib().pem_host_last_error().decode()}")
    n = int(m.nnz)

    def arr(p, dt):
        return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)

    out = dict(rows=m.rows, cols=m.cols, nnz=n, I=arr(m.I, np.int32), J=arr(m.J, np.int32), V=arr(m.V, np.float64),
               symmetric=bool(m.symmetric), field=m.field)
    lib().pem_coo_free(C.byref(m))
    return out


def standin(name, scale=1.0):
    """pem_standin_generate: the seeded C++ stand-in generator behind `pemspgemm --standin` -> rows, cols, I, J, V"""
    m = _Coo()
    rc = lib().pem_standin_generate(name.encode(), C.c_double(float(scale)), C.byref(m))
    if rc != 0:
        raise RuntimeError(f"pem_standin_generate({name!r}, {scale}) -> {rc}")
    n = int(m.nnz)
    I = np.ctypeslib.as_array(m.I, shape=(n,)).astype(np.int32, copy=True)
    J = np.ctypeslib.as_array(m.J, shape=(n,)).astype(np.int32, copy=True)
    V = np.ctypeslib.as_array(m.V, shape=(n,)).astype(np.float64, copy=True)
    rows, cols = m.rows, m.cols
    lib().pem_coo_free(C.byref(m))
    return rows, cols, I, J, V


def standin_names():
    lib().pem_standin_names.restype = C.c_char_p
    return lib().pem_standin_names().decode().split()


def write_result_files(directory, rows, cols, vals):
    rows = np.ascontiguousarray(rows, np.int32)
    cols = np.ascontiguousarray(cols, np.int32)
    vals = np.ascontiguousarray(vals, np.float64)
    rc = lib().pem_write_result_files(directory.encode(), C.c_int64(len(rows)), rows.ctypes.data_as(C.POINTER(C.c_int32)),
                                      cols.ctypes.data_as(C.POINTER(C.c_int32)), vals.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise RuntimeError(lib().pem_host_last_error().decode())


def csv_append(path, extra=None, **fields):
    rec = CsvRecord()
    for k, v in fields.items():
        setattr(rec, k, v.encode() if k == "matrix" else v)
    rc = lib().pem_csv_append(path.encode(), C.byref(rec), extra.encode() if extra else None)
    if rc != 0:
        raise RuntimeError(lib().pem_host_last_error().decode())


def write_mtx_csr(path, rows, cols, rowptr, colidx, vals, comment=""):
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    colidx = np.ascontiguousarray(colidx, np.int32)
    vals = np.ascontiguousarray(vals, np.float64)
    rc = lib().pem_write_mtx_csr(path.encode(), int(rows), int(cols), rowptr.ctypes.data_as(C.POINTER(C.c_int32)),
                                 colidx.ctypes.data_as(C.POINTER(C.c_int32)), vals.ctypes.data_as(C.POINTER(C.c_double)), comment.encode())
    if rc != 0:
        raise RuntimeError(lib().pem_host_last_error().decode())
