"""Independent Python statement of the tiled-format cache file (include/pem_spgemm.h): the tests write
files from the ORACLE's arrays with it, hand them to pem_tiled_load, and compare pem_tiled_save's bytes with it.

Layout (little-endian): 128-byte header, then tile_keys int64[T], tile_nnz_ptr int32[T+1], rowcolidx uint8[nnz],
vals float64[nnz], each zero-padded to a multiple of 64 bytes."""
import struct

import numpy as np

MAGIC = b"PEMTILE1"
M64 = (1 << 64) - 1
MUL = 0xD6E8FEB86659FD93
SEED_PAYLOAD = 0x70656D74696C6531
SEED_HEADER = 0x6865616465723031
HEADER = struct.Struct("<8sIIIIiiqqQqIIQQQ32x")   # 128 bytes
assert HEADER.size == 128


def _step(h, w):
    h = ((h ^ w) * MUL) & M64
    return ((h << 29) | (h >> 35)) & M64


def hash64(data, seed):
    """Four interleaved multiply-rotate lanes over little-endian 8-byte words (tail zero-padded): word i feeds lane i & 3."""
    n = len(data)
    h = [(seed ^ ((n * 0x9E3779B97F4A7C15) & M64) ^ ((k * 0xA0761D6478BD642F) & M64)) & M64 for k in range(4)]
    words = np.frombuffer(bytes(data) + b"\0" * ((-n) % 8), dtype="<u8").tolist()
    for i, w in enumerate(words):
        h[i & 3] = _step(h[i & 3], w)
    r = h[0]
    for k in (1, 2, 3):
        r = _step(r, h[k])
    r ^= r >> 32
    r = (r * MUL) & M64
    r ^= r >> 29
    return r


def _pad64(b):
    return b + b"\0" * ((-len(b)) % 64)


def payload_bytes(tile_keys, tile_nnz_ptr, rowcolidx, vals, value_bytes=8):
    return (_pad64(np.ascontiguousarray(tile_keys, dtype="<i8").tobytes()) +
            _pad64(np.ascontiguousarray(tile_nnz_ptr, dtype="<i4").tobytes()) +
            _pad64(np.ascontiguousarray(rowcolidx, dtype=np.uint8).tobytes()) +
            _pad64(np.ascontiguousarray(vals, dtype="<f8" if value_bytes == 8 else "<f4").tobytes()))


def header_bytes(rows, cols, nnz, ntiles, payload, key=(0, 0, 0), version=1, tile=16, value_bytes=8, payload_hash=None):
    ph = hash64(payload, SEED_PAYLOAD) if payload_hash is None else payload_hash
    fields = [MAGIC, version, tile, value_bytes, 128, rows, cols, nnz, ntiles, key[0], key[1], key[2], 0, len(payload), ph]
    h0 = HEADER.pack(*fields, 0)
    return HEADER.pack(*fields, hash64(h0, SEED_HEADER))


def cache_bytes(rows, cols, tile_keys, tile_nnz_ptr, rowcolidx, vals, key=(0, 0, 0), nnz=None, ntiles=None, value_bytes=8):
    """value_bytes 8: fp64 (the reference's ValueType); 4: fp32 tilings."""
    p = payload_bytes(tile_keys, tile_nnz_ptr, rowcolidx, vals, value_bytes)
    nnz = len(vals) if nnz is None else nnz
    ntiles = len(tile_keys) if ntiles is None else ntiles
    return header_bytes(rows, cols, nnz, ntiles, p, key, value_bytes=value_bytes) + p


def read_cache(path):
    raw = open(path, "rb").read()
    (magic, version, tile, vbytes, hbytes, rows, cols, nnz, ntiles, ksize, kmtime, ktr, _res, pbytes, phash, hhash) = HEADER.unpack(raw[:128])
    assert magic == MAGIC and version == 1 and tile == 16 and vbytes in (4, 8) and hbytes == 128
    z = bytearray(raw[:128])
    z[88:96] = b"\0" * 8
    assert hash64(bytes(z), SEED_HEADER) == hhash, "header checksum"
    payload = raw[128:]
    assert len(payload) == pbytes and hash64(payload, SEED_PAYLOAD) == phash, "payload checksum"
    off = 0

    def take(count, dt):
        nonlocal off
        nb = count * np.dtype(dt).itemsize
        a = np.frombuffer(payload[off:off + nb], dtype=dt).copy()
        off += nb + ((-nb) % 64)
        return a
    out = dict(rows=rows, cols=cols, nnz=nnz, ntiles=ntiles, key=(ksize, kmtime, ktr))
    out["tile_keys"] = take(ntiles, "<i8")
    out["tile_nnz_ptr"] = take(ntiles + 1, "<i4")
    out["rowcolidx"] = take(nnz, np.uint8)
    out["vals"] = take(nnz, "<f8" if vbytes == 8 else "<f4")
    out["value_bytes"] = vbytes
    assert off == len(payload)
    return out
