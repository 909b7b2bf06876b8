"""Minimal pure-Python reader for the Keras ``.h5`` weight files the reference writes
(``model.save_weights``: src/UtilsFiles.py:153-164, naming src/NeRF.py:18-19,342-351) -- no h5py needed.

Layout of such a file: root groups ``model`` (coarse) and ``model_1`` (fine), each
holding one subgroup per Dense layer (``dense``, ``dense_1`` ... numbered in creation order) with float32
datasets ``kernel:0`` (in,out) and ``bias:0``.

Supported subset of HDF5 (what h5py/libhdf5 emit with default settings): superblock version 0/1,
version-1 object headers (with continuation blocks), old-style groups (symbol-table message, v1 B-tree,
local heap), contiguous or compact little-endian IEEE float32/float64 datasets.  Anything else (new-style
groups, chunked/compressed data) raises ``ValueError`` naming what was found.  Nothing in the file is
executed: it is parsed as plain bytes.
"""
from __future__ import annotations

import re
import struct
from typing import Dict, List, Tuple

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _H5:
    def __init__(self, data: bytes):
        self.d = data
        if data[:8] != _SIG:
            raise ValueError("not an HDF5 file (bad signature)")
        ver = data[8]
        if ver not in (0, 1):
            raise ValueError(f"HDF5 superblock version {ver} is not supported (only 0/1: files written with "
                             "libver='earliest', the h5py/Keras default)")
        self.so, self.sl = data[13], data[14]          # size of offsets / lengths
        if self.so != 8 or self.sl != 8:
            raise ValueError("only 8-byte offsets/lengths are supported")
        p = 24 if ver == 0 else 28
        self.base = self.u64(p)
        root_entry = p + 4 * 8
        self.root_header = self.u64(root_entry + 8)

    def u16(self, p): return struct.unpack_from("<H", self.d, p)[0]
    def u32(self, p): return struct.unpack_from("<I", self.d, p)[0]
    def u64(self, p): return struct.unpack_from("<Q", self.d, p)[0]

    # ---- object headers (version 1) ----
    def messages(self, addr: int) -> List[Tuple[int, int, int]]:
        """[(type, data_offset, size)] of the object header at ``addr``, continuation blocks followed."""
        d = self.d
        if d[addr:addr + 4] == b"OHDR":
            raise ValueError("version-2 object headers (libver='latest') are not supported")
        if d[addr] != 1:
            raise ValueError(f"object header version {d[addr]} is not supported")
        nmsg = self.u16(addr + 2)
        size = self.u32(addr + 8)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize = self.u16(p), self.u16(p + 2)
                body = p + 8
                if mtype == 0x0010:                       # continuation
                    blocks.append((self.u64(body), self.u64(body + 8)))
                out.append((mtype, body, msize))
                p = body + msize
        return out

    # ---- groups ----
    def heap_name(self, heap_addr: int, off: int) -> str:
        if self.d[heap_addr:heap_addr + 4] != b"HEAP":
            raise ValueError("bad local heap signature")
        seg = self.u64(heap_addr + 24)
        end = self.d.index(b"\x00", seg + off)
        return self.d[seg + off:end].decode("utf-8")

    def _walk_btree(self, node: int, heap: int, out: Dict[str, int]) -> None:
        d = self.d
        if d[node:node + 4] != b"TREE":
            raise ValueError("bad B-tree signature")
        ntype, level, used = d[node + 4], d[node + 5], self.u16(node + 6)
        if ntype != 0:
            raise ValueError("unexpected B-tree node type for a group")
        p = node + 8 + 16                                  # past siblings
        for i in range(used):
            child = self.u64(p + 8 + i * 16)               # key_i (8) child_i (8) ...
            if level > 0:
                self._walk_btree(child, heap, out)
            else:
                if d[child:child + 4] != b"SNOD":
                    raise ValueError("bad symbol-table node signature")
                nsym = self.u16(child + 6)
                for s in range(nsym):
                    e = child + 8 + s * 40
                    out[self.heap_name(heap, self.u64(e))] = self.u64(e + 8)

    def members(self, header_addr: int) -> Dict[str, int]:
        """name -> object header address of a group's members ({} if the object is not an old-style group)."""
        for mtype, body, _ in self.messages(header_addr):
            if mtype == 0x0011:
                out: Dict[str, int] = {}
                self._walk_btree(self.u64(body), self.u64(body + 8), out)
                return out
            if mtype in (0x0002, 0x0006):
                raise ValueError("new-style groups (link messages) are not supported")
        return {}

    # ---- datasets ----
    def dataset(self, header_addr: int) -> np.ndarray:
        shape = dtype = raw = None
        for mtype, body, size in self.messages(header_addr):
            d = self.d
            if mtype == 0x0001:                            # dataspace
                ver, rank, flags = d[body], d[body + 1], d[body + 2]
                p = body + (8 if ver == 1 else 4)
                shape = tuple(self.u64(p + 8 * i) for i in range(rank))
            elif mtype == 0x0003:                          # datatype
                cls, bits0, tsize = d[body] & 0x0F, d[body + 1], self.u32(body + 4)
                if cls != 1 or (bits0 & 1) != 0 or tsize not in (4, 8):
                    raise ValueError(f"unsupported datatype (class {cls}, size {tsize}): need little-endian float")
                dtype = np.dtype("<f4" if tsize == 4 else "<f8")
            elif mtype == 0x0008:                          # layout
                ver = d[body]
                if ver != 3:
                    raise ValueError(f"data layout message version {ver} is not supported")
                cls = d[body + 1]
                if cls == 1:                               # contiguous
                    addr, n = self.u64(body + 2), self.u64(body + 10)
                    raw = b"" if addr == _UNDEF else d[self.base + addr:self.base + addr + n]
                elif cls == 0:                             # compact
                    n = self.u16(body + 2)
                    raw = d[body + 4:body + 4 + n]
                else:
                    raise ValueError("chunked/compressed datasets are not supported")
            elif mtype == 0x000B:
                raise ValueError("filtered (compressed) datasets are not supported")
        if shape is None or dtype is None or raw is None:
            raise ValueError("object is not a simple dataset")
        n = int(np.prod(shape)) if shape else 1
        return np.frombuffer(raw, dtype=dtype, count=n).reshape(shape).astype(np.float32)


def _suffix(name: str) -> int:
    m = re.search(r"_(\d+)$", name)
    return int(m.group(1)) if m else 0


def read_keras_weights(path: str) -> Dict[str, List[np.ndarray]]:
    """{model group name: [kernel, bias, kernel, bias, ...] in layer-creation order} -- the order of
    Keras ``model.get_weights()`` that ``Context.load_weights`` takes."""
    with open(path, "rb") as f:
        h5 = _H5(f.read())
    out: Dict[str, List[np.ndarray]] = {}
    for gname, gaddr in sorted(h5.members(h5.root_header).items(), key=lambda kv: _suffix(kv[0])):
        layers = h5.members(gaddr)
        weights: List[np.ndarray] = []
        for lname, laddr in sorted(layers.items(), key=lambda kv: _suffix(kv[0])):
            sub = h5.members(laddr)
            # Keras nests the variables one level deeper when names contain '/': <layer>/<layer>/kernel:0
            if len(sub) == 1 and next(iter(sub)) == lname:
                sub = h5.members(next(iter(sub.values())))
            for vname in ("kernel:0", "bias:0"):
                if vname in sub:
                    weights.append(h5.dataset(sub[vname]))
        if weights:
            out[gname] = weights
    return out


def load_nerf_checkpoint(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """(coarse_blob, fine_blob) from ``NeRF_model_epoch_XXX.h5``; fine is None for a coarse-only run."""
    models = read_keras_weights(path)
    names = sorted(models, key=_suffix)
    if not names:
        raise ValueError("no model groups with Dense weights found")
    blobs = [np.concatenate([w.ravel() for w in models[n]]) for n in names[:2]]
    return blobs[0], (blobs[1] if len(blobs) > 1 else None)
