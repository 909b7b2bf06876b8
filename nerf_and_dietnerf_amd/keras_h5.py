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


# ---------------------------------------------------------------------------------------------
# writer: model.save_weights(path) in the layout above (src/UtilsFiles.py:153-164) -- readable by h5py / Keras
# ---------------------------------------------------------------------------------------------
class _H5Writer:
    """Minimal HDF5 emitter: superblock v0, version-1 object headers, old-style groups (one symbol-table node per
    group: the superblock's group-leaf K is raised so that 2K >= the largest group), contiguous little-endian
    float32 datasets, version-1 attributes holding fixed-length strings (scalar or 1-D)."""

    LEAF_K = 32          # up to 64 members per group in one SNOD
    INT_K = 16

    def __init__(self):
        self.buf = bytearray(b"\x00" * 96)     # superblock (56 + 40-byte root symbol-table entry), patched at the end

    def _align(self, n=8):
        self.buf += b"\x00" * (-len(self.buf) % n)

    def _alloc(self, data: bytes) -> int:
        self._align()
        addr = len(self.buf)
        self.buf += data
        return addr

    # ---- messages ----
    @staticmethod
    def _msg(mtype: int, body: bytes) -> bytes:
        body += b"\x00" * (-len(body) % 8)
        return struct.pack("<HHB3x", mtype, len(body), 0) + body

    @staticmethod
    def _dataspace(shape) -> bytes:
        return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(s)) for s in shape)

    @staticmethod
    def _dtype_f32() -> bytes:
        # class 1 (float) version 1; little-endian, implied-msb mantissa, sign bit 31; 32-bit IEEE layout
        return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)

    @staticmethod
    def _dtype_str(n: int) -> bytes:
        return struct.pack("<BBBBI", 0x13, 0x01, 0x00, 0x00, n)       # class 3 (string), null-padded, ASCII

    def _attr(self, name: str, strings, scalar: bool) -> bytes:
        vals = [s.encode("utf-8") for s in strings]
        width = max([1] + [len(v) for v in vals])                      # an empty list gives a zero-length attribute
        nm = name.encode("utf-8") + b"\x00"
        dt = self._dtype_str(width)
        ds = self._dataspace(() if scalar else (len(vals),))
        pad = lambda b: b + b"\x00" * (-len(b) % 8)
        body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + pad(nm) + pad(dt) + pad(ds)
        body += b"".join(v.ljust(width, b"\x00") for v in vals)
        return self._msg(0x000C, body)

    def _object_header(self, msgs) -> int:
        body = b"".join(msgs)
        return self._alloc(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

    # ---- objects ----
    def dataset(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr, dtype="<f4")
        data_addr = self._alloc(arr.tobytes())
        layout = struct.pack("<BBQQ", 3, 1, data_addr, arr.nbytes)
        fill = struct.pack("<BBBB", 2, 2, 0, 0)                        # fill value v2: allocate late, never write, undefined
        return self._object_header([self._msg(0x0001, self._dataspace(arr.shape)), self._msg(0x0003, self._dtype_f32()),
                                    self._msg(0x0005, fill), self._msg(0x0008, layout)])

    def group(self, members: Dict[str, int], attrs=()) -> int:
        """members: name -> object header address.  attrs: [(name, [strings], scalar)]."""
        names = sorted(members)                                        # symbol-table entries are ordered by name
        if len(names) > 2 * self.LEAF_K:
            raise ValueError("group too large for a single symbol-table node")
        heap_data = bytearray(b"\x00" * 8)                             # offset 0: the empty name
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            b = n.encode("utf-8") + b"\x00"
            heap_data += b + b"\x00" * (-len(b) % 8)
        heap_seg = self._alloc(bytes(heap_data))
        heap = self._alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, heap_seg))   # free list: none (1)
        snod = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(names)))
        for n in names:
            snod += struct.pack("<QQII16x", offs[n], members[n], 0, 0)
        snod += b"\x00" * (8 + 2 * self.LEAF_K * 40 - len(snod))
        snod_addr = self._alloc(bytes(snod))
        tree = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, _UNDEF, _UNDEF))
        tree += struct.pack("<Q", 0)                                   # key 0: the empty name
        if names:
            tree += struct.pack("<QQ", snod_addr, offs[names[-1]])     # child 0, key 1 = largest name in it
        tree += b"\x00" * (24 + (2 * self.INT_K + 1) * 8 + 2 * self.INT_K * 8 - len(tree))
        tree_addr = self._alloc(bytes(tree))
        msgs = [self._msg(0x0011, struct.pack("<QQ", tree_addr, heap))]
        msgs += [self._attr(n, v, sc) for n, v, sc in attrs]
        return self._object_header(msgs), tree_addr, heap

    def finish(self, root) -> bytes:
        root_hdr, tree_addr, heap = root
        self._align()
        sb = _SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.INT_K, 0)
        sb += struct.pack("<QQQQ", 0, _UNDEF, len(self.buf), _UNDEF)
        sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", tree_addr, heap)
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write_keras_weights(path: str, models: Dict[str, List[np.ndarray]], first_dense_index: int = 0,
                        keras_version: str = "2.7.0") -> None:
    """``models``: {"model": [kernel, bias, ...], "model_1": [...]} in ``get_weights()`` order -> a Keras-2.7 style
    weights file: groups ``model`` / ``model_1`` (+ the empty ``top_level_model_weights``), per Dense layer a group
    ``dense_<k>`` with ``kernel:0`` and ``bias:0``; Dense layers are numbered consecutively across the models as
    Keras names them; attributes ``layer_names`` / ``backend`` / ``keras_version`` on the root and ``weight_names`` on
    every model group (fixed-length strings)."""
    w = _H5Writer()
    k = first_dense_index
    root_members = {}
    for gname in sorted(models, key=_suffix):
        tensors = models[gname]
        if len(tensors) % 2:
            raise ValueError("expected kernel/bias pairs")
        layers, weight_names = {}, []
        for i in range(0, len(tensors), 2):
            lname = "dense" if k == 0 else f"dense_{k}"
            k += 1
            kaddr, baddr = w.dataset(tensors[i]), w.dataset(tensors[i + 1])
            layers[lname] = w.group({"kernel:0": kaddr, "bias:0": baddr})[0]
            weight_names += [f"{lname}/kernel:0", f"{lname}/bias:0"]
        root_members[gname] = w.group(layers, [("weight_names", weight_names, False)])[0]
    root_members["top_level_model_weights"] = w.group({}, [("weight_names", [], False)])[0]     # empty, as Keras writes
    root = w.group(root_members, [("layer_names", sorted(models, key=_suffix), False), ("backend", ["tensorflow"], True),
                                  ("keras_version", [keras_version], True)])
    with open(path, "wb") as f:
        f.write(w.finish(root))


def save_nerf_checkpoint(path: str, coarse_blob: np.ndarray, fine_blob=None, **shape_kw) -> None:
    """Inverse of load_nerf_checkpoint: flat ``get_weights()``-order blobs -> ``NeRF_model_epoch_XXX.h5``."""
    from .weights import layer_shapes
    models = {}
    for name, blob in (("model", coarse_blob), ("model_1", fine_blob)):
        if blob is None:
            continue
        blob = np.asarray(blob, np.float32).ravel()
        tensors, off = [], 0
        for i, o in layer_shapes(**shape_kw):
            tensors.append(blob[off:off + i * o].reshape(i, o)); off += i * o
            tensors.append(blob[off:off + o]); off += o
        if off != blob.size:
            raise ValueError(f"weight blob has {blob.size} floats, expected {off}")
        models[name] = tensors
    write_keras_weights(path, models)
