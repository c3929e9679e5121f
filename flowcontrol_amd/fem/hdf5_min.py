"""Minimal pure-Python HDF5 dataset reader (no h5py in this image).

Covers exactly what the mesh files shipped with the reference use (see
SURVEY.md Appendix B): superblock v0/v1, symbol-table groups (v1 B-tree +
local heap), v1 object headers with continuation blocks, data layout v3
(contiguous and chunked via v1 chunk B-tree), filter pipeline with deflate
(and byte-shuffle), fixed-point / IEEE-float datatypes of either byte order.

This replaces ``dolfin.XDMFFile.read(mesh)`` (reference
``src/flowcontrol/flowsolver.py:233-240``) for the HDF5 payload.
"""

from __future__ import annotations

import struct
import zlib
from pathlib import Path

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5Error(RuntimeError):
    pass


class MinimalHDF5:
    def __init__(self, path: str | Path):
        self.path = Path(path)
        self.buf = self.path.read_bytes()
        self._parse_superblock()

    # ── low level ────────────────────────────────────────────────────────────
    def _u(self, off: int, n: int) -> int:
        return int.from_bytes(self.buf[off : off + n], "little")

    def _parse_superblock(self) -> None:
        b = self.buf
        if b[:8] != _SIG:
            raise HDF5Error(f"{self.path}: not an HDF5 file")
        ver = b[8]
        if ver not in (0, 1):
            raise HDF5Error(f"superblock version {ver} not supported")
        self.so = b[13]  # size of offsets
        self.sl = b[14]  # size of lengths
        p = 24 if ver == 0 else 28
        self.base = self._u(p, self.so)
        p += 4 * self.so  # base, free-space, eof, driver-info
        # root group symbol-table entry
        p += self.so  # link name offset
        self.root_header = self._u(p, self.so)
        p += self.so
        cache_type = self._u(p, 4)
        p += 8
        self.root_btree = self.root_heap = None
        if cache_type == 1:
            self.root_btree = self._u(p, self.so)
            self.root_heap = self._u(p + self.so, self.so)

    # ── object headers ───────────────────────────────────────────────────────
    def _messages(self, addr: int):
        """Yield (type, flags, payload-bytes) for a version-1 object header."""
        b = self.buf
        if b[addr] != 1:
            raise HDF5Error(f"object header version {b[addr]} not supported")
        nmsg = self._u(addr + 2, 2)
        hsize = self._u(addr + 8, 4)
        blocks = [(addr + 16, hsize)]
        seen = 0
        while blocks and seen < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and seen < nmsg:
                mtype = self._u(p, 2)
                msize = self._u(p + 2, 2)
                flags = b[p + 4]
                body = b[p + 8 : p + 8 + msize]
                p += 8 + msize
                seen += 1
                if mtype == 0x0010:  # continuation
                    blocks.append((self._u_from(body, 0, self.so), self._u_from(body, self.so, self.sl)))
                else:
                    yield mtype, flags, body

    @staticmethod
    def _u_from(body: bytes, off: int, n: int) -> int:
        return int.from_bytes(body[off : off + n], "little")

    # ── groups ───────────────────────────────────────────────────────────────
    def _group_entries(self, btree: int, heap: int) -> dict[str, int]:
        b = self.buf
        if b[heap : heap + 4] != b"HEAP":
            raise HDF5Error("bad local heap")
        heap_data = self._u(heap + 8 + 2 * self.sl, self.so)
        out: dict[str, int] = {}

        def name_at(off: int) -> str:
            s = heap_data + off
            e = b.index(b"\x00", s)
            return b[s:e].decode()

        def walk(node: int) -> None:
            if b[node : node + 4] == b"TREE":
                level = b[node + 5]
                n = self._u(node + 6, 2)
                p = node + 8 + 2 * self.so
                # keys (sl) and children (so) interleaved: key0 child0 key1 ...
                for i in range(n):
                    p += self.sl
                    child = self._u(p, self.so)
                    p += self.so
                    walk(child)
                _ = level
            elif b[node : node + 4] == b"SNOD":
                n = self._u(node + 6, 2)
                p = node + 8
                for i in range(n):
                    noff = self._u(p, self.so)
                    hdr = self._u(p + self.so, self.so)
                    out[name_at(noff)] = hdr
                    p += 2 * self.so + 4 + 4 + 16
            else:
                raise HDF5Error("bad group node")

        walk(btree)
        return out

    def _children(self, header_addr: int) -> dict[str, int]:
        if header_addr == self.root_header and self.root_btree is not None:
            return self._group_entries(self.root_btree, self.root_heap)
        for mtype, _, body in self._messages(header_addr):
            if mtype == 0x0011:  # symbol table message
                bt = self._u_from(body, 0, self.so)
                hp = self._u_from(body, self.so, self.so)
                return self._group_entries(bt, hp)
        raise HDF5Error("object is not an (old-style) group")

    def _resolve(self, name: str) -> int:
        addr = self.root_header
        for part in [p for p in name.split("/") if p]:
            ch = self._children(addr)
            if part not in ch:
                raise KeyError(f"{name!r}: no member {part!r}; have {sorted(ch)}")
            addr = ch[part]
        return addr

    def keys(self, group: str = "/") -> list[str]:
        return sorted(self._children(self._resolve(group)))

    # ── datasets ─────────────────────────────────────────────────────────────
    def read(self, name: str) -> np.ndarray:
        addr = self._resolve(name)
        shape = dtype = layout = None
        filters: list[int] = []
        for mtype, _, body in self._messages(addr):
            if mtype == 0x0001:
                shape = self._parse_dataspace(body)
            elif mtype == 0x0003:
                dtype = self._parse_datatype(body)
            elif mtype == 0x0008:
                layout = self._parse_layout(body)
            elif mtype == 0x000B:
                filters = self._parse_filters(body)
        if shape is None or dtype is None or layout is None:
            raise HDF5Error(f"{name}: incomplete dataset header")
        if layout[0] == "contiguous":
            _, daddr, dsize = layout
            n = int(np.prod(shape)) if shape else 1
            arr = np.frombuffer(self.buf, dtype=dtype, count=n, offset=daddr)
            return arr.reshape(shape).astype(dtype.newbyteorder("="))
        if layout[0] == "compact":
            arr = np.frombuffer(layout[1], dtype=dtype)
            return arr.reshape(shape).astype(dtype.newbyteorder("="))
        _, btree, chunk = layout
        out = np.empty(shape, dtype=dtype.newbyteorder("="))
        self._read_chunks(btree, chunk, filters, dtype, out)
        return out

    def _parse_dataspace(self, body: bytes) -> tuple[int, ...]:
        ver = body[0]
        rank = body[1]
        if ver == 1:
            p = 8
        elif ver == 2:
            p = 4
        else:
            raise HDF5Error(f"dataspace version {ver}")
        return tuple(self._u_from(body, p + i * self.sl, self.sl) for i in range(rank))

    @staticmethod
    def _parse_datatype(body: bytes) -> np.dtype:
        cls = body[0] & 0x0F
        bits0 = body[1]
        size = int.from_bytes(body[4:8], "little")
        order = ">" if (bits0 & 1) else "<"
        if cls == 0:
            signed = (bits0 >> 3) & 1
            return np.dtype(f"{order}{'i' if signed else 'u'}{size}")
        if cls == 1:
            return np.dtype(f"{order}f{size}")
        raise HDF5Error(f"datatype class {cls} not supported")

    def _parse_layout(self, body: bytes):
        ver = body[0]
        if ver != 3:
            raise HDF5Error(f"data layout version {ver} not supported")
        cls = body[1]
        if cls == 1:
            return ("contiguous", self._u_from(body, 2, self.so), self._u_from(body, 2 + self.so, self.sl))
        if cls == 0:
            n = self._u_from(body, 2, 2)
            return ("compact", bytes(body[4 : 4 + n]))
        if cls == 2:
            nd = body[2]
            bt = self._u_from(body, 3, self.so)
            dims = tuple(self._u_from(body, 3 + self.so + 4 * i, 4) for i in range(nd))
            return ("chunked", bt, dims[:-1])  # last dim = element size
        raise HDF5Error(f"layout class {cls}")

    @staticmethod
    def _parse_filters(body: bytes) -> list[int]:
        ver = body[0]
        nf = body[1]
        ids = []
        p = 8 if ver == 1 else 2
        for _ in range(nf):
            fid = int.from_bytes(body[p : p + 2], "little")
            if ver == 1 or fid >= 256:
                nlen = int.from_bytes(body[p + 2 : p + 4], "little")
                ncd = int.from_bytes(body[p + 6 : p + 8], "little")
                p += 8
            else:
                nlen = 0
                ncd = int.from_bytes(body[p + 4 : p + 6], "little")
                p += 6
            if ver == 1:
                nlen = (nlen + 7) // 8 * 8
            p += nlen + 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            ids.append(fid)
        return ids

    def _read_chunks(self, node: int, chunk: tuple[int, ...], filters: list[int], dtype: np.dtype, out: np.ndarray):
        b = self.buf
        if b[node : node + 4] != b"TREE" or b[node + 4] != 1:
            raise HDF5Error("bad chunk B-tree node")
        level = b[node + 5]
        n = self._u(node + 6, 2)
        nd = len(chunk)
        keysz = 8 + 8 * (nd + 1)
        p = node + 8 + 2 * self.so
        for _ in range(n):
            csize = self._u(p, 4)
            fmask = self._u(p + 4, 4)
            offs = tuple(self._u(p + 8 + 8 * i, 8) for i in range(nd))
            child = self._u(p + keysz, self.so)
            p += keysz + self.so
            if level > 0:
                self._read_chunks(child, chunk, filters, dtype, out)
                continue
            raw = bytes(b[child : child + csize])
            for k, fid in reversed(list(enumerate(filters))):
                if fmask & (1 << k):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    a = np.frombuffer(raw, np.uint8).reshape(dtype.itemsize, -1)
                    raw = a.T.tobytes()
                else:
                    raise HDF5Error(f"filter id {fid} not supported")
            block = np.frombuffer(raw, dtype=dtype, count=int(np.prod(chunk))).reshape(chunk)
            sl_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, out.shape))
            sl_in = tuple(slice(0, s.stop - s.start) for s in sl_out)
            out[sl_out] = block[sl_in]


def read_dataset(path: str | Path, name: str) -> np.ndarray:
    return MinimalHDF5(path).read(name)


__all__ = ["MinimalHDF5", "read_dataset", "HDF5Error"]

# keep struct imported for callers that extend the reader
_ = struct


# ──────────────────────────────────────────────────────────────────────────────────────────
# Minimal writer: the same on-disk structures the reader above understands (and that HDF5 1.8+
# reads): superblock v0, old-style groups (v1 B-tree + local heap + one symbol node), v1 object
# headers, contiguous uncompressed little-endian datasets of f8 / i8 / i4 / u1.
# ──────────────────────────────────────────────────────────────────────────────────────────
_INT_K = 16


def _pad8(b: bytes) -> bytes:
    return b + b"\x00" * (-len(b) % 8)


def _dtype_message(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize == 8:
        head = bytes([0x11, 0x20, 0x3F, 0x00]) + struct.pack("<I", 8)
        props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        return head + props
    if dt.kind in "iu" and dt.itemsize in (1, 4, 8):
        head = bytes([0x10, 0x08 if dt.kind == "i" else 0x00, 0x00, 0x00]) + struct.pack("<I", dt.itemsize)
        return head + struct.pack("<HH", 0, 8 * dt.itemsize)
    raise HDF5Error(f"cannot write dtype {dt}")


def _message(mtype: int, body: bytes) -> bytes:
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def _object_header(messages: list[bytes]) -> bytes:
    blob = b"".join(messages)
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(blob)) + blob


class _Writer:
    def __init__(self, leaf_k: int = 4):
        self.buf = bytearray(96)  # superblock v0 with 8-byte offsets/lengths is 96 bytes
        self.leaf_k = leaf_k  # symbols per group node = 2K; a file-level parameter stored in the superblock

    def alloc(self, data: bytes) -> int:
        self.buf += b"\x00" * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def dataset(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        raw = arr.tobytes()
        daddr = self.alloc(raw) if raw else _UNDEF
        space = struct.pack("<BBB5x", 1, arr.ndim, 0) + b"".join(struct.pack("<Q", d) for d in arr.shape)
        layout = struct.pack("<BB", 3, 1) + struct.pack("<QQ", daddr, len(raw))
        fill = struct.pack("<BBBB", 2, 2, 2, 0)  # fill value v2: late allocation, write-if-set, undefined
        return self.alloc(_object_header([_message(0x0001, space), _message(0x0003, _dtype_message(arr.dtype)),
                                          _message(0x0005, fill), _message(0x0008, layout)]))

    def group(self, members: dict[str, int]) -> tuple[int, int, int]:
        """members: name → object header address.  Returns (header, btree, heap) addresses."""
        if len(members) > 2 * self.leaf_k:
            raise HDF5Error("too many members in one group for the minimal writer")
        names = sorted(members)
        heap_data = bytearray(8)  # offset 0: empty string
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += _pad8(n.encode() + b"\x00")
        free_off = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 16)  # one free block: next = 1 (none), size = 16
        data_addr = self.alloc(bytes(heap_data))
        heap = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, data_addr))
        snod = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(names)))
        for n in names:
            snod += struct.pack("<QQII16x", offs[n], members[n], 0, 0)
        snod += b"\x00" * (8 + 2 * self.leaf_k * 40 - len(snod))
        snod_addr = self.alloc(bytes(snod))
        tree = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, _UNDEF, _UNDEF))
        tree += struct.pack("<QQQ", 0, snod_addr, offs[names[-1]] if names else 0)
        tree += b"\x00" * (24 + (2 * _INT_K + 1) * 8 + 2 * _INT_K * 8 - len(tree))
        btree = self.alloc(bytes(tree))
        header = self.alloc(_object_header([_message(0x0011, struct.pack("<QQ", btree, heap))]))
        return header, btree, heap

    def finish(self, root: tuple[int, int, int]) -> bytes:
        header, btree, heap = root
        sb = _SIG + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", self.leaf_k, _INT_K, 0)
        sb += struct.pack("<QQQQ", 0, _UNDEF, len(self.buf), _UNDEF)
        sb += struct.pack("<QQII", 0, header, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write_hdf5(path: str | Path, tree: dict) -> None:
    """Write nested ``{name: array | {…}}`` as groups / contiguous datasets."""
    def widest(node: dict) -> int:
        return max([len(node)] + [widest(v) for v in node.values() if isinstance(v, dict)])

    w = _Writer(leaf_k=max(4, (widest(tree) + 1) // 2))

    def emit(node: dict) -> tuple[int, int, int]:
        members = {}
        for name, val in node.items():
            if "/" in name or not name:
                raise HDF5Error(f"bad member name {name!r}")
            members[name] = emit(val)[0] if isinstance(val, dict) else w.dataset(np.asarray(val))
        return w.group(members)

    data = w.finish(emit(tree))
    Path(path).write_bytes(data)


def read_hdf5_tree(path: str | Path) -> dict:
    """Inverse of :func:`write_hdf5` for files made of old-style groups and simple datasets."""
    f = MinimalHDF5(path)

    def walk(prefix: str) -> dict:
        out = {}
        for k in f.keys(prefix or "/"):
            p = f"{prefix}/{k}"
            try:
                out[k] = f.read(p)
            except HDF5Error:
                out[k] = walk(p)
        return out

    return walk("")


__all__ += ["write_hdf5", "read_hdf5_tree"]
