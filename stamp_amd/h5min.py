"""A minimal HDF5 reader / writer in pure Python + numpy: the third backend of `stamp_amd.h5io` (after h5py and libhdf5-via-ctypes), so that
the feature files of the path -- `feats` f16 [N, D], `coords` f32 [N, 2] and a handful of root attributes (reference
src/stamp/preprocessing/__init__.py:345-367, src/stamp/encoding/encoder/__init__.py:203-229) -- can be written and read on a machine
that has neither library.  Host-side I/O only; no arithmetic.

Scope (what STAMP's files need, nothing more):
  * WRITER: superblock version 0, one root group in the classic form (symbol-table message -> v1 B-tree -> one symbol-table node ->
    local heap; at most 8 datasets), version-1 object headers, CONTIGUOUS little-endian float16/32/64 and (u)int8/16/32/64 datasets,
    scalar root attributes: str (variable-length UTF-8 through one global-heap collection -- what h5py writes for a Python str),
    float (IEEE f64), int / bool (i64).  The result is read back by libhdf5 / h5py / h5dump (tests/test_cpu_h5io.py checks that
    against libhdf5 where it exists).
  * READER: files of that form and what h5py writes with its defaults (libver "earliest"): superblock 0 / 1, v1 object headers with
    continuation blocks, classic groups, contiguous / compact / chunked (v1 chunk B-tree; deflate, shuffle, fletcher32 filters)
    datasets of integer / float / fixed-string type, root attributes of numeric, fixed-string and variable-length-string type.
    Anything else (superblock 2 / 3 with "latest" object headers, compound types, ...) raises `Unsupported` naming the construct, so
    the caller can tell the user to install h5py.

Format reference: "HDF5 File Format Specification Version 3.0" (The HDF Group) -- section numbers in the comments.
"""
from __future__ import annotations

import struct
import zlib
from pathlib import Path

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b"\x89HDF\r\n\x1a\n"


class Unsupported(NotImplementedError):
    pass


def _pad8(n: int) -> int:
    return (n + 7) & ~7


# =====================================================================================================================================
# datatype / dataspace messages (IV.A.2.d, IV.A.2.b)
# =====================================================================================================================================
_FLOAT_PROPS = {2: (15, 10, 5, 0, 10, 15), 4: (31, 23, 8, 0, 23, 127), 8: (63, 52, 11, 0, 52, 1023)}     # sign, exp loc, exp size, mant loc, mant size, bias


def _dtype_msg(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.byteorder == ">":
        raise Unsupported("big-endian arrays: convert with .astype(dt.newbyteorder('<')) first")
    if dt.kind == "f" and dt.itemsize in _FLOAT_PROPS:
        sign, eloc, esz, mloc, msz, bias = _FLOAT_PROPS[dt.itemsize]
        head = struct.pack("<BBBBI", 0x10 | 1, 0x20, sign, 0, dt.itemsize)             # version 1, class 1; mantissa normalisation: implied msb
        return head + struct.pack("<HHBBBBI", 0, 8 * dt.itemsize, eloc, esz, mloc, msz, bias)
    if dt.kind in "iu":
        head = struct.pack("<BBBBI", 0x10 | 0, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize)
        return head + struct.pack("<HH", 0, 8 * dt.itemsize)
    raise Unsupported(f"dtype {dt}")


# variable-length string, null-terminated, UTF-8; base type = one 8-bit character (libhdf5 encodes H5T_C_S1's element as an unsigned byte)
_VLEN_STR = struct.pack("<BBBBI", 0x10 | 9, 0x01, 0x01, 0, 16) + _dtype_msg(np.dtype("u1"))


def _space_msg(shape: tuple) -> bytes:
    return struct.pack("<BBBB4x", 1, len(shape), 0, 0) + b"".join(struct.pack("<Q", int(s)) for s in shape)


def _msg(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = data + b"\0" * (_pad8(len(data)) - len(data))
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(msgs: list[bytes]) -> bytes:
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body        # version 1, #messages, reference count, header size, pad to 8


# =====================================================================================================================================
# writer
# =====================================================================================================================================
def write(path, datasets: dict, attrs: dict) -> None:
    """datasets: name -> array (written contiguous, C order); attrs: name -> str | float | int | bool on the root group."""
    if len(datasets) > 8:
        raise Unsupported("more than 8 datasets in the root group (one symbol-table node)")
    names = sorted(datasets, key=lambda s: s.encode())                        # symbol-table entries are kept in strcmp order
    arrays = {k: np.ascontiguousarray(datasets[k]) for k in names}
    # ---- local heap data segment: "" at offset 0, then the link names
    heap_data = bytearray(b"\0" * 8)
    name_off = {}
    for k in names:
        name_off[k] = len(heap_data)
        b = k.encode("utf-8") + b"\0"
        heap_data += b + b"\0" * (_pad8(len(b)) - len(b))
    free_off = len(heap_data)
    heap_data += struct.pack("<QQ", 1, 32) + b"\0" * 16                      # one free block (next = 1: none), 32 bytes
    # ---- global heap collection holding the attribute strings
    gcol_objs = bytearray()
    str_index = {}
    for k, v in attrs.items():
        if isinstance(v, str):
            b = v.encode("utf-8")
            idx = len(str_index) + 1
            str_index[k] = (idx, len(b))
            gcol_objs += struct.pack("<HH4xQ", idx, 1, len(b)) + b + b"\0" * (_pad8(len(b)) - len(b))
    gcol_size = max(4096, _pad8(16 + len(gcol_objs) + 16))
    # ---- addresses.  Layout: superblock | root header | B-tree | SNOD | heap header | heap data | [GCOL] | dataset headers | raw data
    SB, BT, SN, HH = 96, 24 + 33 * 8 + 32 * 8, 8 + 8 * 40, 32

    def attr_msgs(gcol_addr: int) -> list[bytes]:
        out = []
        for k, v in attrs.items():
            nm = k.encode("utf-8") + b"\0"
            if isinstance(v, str):
                idx, ln = str_index[k]
                t, data = _VLEN_STR, struct.pack("<IQI", ln, gcol_addr, idx)
            elif isinstance(v, (bool, int, np.integer)):
                t, data = _dtype_msg(np.dtype("<i8")), struct.pack("<q", int(v))
            elif isinstance(v, (float, np.floating)):
                t, data = _dtype_msg(np.dtype("<f8")), struct.pack("<d", float(v))
            else:
                raise TypeError(f"attribute {k}: unsupported type {type(v)}")
            s = _space_msg(())
            body = struct.pack("<BxHHH", 1, len(nm), len(t), len(s))
            body += nm + b"\0" * (_pad8(len(nm)) - len(nm)) + t + b"\0" * (_pad8(len(t)) - len(t)) + s + b"\0" * (_pad8(len(s)) - len(s)) + data
            out.append(_msg(0x000C, body))
        return out

    root_len = len(_object_header([_msg(0x0011, b"\0" * 16)] + attr_msgs(0)))
    a_root = SB
    a_bt = _pad8(a_root + root_len)
    a_sn = a_bt + BT
    a_hh = a_sn + SN
    a_hd = a_hh + HH
    a_gc = _pad8(a_hd + len(heap_data))
    pos = a_gc + (gcol_size if str_index else 0)
    ds_hdr, ds_addr = {}, {}
    for k in names:                                                           # dataset headers first (their size does not depend on addresses)
        ds_addr[k] = pos
        pos += len(_object_header([_msg(0x0001, _space_msg(arrays[k].shape)), _msg(0x0003, _dtype_msg(arrays[k].dtype), 1),
                                   _msg(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)), _msg(0x0008, struct.pack("<BBQQ", 3, 1, 0, 0))]))
    raw_addr = {}
    for k in names:
        pos = _pad8(pos)
        raw_addr[k] = pos if arrays[k].nbytes else UNDEF
        pos += arrays[k].nbytes
    eof = pos
    for k in names:
        a = arrays[k]
        ds_hdr[k] = _object_header([_msg(0x0001, _space_msg(a.shape)), _msg(0x0003, _dtype_msg(a.dtype), 1),
                                    _msg(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),
                                    _msg(0x0008, struct.pack("<BBQQ", 3, 1, raw_addr[k], a.nbytes))])
    # ---- assemble
    out = bytearray(eof)
    sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0) + struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, a_root, 1, 0) + struct.pack("<QQ", a_bt, a_hh)          # root symbol-table entry, cached B-tree / heap addresses
    assert len(sb) == SB
    out[0:SB] = sb
    root = _object_header([_msg(0x0011, struct.pack("<QQ", a_bt, a_hh))] + attr_msgs(a_gc))
    assert len(root) == root_len
    out[a_root:a_root + root_len] = root
    # B-tree node (III.A.1): one leaf-level node pointing to the symbol-table node; key 0 = "", key 1 = the largest name
    bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF)
    bt += struct.pack("<QQQ", 0, a_sn, name_off[names[-1]] if names else 0)
    out[a_bt:a_bt + len(bt)] = bt
    sn = b"SNOD" + struct.pack("<BxH", 1, len(names))
    for k in names:
        sn += struct.pack("<QQII16x", name_off[k], ds_addr[k], 0, 0)
    out[a_sn:a_sn + len(sn)] = sn
    out[a_hh:a_hh + HH] = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, a_hd)
    out[a_hd:a_hd + len(heap_data)] = heap_data
    if str_index:
        gc = b"GCOL" + struct.pack("<B3xQ", 1, gcol_size) + bytes(gcol_objs)
        gc += struct.pack("<HH4xQ", 0, 0, gcol_size - len(gc))                          # object 0: the free space (size includes its header)
        out[a_gc:a_gc + len(gc)] = gc
    for k in names:
        out[ds_addr[k]:ds_addr[k] + len(ds_hdr[k])] = ds_hdr[k]
        if arrays[k].nbytes:
            out[raw_addr[k]:raw_addr[k] + arrays[k].nbytes] = arrays[k].tobytes()
    Path(path).write_bytes(bytes(out))


# =====================================================================================================================================
# reader
# =====================================================================================================================================
class _File:
    def __init__(self, path):
        self.b = Path(path).read_bytes()
        b = self.b
        if b[:8] != SIG:
            raise ValueError(f"{path}: not an HDF5 file (no signature at offset 0; user blocks are not supported)")
        ver = b[8]
        if ver not in (0, 1):
            raise Unsupported(f"HDF5 superblock version {ver} (written with libver='latest'?)")
        if b[13] != 8 or b[14] != 8:
            raise Unsupported(f"offset / length size {b[13]} / {b[14]} (only 8 / 8)")
        p = 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from("<Q", b, p)[0]
        ste = p + 32
        self.root_header = struct.unpack_from("<Q", b, ste + 8)[0]
        self._gcol: dict = {}

    # ---- object header, version 1 (IV.A.1.a)
    def messages(self, addr: int) -> list[tuple[int, bytes]]:
        b = self.b
        if b[addr:addr + 4] == b"OHDR":
            raise Unsupported("version-2 object headers (file written with libver='latest')")
        ver, nmsg, _ref, size = struct.unpack_from("<BxHII", b, addr)
        if ver != 1:
            raise Unsupported(f"object header version {ver}")
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(out) < nmsg:
                mt, ms, _fl = struct.unpack_from("<HHB", b, p)
                data = b[p + 8:p + 8 + ms]
                p += 8 + ms
                if mt == 0x0010:                                              # continuation
                    ca, cl = struct.unpack_from("<QQ", data)
                    blocks.append((ca, cl))
                out.append((mt, data))
        return out

    # ---- classic group: B-tree v1 of symbol-table nodes + local heap (III.A.1, III.C, III.D)
    def links(self, btree: int, heap: int) -> dict[str, int]:
        b = self.b
        if b[heap:heap + 4] != b"HEAP":
            raise ValueError("bad local heap signature")
        seg = struct.unpack_from("<Q", b, heap + 24)[0]
        out: dict[str, int] = {}

        def name(off: int) -> str:
            e = b.index(b"\0", seg + off)
            return b[seg + off:e].decode("utf-8")

        def walk(a: int):
            if b[a:a + 4] == b"TREE":
                _t, _lvl, n = struct.unpack_from("<BBH", b, a + 4)
                for i in range(n):
                    walk(struct.unpack_from("<Q", b, a + 24 + 8 + 16 * i)[0])
            elif b[a:a + 4] == b"SNOD":
                n = struct.unpack_from("<H", b, a + 6)[0]
                for i in range(n):
                    no, oh = struct.unpack_from("<QQ", b, a + 8 + 40 * i)
                    out[name(no)] = oh
            else:
                raise ValueError("bad group B-tree node")

        walk(btree)
        return out

    # ---- datatype (IV.A.2.d) -> numpy dtype | ("vlen_str",) | ("str", n)
    def dtype(self, d: bytes):
        cls, ver = d[0] & 15, d[0] >> 4
        bf = d[1] | (d[2] << 8) | (d[3] << 16)
        size = struct.unpack_from("<I", d, 4)[0]
        order = ">" if bf & 1 else "<"
        if cls == 0:
            return np.dtype(f"{order}{'i' if bf & 8 else 'u'}{size}")
        if cls == 1:
            if size not in (2, 4, 8):
                raise Unsupported(f"{size}-byte float")
            return np.dtype(f"{order}f{size}")
        if cls == 3:
            return ("str", size)
        if cls == 9 and (bf & 15) == 1:
            return ("vlen_str",)
        if cls == 8:                                                          # enum (h5py stores numpy bool as an enum over int8): the base type
            return self.dtype(d[8:])
        raise Unsupported(f"datatype class {cls} (version {ver})")

    @staticmethod
    def shape(d: bytes) -> tuple:
        ver, rank = d[0], d[1]
        if ver == 1:
            return tuple(struct.unpack_from("<Q", d, 8 + 8 * i)[0] for i in range(rank))
        if ver == 2:
            if d[3] == 2:
                raise Unsupported("null dataspace")
            return tuple(struct.unpack_from("<Q", d, 4 + 8 * i)[0] for i in range(rank))
        raise Unsupported(f"dataspace version {ver}")

    def gheap(self, addr: int, index: int) -> bytes:
        if addr not in self._gcol:
            b = self.b
            if b[addr:addr + 4] != b"GCOL":
                raise ValueError("bad global heap signature")
            size = struct.unpack_from("<Q", b, addr + 8)[0]
            objs, p = {}, addr + 16
            while p + 16 <= addr + size:
                idx, _rc, n = struct.unpack_from("<HH4xQ", b, p)
                if idx == 0:
                    break
                objs[idx] = b[p + 16:p + 16 + n]
                p += 16 + _pad8(n)
            self._gcol[addr] = objs
        return self._gcol[addr][index]

    def value(self, t, shape: tuple, raw: bytes):
        n = int(np.prod(shape)) if shape else 1
        if isinstance(t, np.dtype):
            a = np.frombuffer(raw, dtype=t, count=n).reshape(shape)
            return a.astype(t.newbyteorder("="), copy=True)
        if t[0] == "str":
            items = [raw[i * t[1]:(i + 1) * t[1]].split(b"\0")[0].decode("utf-8") for i in range(n)]
        else:
            items = []
            for i in range(n):
                ln, ga, gi = struct.unpack_from("<IQI", raw, 16 * i)
                items.append(self.gheap(ga, gi)[:ln].decode("utf-8") if ln else "")
        return items[0] if not shape else np.array(items, dtype=object).reshape(shape)

    def attribute(self, d: bytes):
        ver = d[0]
        if ver == 1:
            ns, ts, ss = struct.unpack_from("<HHH", d, 2)
            p = 8
            name = d[p:p + ns].split(b"\0")[0].decode("utf-8"); p += _pad8(ns)
            t = self.dtype(d[p:p + ts]); p += _pad8(ts)
            shp = self.shape(d[p:p + ss]); p += _pad8(ss)
        elif ver in (2, 3):
            ns, ts, ss = struct.unpack_from("<HHH", d, 2)
            p = 8 + (1 if ver == 3 else 0)
            name = d[p:p + ns].split(b"\0")[0].decode("utf-8"); p += ns
            t = self.dtype(d[p:p + ts]); p += ts
            shp = self.shape(d[p:p + ss]); p += ss
        else:
            raise Unsupported(f"attribute message version {ver}")
        v = self.value(t, shp, d[p:])
        if isinstance(v, np.ndarray) and v.shape == ():
            v = v.item()
        return name, v

    # ---- dataset
    def dataset(self, addr: int) -> np.ndarray:
        b = self.b
        t = shp = layout = None
        filters: list[tuple[int, tuple]] = []
        for mt, d in self.messages(addr):
            if mt == 0x0001:
                shp = self.shape(d)
            elif mt == 0x0003:
                t = self.dtype(d)
            elif mt == 0x0008:
                layout = d
            elif mt == 0x000B:
                filters = self._filters(d)
        if t is None or shp is None or layout is None:
            raise ValueError("dataset without datatype / dataspace / layout message")
        if not isinstance(t, np.dtype):
            if t[0] == "str":
                t = np.dtype(f"S{t[1]}")
            else:
                raise Unsupported("variable-length string datasets")
        n = int(np.prod(shp)) if shp else 1
        if layout[0] != 3:
            raise Unsupported(f"data layout message version {layout[0]}")
        cls = layout[1]
        if cls == 0:                                                          # compact
            sz = struct.unpack_from("<H", layout, 2)[0]
            raw = layout[4:4 + sz]
        elif cls == 1:                                                        # contiguous
            a, sz = struct.unpack_from("<QQ", layout, 2)
            raw = b"\0" * (n * t.itemsize) if a == UNDEF else b[self.base + a:self.base + a + n * t.itemsize]
        elif cls == 2:
            return self._chunked(layout, t, shp, filters)
        else:
            raise Unsupported(f"layout class {cls}")
        return np.frombuffer(raw, dtype=t, count=n).reshape(shp).astype(t.newbyteorder("="), copy=True)

    @staticmethod
    def _filters(d: bytes) -> list[tuple[int, tuple]]:
        ver, nf = d[0], d[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(nf):
            fid, nlen = struct.unpack_from("<HH", d, p) if (ver == 1 or struct.unpack_from("<H", d, p)[0] >= 256) else (struct.unpack_from("<H", d, p)[0], 0)
            if ver == 1 or fid >= 256:
                _flags, ncv = struct.unpack_from("<HH", d, p + 4)
                p += 8
                p += _pad8(nlen) if ver == 1 else nlen
            else:
                _flags, ncv = struct.unpack_from("<HH", d, p + 2)
                p += 6
            cv = struct.unpack_from(f"<{ncv}I", d, p)
            p += 4 * ncv
            if ver == 1 and ncv % 2:
                p += 4
            out.append((fid, cv))
        return out

    def _chunked(self, layout: bytes, t: np.dtype, shp: tuple, filters) -> np.ndarray:
        b = self.b
        nd = layout[2]                                                        # rank + 1
        bt = struct.unpack_from("<Q", layout, 3)[0]
        cdims = struct.unpack_from(f"<{nd}I", layout, 11)[:-1]
        out = np.zeros(shp, dtype=t.newbyteorder("="))
        if bt == UNDEF:
            return out

        def walk(a: int):
            if b[a:a + 4] != b"TREE":
                raise ValueError("bad chunk B-tree node")
            ntype, lvl, n = struct.unpack_from("<BBH", b, a + 4)
            if ntype != 1:
                raise ValueError("not a chunk B-tree")
            ksz = 8 + 8 * nd
            p = a + 24
            for _ in range(n):
                csize, fmask = struct.unpack_from("<II", b, p)
                offs = struct.unpack_from(f"<{nd}Q", b, p + 8)[:-1]
                child = struct.unpack_from("<Q", b, p + ksz)[0]
                p += ksz + 8
                if lvl > 0:
                    walk(child)
                    continue
                raw = b[self.base + child:self.base + child + csize]
                for i, (fid, cv) in reversed(list(enumerate(filters))):
                    if fmask & (1 << i):
                        continue
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:
                        es = cv[0] if cv else t.itemsize
                        raw = np.frombuffer(raw, np.uint8).reshape(es, -1).T.tobytes()
                    elif fid == 3:
                        raw = raw[:-4]
                    else:
                        raise Unsupported(f"filter id {fid}")
                chunk = np.frombuffer(raw, dtype=t, count=int(np.prod(cdims))).reshape(cdims)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shp))
                out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]

        walk(bt)
        return out


def read(path, want=None) -> tuple[dict[str, np.ndarray], dict]:
    """(datasets of the root group [only the names in `want` if given], root attributes)."""
    f = _File(path)
    attrs, bt = {}, None
    for mt, d in f.messages(f.root_header):
        if mt == 0x0011:
            bt = struct.unpack_from("<QQ", d)
        elif mt == 0x000C:
            k, v = f.attribute(d)
            attrs[k] = v
        elif mt in (0x0002, 0x0006, 0x0015):
            raise Unsupported("new-style group (link messages): file written with libver='latest'")
    if bt is None:
        raise Unsupported("root group without a symbol-table message")
    links = f.links(*bt)
    data = {k: f.dataset(a) for k, a in links.items() if want is None or k in want}
    return data, attrs
