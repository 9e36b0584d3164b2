"""A minimal reader for the netCDF-4 (HDF5) files of the reference's `in_fmt SeaNetCDF` inputs (data/3D/large_tests/white_sea):
root group with compact links -> datasets with a version-2 object header, IEEE little-endian floats, chunked layout (version-1
B-tree index) and the deflate / shuffle filters.  That is what the netCDF-4 library writes for such a file; anything else raises.
The reference reads these files through libnetcdf (Grid3D::LoadNetCDF, FluidSolver3D/Grid3D.cpp:433-486); neither libnetcdf nor
libhdf5 exists in this image.  Python twin of cmc_fluid_solver_amd/host/Hdf5Min.h.  Format: the HDF5 File Format Specification
version 3.0 (superblock version 2/3, object header version 2, data layout message version 3, B-tree version 1 node type 1).
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(ValueError):
    pass


class Hdf5File:
    def __init__(self, path):
        self.d = open(path, "rb").read()
        d = self.d
        if d[:8] != b"\x89HDF\r\n\x1a\n":
            raise Hdf5Error("not an HDF5 file (a classic netCDF file starts with 'CDF')")
        if d[8] not in (2, 3) or d[9] != 8 or d[10] != 8:
            raise Hdf5Error("HDF5 superblock version %d / offset size %d: not supported" % (d[8], d[9]))
        self.base, _, _, root = struct.unpack("<QQQQ", d[12:44])
        self.links = {}
        for t, b in self._messages(root):
            if t == 0x06:
                name, addr = self._link(b)
                if addr is not None:
                    self.links[name] = addr
            elif t == 0x02 and struct.unpack("<Q", b[-16:-8])[0] != UNDEF and not any(m[0] == 0x06 for m in self._messages(root)):
                raise Hdf5Error("root group with dense link storage: not supported")

    # ---- object headers (version 2) ---------------------------------------------------------------------------------
    def _messages(self, addr):
        d = self.d
        addr += self.base
        if d[addr:addr + 4] != b"OHDR" or d[addr + 4] != 2:
            raise Hdf5Error("object header version 2 expected at %d" % addr)
        flags = d[addr + 5]
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        szf = 1 << (flags & 3)
        chunk0 = int.from_bytes(d[p:p + szf], "little")
        p += szf
        out = []

        def walk(p, end):
            while p + 4 <= end:
                t, sz = d[p], struct.unpack("<H", d[p + 1:p + 3])[0]
                p += 4 + (2 if flags & 0x04 else 0)
                body = d[p:p + sz]
                if t == 0x10:                                   # continuation
                    ca, cl = struct.unpack("<QQ", body[:16])
                    ca += self.base
                    if d[ca:ca + 4] != b"OCHK":
                        raise Hdf5Error("bad object header continuation")
                    walk(ca + 4, ca + cl - 4)
                elif t != 0:
                    out.append((t, body))
                p += sz
        walk(p, p + chunk0)
        return out

    @staticmethod
    def _link(b):
        if b[0] != 1:
            raise Hdf5Error("link message version %d" % b[0])
        fl, p = b[1], 2
        ltype = 0
        if fl & 0x08:
            ltype = b[p]; p += 1
        if fl & 0x04:
            p += 8
        if fl & 0x10:
            p += 1
        n = 1 << (fl & 3)
        ln = int.from_bytes(b[p:p + n], "little"); p += n
        name = b[p:p + ln].decode(); p += ln
        return name, (struct.unpack("<Q", b[p:p + 8])[0] if ltype == 0 else None)

    # ---- datasets ---------------------------------------------------------------------------------------------------
    def shape(self, name):
        return self._info(name)[0]

    def _info(self, name):
        if name not in self.links:
            raise Hdf5Error("no dataset %r in the root group" % name)
        dims = dtype = layout = None
        filters = []
        for t, b in self._messages(self.links[name]):
            if t == 0x01:                                       # dataspace
                if b[0] == 2:
                    rank, p = b[1], 4
                elif b[0] == 1:
                    rank, p = b[1], 8
                else:
                    raise Hdf5Error("dataspace message version %d" % b[0])
                dims = struct.unpack("<%dQ" % rank, b[p:p + 8 * rank])
            elif t == 0x03:                                     # datatype
                cls, size = b[0] & 0x0F, struct.unpack("<I", b[4:8])[0]
                if cls != 1 or (b[1] & 1) or size not in (4, 8):
                    raise Hdf5Error("only little-endian IEEE float32 / float64 datasets are supported")
                dtype = np.dtype("<f%d" % size)
            elif t == 0x0B:                                     # filter pipeline
                if b[0] != 2:
                    raise Hdf5Error("filter pipeline message version %d" % b[0])
                p = 2
                for _ in range(b[1]):
                    fid, = struct.unpack("<H", b[p:p + 2]); p += 2
                    if fid >= 256:
                        nl, = struct.unpack("<H", b[p:p + 2]); p += 2
                    else:
                        nl = 0
                    p += 2
                    ncv, = struct.unpack("<H", b[p:p + 2]); p += 2 + nl + 4 * ncv
                    filters.append(fid)
            elif t == 0x08:                                     # data layout
                if b[0] != 3:
                    raise Hdf5Error("data layout message version %d" % b[0])
                layout = b
        if dims is None or dtype is None or layout is None:
            raise Hdf5Error("dataset %r: incomplete object header" % name)
        return dims, dtype, layout, filters

    def read(self, name):
        dims, dtype, lay, filters = self._info(name)
        d = self.d
        rank = len(dims)
        if lay[1] == 1:                                         # contiguous
            addr, size = struct.unpack("<QQ", lay[2:18])
            return np.frombuffer(d[self.base + addr:self.base + addr + size], dtype).reshape(dims).copy()
        if lay[1] != 2:
            raise Hdf5Error("data layout class %d" % lay[1])
        if lay[2] != rank + 1:
            raise Hdf5Error("chunk rank")
        btree, = struct.unpack("<Q", lay[3:11])
        cdims = struct.unpack("<%dI" % (rank + 1), lay[11:11 + 4 * (rank + 1)])[:rank]
        if any(f not in (1, 2) for f in filters):
            raise Hdf5Error("filters %r: only shuffle and deflate are supported" % (filters,))
        if any(n == 0 for n in dims + cdims) or int(np.prod(dims, dtype=object)) > 1 << 28 or int(np.prod(cdims, dtype=object)) > 1 << 28:
            raise Hdf5Error("dataset or chunk too large for this reader (or a corrupt header)")
        out = np.zeros(dims, dtype)
        if btree == UNDEF:
            return out
        for offs, addr, size, mask in self._chunks(btree, rank):
            if self.base + addr + size > len(d):
                raise Hdf5Error("chunk data past the end of the file")
            raw = d[self.base + addr:self.base + addr + size]
            for k, f in reversed(list(enumerate(filters))):     # undo the pipeline back to front
                if mask & (1 << k):
                    continue
                if f == 1:
                    raw = zlib.decompress(raw)
                else:                                           # shuffle: byte planes -> elements
                    es = dtype.itemsize
                    raw = np.frombuffer(raw, np.uint8).reshape(es, -1).T.tobytes()
            chunk = np.frombuffer(raw, dtype).reshape(cdims)
            sel_o = tuple(slice(o, min(o + c, n)) for o, c, n in zip(offs, cdims, dims))
            sel_c = tuple(slice(0, s.stop - s.start) for s in sel_o)
            out[sel_o] = chunk[sel_c]
        return out

    def _chunks(self, addr, rank, parent_level=-1, depth=0):
        d = self.d
        p = self.base + addr
        if depth > 8 or p + 24 > len(d) or d[p:p + 4] != b"TREE" or d[p + 4] != 1:
            raise Hdf5Error("version-1 B-tree of raw data chunks expected")
        level, n = d[p + 5], struct.unpack("<H", d[p + 6:p + 8])[0]
        if parent_level >= 0 and level != parent_level - 1:          # a node that names itself must not recurse for ever
            raise Hdf5Error("B-tree child is not one level below its parent")
        p += 24
        ks = 8 + 8 * (rank + 1)
        if p + n * (ks + 8) > len(d):
            raise Hdf5Error("B-tree node past the end of the file")
        for _ in range(n):
            size, mask = struct.unpack("<II", d[p:p + 8])
            offs = struct.unpack("<%dQ" % rank, d[p + 8:p + 8 + 8 * rank])
            child, = struct.unpack("<Q", d[p + ks:p + ks + 8])
            p += ks + 8
            if level == 0:
                yield offs, child, size, mask
            else:
                yield from self._chunks(child, rank, level, depth + 1)
