"""Read the integer matrices of the package's bundled data sets (`data/*.RData`, what `data(K2_N100_P5)`
loads in R: /root/reference/R/bmm-mcmc.R:10-55) without R.

An .RData file of this kind is gzip'd R serialisation, format "RDX3" in XDR (big-endian): a pairlist
whose tags are the object names.  Only bytes are parsed -- nothing in the file is evaluated -- and only
the item types such a file needs are understood (pairlist, symbol, character, integer / logical / double
vectors with attributes); anything else raises.
"""
import gzip
import struct

import numpy as np

_NIL, _REF = 254, 255
_LIST, _SYM, _CHAR, _LGL, _INT, _REAL, _STR = 2, 1, 9, 10, 13, 14, 16


class _Reader:
    def __init__(self, raw):
        self.b, self.o, self.refs = raw, 0, []

    def int(self):
        v = struct.unpack_from(">i", self.b, self.o)[0]
        self.o += 4
        return v

    def bytes(self, n):
        v = self.b[self.o:self.o + n]
        if len(v) != n:
            raise ValueError("truncated RData stream")
        self.o += n
        return v

    def item(self):
        flags = self.int()
        ty, has_attr, has_tag = flags & 0xFF, bool(flags & 0x200), bool(flags & 0x400)
        if ty == _NIL:
            return None
        if ty == _REF:
            idx = flags >> 8
            if idx == 0:
                idx = self.int()
            return self.refs[idx - 1]
        if ty == _SYM:
            name = self.item()
            self.refs.append(name)
            return name
        if ty == _CHAR:
            n = self.int()
            return None if n == -1 else self.bytes(n).decode("latin-1")
        if ty == _LIST:  # pairlist: [attributes] [tag] car cdr -> list of (tag, value)
            out = []
            while True:
                if has_attr:
                    self.item()
                tag = self.item() if has_tag else None
                out.append((tag, self.item()))
                flags = self.int()
                ty, has_attr, has_tag = flags & 0xFF, bool(flags & 0x200), bool(flags & 0x400)
                if ty == _NIL:
                    return out
                if ty != _LIST:
                    raise ValueError("unexpected item type %d in a pairlist" % ty)
        if ty in (_INT, _LGL, _REAL):
            n = self.int()
            if n == -1:
                raise ValueError("long vectors are not supported")
            dt, size = (">f8", 8) if ty == _REAL else (">i4", 4)
            v = np.frombuffer(self.bytes(n * size), dtype=dt).astype(np.float64 if ty == _REAL else np.int32)
            attrs = dict(self.item()) if has_attr else {}
            return v, attrs
        if ty == _STR:
            n = self.int()
            v = [self.item() for _ in range(n)]
            attrs = dict(self.item()) if has_attr else {}
            return v, attrs
        raise ValueError("RData item type %d is not supported by this reader" % ty)


def read_rdata(path):
    """{object name: array} for the vector objects of an RDX3 .RData file; matrices come back shaped
    (column-major, as R stores them)."""
    raw = gzip.open(path, "rb").read()
    if raw[:5] != b"RDX3\n":
        raise ValueError("not an RDX3 .RData file")
    r = _Reader(raw)
    r.o = 5
    if r.bytes(2) != b"X\n":
        raise ValueError("only XDR (binary big-endian) serialisation is supported")
    version = r.int()
    r.int(); r.int()                      # writer version, minimal reader version
    if version == 3:
        r.bytes(r.int())                  # native encoding name
    elif version != 2:
        raise ValueError("unknown serialisation version %d" % version)
    out = {}
    for tag, val in r.item() or []:
        if isinstance(val, tuple) and isinstance(val[0], np.ndarray):
            v, attrs = val
            dim = attrs.get("dim")
            if dim is not None:
                v = v.reshape(tuple(int(d) for d in dim[0]), order="F")
            out[tag] = v
    return out


def read_rdata_matrix(path, name=None):
    """The N x P integer matrix of a bundled data set, Fortran-ordered int32 -- what gibbs_*() take."""
    objs = read_rdata(path)
    if name is None:
        mats = [k for k, v in objs.items() if v.ndim == 2]
        if len(mats) != 1:
            raise ValueError("expected one matrix in %s, found %s" % (path, sorted(objs)))
        name = mats[0]
    return np.asfortranarray(objs[name], dtype=np.int32)
