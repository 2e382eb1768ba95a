"""Re-encode the reference's bundled datasets (data/*.RData) as plain-text fixtures.

Run once in the build container (needs /root/reference); the outputs are committed and
the tests never touch /root/reference.  The .RData files are gzip'd R serialisation
(RDX3, XDR): a pairlist with one INTSXP matrix.  Only bytes are parsed here -- nothing
is executed from the file.  Output: one row per observation, P characters '0'/'1'.
"""
import gzip
import os
import struct

import numpy as np

SRC = "/root/reference/data"
HERE = os.path.dirname(os.path.abspath(__file__))


def decode(name):
    raw = gzip.open(os.path.join(SRC, name + ".RData"), "rb").read()
    assert raw[:7] == b"RDX3\nX\n"
    off = raw.index(name.encode()) + len(name)
    flags, n = struct.unpack(">II", raw[off:off + 8])
    assert flags & 0xFF == 13  # INTSXP
    data = np.frombuffer(raw[off + 8:off + 8 + 4 * n], dtype=">i4").astype(np.int32)
    tail = raw[off + 8 + 4 * n:]
    k = tail.index(b"dim") + 3
    ty, ln, nr, nc = struct.unpack(">IIII", tail[k:k + 16])
    assert ty & 0xFF == 13 and ln == 2 and nr * nc == n
    return data.reshape((nr, nc), order="F")


for name in ("K2_N100_P5", "K2_N1000_P5", "K3_N1000_P5"):
    X = decode(name)
    assert set(np.unique(X)) <= {0, 1}
    with open(os.path.join(HERE, name + ".txt"), "w") as f:
        for row in X:
            f.write("".join(str(int(v)) for v in row) + "\n")
    print(name, X.shape, int(X.sum()), X.mean(axis=0).round(3))
