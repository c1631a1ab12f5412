"""Reader for tests/golden/ref_vectors.bin (written by oracle/ref_vectors.cc)."""
import struct

import numpy as np

_DT = {b"d": np.float64, b"i": np.int32, b"u": np.uint32, b"f": np.float32}


def read_vectors(path):
    out = {}
    with open(path, "rb") as f:
        assert f.read(4) == b"FJGV"
        while True:
            h = f.read(4)
            if len(h) < 4:
                break
            (nl,) = struct.unpack("<I", h)
            name = f.read(nl).decode("ascii")
            dt = _DT[f.read(1)]
            (nd,) = struct.unpack("<I", f.read(4))
            dims = struct.unpack("<%dI" % nd, f.read(4 * nd))
            n = int(np.prod(dims)) if nd else 1
            out[name] = np.frombuffer(f.read(n * np.dtype(dt).itemsize), dtype=dt).reshape(dims).copy()
    return out
