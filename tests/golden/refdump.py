"""Reader for the oracle harness dump format (oracle/ref/driver.F90, module oracle_dump):
records of  name(32 bytes) kind(i4: 8=real64, 4=int32) ndim(i4) dims(3 x i4) data (column-major)."""
import numpy as np


def read_dump(path, want=None, tolerate_truncation=False):
    out = {}
    with open(path, "rb") as f:
        buf = f.read()
    pos, n = 0, len(buf)
    while pos + 52 <= n:
        name = buf[pos:pos + 32].decode().strip()
        kind, ndim, d1, d2, d3 = np.frombuffer(buf, dtype="<i4", count=5, offset=pos + 32)
        pos += 52
        cnt = int(d1) * int(d2) * int(d3)
        dt = np.dtype("<f8") if kind == 8 else np.dtype("<i4")
        nbytes = cnt * dt.itemsize
        if pos + nbytes > n:
            if tolerate_truncation:
                break
            raise ValueError(f"truncated record {name} in {path}")
        if want is None or name in want:
            a = np.frombuffer(buf, dtype=dt, count=cnt, offset=pos)
            dims = [int(d1), int(d2), int(d3)][:int(ndim)]
            out[name] = a.reshape(dims[::-1])      # C view of the Fortran array: last Fortran index first
        else:
            out.setdefault("_skipped", []).append(name)
        pos += nbytes
    return out
