"""TEST INFRASTRUCTURE - not a product path.  PARITY UNPINNED (see oracle/README.md).

Pure-Python/numpy restatement of the Kaldi table/object I/O that the hot path touches
(`SequentialBaseFloatMatrixReader` on the feature rspecifier and `BaseFloatVectorWriter` on the
`ark,scp:` wspecifier, reference call sites: egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:79,93).

Kaldi itself is not vendored under /root/reference; the binary layouts below are the
published Kaldi formats.  The *text* matrix form is the one the reference's own
Python helpers read and write (egs/sre/v2/steps/libs/common.py:354-470), which is the only in-tree pin.

Used by tests/ as an independent implementation against which the C++ `kio` layer is checked, and to
write fixtures.  Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import it.
"""
import io
import struct

import numpy as np


# ----------------------------------------------------------------------------- primitives
def write_token(f, tok, binary=True):
    f.write(tok.encode() + b" ")


def read_token(f):
    tok = b""
    while True:
        c = f.read(1)
        if c == b"":
            break
        if c.isspace():
            if tok:
                break
            continue
        tok += c
    return tok.decode()


def peek(f, n=1):
    pos = f.tell()
    b = f.read(n)
    f.seek(pos)
    return b


def write_int32(f, v, binary=True):
    if binary:
        f.write(b"\x04" + struct.pack("<i", v))
    else:
        f.write(("%d " % v).encode())


def write_float(f, v, binary=True):
    if binary:
        f.write(b"\x04" + struct.pack("<f", v))
    else:
        f.write(("%.9g " % v).encode())


def write_double(f, v, binary=True):
    if binary:
        f.write(b"\x08" + struct.pack("<d", v))
    else:
        f.write(("%.17g " % v).encode())


def write_bool(f, v, binary=True):
    f.write(b"T" if v else b"F")
    if not binary:
        f.write(b" ")


def read_basic(f, binary=True):
    """Self-describing scalar: int32/float share the \\x04 size byte, double has \\x08."""
    if binary:
        n = f.read(1)[0]
        return f.read(n)
    return read_token(f)


def read_int32(f, binary=True):
    if binary:
        b = read_basic(f)
        assert len(b) == 4
        return struct.unpack("<i", b)[0]
    return int(read_token(f))


def read_float_or_double(f, binary=True):
    if binary:
        b = read_basic(f)
        return struct.unpack("<f" if len(b) == 4 else "<d", b)[0]
    return float(read_token(f))


def read_bool(f, binary=True):
    if binary:
        c = f.read(1)
    else:
        c = read_token(f).encode()
    assert c in (b"T", b"F"), c
    return c == b"T"


# ----------------------------------------------------------------------------- vectors / matrices
def write_vector(f, v, binary=True, double=False):
    v = np.asarray(v)
    if binary:
        dt = "<f8" if double else "<f4"
        f.write(b"DV " if double else b"FV ")
        f.write(b"\x04" + struct.pack("<i", v.shape[0]))
        f.write(v.astype(dt).tobytes())
    else:
        f.write((" [ " + " ".join("%.9g" % x for x in v) + " ]\n").encode())


def write_matrix(f, m, binary=True, double=False):
    m = np.asarray(m)
    assert m.ndim == 2
    if binary:
        dt = "<f8" if double else "<f4"
        f.write(b"DM " if double else b"FM ")
        f.write(b"\x04" + struct.pack("<i", m.shape[0]))
        f.write(b"\x04" + struct.pack("<i", m.shape[1]))
        f.write(np.ascontiguousarray(m, dtype=dt).tobytes())
    else:
        if m.shape[0] == 0:
            f.write(b" [ ]\n")
            return
        f.write(b" [\n")
        for i, row in enumerate(m):
            f.write(("  " + " ".join("%.9g" % x for x in row)).encode())
            f.write(b" ]\n" if i == m.shape[0] - 1 else b"\n")


def _read_text_vector_or_matrix(f):
    # after optional whitespace: '[' ... ']' with rows separated by newlines
    c = f.read(1)
    while c.isspace():
        c = f.read(1)
    assert c == b"[", c
    rows, cur, tok = [], [], b""
    while True:
        c = f.read(1)
        assert c != b"", "EOF in text matrix"
        if c in b" \t\r\n]":
            if tok:
                cur.append(float(tok))
                tok = b""
            if c == b"\n" or c == b"]":
                if cur:
                    rows.append(cur)
                    cur = []
            if c == b"]":
                break
        else:
            tok += c
    # consume the rest of the line
    nxt = peek(f)
    if nxt == b"\n":
        f.read(1)
    return rows


def read_vector(f, binary=True):
    if binary:
        tag = read_token(f)
        assert tag in ("FV", "DV"), tag
        n = read_int32(f)
        dt = "<f4" if tag == "FV" else "<f8"
        return np.frombuffer(f.read(n * int(dt[2])), dtype=dt).astype(np.float64 if tag == "DV" else np.float32)
    rows = _read_text_vector_or_matrix(f)
    return np.array(rows[0] if rows else [], dtype=np.float32)


def _uint16_to_float(h_min, h_range, v):
    return h_min + h_range * 1.52590218966964e-05 * v


def _read_compressed(f, tag):
    h_min, h_range, rows, cols = struct.unpack("<ffii", f.read(16))
    if tag == "CM":
        hdr = np.frombuffer(f.read(cols * 8), dtype="<u2").reshape(cols, 4).astype(np.float64)
        data = np.frombuffer(f.read(rows * cols), dtype=np.uint8).reshape(cols, rows).astype(np.float64)
        p0 = _uint16_to_float(h_min, h_range, hdr[:, 0])[:, None]
        p25 = _uint16_to_float(h_min, h_range, hdr[:, 1])[:, None]
        p75 = _uint16_to_float(h_min, h_range, hdr[:, 2])[:, None]
        p100 = _uint16_to_float(h_min, h_range, hdr[:, 3])[:, None]
        out = np.where(data <= 64, p0 + (p25 - p0) * data * (1 / 64.0),
                       np.where(data <= 192, p25 + (p75 - p25) * (data - 64) * (1 / 128.0),
                                p75 + (p100 - p75) * (data - 192) * (1 / 63.0)))
        return out.T.astype(np.float32)
    if tag == "CM2":
        data = np.frombuffer(f.read(rows * cols * 2), dtype="<u2").reshape(rows, cols).astype(np.float64)
        return (h_min + h_range * (1.0 / 65535.0) * data).astype(np.float32)
    if tag == "CM3":
        data = np.frombuffer(f.read(rows * cols), dtype=np.uint8).reshape(rows, cols).astype(np.float64)
        return (h_min + h_range * (1.0 / 255.0) * data).astype(np.float32)
    raise ValueError(tag)


def read_matrix(f, binary=True):
    if binary:
        tag = read_token(f)
        if tag in ("CM", "CM2", "CM3"):
            return _read_compressed(f, tag)
        assert tag in ("FM", "DM"), tag
        r = read_int32(f)
        c = read_int32(f)
        dt = "<f4" if tag == "FM" else "<f8"
        return np.frombuffer(f.read(r * c * int(dt[2])), dtype=dt).reshape(r, c).astype(np.float32)
    rows = _read_text_vector_or_matrix(f)
    if not rows:
        return np.zeros((0, 0), dtype=np.float32)
    return np.array(rows, dtype=np.float32)


def write_compressed_matrix(f, m, method="CM2"):
    """Writer for compressed matrices (fixtures only): CM (per-column percentiles + uint8),
    CM2 (uint16), CM3 (uint8).  Returns the matrix a conforming reader must reconstruct."""
    m = np.asarray(m, dtype=np.float32)
    rows, cols = m.shape
    h_min = float(m.min())
    h_range = float(m.max() - m.min())
    if h_range == 0:
        h_range = 1.0
    buf = io.BytesIO()
    buf.write(method.encode() + b" ")
    buf.write(struct.pack("<ffii", h_min, h_range, rows, cols))
    if method == "CM2":
        q = np.clip(np.floor((m - h_min) / h_range * 65535.0 + 0.499), 0, 65535).astype("<u2")
        buf.write(q.tobytes())
    elif method == "CM3":
        q = np.clip(np.floor((m - h_min) / h_range * 255.0 + 0.499), 0, 255).astype(np.uint8)
        buf.write(q.tobytes())
    elif method == "CM":
        def f2u(x):
            return np.clip(np.floor((x - h_min) / h_range * 65535.0 + 0.499), 0, 65535).astype(np.int64)
        hdr = np.zeros((cols, 4), dtype="<u2")
        data = np.zeros((cols, rows), dtype=np.uint8)
        for c in range(cols):
            col = np.sort(m[:, c])
            q = [col[0], col[rows // 4], col[(3 * rows) // 4], col[-1]]
            u = f2u(np.array(q))
            # percentiles must be strictly increasing so that the byte code is invertible
            u[1] = min(max(u[1], u[0] + 1), 65533)
            u[2] = min(max(u[2], u[1] + 1), 65534)
            u[3] = max(u[3], u[2] + 1)
            hdr[c] = u
            p = _uint16_to_float(h_min, h_range, u.astype(np.float64))
            v = m[:, c].astype(np.float64)
            b = np.where(v < p[1], np.clip(np.floor((v - p[0]) / (p[1] - p[0]) * 64 + 0.5), 0, 64),
                         np.where(v < p[2], np.clip(np.floor((v - p[1]) / (p[2] - p[1]) * 128 + 64.5), 64, 192),
                                  np.clip(np.floor((v - p[2]) / (p[3] - p[2]) * 63 + 192.5), 192, 255)))
            data[c] = b.astype(np.uint8)
        buf.write(hdr.tobytes())
        buf.write(data.tobytes())
    else:
        raise ValueError(method)
    f.write(buf.getvalue())
    buf.seek(len(method) + 1)
    return _read_compressed(buf, method)


# ----------------------------------------------------------------------------- archives
def write_ark_matrices(path, items, binary=True, scp_path=None, compressed=None):
    """items: iterable of (key, 2-D array). Returns {key: byte offset of the object}."""
    offs = {}
    with open(path, "wb") as f:
        for key, m in items:
            f.write(key.encode() + b" ")
            offs[key] = f.tell()
            if binary:
                f.write(b"\x00B")
                if compressed:
                    write_compressed_matrix(f, m, compressed)
                else:
                    write_matrix(f, m, True)
            else:
                write_matrix(f, m, False)
    if scp_path:
        with open(scp_path, "w") as s:
            for k, o in offs.items():
                s.write("%s %s:%d\n" % (k, path, o))
    return offs


def write_ark_vectors(path, items, binary=True, scp_path=None):
    offs = {}
    with open(path, "wb") as f:
        for key, v in items:
            f.write(key.encode() + b" ")
            offs[key] = f.tell()
            if binary:
                f.write(b"\x00B")
            write_vector(f, v, binary)
    if scp_path:
        with open(scp_path, "w") as s:
            for k, o in offs.items():
                s.write("%s %s:%d\n" % (k, path, o))
    return offs


def _read_key(f):
    key = b""
    while True:
        c = f.read(1)
        if c == b"":
            return None
        if c.isspace():
            if key:
                return key.decode()
            continue
        key += c


def read_ark(path_or_file, kind="matrix"):
    """Yield (key, array) from a Kaldi archive of matrices or vectors (binary or text)."""
    f = open(path_or_file, "rb") if isinstance(path_or_file, str) else path_or_file
    try:
        while True:
            key = _read_key(f)
            if key is None:
                return
            binary = peek(f, 2) == b"\x00B"
            if binary:
                f.read(2)
            obj = read_matrix(f, binary) if kind == "matrix" else read_vector(f, binary)
            yield key, obj
    finally:
        if isinstance(path_or_file, str):
            f.close()


def read_scp(path, kind="matrix"):
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        key, rx = line.split(None, 1)
        fn, off = rx.rsplit(":", 1) if ":" in rx else (rx, "0")
        with open(fn, "rb") as f:
            f.seek(int(off))
            binary = peek(f, 2) == b"\x00B"
            if binary:
                f.read(2)
            yield key, (read_matrix(f, binary) if kind == "matrix" else read_vector(f, binary))
