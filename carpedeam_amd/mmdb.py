"""Minimal reader/writer for the MMseqs2 on-disk DB format as modified by the
CarpeDeam fork (4-column index: key, offset, length, wasExtended).

Reference: lib/mmseqs/src/commons/DBReader.cpp:773-838 (index parse),
DBWriter.cpp:193-213 (dbtype), :415-427 (index line).  Used by tests, bench.py and
tests/golden/make_golden.py; the product's own reader/writer is C++
(carpedeam_amd/csrc/host/mmdb.cpp).
"""
import os
import struct

DBTYPE_NUCLEOTIDES = 1
DBTYPE_ALIGNMENT_RES = 5
DBTYPE_PREFILTER_REV_RES = 14


def write_db(path, entries, dbtype, ext=None):
    """entries: iterable of (key:int, payload:bytes); payload is written followed by NUL.

    For sequence DBs pass payload = sequence + b"\\n"; length in the index includes the NUL.
    ext: optional dict key -> wasExtended flag (0/1).
    """
    off = 0
    with open(path, "wb") as d, open(path + ".index", "w") as ix:
        for key, payload in entries:
            d.write(payload)
            d.write(b"\0")
            n = len(payload) + 1
            e = 0 if ext is None else int(ext.get(key, 0))
            ix.write("%d\t%d\t%d\t%d\n" % (key, off, n, e))
            off += n
    with open(path + ".dbtype", "wb") as t:
        t.write(struct.pack("<i", dbtype))


def write_seqdb(path, seqs, keys=None, ext=None):
    if keys is None:
        keys = range(len(seqs))
    def gen():
        for k, s in zip(keys, seqs):
            if isinstance(s, str):
                s = s.encode()
            yield k, bytes(s) + b"\n"
    write_db(path, gen(), DBTYPE_NUCLEOTIDES, ext)


def _data_files(path):
    if os.path.exists(path):
        return [path]
    files = []
    i = 0
    while os.path.exists("%s.%d" % (path, i)):
        files.append("%s.%d" % (path, i))
        i += 1
    return files


def read_db(path):
    """Return dict key -> (payload bytes without the trailing NUL, wasExtended)."""
    blobs = [open(f, "rb").read() for f in _data_files(path)]
    data = b"".join(blobs)
    out = {}
    with open(path + ".index") as ix:
        for line in ix:
            c = line.rstrip("\n").split("\t")
            if len(c) < 3:
                continue
            key, off, ln = int(c[0]), int(c[1]), int(c[2])
            e = int(c[3]) if len(c) > 3 else 0
            out[key] = (data[off:off + ln - 1], e)
    return out


def read_dbtype(path):
    with open(path + ".dbtype", "rb") as t:
        return struct.unpack("<i", t.read(4))[0]


def dump_keyed(path):
    """Canonical text dump: one block per key in key order (thread-order independent)."""
    db = read_db(path)
    lines = []
    for k in sorted(db):
        payload, e = db[k]
        lines.append("#%d\t%d" % (k, e))
        lines.append(payload.decode("latin1").rstrip("\n"))
    return "\n".join(lines) + "\n"


def load_keyed(path):
    """Inverse of dump_keyed (plain or .gz): dict key -> (payload bytes incl. trailing newline(s), wasExtended)."""
    import gzip
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        text = f.read().decode("latin1")
    out = {}
    key = None
    buf = []
    for line in text.split("\n"):
        if line.startswith("#"):
            if key is not None:
                out[key[0]] = (_join(buf), key[1])
            k, e = line[1:].split("\t")
            key = (int(k), int(e))
            buf = []
        elif key is not None:
            buf.append(line)
    if key is not None:
        if buf and buf[-1] == "":
            buf.pop()
        out[key[0]] = (_join(buf), key[1])
    return out


def _join(lines):
    body = "\n".join(lines)
    return (body + "\n").encode("latin1") if body != "" else b""


def write_from_keyed(path, keyed, dbtype):
    write_db(path, ((k, keyed[k][0]) for k in sorted(keyed)), dbtype, {k: v[1] for k, v in keyed.items()})


def canon(db):
    """Canonical comparable form of read_db()/load_keyed() output."""
    return {k: (v[0].rstrip(b"\n"), v[1]) for k, v in db.items()}
