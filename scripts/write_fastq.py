"""Synthetic reads (the generator of carpedeam_amd/synth.py, on the device) as a FASTQ file, built in one numpy buffer - no Python
object per read, so 50 M reads take seconds:   python scripts/write_fastq.py <reads> <len> <out.fq> [seed]
(fixed read length; every record is "@r\\nSEQ\\n+\\nIII...\\n"; also writes <out.fq>_dhigh{5p,3p}.prof)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carpedeam_amd import capi, synth  # noqa: E402

n, L, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
synth.write_dhigh_profiles(out + "_dhigh")
row = np.frombuffer(b"@r\n" + b"N" * L + b"\n+\n" + b"I" * L + b"\n", np.uint8)
ctx = capi.Ctx(0)
with open(out, "wb") as f:
    for first in range(0, n, 10_000_000):       # 10 M reads (2 GB of text) at a time
        m = min(10_000_000, n - first)
        db = ctx.synth(m, L, L, seed, n_total=n, first=first)
        tight = np.empty(m * (L + 1), np.uint8)            # "SEQ\n" per read (the library writes the whole range it is given)
        db.download_into(tight, np.arange(m, dtype=np.uint64) * np.uint64(L + 1))
        buf = np.tile(row, m).reshape(m, row.size)
        buf[:, 3:3 + L] = tight.reshape(m, L + 1)[:, :L]
        buf.tofile(f)
        del db, buf, tight
print("wrote %d reads of %d letters to %s" % (n, L, out))
