"""Writes <reads> synthetic reads (lengths lo..hi, 20x, seed 1) as a sequence DB + the dhigh damage profiles:
python scripts/write_reads_db.py <reads> <lo> <hi> <out prefix>   ->  <out>, <out>.index, <out>.dbtype, <out>_dhigh{5p,3p}.prof"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carpedeam_amd import capi, mmdb, synth  # noqa: E402

n, lo, hi, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
synth.write_dhigh_profiles(out + "_dhigh")
ctx = capi.Ctx(0)
seqs, _, _ = ctx.synth(n, lo, hi, 1).download()
mmdb.write_seqdb(out, seqs)
print("wrote %d reads to %s" % (n, out))
