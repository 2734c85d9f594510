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
synth.write_fastq_device(capi.Ctx(0), n, L, out, seed)
print("wrote %d reads of %d letters to %s" % (n, L, out))
