"""Correction kernel time on a synthetic read set, under the environment's CDM_CORRECT_* switches: scripts/probes/correct_probe.py [reads] [lo hi]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from carpedeam_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100, 100)
d = tempfile.mkdtemp()
synth.write_dhigh_profiles(os.path.join(d, "dhigh"))
ctx = capi.Ctx(0)
ctx.damage_load(os.path.join(d, "dhigh"))
db = ctx.synth(n, lo, hi, 1)
alns = ctx.rescore(db, ctx.kmermatch(db))
ms = []
for i in range(3):
    corr = ctx.correct(db, alns)
    ms.append(round(ctx.last_kernel_ms(0), 2))
print({k: v for k, v in os.environ.items() if k.startswith("CDM_")}, n, lo, hi, "correction kernels ms", ms)
