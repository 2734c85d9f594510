"""Extension kernel time on a synthetic read set, under the environment's CDM_EXTEND / CDM_XR_* switches: scripts/probes/extend_probe.py [reads]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from carpedeam_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
d = tempfile.mkdtemp()
synth.write_dhigh_profiles(os.path.join(d, "dhigh"))
ctx = capi.Ctx(0)
ctx.damage_load(os.path.join(d, "dhigh"))
db = ctx.synth(n, 100, 100, 1)
alns = ctx.rescore(db, ctx.kmermatch(db))
corr = ctx.correct(db, alns)
ms = []
for i in range(3):
    asm = ctx.extend(corr, alns)
    ms.append(round(ctx.last_kernel_ms(4), 2))
print({k: v for k, v in os.environ.items() if k.startswith("CDM_")}, "extension kernels ms", ms)
