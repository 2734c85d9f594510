import sys; sys.path.insert(0, "/root/repo")
from carpedeam_amd import capi
ctx = capi.Ctx(0)
db = ctx.synth(int(sys.argv[1]), 60, 150, 2)
print("ok", db.n)
