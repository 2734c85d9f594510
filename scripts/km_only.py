import sys, time, os
sys.path.insert(0, "/root/repo")
from carpedeam_amd import capi
n = int(sys.argv[1])
ctx = capi.Ctx(0)
db = ctx.synth(n, 100, 100, 1)
for i in range(2):
    t = time.time(); h = ctx.kmermatch(db); dt = time.time() - t
    print("kmermatch", n, "hits", h.count, "ms", dt * 1e3, [round(ctx.last_kernel_ms(i), 1) for i in range(8)], flush=True)
    del h
