"""A second build of the library with other compile-time geometry, for A/B runs on the GPU box (CDM_LIB=<path> selects it):
python scripts/build_variant.py <tag> <file.hip>[,<file2.hip>...] -DX=1 ...   ->  carpedeam_amd/_variants/libcarpedeam_hip_<tag>.so
Only the named files are recompiled (with the extra flags; name EVERY file that includes the header the macro lives in); the other
objects are the regular build's."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carpedeam_amd import build as B  # noqa: E402

tag, srcs, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
B.build()
out_dir = os.path.join(B.HERE, "_variants")
os.makedirs(out_dir, exist_ok=True)
mine = []
for src in srcs:
    obj = os.path.join(out_dir, "%s_%s.o" % (src, tag))
    subprocess.run([B.HIPCC] + B.HIP_FLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj], check=True, capture_output=True)
    mine.append(obj)
objs = [os.path.join(B.OBJ, f) for f in sorted(os.listdir(B.OBJ)) if f.endswith(".o") and f[:-2] not in srcs] + mine
lib = os.path.join(out_dir, "libcarpedeam_hip_%s.so" % tag)
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lgomp"], check=True, capture_output=True)
print(lib)
