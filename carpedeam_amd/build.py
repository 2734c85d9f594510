"""Build the gfx950 shared library (C ABI, include/carpedeam_hip.h) in-tree.

    python -m carpedeam_amd.build            # -> carpedeam_amd/libcarpedeam_hip.so (+ carpedeam_amd/carpedeam host binary)

hipcc cross-compiles for gfx950 without a GPU.  Host numerics (damage tables, E-values) are compiled with g++ at the
reference recipe's flags because their last bits matter (x87 long double, FMA contraction); everything else with hipcc.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libcarpedeam_hip.so")
BIN = os.path.join(HERE, "carpedeam_mi355x")     # the device module binary (host/main.cpp)
FRONT = os.path.join(HERE, "carpedeam")          # the front end that takes the reference binary's place (host/front.c; no HIP inside)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ROCM_INC = "/opt/rocm/include"

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]
GXX_FLAGS = ["-O3", "-march=x86-64-v3", "-std=c++17", "-fPIC", "-fsigned-char", "-Wall",
             "-D__HIP_PLATFORM_AMD__", "-I" + ROCM_INC]


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(f) > t for f in (src,) + tuple(extra))


def build(verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = tuple(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")) + (
        os.path.join(os.path.dirname(HERE), "include", "carpedeam_hip.h"),)
    objs = []
    jobs = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip"):
            src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f + ".o")
            objs.append(obj)
            if _newer(src, obj, headers):
                jobs.append([HIPCC] + HIP_FLAGS + ["-c", src, "-o", obj])
    hostdir = os.path.join(CSRC, "host")
    host_lib_srcs = ["damage.cpp", "evalue.cpp", "contigmerge.cpp"]
    for f in host_lib_srcs:
        src, obj = os.path.join(hostdir, f), os.path.join(OBJ, f + ".o")
        objs.append(obj)
        if _newer(src, obj, headers):
            jobs.append(["g++"] + GXX_FLAGS + ["-fopenmp", "-c", src, "-o", obj])
    procs = [(j, subprocess.Popen(j, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)) for j in jobs]
    for j, p in procs:
        out, _ = p.communicate()
        if verbose or p.returncode:
            sys.stderr.write(" ".join(j) + "\n" + out)
        if p.returncode:
            raise RuntimeError("compile failed: " + " ".join(j))
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lgomp"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    # host multi-call binary (module surface of the reference): built when its sources exist
    main_src = os.path.join(hostdir, "main.cpp")
    if os.path.exists(main_src):
        srcs = [os.path.join(hostdir, f) for f in sorted(os.listdir(hostdir)) if f.endswith(".cpp") and f not in host_lib_srcs]
        host_headers = headers + tuple(os.path.join(hostdir, h) for h in os.listdir(hostdir) if h.endswith(".h"))
        if any(_newer(s, BIN, host_headers) for s in srcs) or _newer(LIB, BIN):
            cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-Wall", "-I" + os.path.join(os.path.dirname(HERE), "include"), "-o", BIN] + srcs + [
                "-L" + HERE, "-lcarpedeam_hip", "-lz", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + HERE]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError("host binary link failed")
    front_src = os.path.join(hostdir, "front.c")
    if os.path.exists(front_src) and _newer(front_src, FRONT):
        r = subprocess.run(["gcc", "-O2", "-Wall", "-o", FRONT, front_src], capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("front end build failed")
    return LIB


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
