"""Debug helper: kmermatcher on the bucket-path test database under each environment variant, one subprocess per variant,
so that a crash names its variant.  python scripts/probes/probe_envs.py [child <json env>]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ENVS = [{}, {"CDM_BUCKET_CAP": "64"}, {"CDM_BUCKET_CAP": "5"}, {"CDM_BUCKET_CAP": "3,17"}, {"CDM_BUCKET_CAP": "1"}, {"CDM_BUCKET_CAP": "512,40"},
        {"CDM_KMER_SORT": "lsd"}, {"CDM_KMER_LAYOUT": "wide"}, {"CDM_KMER_LAYOUT": "wide", "CDM_BUCKET_CAP": "6,100"},
        {"CDM_KMER_SORT2": "radix"}, {"CDM_KMER_SORT2": "check"}, {"CDM_KMER_SORT2": "check", "CDM_UNIT_CAP": "5"},
        {"CDM_KMER_SORT2": "check", "CDM_UNIT_CAP": "5", "CDM_BLOCK_CAP": "8"}, {"CDM_UNIT_CAP": "1", "CDM_BLOCK_CAP": "0"},
        {"CDM_KMER_SORT2": "check", "CDM_UNIT_SUB": "3"}]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from carpedeam_amd import capi, synth
    import numpy as np
    seqs = synth.generate_strings(6000, seed=11, mixed=(40, 160)) + ["ACGTTGCA" * 12] * 40 + ["AC" * 50, ""]
    ctx = capi.Ctx(0)
    db = ctx.upload_seqs([s.encode() for s in seqs])
    off, rec = ctx.kmermatch(db).download()
    import zlib
    print("ok hits %d crc %08x" % (len(rec), zlib.crc32(np.ascontiguousarray(rec).tobytes())))
    sys.exit(0)
for env in ENVS:
    e = dict(os.environ); e.update(env); e["CDM_BUCKET_STATS"] = "1"
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True)
    print(json.dumps(env), "rc", r.returncode, "|", r.stdout.strip()[-200:], "|", " ".join(r.stderr.strip().splitlines()[-6:])[-900:])
