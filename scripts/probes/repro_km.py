"""Reproduce a kmermatcher difference on a list of reads (one per line): device vs oracle, prints differing records.
    python scripts/probes/repro_km.py <reads.txt> [ENV=VALUE ...]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1); os.environ[k] = v
from carpedeam_amd import capi, mmdb
from stageflags import K_FLAGS

seqs = [l.rstrip("\n") for l in open(sys.argv[1])]
d = tempfile.mkdtemp(); t = lambda s: os.path.join(d, s)
mmdb.write_seqdb(t("in"), seqs)
subprocess.run([os.path.join(ROOT, "oracle", "_build", "cdm_oracle"), "kmermatcher", t("in"), t("pref")] + K_FLAGS + ["--threads", "1"], check=True, capture_output=True)
exp = {k: v[0] for k, v in mmdb.read_db(t("pref")).items()}
ctx = capi.Ctx(0)
db = ctx.upload_seqs(seqs)
_, keys, _ = db.meta()
off, rec = ctx.kmermatch(db).download()
got = capi.hits_to_text(off, rec, keys)
for k in sorted(exp):
    if got.get(k) != exp[k]:
        print("key", k, "\n got", got.get(k), "\n exp", exp[k])
print("done")
