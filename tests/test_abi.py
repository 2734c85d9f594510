"""CPU-side checks of the drop-in boundary: the shared library builds, loads and exports every symbol that
include/carpedeam_hip.h declares; without a GPU the compute path fails loudly instead of falling back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from carpedeam_amd import build
    return build.build()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "carpedeam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cdm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(libpath):
    lib = ctypes.CDLL(libpath)
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_lists_every_symbol(libpath):
    from carpedeam_amd import capi
    assert sorted(capi.EXPORTS) == declared_symbols()
    capi.lib()


def test_struct_layouts_match_header():
    from carpedeam_amd import capi
    assert capi.HIT_DTYPE.itemsize == 12 and capi.ALN_DTYPE.itemsize == 32
    assert ctypes.sizeof(capi.KmerParams) == 40 and ctypes.sizeof(capi.RescoreParams) == 32 and ctypes.sizeof(capi.AncientParams) == 40


def test_no_cpu_fallback(libpath):
    """On a box without a GPU creating a context must fail with a clear error (never a silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from carpedeam_amd import capi
    with pytest.raises(capi.CdmError) as e:
        capi.Ctx(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_host_evalue_matches_reference_table(libpath):
    """cdm_evalue / cdm_bit_score (host code of the library) against the reference's ALP numbers."""
    import gzip
    from carpedeam_amd import capi
    l = capi.lib()
    rows = [r.rstrip("\n").split("\t") for r in gzip.open(os.path.join(ROOT, "tests", "golden", "functions", "evalue.tsv.gz"), "rt")]
    for a, b in rows:
        db, score, qlen = a.split(" ")
        exp = b.split(" ")
        ev = l.cdm_evalue(float(score), float(qlen), int(db))
        assert "%.3E" % ev == exp[2], (a, ev, exp)
        assert l.cdm_bit_score(float(score)) == int(float.fromhex(exp[1]) + 0.5), a


def test_host_binary_fails_loudly_without_gpu(libpath, tmp_path):
    """The module binary has no CPU path either: on a box without a GPU it exits non-zero with an error message."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from carpedeam_amd import mmdb
    exe = os.path.join(ROOT, "carpedeam_amd", "carpedeam")
    assert os.path.exists(exe)
    mmdb.write_seqdb(str(tmp_path / "s"), ["ACGT" * 10, "ACGTT" * 8])
    r = subprocess.run([exe, "kmermatcher", str(tmp_path / "s"), str(tmp_path / "p"), "-k", "20"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_host_binary_flag_lists(libpath, tmp_path):
    """Flags are checked against the reference's per-module lists before anything else happens (no GPU needed): an unknown flag
    is "Unrecognized parameter" (Parameters.cpp:1703), a known flag with a value the MI355X path would compute differently is
    refused with a message, a flag the path really has no use for is accepted."""
    import subprocess
    from carpedeam_amd import mmdb
    exe = os.path.join(ROOT, "carpedeam_amd", "carpedeam")
    mmdb.write_seqdb(str(tmp_path / "s"), ["ACGT" * 10, "ACGTT" * 8])
    s, p = str(tmp_path / "s"), str(tmp_path / "p")

    def run(*a):
        r = subprocess.run([exe] + list(a), capture_output=True, text=True)
        return r.returncode, r.stderr

    for mod, args, bad in (("kmermatcher", [s, p], ["--frobnicate", "1"]), ("kmermatcher", [s, p], ["--min-aln-len", "3"]),       # rescorediagonal's flag
                           ("rescorediagonal", [s, s, p, p + "2"], ["--kmer-per-seq", "20"]), ("ancient_correction", [s, p, p + "3"], ["-k", "20"]),
                           ("ancient_read_assemble", [s, p, p + "3"], ["--sort-results", "0"])):
        rc, err = run(mod, *args, *bad)
        assert rc != 0 and 'Unrecognized parameter "%s"' % bad[0] in err, (mod, bad, err)
    for mod, args, bad in (("kmermatcher", [s, p], ["--mask", "1"]), ("kmermatcher", [s, p], ["--spaced-kmer-mode", "1"]), ("kmermatcher", [s, p], ["--adjust-kmer-len", "1"]),
                           ("kmermatcher", [s, p], ["--sub-mat", "blosum62.out"]), ("rescorediagonal", [s, s, p, p + "2"], ["--rescore-mode", "2"]),
                           ("rescorediagonal", [s, s, p, p + "2"], ["--rescore-mode", "3", "--sort-results", "1"]), ("rescorediagonal", [s, s, p, p + "2"], ["--rescore-mode", "3", "-a", "1"]),
                           ("rescorediagonal", [s, s, p, p + "2"], ["--rescore-mode", "3", "--filter-hits", "1"]), ("ancient_read_assemble", [s, p, p + "3"], ["--rescore-mode", "0"])):
        rc, err = run(mod, *args, *bad)
        assert rc != 0 and "is not supported by the MI355X path" in err, (mod, bad, err)
    # accepted flags get as far as the device (or, with a GPU, through): the error, if any, is not about flags
    rc, err = run("kmermatcher", s, p, "-k", "20", "--mask", "0", "--alph-size", "21", "--threads", "3", "--sub-mat", "nucl:nucleotide.out,aa:blosum62.out", "--split-memory-limit", "1G")
    assert "Unrecognized" not in err and "not supported" not in err
