"""Deterministic synthetic ancient-DNA read generator.

The generator is *counter based* so that numpy (tests, golden fixtures), the C++
host and the HIP device generator used by bench.py produce identical reads for
the same (seed, n, L) without sharing state:

    mix(x)       = splitmix64 finaliser of the 64-bit word x
    stream(s, i) = mix(mix(seed * 0x9E3779B97F4A7C15 + s) + i)        (all mod 2^64)

    genome base i           = stream(0, i) & 3                -> "ACGT"[.]
    read r start            = stream(1, r) mod (G - L_r + 1)
    read r strand           = stream(2, r) & 1                (1 = reverse complement)
    read r length (mixed)   = lo + stream(5, r) mod (hi - lo + 1)
    5' damage, k = 0..4     : u = stream(3, 8 r + k) / 2^64 < p5[k]  and base k       == C -> T
    3' damage, k = 0..4     : u = stream(4, 8 r + k) / 2^64 < p3[k]  and base L-1-k   == G -> A

G = n * mean(L) / coverage (coverage 20).  p5/p3 are the C>T / G>A columns of the
reference's example/dhigh5p.prof and example/dhigh3p.prof, rows 1..5 (SURVEY.md 8(d):
p3[k] applies to position L-1-k).  The strand flip happens before the damage, i.e.
damage is applied to the read as sequenced.
"""
import numpy as np

P5_DHIGH = (0.329405, 0.221745, 0.187678, 0.161196, 0.144011)      # example/dhigh5p.prof, C>T, rows 1..5
P3_DHIGH = (0.32891, 0.223405, 0.188599, 0.164419, 0.146352)  # example/dhigh3p.prof, G>A, rows 1..5

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def mix(x):
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + GOLDEN
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def stream(seed, s, idx):
    with np.errstate(over="ignore"):
        base = mix(np.uint64(seed) * GOLDEN + np.uint64(s))
        return mix(base + np.asarray(idx, dtype=np.uint64))


def threshold_u64(p):
    """u/2^64 < p  <=>  u < floor(p * 2^64) (p given as a decimal literal; exact in Python ints)."""
    from fractions import Fraction
    return int(Fraction(str(p)) * (1 << 64))


def genome_codes(seed, G):
    return (stream(seed, 0, np.arange(G, dtype=np.uint64)) & np.uint64(3)).astype(np.uint8)


def generate(n, L=100, seed=1, coverage=20, mixed=None, p5=P5_DHIGH, p3=P3_DHIGH):
    """Return (list of ASCII read strings as a uint8 matrix/ragged list, lengths).

    mixed=(lo, hi) draws lengths uniformly in [lo, hi]; otherwise all reads have length L.
    Output: (codes, lens) where codes is a list of uint8 arrays holding ASCII letters.
    """
    r = np.arange(n, dtype=np.uint64)
    if mixed is None:
        lens = np.full(n, L, dtype=np.int64)
        meanL = L
    else:
        lo, hi = mixed
        lens = (lo + (stream(seed, 5, r) % np.uint64(hi - lo + 1)).astype(np.int64))
        meanL = (lo + hi) / 2.0
    G = max(int(n * meanL / coverage), int(lens.max()) + 1)
    g = genome_codes(seed, G)
    start = (stream(seed, 1, r) % (np.uint64(G) - lens.astype(np.uint64) + np.uint64(1))).astype(np.int64)
    strand = (stream(seed, 2, r) & np.uint64(1)).astype(np.uint8)
    t5 = [np.uint64(threshold_u64(p)) for p in p5]
    t3 = [np.uint64(threshold_u64(p)) for p in p3]
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    # vectorised over reads of equal length
    for Lr in np.unique(lens):
        sel = np.nonzero(lens == Lr)[0]
        idx = start[sel][:, None] + np.arange(Lr)[None, :]
        c = g[idx]                                   # codes A,C,G,T = 0..3
        rev = strand[sel] == 1
        c[rev] = (3 - c[rev])[:, ::-1]
        rr = sel.astype(np.uint64)
        for k in range(5):
            u5 = stream(seed, 3, rr * np.uint64(8) + np.uint64(k))
            hit = (u5 < t5[k]) & (c[:, k] == 1)
            c[hit, k] = 3
        for k in range(5):
            u3 = stream(seed, 4, rr * np.uint64(8) + np.uint64(k))
            pos = Lr - 1 - k
            hit = (u3 < t3[k]) & (c[:, pos] == 2)
            c[hit, pos] = 0
        asc = letters[c]
        for i, s in enumerate(sel):
            out.append((int(s), asc[i]))
    out.sort(key=lambda t: t[0])
    return [a for _, a in out], lens


def generate_strings(n, **kw):
    seqs, _ = generate(n, **kw)
    return [s.tobytes().decode() for s in seqs]


PROF_HEADER = "A>C\tA>G\tA>T\tC>A\tC>G\tC>T\tG>A\tG>C\tG>T\tT>A\tT>C\tT>G"


def write_dhigh_profiles(prefix, p5=P5_DHIGH, p3=P3_DHIGH):
    """Write <prefix>5p.prof / <prefix>3p.prof byte-identical to the reference's example/dhigh{5p,3p}.prof
    (12 tab-separated substitution-rate columns, header line, 5 rows; C>T is column 5, G>A column 6)."""
    with open(prefix + "5p.prof", "w") as f:
        f.write(PROF_HEADER + "\n")
        for p in p5:
            f.write("\t".join(["0"] * 5 + [repr(p)] + ["0"] * 6) + "\n")
    with open(prefix + "3p.prof", "w") as f:
        f.write(PROF_HEADER + "\n")
        for p in p3:
            f.write("\t".join(["0"] * 6 + [repr(p)] + ["0"] * 5) + "\n")


def write_fastq_device(ctx, n, L, path, seed=1, chunk=10_000_000):
    """The n synthetic reads of (L, seed) - generated on the device by cdm_seqdb_synth, the same reads generate() makes - as a FASTQ
    file built in numpy buffers (no Python object per read: 50 M reads take seconds).  Every record is "@r\\nSEQ\\n+\\nIII...\\n"."""
    row = np.frombuffer(b"@r\n" + b"N" * L + b"\n+\n" + b"I" * L + b"\n", np.uint8)
    with open(path, "wb") as f:
        for first in range(0, n, chunk):
            m = min(chunk, n - first)
            db = ctx.synth(m, L, L, seed, n_total=n, first=first)
            tight = np.empty(m * (L + 1), np.uint8)            # "SEQ\n" per read (the library writes the whole range it is given)
            db.download_into(tight, np.arange(m, dtype=np.uint64) * np.uint64(L + 1))
            buf = np.tile(row, m).reshape(m, row.size)
            buf[:, 3:3 + L] = tight.reshape(m, L + 1)[:, :L]
            buf.tofile(f)
            del db, buf, tight
