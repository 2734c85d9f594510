"""Contig sets for cyclecheck (src/assembler/cyclecheck.cpp): linear, circular (a genome plus its own start again), tandem
repeats, low complexity, N runs, sequences around the 22-mer / three-thirds edge cases.  Deterministic (RandomState)."""
import numpy as np

ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)


def rnd(rs, n):
    return ALPHA[rs.randint(0, 4, size=n)].copy()


def mutate(rs, a, rate):
    a = a.copy()
    m = rs.random_sample(a.size) < rate
    a[m] = ALPHA[rs.randint(0, 4, size=int(m.sum()))]
    return a


def cases(seed=7, scale=1.0):
    rs = np.random.RandomState(seed)
    out = []

    def add(a):
        out.append(bytes(bytearray(a)).decode("ascii"))
    for L in (0, 1, 5, 21, 22, 23, 24, 25, 44, 45, 65, 66, 67, 68, 69, 70, 71, 100):
        add(rnd(rs, L))
        add(np.tile(rnd(rs, 3), L // 3 + 1)[:L])
    for _ in range(int(30 * scale)):                       # linear
        add(rnd(rs, rs.randint(66, 4000)))
    for _ in range(int(40 * scale)):                       # circular: genome + the start again (0.1 .. 2.2 laps more), some noise
        g = rnd(rs, rs.randint(40, 3000))
        laps = rs.choice([0.1, 0.3, 0.5, 0.9, 1.0, 1.3, 2.2])
        full = np.tile(g, 4)[: int(g.size * (1 + laps))]
        add(mutate(rs, full, rs.choice([0.0, 0.0, 0.01, 0.03, 0.08])))
    for _ in range(int(20 * scale)):                       # tandem repeats of short units, homopolymers
        unit = rnd(rs, rs.choice([1, 2, 3, 7, 11, 22, 23, 50, 120]))
        add(mutate(rs, np.tile(unit, 4000 // unit.size + 1)[: rs.randint(70, 3000)], rs.choice([0.0, 0.005, 0.02])))
    for _ in range(int(15 * scale)):                       # N: runs and sprinkles, in circular and linear contigs
        g = rnd(rs, rs.randint(100, 1500))
        a = np.tile(g, 3)[: int(g.size * rs.choice([1.0, 1.6, 2.5]))]
        for _ in range(rs.randint(1, 6)):
            p = rs.randint(0, a.size)
            a[p: p + rs.choice([1, 1, 2, 30])] = ord("N")
        add(a)
    for _ in range(int(8 * scale)):                        # repeat inside the first third only / between middle and back only
        L = rs.randint(600, 3000)
        a = rnd(rs, L)
        blk = rnd(rs, rs.randint(60, L // 7))
        if rs.randint(2):
            a[10: 10 + blk.size] = blk; a[10 + blk.size + 5: 10 + 2 * blk.size + 5] = blk
        else:
            a[L // 2: L // 2 + blk.size] = blk; a[L - blk.size - 3: L - 3] = blk
        add(a)
    g = rnd(rs, 9000)
    add(np.concatenate([g, g[:2500]]))                     # a long circular one, and a long linear one
    add(rnd(rs, 15000))
    return out


def circular_reads(seed=11, genomes=(420, 610, 800, 1500), coverage=30, lo=60, hi=120, linear=2500):
    """Reads (both strands, C->T / G->A damage at the ends) from circular genomes and one linear one: after a few
    iterations of the assembler the circular ones close on themselves."""
    rs = np.random.RandomState(seed)
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    reads = []
    for gi, G in enumerate(list(genomes) + [linear]):
        g = rnd(rs, G)
        circ = gi < len(genomes)
        for _ in range(int(G * coverage / ((lo + hi) / 2))):
            L = rs.randint(lo, hi + 1)
            s = rs.randint(0, G if circ else G - L + 1)
            r = np.take(g, np.arange(s, s + L), mode="wrap").copy()
            if rs.randint(2):
                r = comp[r][::-1].copy()
            for k, p in enumerate((0.3, 0.15, 0.08)):
                if r[k] == ord("C") and rs.random_sample() < p:
                    r[k] = ord("T")
                if r[L - 1 - k] == ord("G") and rs.random_sample() < p:
                    r[L - 1 - k] = ord("A")
            reads.append(bytes(bytearray(r)).decode("ascii"))
    order = rs.permutation(len(reads))
    return [reads[i] for i in order]


def letter_cases(seed=9):
    """the contig set above (another seed, half the size) with what real FASTA carries beside ACGTN: soft-masked stretches,
    IUPAC codes, a few bytes that are no letters (cyclecheck indexes what NucleotideMatrix maps them to and writes them out as
    they are)"""
    rs = np.random.RandomState(seed)
    odd = b"RYSWKMBDHVUNXryswkmbdhvunx*-.1"
    out = []
    for s in cases(seed=seed, scale=0.5):
        b = bytearray(s.encode())
        r = rs.random_sample()
        if b and r < 0.4:
            a = rs.randint(0, len(b)); e = min(len(b), a + rs.randint(1, 400))
            b[a:e] = bytes(b[a:e]).lower()
        elif b and r < 0.7:
            for _ in range(rs.randint(1, 6)):
                b[rs.randint(0, len(b))] = odd[rs.randint(0, len(odd))]
        elif r < 0.8:
            b = bytearray(bytes(b).lower())
        out.append(b.decode("ascii"))
    return out
