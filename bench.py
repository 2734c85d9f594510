#!/usr/bin/env python3
"""Headline benchmark: corrected bases/s of the MI355X hot path on synthetic dhigh reads (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--len L]

A "step" is one pass of  kmermatcher -> rescorediagonal -> ancient_correction -> ancient_read_assemble  over one batch of
R synthetic reads per GPU that is already resident in HBM (generated on the device before the timed region).  Ranks are
independent partitions (weak scaling: every rank gets its own R-read corpus, seed + rank); after the last timed step the
per-shard contigs are all-gathered over RCCL (N > 1).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def baseline_metric():
    """metric string of BASELINE.json (the driver compares it verbatim)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "corrected bases/sec on 50M\u00d7100bp synthetic reads (dhigh), 1/2/4/8 GPU"


def cpu_baseline(n_reads, L, seed, threads):
    """Time the four stages of the reference's own object code (oracle/_ref, built by oracle/Makefile.ref) - or, when that
    binary is absent, the CPU restatement oracle/cdm_oracle.cpp - on a bounded sample of the same synthetic workload."""
    from carpedeam_amd import capi, mmdb, synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from stageflags import A_FLAGS, K_FLAGS, R_FLAGS
    ref = os.path.join(ROOT, "oracle", "_ref", "carpedeam_ref")
    kind = "reference"
    exe = ref
    if not os.path.exists(ref):
        kind = "port"
        exe = os.path.join(ROOT, "oracle", "_build", "cdm_oracle")
        if not os.path.exists(exe):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    ctx = capi.Ctx(0)
    seqs, _, _ = ctx.synth(n_reads, L, L, seed).download()
    del ctx
    with tempfile.TemporaryDirectory() as d:
        p = lambda s: os.path.join(d, s)
        mmdb.write_seqdb(p("reads"), seqs)
        synth.write_dhigh_profiles(p("dhigh"))
        th = ["--threads", str(threads)]
        dmg = ["--ancient-damage", p("dhigh")]
        stages = [("kmermatcher", [p("reads"), p("pref")] + K_FLAGS + th),
                  ("rescorediagonal", [p("reads"), p("reads"), p("pref"), p("aln")] + R_FLAGS + th),
                  ("ancient_correction", [p("reads"), p("aln"), p("corr")] + A_FLAGS + dmg + th),
                  ("ancient_read_assemble", [p("corr"), p("aln"), p("asm")] + A_FLAGS + dmg + th)]
        times = {}
        for name, args in stages:
            t0 = time.perf_counter()
            r = subprocess.run([exe, name] + args, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("cpu baseline stage %s failed: %s" % (name, r.stderr[-500:]))
            times[name] = time.perf_counter() - t0
    total = sum(times.values())
    return {"value": n_reads * L / total, "unit": "corrected bases/s", "cores": threads, "kind": kind,
            "sample": "%d synthetic %d bp dhigh reads (seed %d, 20x coverage), four-stage chain incl. DB read/parse/write, %d threads; stage s: %s"
                      % (n_reads, L, seed, threads, ", ".join("%s %.2f" % (k, v) for k, v in times.items()))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("CDM_BENCH_READS", 50_000_000)), help="reads per GPU")
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-reads", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    import torch
    dist = None
    if world > 1 or os.environ.get("CDM_FORCE_DIST"):   # CDM_FORCE_DIST: exercise the collective path on one GPU
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from carpedeam_amd import build, capi, synth
    if rank == 0:
        build.build()
    if dist is not None:
        dist.barrier()
    ctx = capi.Ctx(local_rank)
    with tempfile.TemporaryDirectory() as d:
        synth.write_dhigh_profiles(os.path.join(d, "dhigh"))
        ctx.damage_load(os.path.join(d, "dhigh"))
    n, L = args.reads, args.len
    from carpedeam_amd import dist as cd
    plan = cd.shard_plan(rank, world, n, args.seed)
    db = ctx.synth(plan["n"], L, L, plan["seed"])      # resident in HBM before the timed region
    residues = db.residues

    def step():
        hits = ctx.kmermatch(db)
        alns = ctx.rescore(db, hits)
        stats = (hits.count, alns.count)
        del hits
        corr = ctx.correct(db, alns)
        asm = ctx.extend(corr, alns)
        ms = [ctx.last_kernel_ms(i) for i in range(8)]
        return asm, stats, ms

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
        del out
    sync()
    t0 = time.perf_counter()
    kernel_ms = [0.0] * 8
    stats = (0, 0)
    asm = None
    for _ in range(args.steps):
        del asm
        asm, stats, ms = step()
        kernel_ms = [a + b for a, b in zip(kernel_ms, ms)]
    sync()
    dt = time.perf_counter() - t0
    gathered = None
    if dist is not None:
        dt = cd.max_over_ranks(dist, dt, device="cuda")
        # the single data-path collective of the north star: RCCL all-gather of the per-shard contigs (packed bases,
        # lengths, keys); every rank ends up holding the contigs of all shards as one device DB
        t1 = time.perf_counter()
        allc = cd.allgather_contigs(dist, ctx, asm, world, key_stride=n)
        torch.cuda.synchronize()
        gathered = {"contigs": allc.n, "bases": allc.residues, "seconds": time.perf_counter() - t1}
    if rank == 0:
        total_bases = residues * args.steps * world
        k_ms = [m / args.steps for m in kernel_ms]
        # Dominant kernel (rocprofv3 --stats, profiles/): rocPRIM's radix_sort_onesweep_iteration<u64 key, u32 value> in the 9-bit
        # configuration kmermatcher's sort 1 uses on the packed 12-byte tuples of the k-mer slots: the top 27 of their
        # 2k + 1 = 41 sort bits go through 3 passes (the low 14 bits are finished on chip by k_bucket_groups); rocPRIM sorts at
        # most 2^30 items per launch, so a pass is ceil(items / 2^30) launches.  (The n whole-sequence hash tuples are sorted by
        # the default 8-bit configuration, a different kernel: 8 short launches, 3 ms per step.)
        # Algorithmic bytes of one launch: its 12-byte tuples read once and written once.  The average launch
        # duration comes from the HIP-event time of the sort call on the library's stream; the call also runs one histogram
        # launch per 2^30 items, which costs about half an iteration launch (it reads the 8-byte keys once).
        tuples_per_read = L - 20 + 2
        n1 = tuples_per_read * n
        c1 = -(-n1 // (1 << 30))
        p1 = -(-min(27, 2 * 20 + 1) // 9)
        launches = p1 * c1
        t1 = k_ms[5] * launches / (launches + 0.5 * c1) if k_ms[5] > 0 else 0.0      # ms spent in the iteration launches
        bytes_all = 2.0 * 12.0 * p1 * n1                                   # summed over this kernel's launches in one step
        iter_ms = t1 / launches if launches else 0.0                       # = rocprof's AverageNs for this kernel
        sort_bytes = bytes_all / launches if launches else 0.0
        achieved = bytes_all / (t1 * 1e-3) / 1e9 if t1 > 0 else 0.0
        # HBM traffic of that kernel from the PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3 --pmc runs of
        # this very command at the default size; profiles/r01_pmc_50M.json) - only quoted for the workload it was measured on
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_50M_l.json")
        if n == 50_000_000 and L == 100 and os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc))["kernels"]["rocprim onesweep_iteration 9-bit <u64, u32>"]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        line = {
            "metric": baseline_metric(), "value": total_bases / dt, "unit": "corrected bases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8 (2-bit packed bases; int32 scores; software x87 f80 likelihood sums)", "data": "synthetic",
            "config": {"workload": "%d synthetic %d bp reads per GPU, dhigh, full correction + kmermatcher/rescorediagonal/ancient_read_assemble (BASELINE.json configs[2])" % (n, L),
                       "reads_per_gpu": n, "read_len": L, "seed": args.seed, "prefilter_hits": stats[0], "alignments": stats[1],
                       "stage_kernel_ms": {"kmer_extract": k_ms[3], "kmer_sort1_call": k_ms[5], "kmer_sort1_hash_call": k_ms[7], "kmer_sort2_call": k_ms[6], "rescore": k_ms[1],
                                           "correct": k_ms[0], "extend": k_ms[4]}},
            "roofline": {"bound": "hbm", "kernel": "rocprim radix_sort_onesweep_iteration<u64 key, u32 value> (9-bit configuration; kmermatcher sort 1: 3 passes over the k-mer slots)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": iter_ms, "launches_per_step": launches, "algorithmic_bytes_per_launch": sort_bytes,
                         "stage_level": {"what": "whole kmermatcher stage against its algorithmic bytes (26.8 B/base, SURVEY.md 8(d))",
                                         "achieved": 26.8 * n * L / ((k_ms[3] + k_ms[5] + k_ms[6] + k_ms[7]) * 1e-3) / 1e9 if (k_ms[3] + k_ms[5] + k_ms[6]) > 0 else 0.0,
                                         "unit": "GB/s"}},
        }
        if gathered is not None:
            line["config"]["allgather_contigs"] = gathered
        if not args.no_cpu_baseline and world == 1:
            try:
                # the pool gives a one-GPU job 16 host cores; the reference's kmermatcher slows down when oversubscribed
                line["cpu_baseline"] = cpu_baseline(args.cpu_reads, L, args.seed, min(os.cpu_count() or 1, int(os.environ.get("CDM_CPU_THREADS", 16))))
            except Exception as e:   # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "corrected bases/s", "cores": os.cpu_count(), "kind": "unavailable", "sample": str(e)[:200]}
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
