#!/usr/bin/env python3
"""Headline benchmark: corrected bases/s of the MI355X hot path on synthetic dhigh reads (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--len L] [--config 2|3] [--scaling strong|weak]

A "step" is one pass of  kmermatcher -> rescorediagonal -> ancient_correction -> ancient_read_assemble  (--config 3, default:
BASELINE.json configs[2]; with N > 1 configs[3]) or of ancient_correction alone (--config 2: configs[1], 5 M reads) over
synthetic reads that are already resident in HBM (generated on the device before the timed region).

N > 1 (one process per GPU; the driver launches them with torch.distributed.run, `--gpus N` alone spawns them): ONE corpus of R
reads (--scaling strong, default).  --scheme exact (default): every rank holds the corpus and the LIBRARY splits the work over RCCL
(csrc/dist.hip: kmermatcher's extraction by blocks of the reads, its sorts by k-mer range and by owner of the representative, the other stages on the owned queries, the new DBs
all-gathered) - the result is the single-device one.  --scheme reads: rank r owns reads [r R/N, (r+1) R/N), runs the stages on them
alone and the per-shard contigs are all-gathered in ONE collective inside the timed step - the north star's wording, but not the
single-device result (the JSON line says so).  --scaling weak gives every rank its own R-read corpus, seed + rank.
Rank 0 prints ONE JSON line.  (CDM_BENCH_BACKEND=gloo: the ranks rendezvous over gloo and the library's calls go over a gloo transport
instead of librccl, so that several ranks can share one device - tests/test_gpu_two_ranks.py; RCCL wants a device per rank.)
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys
import tempfile
import time

# The host side of the library (the contig phase's queue, the text codecs of the module legs) runs OpenMP loops: a GPU box shows all
# of the machine's hardware threads (256) but gives one GPU's job a share of 16 cores - without a cap every loop oversubscribes it
os.environ.setdefault("OMP_NUM_THREADS", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# algorithmic bytes per base, SURVEY.md 8(d) (L = 100, k = 20): kmermatcher, rescorediagonal, ancient_correction, extender
ALG_B_PER_BASE = {"kmermatcher": 26.8, "rescorediagonal": 2.1, "ancient_correction": 2.0, "ancient_read_assemble": 1.3}


def baseline_metric():
    """metric string of BASELINE.json (the driver compares it verbatim)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "corrected bases/sec on 50M×100bp synthetic reads (dhigh), 1/2/4/8 GPU"


def host_ram():
    try:
        kb = {l.split(":")[0]: int(l.split()[1]) for l in open("/proc/meminfo") if l.split(":")[0] in ("MemTotal", "MemAvailable")}
        return "%.0f GB total, %.0f GB available" % (kb["MemTotal"] / 1e6, kb["MemAvailable"] / 1e6)
    except Exception:
        return "unknown"


def stage_cmds(p, exe_threads, dmg):
    from carpedeam_amd.stageflags import A_FLAGS, K_FLAGS, R_FLAGS
    th = ["--threads", str(exe_threads)]
    return [("kmermatcher", [p("reads"), p("pref")] + K_FLAGS + th),
            ("rescorediagonal", [p("reads"), p("reads"), p("pref"), p("aln")] + R_FLAGS + th),
            ("ancient_correction", [p("reads"), p("aln"), p("corr")] + A_FLAGS + dmg + th),
            ("ancient_read_assemble", [p("corr"), p("aln"), p("asm")] + A_FLAGS + dmg + th)]


def module_walls(n_reads, L, seed, threads):
    """Module-wall figures on DB FILES of one bounded sample: the four stages of the reference's own
    object code (oracle/_ref; the CPU restatement oracle/cdm_oracle.cpp when that binary is absent) on `threads` host threads
    = cpu_baseline, and the same four modules + the fused reads loop of the MI355X host binary carpedeam_amd/carpedeam_mi355x
    (DB read/parse, upload, kernels, download, text, DB write all inside) = gpu_module_wall."""
    from carpedeam_amd import capi, mmdb, synth
    ref = os.path.join(ROOT, "oracle", "_ref", "carpedeam_ref")
    kind, exe = "reference", ref
    if not os.path.exists(ref):
        kind, exe = "port", os.path.join(ROOT, "oracle", "_build", "cdm_oracle")
        if not os.path.exists(exe):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    gpu_bin = os.path.join(ROOT, "carpedeam_amd", "carpedeam_mi355x")
    out = {}
    # the DB files of BOTH legs live on the same file system: RAM-backed (/dev/shm) where the box has one with room - the scratch a
    # deployment gives MMseqs2-style tools -, else the default temporary directory; the JSON line says which
    scratch = None
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > 40 * n_reads * L + (8 << 30) and not os.environ.get("CDM_BENCH_TMPDIR"):
            scratch = "/dev/shm"
    except OSError:
        pass
    scratch = os.environ.get("CDM_BENCH_TMPDIR", scratch)
    with tempfile.TemporaryDirectory(dir=scratch) as d:
        p = lambda s: os.path.join(d, s)
        ctx = capi.Ctx(0)
        synth.write_fastq_device(ctx, n_reads, L, p("reads.fq"), seed)
        del ctx
        r = subprocess.run([gpu_bin, "createdb", p("reads.fq"), p("reads"), "--shuffle", "0", "--threads", str(threads)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("createdb failed: " + r.stderr[-400:])
        os.remove(p("reads.fq"))
        synth.write_dhigh_profiles(p("dhigh"))
        dmg = ["--ancient-damage", p("dhigh")]
        for label, binary in (("cpu", exe), ("gpu", gpu_bin)):
            times = {}
            for name, a in stage_cmds(p, threads, dmg):
                t0 = time.perf_counter()
                r = subprocess.run([binary, name] + a, capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError("%s module %s failed: %s" % (label, name, r.stderr[-400:]))
                times[name] = time.perf_counter() - t0
                if os.environ.get("CDM_TIMING") and label == "gpu":      # where a module's wall time goes (its own laps), on this process's stderr
                    sys.stderr.write("== %s %.3f s\n%s" % (name, times[name], r.stderr))
            out[label] = times
        t0 = time.perf_counter()
        r = subprocess.run([gpu_bin, "ancient_reads_loop", p("reads"), p("loop_out")] + dmg + ["--num-iter-reads-only", "1"], capture_output=True, text=True)
        loop_s = time.perf_counter() - t0 if r.returncode == 0 else None
    fmt = lambda t: ", ".join("%s %.2f" % kv for kv in t.items())
    sample = "%d synthetic %d bp dhigh reads (seed %d, 20x coverage), four modules on DB files (in %s) incl. DB read/parse/write" % (
        n_reads, L, seed, (scratch + ", RAM-backed") if scratch == "/dev/shm" else (scratch or tempfile.gettempdir()))
    cpu = {"value": n_reads * L / sum(out["cpu"].values()), "unit": "corrected bases/s", "cores": threads, "kind": kind,
           "sample": sample + ", %d threads; stage s: %s" % (threads, fmt(out["cpu"])),
           "ancient_correction_only_value": n_reads * L / out["cpu"]["ancient_correction"]}
    gpu = {"value": n_reads * L / sum(out["gpu"].values()), "unit": "corrected bases/s", "what": "carpedeam_amd/carpedeam_mi355x, one process per module (context creation, DB files, upload, text codecs included)",
           "sample": sample + "; stage s: %s" % fmt(out["gpu"]),
           "fused_reads_loop_value": (n_reads * L / loop_s) if loop_s else None}
    return cpu, gpu


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (before anything touches the GPU) and exit with their code"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


def run_loop_config(args, ctx, capi, cd, dist, rank, world, plan, torch, comm=None):
    """--config 5 (BASELINE.json configs[4], at a reduced size by default): the workflow loop of data/nuclassemble.sh:96-199 on
    mixed-length reads - 5 read iterations (kmermatcher, rescorediagonal, ancient_correction, ancient_read_assemble) and 7 contig
    iterations (kmermatcher -k 22 --include-only-extendable 1, rescorediagonal, ancient_correction, ancient_contig_merge,
    cyclecheck) - through the C ABI with every intermediate resident in HBM.  comm (N > 1, --scheme exact): every rank holds the corpus and
    the library splits every iteration over the ranks (cdm_reads_iteration_dist / cdm_contig_iteration_dist: the single-device result);
    without it (--scheme reads) a rank works on its own shard of the reads.
    value = bases fed to ancient_correction over all iterations / wall time of the whole job."""
    capi.lib().cdm_pool_headroom(1.6)      # the contig iterations' buffers grow ~1.5x per iteration (the loop binary does the same)
    n = plan["n"]
    db0 = ctx.synth(n, 60, 150, plan["seed"], n_total=plan["n_total"], first=plan["first"])
    kp = capi.KmerParams.reads_default()
    kc = capi.KmerParams.reads_default()
    kc.kmer_size, kc.include_only_extendable = 22, 1
    par = capi.AncientParams.default()
    par.max_seq_len = 200000

    def job():
        db, bases, per_it, circular = db0, 0, [], 0
        for it in range(12):
            t0 = time.perf_counter()
            bases += db.residues
            if comm is not None:
                if it < 5:
                    _, _, _, nxt = comm.reads_iteration(db, kp, None, par)
                else:
                    _, _, merged = comm.contig_iteration(db, kc, None, par)
                    cyc, nxt, _ = ctx.cyclecheck(merged, 200000, True)
                    circular += cyc.n
                    del merged, cyc
                db = nxt
                per_it.append(round(time.perf_counter() - t0, 3))
                if db.n == 0:
                    break
                continue
            hits = ctx.kmermatch(db, kp if it < 5 else kc)
            alns = ctx.rescore(db, hits)
            del hits
            corr = ctx.correct(db, alns, par)
            if it < 5:
                nxt = ctx.extend(corr, alns, par)
            else:
                merged = ctx.contig_merge(corr, alns, par)
                cyc, nxt, _ = ctx.cyclecheck(merged, 200000, True)
                circular += cyc.n
                del merged, cyc
            del corr, alns
            db = nxt
            per_it.append(round(time.perf_counter() - t0, 3))
            if db.n == 0:
                break
        return bases, db, per_it, circular

    gather_info = {}

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        job()
    sync()
    t0 = time.perf_counter()
    bases = 0
    for _ in range(args.steps):
        b, out, per_it, circular = job()
        bases += b
    sync()
    dt = time.perf_counter() - t0
    total = float(bases)
    if dist is not None:
        dt = cd.max_over_ranks(dist, dt, device=("cpu" if dist.get_backend() == "gloo" else "cuda"))
        if comm is None:        # (exact: every rank counted the one shared corpus)
            tb = torch.tensor([total], dtype=torch.float64, device=("cpu" if dist.get_backend() == "gloo" else "cuda"))
            dist.all_reduce(tb)
            total = float(tb.item())
    if rank == 0:
        chain_b = sum(ALG_B_PER_BASE.values())
        ach = chain_b * total / dt / 1e9
        print(json.dumps({
            "metric": baseline_metric(), "value": total / dt, "unit": "corrected bases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u8 (2-bit packed bases; int32 scores; software x87 f80 likelihood sums)", "data": "synthetic",
            "config": {"workload": "%d synthetic reads of 60-150 bp, dhigh, 12 iterations: 5 x (kmermatcher, rescorediagonal, ancient_correction, ancient_read_assemble) + 7 x "
                                   "(kmermatcher -k 22, rescorediagonal, ancient_correction, ancient_contig_merge, cyclecheck) (BASELINE.json configs[4]%s; ancient_contig_merge's queue and extension loop run on the "
                                   "device)" % (args.reads, ": its 25 M reads per GPU" if n >= 25_000_000 else " at reduced size: %d reads per GPU instead of 25 M" % n),
                       "reads_rank0": n, "seed": args.seed, "multi_gpu_scheme": (None if world == 1 else "exact: every iteration split over the ranks by the library (csrc/dist.hip cdm_reads_iteration_dist / cdm_contig_iteration_dist), the single-device result" if comm is not None else "reads: every rank runs the loop on its own shard of the reads (not the single-device result)"),
                       "equivalent_to_single_device": bool(world == 1 or comm is not None), "seconds_per_iteration_rank0": per_it, "final_sequences_rank0": out.n, "final_residues_rank0": out.residues,
                       "circular_contigs_set_aside_rank0": circular, "value_is": "whole job, reads resident in HBM at the start"},
            "roofline": {"bound": "hbm", "kernel": "whole chain (no single kernel dominates the 12 iterations: kmermatcher's passes over the grown contigs, the cycle check, the merge)", "achieved": ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None}}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3, 5), help="BASELINE.json configs index + 1: 2 = ancient_correction only on 5 M reads, 3 = full chain on 50 M reads, 5 = the 12-iteration loop (5 read + 7 contig iterations with cyclecheck) on mixed-length reads, 2 M per GPU unless --reads says otherwise")
    ap.add_argument("--reads", type=int, default=None, help="reads of the corpus (strong scaling) / per GPU (weak)")
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--seed", type=int, default=None, help="generator seed (default: 1, and 2 for --config 5, as SURVEY.md 8(d) specifies the corpora)")
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"))
    ap.add_argument("--scheme", default="exact", choices=("exact", "reads"),
                    help="N > 1: exact (default) = the library's own multi-GPU calls over RCCL (csrc/dist.hip: k-mer-range split + one all-to-all of group keys + "
                         "query-sharded stages + all-gathers of the new DBs), bit-identical to one device; reads = every rank runs the stages on its own read shard and "
                         "the contigs are all-gathered (the north star's wording; NOT equivalent to the single-device run: a shard sees 1/N of every pile-up)")
    ap.add_argument("--cpu-reads", type=int, default=10_000_000, help="reads of the sample the reference's modules, the device modules and the fused loop are timed on (same DB files)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.seed is None:
        args.seed = 2 if args.config == 5 else 1
    if args.steps is None:
        args.steps = 1 if args.config == 5 else 3
    if args.warmup is None:
        args.warmup = 0 if args.config == 5 else 1
    if args.reads is None:
        args.reads = int(os.environ.get("CDM_BENCH_READS", 5_000_000 if args.config == 2 else 50_000_000 if args.config == 3 else 2_000_000 * max(1, int(os.environ.get("WORLD_SIZE", args.gpus)))))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus != world:      # a line that says n_gpus = N must come from N ranks
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    import torch
    dist = None
    if world > 1 or os.environ.get("CDM_FORCE_DIST"):   # CDM_FORCE_DIST: exercise the collective path on one GPU
        import torch.distributed as dist
        # CDM_BENCH_BACKEND=gloo (tests): the ranks may then share ONE device - RCCL wants a device per rank - and the library's calls go
        # over a gloo transport (carpedeam_amd/shard.py GlooTransport) instead of librccl; everything else of the run is the same code
        backend = os.environ.get("CDM_BENCH_BACKEND", "nccl")
        if backend == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from carpedeam_amd import build, capi, synth
    if rank == 0:
        build.build()
    if dist is not None:
        dist.barrier()
    ctx = capi.Ctx(local_rank)
    with tempfile.TemporaryDirectory() as d:
        synth.write_dhigh_profiles(os.path.join(d, "dhigh"))
        ctx.damage_load(os.path.join(d, "dhigh"))
    L = args.len
    from carpedeam_amd import dist as cd
    plan = cd.shard_plan(rank, world, args.reads, args.seed, args.scaling)
    exact = args.scheme == "exact" and dist is not None and args.config in (3, 5)
    if exact:      # every rank holds the whole corpus; the work is split inside the stages, by the library itself over RCCL
        plan = dict(plan, first=0, n=plan["n_total"])
        if dist.get_backend() == "gloo":
            from carpedeam_amd import shard
            comm = capi.Comm.from_transport(ctx, rank, world, shard.GlooTransport(dist, rank, world))
        else:
            uid = [capi.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)          # (torch.distributed only launches, times and hands the id round: the data path is librccl under libcarpedeam_hip)
            comm = capi.Comm.rccl(ctx, rank, world, uid[0])
    if args.config == 5:
        return run_loop_config(args, ctx, capi, cd, dist, rank, world, plan, torch, comm if exact else None)
    db = ctx.synth(plan["n"], L, L, plan["seed"], n_total=plan["n_total"], first=plan["first"])      # resident in HBM before the timed region
    n = plan["n"]
    residues = db.residues

    pre = None
    if args.config == 2:      # the alignment set ancient_correction works on, made once outside the timed region
        hits = ctx.kmermatch(db)
        pre = ctx.rescore(db, hits)
        del hits

    def step():
        if exact:
            hits, alns, corr, asm = comm.reads_iteration(db)
            return asm, (hits.count, alns.count), [ctx.last_kernel_ms(i) for i in range(16)]
        if args.config == 2:
            corr = ctx.correct(db, pre)
            return corr, (0, pre.count), [ctx.last_kernel_ms(i) for i in range(16)]
        hits = ctx.kmermatch(db)
        alns = ctx.rescore(db, hits)
        stats = (hits.count, alns.count)
        del hits
        corr = ctx.correct(db, alns)
        asm = ctx.extend(corr, alns)
        ms = [ctx.last_kernel_ms(i) for i in range(16)]
        if dist is not None:
            # configs[3]: the single data-path collective of the north star is part of the step - RCCL all-gather of the per-shard
            # contigs (packed bases, N planes, lengths, keys in ONE buffer); every rank ends the step holding the contigs of all shards
            t1 = time.perf_counter()
            allc = cd.allgather_contigs(dist, ctx, asm, world, key_base=plan["first"] if args.scaling == "strong" else rank * args.reads)
            torch.cuda.synchronize()
            gather_info.update(contigs=allc.n, bases=allc.residues, seconds=gather_info.get("seconds", 0.0) + time.perf_counter() - t1, calls=gather_info.get("calls", 0) + 1)
            del allc
        return asm, stats, ms

    gather_info = {}

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
        del out
    gather_info.clear()
    sync()
    t0 = time.perf_counter()
    kernel_ms = [0.0] * 16
    stats = (0, 0)
    asm = None
    for _ in range(args.steps):
        del asm
        asm, stats, ms = step()
        kernel_ms = [a + b for a, b in zip(kernel_ms, ms)]
    sync()
    dt = time.perf_counter() - t0
    gathered = None
    total_bases = residues * args.steps
    if dist is not None:
        dt = cd.max_over_ranks(dist, dt, device=("cpu" if dist.get_backend() == "gloo" else "cuda"))
        if exact:           # a rank's results hold the records of the queries it owns + one self record for every other query:
            # summed over the ranks (self records of the others taken out) they are the single-device counts - the same two numbers
            # at every N, which is what "equivalent_to_single_device" claims
            t2 = torch.tensor([float(stats[0] - n), float(stats[1] - n)], dtype=torch.float64, device=("cpu" if dist.get_backend() == "gloo" else "cuda"))
            dist.all_reduce(t2)
            stats = (int(t2[0].item()) + n, int(t2[1].item()) + n)
        if not exact:       # (exact: every rank counted the one shared corpus)
            tb = torch.tensor([float(total_bases)], dtype=torch.float64, device=("cpu" if dist.get_backend() == "gloo" else "cuda"))
            dist.all_reduce(tb)
            total_bases = float(tb.item())
        if args.config == 3 and not exact and gather_info:
            gathered = {"contigs": gather_info["contigs"], "bases": gather_info["bases"], "seconds_per_step": gather_info["seconds"] / max(1, gather_info["calls"]),
                        "inside_timed_region": True}
    if rank == 0:
        k_ms = [m / args.steps for m in kernel_ms]
        step_ms = 1e3 * dt / args.steps
        bases_step = n * L
        gbs = lambda b_per_base, ms: (b_per_base * bases_step / (ms * 1e-3) / 1e9) if ms > 0 else 0.0
        if args.config == 2:
            # ancient_correction only: the dominant kernel is k_correct_fast (one launch per step, HIP events around it on the
            # context stream); algorithmic bytes 2.0 B/base
            ach = gbs(ALG_B_PER_BASE["ancient_correction"], k_ms[0])
            roof = {"bound": "hbm", "kernel": "k_correct_fast (pile-up + per-base call, one launch per step)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": k_ms[0], "launches_per_step": 1,
                    "algorithmic_bytes_per_launch": ALG_B_PER_BASE["ancient_correction"] * bases_step,
                    "stage_level": {"ancient_correction": {"ms": k_ms[10], "achieved": gbs(ALG_B_PER_BASE["ancient_correction"], k_ms[10]), "unit": "GB/s"}}}
            workload = "%d synthetic %d bp reads, dhigh, ancient_correction only on the alignments of one kmermatcher + rescorediagonal pass (BASELINE.json configs[1])" % (n, L)
        else:
            # Dominant kernel by rocprofv3 --stats (profiles/): rx::k_rx_pass (carpedeam_amd/csrc/radix.h), the hand-written onesweep pass of
            # kmermatcher's sort 1: 3 launches per step over the k-mer slots.  Reads of one length (this workload) go through them as
            # 8-byte slot keys / slot tuples (round 5: a head pass that drops the empty slots + two passes inside the head digit's
            # segments); other DBs as 12-byte (u64 key, u32 value) pairs.  A launch's algorithmic traffic = everything it sorts read once and
            # written once; the library sums it (cdm_ctx_last_kernel_ms(15), GB) next to the launches' time, measured live with HIP
            # events on its stream around each pass launch ((13) = ms, (14) = launches).
            launches = max(1, int(round(k_ms[14]))) if k_ms[14] > 0 else 3
            iter_ms = k_ms[13] / launches if k_ms[13] > 0 else 0.0
            sort_bytes = (k_ms[15] * 1e9 / launches) if k_ms[15] > 0 else 2.0 * 12.0 * (L - 20 + 2) * n
            achieved = sort_bytes / (iter_ms * 1e-3) / 1e9 if iter_ms > 0 else 0.0
            slot_layout = k_ms[15] > 0 and k_ms[15] * 1e9 / launches < 2.0 * 10.0 * (L - 20 + 2) * n
            traffic, traffic_total, traffic_from = None, None, None
            pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_50M*.json")))
            if n == 50_000_000 and L == 100 and world == 1 and pmcs:
                try:
                    pj = json.load(open(pmcs[-1]))
                    names = [k for k in pj["kernels"] if k.startswith("rx::k_rx_pass<unsigned long, " + ("rx::NoValue" if slot_layout else "unsigned int")) and "small launches" not in k
                             and (not slot_layout or k.rstrip().endswith((", 1>", ", 2>")))]          # (slot layout: the head pass and the passes inside its segments, not the plain key-only passes of other sorts)
                    by = sum(pj["kernels"][k]["hbm_bytes_per_launch"] * pj["kernels"][k]["launches"] for k in names)
                    ln = sum(pj["kernels"][k]["launches"] for k in names)
                    traffic = by / ln if ln else None
                    traffic_total = {"file": os.path.basename(pmcs[-1]), "hbm_bytes_per_step": pj.get("stages")}
                    traffic_from = "profiles/" + os.path.basename(pmcs[-1]) + " (rocprofv3 --pmc passes of this workload, committed; not collected in this run)"
                except Exception:
                    traffic = None
            stages = {"kmermatcher": k_ms[8], "rescorediagonal": k_ms[9], "ancient_correction": k_ms[10], "ancient_read_assemble": k_ms[11]}
            roof = {"bound": "hbm", "kernel": ("rx::k_rx_pass<u64 slot keys / slot tuples> (hand-written onesweep radix passes of kmermatcher's sort 1 on 8-byte tuples: head pass + 2 passes inside its segments; average over the 3 launches)"
                                               if slot_layout else "rx::k_rx_pass<u64 key, u32 value> (hand-written onesweep radix pass, 9 bits; kmermatcher sort 1: 3 passes over the k-mer slots)"),
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_from": traffic_from,
                    # frac is the dominant kernel's per-launch figure; frac_8d prices the whole step on SURVEY.md 8(d)'s bytes (32.2 B/base:
                    # every radix pass beyond one write + one read of the tuples is overhead there) - the smaller, stricter number
                    "frac_8d": gbs(sum(ALG_B_PER_BASE.values()), step_ms) / HBM_PEAK_GBS,
                    "avg_launch_ms": iter_ms, "launches_per_step": launches, "algorithmic_bytes_per_launch": sort_bytes,
                    "how": "HIP events around the %d pass launches of a step: %.2f ms in total (the whole sort-1 call incl. histogram and status resets: %.2f ms)" % (launches, k_ms[13], k_ms[5]),
                    # the figures that price the radix passes as overhead (the tuple array is written once and read
                    # once): every stage's whole call (HIP events on the context stream around everything it launched) against its
                    # algorithmic bytes, and the chain against 32.2 B/base
                    "stage_level": {s: {"ms": ms, "achieved": gbs(ALG_B_PER_BASE[s], ms), "unit": "GB/s", "frac": gbs(ALG_B_PER_BASE[s], ms) / HBM_PEAK_GBS} for s, ms in stages.items()},
                    "chain": {"alg_bytes_per_base": sum(ALG_B_PER_BASE.values()), "achieved": gbs(sum(ALG_B_PER_BASE.values()), step_ms), "unit": "GB/s",
                              "frac": gbs(sum(ALG_B_PER_BASE.values()), step_ms) / HBM_PEAK_GBS},
                    "traffic_total": traffic_total}
            workload = "%d synthetic %d bp reads%s, dhigh, full correction + kmermatcher/rescorediagonal/ancient_read_assemble (BASELINE.json configs[%d])" % (
                args.reads, L, (" split over %d GPUs (%d per GPU)" % (world, n)) if world > 1 and args.scaling == "strong" else (" per GPU" if world > 1 else ""), 3 if world > 1 else 2)
        line = {
            "metric": baseline_metric(), "value": total_bases / dt, "unit": "corrected bases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u8 (2-bit packed bases; int32 scores; software x87 f80 likelihood sums)", "data": "synthetic",
            "config": {"workload": workload, "reads_rank0": n, "read_len": L, "seed": args.seed, "prefilter_hits": stats[0], "alignments": stats[1],
                       "multi_gpu_scheme": (None if world == 1 and not exact else
                                            ("exact: the library's own RCCL calls (csrc/dist.hip), bit-identical to one device (prefilter_hits / alignments are summed over the ranks: the single-device counts at every N); kmermatcher this run: " +
                                             {"replicate": "every rank ran it whole (two ranks, or a DB that takes the wide group key) and kept its owned representatives' hits - no exchange", "all": "every rank extracted all reads, kept its range of the k-mer space, one all-to-all of group keys to the owners of their representatives", "split": "every rank extracted its block of the reads, all-to-alls carried the k-mer tuples to the rank of their k-mer range and the group keys to the owners of their representatives", "part": "equal slices of the k-mer space by value, one all-to-all of group keys", "ranges": "extraction and sort 1 / grouping by ranges of the k-mer space, the kept group keys all-gathered, sort 2 and the vote on every rank", None: "(not run)"}[comm.last_path()] +
                                             "; rescorediagonal / ancient_correction / ancient_read_assemble on the owned queries, the new sequences all-gathered") if exact else
                                            "reads: every rank runs the stages on its own read shard, no data-path collective, one all-gather of contigs at the end; NOT equivalent to the single-device run (a shard sees 1/N of every pile-up)"),
                       "equivalent_to_single_device": bool(world == 1 or exact),
                       "value_is": "kernel-resident: reads already in HBM, no DB files (the module-wall figure is gpu_module_wall)",
                       "stage_kernel_ms": {"kmer_extract": k_ms[3], "kmer_sort1_call": k_ms[5], "kmer_sort1_hash_call": k_ms[7], "kmer_sort2_call": k_ms[6], "rescore": k_ms[1],
                                           "correct": k_ms[0], "extend": k_ms[4]}},
            "roofline": roof,
        }
        if gathered is not None:
            line["config"]["allgather_contigs"] = gathered
        if not args.no_cpu_baseline and world == 1:
            # the module legs start their own processes on this device: everything this process holds there goes first (sequences,
            # results, the library's cache of device blocks, torch's), so that they meet the device a deployment's module meets
            import gc
            asm = db = pre = None
            ctx = None
            gc.collect()
            torch.cuda.empty_cache()
            try:
                # the pool gives a one-GPU job 16 host cores; the reference's kmermatcher slows down when oversubscribed
                cpu, gpu = module_walls(args.cpu_reads, L, args.seed, min(os.cpu_count() or 1, int(os.environ.get("CDM_CPU_THREADS", 16))))
                if args.cpu_reads != args.reads:
                    # why a sample and not the metric's own corpus: the contract bounds the CPU leg (10-30 s of CPU work per stage set, the
                    # default run within minutes); the reference's four modules take ~6 s per million reads on 16 threads, i.e. ~5 min at
                    # 50 M, and its kmermatcher alone wants 16 B x 81 tuples per read = 65 GB of host RAM there (--cpu-reads 50000000 runs it)
                    cpu["sample_is"] = "%d of the workload's %d reads (same generator and seed); host RAM %s; --cpu-reads %d would run the whole corpus" % (
                        args.cpu_reads, args.reads, host_ram(), args.reads)
                line["cpu_baseline"] = cpu
                line["gpu_module_wall"] = gpu
                # like for like (same DB files, same host threads, DB read/parse/write inside both): what the north star's ">= 20x the CPU
                # baseline" is measured by.  vs_baseline stays null: BASELINE.md holds no published number for this metric.
                if cpu.get("value"):
                    line["vs_cpu_baseline"] = {"device_modules_on_db_files": gpu["value"] / cpu["value"],
                                               "fused_reads_loop_on_db_files": (gpu["fused_reads_loop_value"] / cpu["value"]) if gpu.get("fused_reads_loop_value") else None,
                                               "sample_reads": args.cpu_reads}
                    # (no ratio of `value` to this baseline: value is kernel-resident on the whole corpus, the baseline a run on DB files of
                    #  the sample - the two legs above are the like-for-like ones)
            except Exception as e:   # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "corrected bases/s", "cores": os.cpu_count(), "kind": "unavailable", "sample": str(e)[:300]}
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
