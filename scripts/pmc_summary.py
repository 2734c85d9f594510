"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.

    python scripts/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<command>"

Counter values are KiB.  What FETCH_SIZE means on this machine was calibrated in round 5 (scripts/probes/pmc_calib.hip,
profiles/r05_pmc_calibration.json): the counter is 64 bytes per request of the L2 to the fabric WHATEVER the request's size -
  * a streaming read (4, 8 or 16 bytes per lane, consecutive lanes) goes out as 128-byte requests: the counter shows HALF the bytes
    (factor 2.000 on all three widths, and on aligned 128-byte segments gathered at random);
  * a random segment of 32 or 64 bytes is one 64-byte request: the counter shows 64 bytes per segment (1.0 x a 64-byte segment, 2 x
    the useful bytes of a 32-byte one; a random 4-byte word: 64 bytes);
  * WRITE_SIZE is exact for streaming stores and for scattered runs of 64 bytes and more.
So the correction is per kernel CLASS, not global (round 4 doubled every kernel): streaming kernels x 2; gather kernels (a thread or a
few lanes per randomly placed record / target / segment: k_rescore, k_correct*, k_xr_*, k_extend, the record and entry walkers) x 1 -
their requests are single 64-byte lines for the most part; where a gather covers both halves of a 128-byte line the true figure lies
between x 1 and x 2 (`fetch_bytes_upper`).  `hbm_bytes_per_launch` = (factor * fetch + write) / launches.
"""
import collections
import csv
import json
import re
import sys


def _split_top(t):
    out, depth, cur = [], 0, ""
    for ch in t:
        if ch in "<(":
            depth += 1
        elif ch in ">)":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    m = re.search(r"wrapped_(\w+?)_config<(.*?)>, \(rocprim::\w+::detail::target_arch\)", n)
    if "rocprim" in n and m:
        parts = _split_top(m.group(2))
        sub = "onesweep_iteration" if "onesweep_iteration" in n else ("onesweep_digit_histograms" if "onesweep_global_offsets" in n else ("onesweep_scan_histograms" if "onesweep_scan" in n else m.group(1)))
        bits = re.search(r">, (\d+)u, \(", parts[0])               # a non-default onesweep configuration
        types = ", ".join(parts[1:]).replace("unsigned long long", "u64").replace("unsigned long", "u64").replace("unsigned int", "u32")
        types = re.sub(r"rocprim::\w+::empty_type", "-", types)
        return "rocprim %s%s <%s>" % (sub, " %s-bit" % bits.group(1) if bits else "", types)
    n = re.sub(r"^void ", "", n)
    return re.split(r"[(<]", n)[0] + ("<" + n.split("<", 1)[1].split(">")[0] + ">" if "<" in n.split("(")[0] else "")


def load(path, counter):
    """-> (bytes per kernel, launches per kernel).  Launches of one kernel whose grid is less than a quarter of its largest
    grid are kept apart ("<name> [small launches]"): the radix pass runs on the 4 G k-mer slots and on the 50 M hash tuples."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    big = collections.defaultdict(float)
    for r in rows:
        k = short(r["Kernel_Name"])
        big[k] = max(big[k], float(r.get("Grid_Size", 0) or 0))
    tot = collections.defaultdict(float)
    launches = collections.defaultdict(set)
    for r in rows:
        k = short(r["Kernel_Name"])
        if float(r.get("Grid_Size", 0) or 0) * 4 < big[k]:
            k += " [small launches]"
        tot[k] += float(r["Counter_Value"]) * 1024.0
        launches[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in launches.items()}


# kernels whose reads are gathers of records / targets / segments at random places (one 64-byte request each for the most part)
GATHER = ("k_rescore", "k_correct", "k_xr_score", "k_xr_extend", "k_extend", "k_hamming", "k_contig_stats", "aggv::k_unit_agg", "aggv::k_vote_entries", "aggv::k_rle_segment",
          "runsort::k_unit_bounds", "runsort::k_seg_list", "runsort::k_unit_sort", "runsort::k_gather_ranges", "runsort::k_run_gather", "k_seg_place", "k_write", "k_cyc_hits",
          "k_rec_place_big", "k_big_", "bucket::k_big_copy", "k_stale_tail", "k_out_meta", "k_mark_active")


def fetch_factor(kernel):
    return 1.0 if any(kernel.startswith(g) or ("::" not in g and kernel.split("<")[0] == g) for g in GATHER) else 2.0


def main():
    fetch, nl = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), " + sys.argv[4],
           "units": "bytes; FETCH_SIZE = 64 B per request whatever its size (profiles/r05_pmc_calibration.json): x 2 for streaming kernels (128-byte requests), "
                    "x 1 for gather kernels (single 64-byte lines; fetch_bytes_upper = x 2 bounds them from above)",
           "kernels": {}}
    corrected = {k: fetch_factor(k) * fetch[k] for k in fetch}
    for k in sorted(fetch, key=lambda k: -(corrected[k] + write.get(k, 0))):
        f, w, n = corrected[k], write.get(k, 0.0), max(1, nl.get(k, 1))
        if f + w < 1e8:
            continue
        out["kernels"][k] = {"launches": n, "class": "gather" if fetch_factor(k) == 1.0 else "stream", "fetch_counter_bytes": fetch[k], "fetch_bytes": f, "fetch_bytes_upper": 2 * fetch[k],
                             "write_bytes": w, "hbm_bytes_per_launch": (f + w) / n}
    # per-stage totals of the chain (one step: the PMC passes run `bench.py --steps 1 --warmup 0`)
    stage_of = (("kmermatcher", ("k_seq_hash", "k_extract", "rocprim", "rx::", "k_bucket_groups", "k_groups", "runsort::", "k_unit_sort", "k_block_sort", "k_bucket_sort", "k_seg_",
                                 "aggv::", "k_self", "k_offsets", "k_len_keys", "k_slot_", "k_live_count", "k_stale_tail", "k_reduce_stats", "k_big_", "k_head_segment", "k_count_hash", "cdmscan", "k_block_heads", "k_rec_")),
                ("rescorediagonal", ("k_rescore", "k_expand", "k_count_valid", "k_scatter", "k_min_score")),
                ("ancient_correction", ("k_correct", "k_mark_active<", "k_mark_active(")),
                ("ancient_read_assemble", ("k_extend", "k_xr_", "k_write", "k_out_meta", "k_mark_active2")))
    stages = {name: 0.0 for name, _ in stage_of}
    stages["other (synthetic reads, metadata, copies)"] = 0.0
    for k in fetch:
        b = corrected[k] + write.get(k, 0.0)
        for name, pats in stage_of:
            if any(p in k for p in pats) and not (name == "ancient_correction" and "k_mark_active2" in k):
                stages[name] += b
                break
        else:
            stages["other (synthetic reads, metadata, copies)"] += b
    out["stages"] = stages
    out["total"] = sum(stages.values())
    json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
