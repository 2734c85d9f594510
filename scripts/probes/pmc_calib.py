"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this machine for the access patterns of the library's kernels.

    python scripts/probes/pmc_calib.py <probe stdout> <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

The probe (pmc_calib.hip) prints "CAL <kernel> bytes <moved> ms <t> GBps <r>" per kernel; the counter files come from the same
program under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`.  Counter values are KiB.  out.json:
per kernel the known bytes, the raw counter bytes and `factor` = known / counter - what a counter value of that access pattern has to
be multiplied with.  scripts/pmc_summary.py reads the factors from profiles/r05_pmc_calibration.json.
"""
import csv
import json
import re
import sys


def counters(path, name):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = re.sub(r"^void ", "", r["Kernel_Name"])
        k = re.split(r"\(", k)[0].strip()
        out[k] = out.get(k, 0.0) + float(r["Counter_Value"]) * 1024.0
    return out


def main():
    known = {}
    for line in open(sys.argv[1]):
        m = re.match(r"CAL (\S+) bytes (\d+) ms ([\d.]+) GBps ([\d.]+)", line)
        if m:
            known[m.group(1)] = {"bytes": int(m.group(2)), "ms": float(m.group(3)), "GBps": float(m.group(4))}
    fetch, write = counters(sys.argv[2], "FETCH_SIZE"), counters(sys.argv[3], "WRITE_SIZE")
    out = {"source": "scripts/probes/pmc_calib.hip under rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)", "kernels": {}}
    for k, v in known.items():
        is_write = k.startswith("cal_write") or k.startswith("cal_scatter")
        c = (write if is_write else fetch).get(k)
        e = dict(v)
        e["counter"] = "WRITE_SIZE" if is_write else "FETCH_SIZE"
        e["counter_bytes"] = c
        e["factor"] = (v["bytes"] / c) if c else None
        # what the OTHER counter saw (reads of a write kernel: partial-line read-modify-write would show here)
        other = (fetch if is_write else write).get(k)
        e["other_counter_bytes"] = other
        out["kernels"][k] = e
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    for k, e in out["kernels"].items():
        print("%-24s known %8.2f GB  %s %8.2f GB  factor %s  (%.0f GB/s)" % (k, e["bytes"] / 1e9, e["counter"], (e["counter_bytes"] or 0) / 1e9,
                                                                          "%.3f" % e["factor"] if e["factor"] else "-", e["GBps"]))


if __name__ == "__main__":
    main()
