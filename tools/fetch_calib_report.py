#!/usr/bin/env python3
"""Turn a tools/calib_round.sh output tree into profiles/<prefix>.{md,json}: rocprofv3's FETCH_SIZE against known byte counts for the
access shapes of the path tracer's kernels.  usage: fetch_calib_report.py gpurun_out/calib profiles/r02_fetch_calibration"""
import csv
import glob
import json
import re
import sys


def main(src, dst):
    rows, out = [], {"source": src, "tables": {}}
    for lg in (24, 27, 31):
        t = {}
        for ln in open(f"{src}/time_{lg}.log"):
            m = re.match(r"CALIB (\S+)\s+table_bytes (\d+) record_bytes (\d+) records (\d+) bytes (\d+) ms ([\d.]+) GB/s ([\d.]+) Grecords/s ([\d.]+)", ln)
            if m:  # the second round of each kernel overwrites the first (cold TLBs)
                t[m.group(1)] = {"record_bytes": int(m.group(3)), "records": int(m.group(4)), "bytes": int(m.group(5)), "ms": float(m.group(6)),
                                 "GBps": float(m.group(7)), "Grecords_per_s": float(m.group(8))}
        f = glob.glob(f"{src}/pmc_{lg}/**/*counter_collection.csv", recursive=True)
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] != "FETCH_SIZE":
                continue
            k = r["Kernel_Name"].split("(")[0]
            if k in t:
                t[k]["fetch_size_bytes"] = float(r["Counter_Value"]) * 1024.0  # last dispatch wins: the warm round
        for k, v in t.items():
            v["requested_over_reported"] = v["bytes"] / v["fetch_size_bytes"]
            v["reported_bytes_per_record"] = v["fetch_size_bytes"] / v["records"]
            rows.append((lg, k, v))
        out["tables"][f"2^{lg}"] = t
    big = out["tables"]["2^31"]
    out["conclusion"] = {
        "fetch_size_counts": "one 64-byte tally per 128-byte line request that leaves L2, for every access shape tried",
        "fabric_bytes": "2 x FETCH_SIZE for every shape (a line request moves a 128-byte line: 64 B and 128 B random records run at the same RECORD rate, "
                        f"{big['k_calib_gather64']['Grecords_per_s']:.1f} vs {big['k_calib_gather128']['Grecords_per_s']:.1f} G records/s beyond the Infinity Cache)",
        "requested_bytes": {"16 B/lane coalesced stream": "2.00 x FETCH_SIZE", "128 B random records": "2.00 x FETCH_SIZE", "64 B random records": "0.97-1.06 x FETCH_SIZE",
                            "16 B random records": "0.24-0.31 x FETCH_SIZE"},
        "infinity_cache": "a 16 MiB table (resident in the 256 MiB Infinity Cache after the first sweep) reports the same FETCH_SIZE as a 2 GiB one: the counter sits "
                          "on the L2's fabric side and includes Infinity-Cache hits, so 2 x FETCH_SIZE + WRITE_SIZE is L2-miss (fabric) traffic, an UPPER bound of HBM traffic",
    }
    json.dump(out, open(dst + ".json", "w"), indent=1)
    md = ["# FETCH_SIZE calibration on gfx950 (tools/micro/fetch_calib.hip, tools/calib_round.sh)", "",
          "Every kernel reads each record of the table exactly once (bijective index scramble), so the requested byte count is known.",
          "`rocprofv3 --pmc FETCH_SIZE` in its own pass; timings from a separate un-profiled run (second round of each kernel).", "",
          "| table | kernel | record B | requested bytes | FETCH_SIZE bytes | requested / reported | reported B per record | GB/s (requested) | G records/s |", "|---|---|---|---|---|---|---|---|---|"]
    for lg, k, v in rows:
        md.append(f"| 2^{lg} B | `{k}` | {v['record_bytes']} | {v['bytes']} | {v['fetch_size_bytes']:.0f} | {v['requested_over_reported']:.3f} | {v['reported_bytes_per_record']:.1f} | {v['GBps']:.0f} | {v['Grecords_per_s']:.1f} |")
    md += ["", "## Reading", ""]
    for k, v in out["conclusion"].items():
        md.append(f"* **{k}**: {v if isinstance(v, str) else '; '.join(f'{a}: {b}' for a, b in v.items())}")
    md += ["",
           "Consequence for the profiles of this repo: `2 x FETCH_SIZE + WRITE_SIZE` is the right *fabric-side* byte count for k_extend / k_shadow / k_shade as well "
           "(VERDICT r1 asked whether the doubling, established for wide streaming reads only, also holds for scattered 64-byte records: it does, because the unit "
           "the fabric moves is the 128-byte line).  What the figure is NOT is HBM traffic proper: Infinity-Cache hits are included, which is how k_shade's "
           "6.8 TB/s could exceed the 6.29 TB/s copy peak of HBM.  A 64-byte record therefore costs a full 128-byte line per L2 miss -- half of it wasted unless "
           "the neighbouring record is used too -- and a 16-byte texel gather costs eight times its size."]
    open(dst + ".md", "w").write("\n".join(md) + "\n")
    print("wrote", dst + ".md")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
