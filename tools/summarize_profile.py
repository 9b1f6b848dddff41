#!/usr/bin/env python3
"""Condense a rocprofv3 output tree (kernel-trace --stats and --pmc passes) into small CSV/markdown files for profiles/.
usage: summarize_profile.py <gpurun_out/prof_dir> <profiles/prefix>"""
import collections
import csv
import glob
import json
import sys


def short(name):
    n = name.replace("void ", "").split("(")[0]
    return n[:80]


def main(src, dst):
    lines = ["# rocprofv3 summary (" + src + ")", ""]
    traffic = {}
    st = glob.glob(f"{src}/trace/**/*_kernel_stats.csv", recursive=True)
    if st:
        lines += ["## --kernel-trace --stats (all dispatches of the profiled command)", "", "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
        for r in csv.DictReader(open(st[0])):
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.4f} | {float(r['Percentage']):.2f} |")
        lines.append("")
    for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        fs = glob.glob(f"{src}/{tag}/**/*_counter_collection.csv", recursive=True)
        if not fs:
            continue
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] != ctr:
                continue
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        lines += [f"## --pmc {ctr} (separate pass; rocprofv3 reports KiB per dispatch, summed over the XCDs)", "",
                  "| kernel | dispatches | sum KiB | GiB per dispatch |", "|---|---|---|---|"]
        for k, (n, v) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
            lines.append(f"| `{k}` | {n} | {v:.0f} | {v / n / 1048576:.3f} |")
            traffic.setdefault(k, {})[ctr] = v * 1024.0 / n  # bytes per dispatch, as reported
        lines.append("")
    sq = glob.glob(f"{src}/pmc_sq/**/*_counter_collection.csv", recursive=True)
    if sq:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(sq[0])):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        lines += ["## --pmc SQ_* (separate pass, one frame; WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall)", "",
                  "| kernel | WAVE_CYCLES | ACTIVE_INST_ANY | WAIT_INST_ANY | WAIT_ANY | INSTS_VALU | INSTS_VMEM_RD |", "|---|---|---|---|---|---|---|"]
        for k, v in agg.items():
            wc = v.get("SQ_WAVE_CYCLES", 0.0)
            if wc > 1e9:
                lines.append(f"| `{k}` | {wc:.3g} | {v['SQ_ACTIVE_INST_ANY'] / wc:.1%} | {v['SQ_WAIT_INST_ANY'] / wc:.1%} | {v['SQ_WAIT_ANY'] / wc:.1%} | {v['SQ_INSTS_VALU']:.3g} | {v['SQ_INSTS_VMEM_RD']:.3g} |")
        lines.append("")
    for log in sorted(glob.glob(f"{src}/bench*.log")):
        for ln in open(log):
            if ln.startswith("{"):
                lines += [f"## bench line under `{log.split('/')[-1]}`", "", "```json", ln.strip(), "```", ""]
    if traffic:
        # MI355X_MICROARCH.md §HBM: the counters report KiB; on gfx950 FETCH_SIZE tallies 16 B/lane reads at half their bytes
        # (every load in these kernels is a 16-byte lane request) -> doubled; WRITE_SIZE is exact for 16 B/lane stores.
        out = {k: {"fetch_bytes_raw": v.get("FETCH_SIZE"), "write_bytes": v.get("WRITE_SIZE"),
                   "hbm_bytes_per_launch": 2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)} for k, v in traffic.items()}
        json.dump({"source": src, "correction": "2*FETCH_SIZE + WRITE_SIZE (KiB->bytes; gfx950 16 B/lane read correction)", "kernels": out},
                  open(dst + "_traffic.json", "w"), indent=1)
    open(dst + ".md", "w").write("\n".join(lines))
    print("wrote", dst + ".md")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
