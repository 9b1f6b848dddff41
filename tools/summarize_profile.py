#!/usr/bin/env python3
"""Condense a rocprofv3 output tree (kernel-trace --stats and --pmc passes) into small CSV/markdown files for profiles/.
usage: summarize_profile.py <gpurun_out/prof_dir> <profiles/prefix>"""
import collections
import csv
import glob
import sys


def short(name):
    n = name.replace("void ", "").split("(")[0]
    return n[:80]


def main(src, dst):
    lines = ["# rocprofv3 summary (" + src + ")", ""]
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
        lines.append("")
    for log in sorted(glob.glob(f"{src}/bench_*.log")):
        for ln in open(log):
            if ln.startswith("{"):
                lines += [f"## bench line under `{log.split('/')[-1]}`", "", "```json", ln.strip(), "```", ""]
    open(dst + ".md", "w").write("\n".join(lines))
    print("wrote", dst + ".md")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
