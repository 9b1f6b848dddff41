#!/usr/bin/env python3
"""Condense a rocprofv3 output tree (kernel-trace --stats and --pmc passes) into small CSV/markdown files for profiles/.
usage: summarize_profile.py <gpurun_out/prof_dir> <profiles/prefix>"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    return n[:80]


def main(src, dst):
    lines = ["# rocprofv3 summary (" + src + ")", ""]
    traffic = {}
    avg_ms = {}
    st = glob.glob(f"{src}/trace/**/*_kernel_stats.csv", recursive=True)
    if st:
        lines += ["## --kernel-trace --stats (all dispatches of the profiled command)", "", "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
        for r in csv.DictReader(open(st[0])):
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.4f} | {float(r['Percentage']):.2f} |")
            avg_ms[short(r["Name"])] = float(r["AverageNs"]) / 1e6
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
    sqinfo = {}
    if sq:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        span = collections.defaultdict(dict)  # kernel -> dispatch id -> (start, end) ns of the profiled dispatch itself
        for r in csv.DictReader(open(sq[0])):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                span[k][r["Dispatch_Id"]] = (float(r["Start_Timestamp"]), float(r["End_Timestamp"]))
        lines += ["## --pmc SQ_* + GRBM_GUI_ACTIVE (separate pass, one frame; WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall)", "",
                  "Effective clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch time (MI355X_MICROARCH.md, DVFS); VALU issue = SQ_INSTS_VALU (wave instructions) per",
                  "clock per SIMD (1024 SIMDs); a wave64 VALU instruction holds its SIMD-32 for two cycles, so the ceiling is 0.5.", "",
                  "| kernel | WAVE_CYCLES | ACTIVE_INST_ANY | WAIT_INST_ANY | WAIT_ANY | INSTS_VALU | INSTS_VMEM_RD | time ms | clock GHz | VALU / clk / SIMD | of 0.5 |", "|---|---|---|---|---|---|---|---|---|---|---|"]
        for k, v in agg.items():
            wc = v.get("SQ_WAVE_CYCLES", 0.0)
            if wc > 1e9:
                t_ns = sum(e - s for s, e in span[k].values())
                clk = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / t_ns if t_ns > 0 and v.get("GRBM_GUI_ACTIVE") else 0.0  # GHz
                use_clk = clk if 1.0 < clk < 2.6 else 2.4
                vpc = v["SQ_INSTS_VALU"] / (t_ns * use_clk * 1024.0) if t_ns > 0 else 0.0
                sqinfo[k] = {"valu_per_clk_per_simd": round(vpc, 4), "clock_ghz": round(use_clk, 3), "wait_any": round(v["SQ_WAIT_ANY"] / wc, 4),
                             "wait_inst_any": round(v["SQ_WAIT_INST_ANY"] / wc, 4), "active_inst_any": round(v["SQ_ACTIVE_INST_ANY"] / wc, 4)}
                lines.append(f"| `{k}` | {wc:.3g} | {v['SQ_ACTIVE_INST_ANY'] / wc:.1%} | {v['SQ_WAIT_INST_ANY'] / wc:.1%} | {v['SQ_WAIT_ANY'] / wc:.1%} | {v['SQ_INSTS_VALU']:.3g} | "
                             f"{v['SQ_INSTS_VMEM_RD']:.3g} | {t_ns / 1e6:.2f} | {use_clk:.2f} | {vpc:.3f} | {vpc / 0.5:.1%} |")
        lines.append("")
    l2 = glob.glob(f"{src}/pmc_l2/**/*_counter_collection.csv", recursive=True)
    l2info = {}
    if l2:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(l2[0])):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        lines += ["## --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum (separate pass, one frame): where the gathers are served", "",
                  "L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS) (MI355X_MICROARCH.md); TCP_TCC_READ_REQ = read requests that missed the CU's L1 (vector cache) and",
                  "went to L2; TCC_EA0_RDREQ = read requests L2 sent on to the fabric (Infinity Cache / HBM).", "",
                  "| kernel | TCC_HIT | TCC_MISS | L2 hit rate | TCP_TCC_READ_REQ | TCC_EA0_RDREQ | fabric requests per L2 read request |", "|---|---|---|---|---|---|---|"]
        for k, v in sorted(agg.items(), key=lambda x: -(x[1].get("TCC_HIT_sum", 0.0) + x[1].get("TCC_MISS_sum", 0.0)))[:8]:
            hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
            if hit + miss <= 0:
                continue
            rq, ea = v.get("TCP_TCC_READ_REQ_sum", 0.0), v.get("TCC_EA0_RDREQ_sum", 0.0)
            l2info[k] = {"l2_hit_rate": round(hit / (hit + miss), 4), "tcc_hit": hit, "tcc_miss": miss, "tcp_tcc_read_req": rq, "tcc_ea0_rdreq": ea}
            lines.append(f"| `{k}` | {hit:.4g} | {miss:.4g} | {hit / (hit + miss):.1%} | {rq:.4g} | {ea:.4g} | {(ea / rq if rq else 0.0):.3f} |")
        lines.append("")
    for log in sorted(glob.glob(f"{src}/bench*.log")):
        for ln in open(log):
            if ln.startswith("{"):
                try:
                    j = json.loads(ln)
                    brief = {"value": j.get("value"), "unit": j.get("unit"), "ms_per_step": j.get("ms_per_step"), "steps": j.get("steps"),
                             "ms_per_frame": (j.get("roofline") or {}).get("ms_per_frame"), "source_hash": j.get("source_hash")}
                    lines += [f"## bench line under `{log.split('/')[-1]}` (per-kernel HIP events; the full line is in the matching `*_bench.log`)", "", "```json", json.dumps(brief), "```", ""]
                except ValueError:
                    pass
    if traffic:
        # profiles/r02_fetch_calibration.md: FETCH_SIZE tallies 64 B per 128-byte line request leaving L2, whatever the access shape
        # (coalesced streams, 16 / 64 / 128-byte random records), Infinity-Cache hits included; WRITE_SIZE is exact.  So
        # 2 x FETCH_SIZE + WRITE_SIZE = fabric-side (L2-miss) bytes, an upper bound of HBM traffic, for gathers as for streams.
        out = {}
        for k, v in traffic.items():
            e = {"fetch_bytes_raw": v.get("FETCH_SIZE", 0.0), "write_bytes": v.get("WRITE_SIZE", 0.0),
                 "hbm_bytes_per_launch": 2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0), "avg_ms": avg_ms.get(k)}
            e.update(sqinfo.get(k, {}))
            e.update(l2info.get(k, {}))
            out[k] = e
        from raytracer3_amd._lib import kernel_source_hash

        json.dump({"source": src, "source_hash": kernel_source_hash(), "correction": "2*FETCH_SIZE + WRITE_SIZE = fabric-side (L2-miss) bytes incl. Infinity-Cache hits; the factor 2 is calibrated for streams AND "
                                                "16/64/128 B gathers in profiles/r02_fetch_calibration.md", "kernels": out}, open(dst + "_traffic.json", "w"), indent=1)
    open(dst + ".md", "w").write("\n".join(lines))
    print("wrote", dst + ".md")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
