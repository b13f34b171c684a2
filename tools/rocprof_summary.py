#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases into a small JSON/Markdown pair under profiles/.

  python tools/rocprof_summary.py <tag> <stats.db> [<pmc_fetch.db> <pmc_write.db>]

Kernel times come from the --kernel-trace --stats run; FETCH_SIZE / WRITE_SIZE from their own --pmc
passes (they do not fit one pass: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2).  Units: the
counters are in KiB.  Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports half
of the bytes of a streaming read -- our own calibration point is finalize_kernel, which reads and writes
exactly one plane set (W*H*2 bytes per plane): WRITE_SIZE matches exactly, FETCH_SIZE reads 0.50x.
So traffic_bytes = factor*FETCH_SIZE*1024 + WRITE_SIZE*1024 with the factor calibrated per kernel (bench.py FETCH_FACTOR: 1 for
dwt_tile_kernel -- its stage 0 reads a known byte count and FETCH_SIZE matches it uncorrected -- and family_events_kernel, 2 elsewhere).
"""
import json
import os
import sqlite3
import sys


def short(name: str) -> str:
    n = name.replace("(anonymous namespace)::", "").split("(")[0]
    head = n.split("<")[0]                                  # (template arguments may hold namespaces of their own)
    return head.split("::")[-1].replace("void ", "") + n[len(head):].replace("icer::", "")


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    out = {"tag": tag, "kernels": {}}
    cur = sqlite3.connect(stats).cursor()
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        out["kernels"][short(name)] = {"calls": calls, "avg_us": round(avg, 3), "total_us": round(total, 3), "pct": round(pct, 4)}
    for path, ctr in zip(sys.argv[3:5], ("FETCH_SIZE", "WRITE_SIZE")):
        c = sqlite3.connect(path).cursor()
        q = "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name"
        for name, n, avg in c.execute(q, (ctr,)):
            out["kernels"].setdefault(short(name), {})[ctr + "_KiB_per_launch"] = round(avg, 3)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import FETCH_FACTOR
    for k, v in out["kernels"].items():
        if "FETCH_SIZE_KiB_per_launch" in v and "WRITE_SIZE_KiB_per_launch" in v:
            v["fetch_factor"] = FETCH_FACTOR.get(k.split("<")[0], 2.0)
            v["traffic_bytes_per_launch"] = int(v["fetch_factor"] * v["FETCH_SIZE_KiB_per_launch"] * 1024 + v["WRITE_SIZE_KiB_per_launch"] * 1024)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", f"{tag}_rocprof.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    lines = [f"# rocprofv3 summary `{tag}`", "", "| kernel | calls | avg us | % | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic MB (factor*F+W) |", "|---|---|---|---|---|---|---|"]
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("total_us", 0)):
        lines.append(f"| {k} | {v.get('calls','')} | {v.get('avg_us','')} | {v.get('pct','')} | {v.get('FETCH_SIZE_KiB_per_launch','')} | "
                     f"{v.get('WRITE_SIZE_KiB_per_launch','')} | {round(v['traffic_bytes_per_launch']/1e6,2) if 'traffic_bytes_per_launch' in v else ''} |")
    with open(os.path.join(root, "profiles", f"{tag}_rocprof.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
