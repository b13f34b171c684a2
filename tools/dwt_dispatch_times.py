#!/usr/bin/env python3
"""Per-dispatch durations of dwt_tile_kernel (one dispatch per stage) for the headline frame, from a rocprofv3 kernel trace of
a child bench.py.   python tools/dwt_dispatch_times.py [--config C2|C4|C5]      (needs a GPU and rocprofv3)"""
import os
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    cfg = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "C2"
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--batched-probe", "0",
             "--no-batch-configs", "--no-traffic", "--no-extras", "--config", cfg]
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        subprocess.run(["rocprofv3", "--kernel-trace", "-d", td, "-o", "r", "--"] + child, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=200)
        db = [os.path.join(dp, f) for dp, _, fs in os.walk(td) for f in fs if f.endswith(".db")][0]
        cur = sqlite3.connect(db).cursor()
        tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
        view = "kernels" if "kernels" in tables else [t for t in tables if "kernel_dispatch" in t][0]
        cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
        name_col = "name" if "name" in cols else "kernel_name"
        rows = list(cur.execute(f"select {name_col}, start, end, grid_size_x, grid_size_y, grid_size_z from {view} order by start")) if "grid_size_x" in cols else \
            [(r[0], r[1], r[2], 0, 0, 0) for r in cur.execute(f"select {name_col}, start, end from {view} order by start")]
    d = [(r[2] - r[1], r[3], r[4], r[5]) for r in rows if "dwt_tile" in r[0]]
    stages = {"C2": 5, "C4": 4, "C5": 6}[cfg]
    last = d[-stages:]
    print(f"{cfg}: dwt_tile_kernel dispatches of the last encode (ns, grid): " + "  ".join(f"{x[0]} ({x[1]}x{x[2]}x{x[3]})" for x in last) + f"   sum {sum(x[0] for x in last) / 1e3:.1f} us")
    names = {}
    for r in rows:
        names.setdefault(r[0].split("(")[0][:60], []).append(r[2] - r[1])
    for n, v in sorted(names.items(), key=lambda kv: -sum(kv[1])):
        print(f"   {n:60s} n {len(v):4d}  mean {sum(v) / len(v) / 1e3:9.1f} us")


if __name__ == "__main__":
    main()
