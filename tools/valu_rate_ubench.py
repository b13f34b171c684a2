#!/usr/bin/env python3
"""Builds and runs tools/valu_rate_ubench.hip on the GPU of this box and writes
     profiles/r03_valu_rate_ubench.json   raw results + `bench_constants` (what bench.py's issue roofline uses)
     profiles/r03_valu_rate_ubench.md     the table
   python tools/valu_rate_ubench.py [outdir]        (needs a GPU; ~20 s)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
    os.makedirs(outdir, exist_ok=True)
    exe = "/tmp/valu_rate_ubench"
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", os.path.join(ROOT, "tools", "valu_rate_ubench.hip"), "-o", exe])
    res = json.loads(subprocess.check_output([exe], text=True))
    by = {}
    for c in res["cases"]:
        by.setdefault(c["case"], {})[c["waves_per_simd"]] = c
    # the coder's VALU mix is 32-bit integer ALU: price it at the mean saturated cost (8 waves per SIMD, independent streams) of the
    # plain integer classes
    plain = ["k_and", "k_add", "k_or", "k_sub", "k_lshl", "k_lshr", "k_bfe", "k_cndmask_sgpr", "k_mbcnt_lo", "k_mbcnt_hi", "k_bcnt", "k_lshl_or", "k_and_or"]
    sat = [by[f"{k} x8 independent"][8]["simd_cycles_per_wave_inst"] for k in plain]
    res["bench_constants"] = {"valu_int32": round(sum(sat) / len(sat), 3),
                              "valu_fma_f32": by["k_fma x8 independent"][8]["simd_cycles_per_wave_inst"],
                              "lone_wave_dependent_int32": by["k_and dependent"][1]["wave_cycles_per_inst_s_memtime"],
                              "lds_round_trip_lone_wave": by["k_lds_chase dependent"][1]["wave_cycles_per_inst_s_memtime"],
                              "source": "profiles/r03_valu_rate_ubench.md (tools/valu_rate_ubench.py on this chip: mean saturated SIMD cycles per "
                                        "wave64 instruction of v_and/v_add/v_or/v_sub/v_lshl/v_lshr/v_bfe/v_cndmask/v_mbcnt/v_bcnt/v_lshl_or/v_and_or at 8 waves per SIMD)"}
    with open(os.path.join(outdir, "r03_valu_rate_ubench.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    lines = ["# Cost of one wave64 instruction on gfx950, by class and occupancy (tools/valu_rate_ubench.hip)", "",
             f"Device {res['device']}, {res['cus']} CUs, 2.4 GHz assumed for the kernel-duration column.  `SIMD cyc` = kernel duration x 2.4 GHz x 1024 SIMDs / "
             "wave-instructions issued (what a SIMD is occupied per instruction once it is saturated); `wave cyc` = s_memtime around the loop / instructions "
             "(what ONE wave waits per instruction at that occupancy).", "",
             "| case | instruction(s) | SIMD cyc @1 | @2 | @4 | @8 waves/SIMD | wave cyc @1 | @2 | @4 | @8 |", "|---|---|---|---|---|---|---|---|---|---|"]
    for name, d in by.items():
        lines.append(f"| {name} | {d[1]['what']} | " + " | ".join(f"{d[w]['simd_cycles_per_wave_inst']:.2f}" for w in (1, 2, 4, 8)) + " | " +
                     " | ".join(f"{d[w]['wave_cycles_per_inst_s_memtime']:.1f}" for w in (1, 2, 4, 8)) + " |")
    lines += ["", "bench.py constants: `" + json.dumps(res["bench_constants"]) + "`"]
    with open(os.path.join(outdir, "r03_valu_rate_ubench.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
