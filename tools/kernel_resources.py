#!/usr/bin/env python3
"""Register / LDS / spill table of every kernel of a HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
   python tools/kernel_resources.py [api.hip|decoder.hip ...] [-D...]      -> markdown table on stdout"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = [("VGPRs", "VGPRs"), ("SGPRs", "TotalSGPRs"), ("scratch B/lane", r"ScratchSize \[bytes/lane\]"), ("VGPR spills", "VGPRs Spill"),
        ("SGPR spills", "SGPRs Spill"), ("LDS B", r"LDS Size \[bytes/block\]"), ("occupancy", r"Occupancy \[waves/SIMD\]")]


def main():
    srcs = [a for a in sys.argv[1:] if not a.startswith("-")] or ["api.hip", "decoder.hip"]
    flags = [a for a in sys.argv[1:] if a.startswith("-")]
    print("| kernel | " + " | ".join(k for k, _ in KEYS) + " |\n|---|" + "---|" * len(KEYS))
    for src in srcs:
        p = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                            os.path.join(ROOT, "icer_compression_amd", "csrc", src), "-o", "/dev/null"] + flags, capture_output=True, text=True)
        for b in re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]:
            name = subprocess.run(["c++filt", b.split(" ")[0]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name).replace("icer::", "").replace("(anonymous namespace)::", "").replace("void ", "")
            vals = []
            for _, pat in KEYS:
                m = re.search(pat + r": (\d+)", b)
                vals.append(m.group(1) if m else "?")
            print(f"| `{name}` | " + " | ".join(vals) + " |")


if __name__ == "__main__":
    main()
