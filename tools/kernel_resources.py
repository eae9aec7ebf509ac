#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the HIP library (compiler remarks, no GPU needed).

    python tools/kernel_resources.py [--scratch] [substring ...]

A kernel that shows scratch bytes, or whose VGPR count jumped, regressed at compile time: by-value
argument structs indexed with a run-time index get copied to scratch.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only_scratch = "--scratch" in sys.argv
    src = os.path.join(ROOT, "sdqlpy_amd", "csrc", "sdqh_hip.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-munsafe-fp-atomics", "-fno-gpu-rdc", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(src),
           "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_kernel_resources.so", src]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        text = m.group(1).strip()
        if text.startswith("Function Name:"):
            cur = {"name": text.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in text:
            k, v = text.split(":", 1)
            cur[k.strip()] = v.strip()
    names = demangle([r["name"] for r in rows])
    print("%-6s %-6s %-8s %-4s %s" % ("VGPR", "SGPR", "scratch", "occ", "kernel"))
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("void sdqh::", "").replace("sdqh::", "")
        if args and not any(a in n for a in args):
            continue
        scratch = int(r.get("ScratchSize [bytes/lane]", "0"))
        if only_scratch and not scratch:
            continue
        print("%-6s %-6s %-8d %-4s %s" % (r.get("VGPRs", "?"), r.get("TotalSGPRs", "?"), scratch, r.get("Occupancy [waves/SIMD]", "?"), n))


if __name__ == "__main__":
    main()
