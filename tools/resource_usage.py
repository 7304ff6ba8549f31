#!/usr/bin/env python3
"""Table of hipcc's -Rpass-analysis=kernel-resource-usage remarks (read from a file or stdin): one line per kernel.
  hipcc ... --cuda-device-only -Rpass-analysis=kernel-resource-usage -c unit.hip -o /dev/null 2> remarks.txt; python tools/resource_usage.py remarks.txt [filter]"""
import re, subprocess, sys


def parse(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (.*?)(?: \[-Rpass)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" [")[0]] = v
    return rows


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.splitlines()
        return [re.sub(r"\(.*", "", o) for o in out]
    except OSError:
        return names


if __name__ == "__main__":
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 and sys.argv[1] != "-" else sys.stdin.read()
    rows = parse(text)
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    names = demangle([r["name"] for r in rows])
    print("%-70s %5s %5s %7s %4s %6s %6s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "Sspill", "Vspill", "LDS"))
    for r, n in zip(rows, names):
        if flt and flt not in n:
            continue
        print("%-70s %5s %5s %7s %4s %6s %6s %7s" % (n[-70:], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size")))
