#!/usr/bin/env python3
"""Per-kernel resource table of libhk's translation units: code bytes, VGPRs (arch + accumulation), SGPRs, spills, scratch,
static LDS and the waves per SIMD that follow — read from the code-object metadata of a device-only compile.

  python tools/kernel_stats.py [hk_ga4.hip ...] [--defs -DX ...]  > profiles/r02_kernel_resources.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
LLVM = "/opt/rocm/lib/llvm/bin"


def stats(unit, defs):
    co = "/tmp/hk_stats_%s.co" % unit.replace(".hip", "")
    if not os.environ.get("HK_STATS_REUSE") or not os.path.exists(co):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + ge.HIPCC_FLAGS + ge._backend_flags(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) + defs + ["--cuda-device-only", "-c", os.path.join(ge.CSRC, unit), "-o", co])
    # the device-only output is an offload bundle: take the gfx950 code object out of it
    elf = co + ".elf"
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + co, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
    co = elf
    sym = subprocess.run([LLVM + "/llvm-readelf", "-sW", co], stdout=subprocess.PIPE, text=True).stdout
    size = {}
    for ln in sym.splitlines():
        f = ln.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size[f[7]] = int(f[2])
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], stdout=subprocess.PIPE, text=True).stdout
    out = []
    for blk in notes.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "0"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void ", "")
        vg, ag = int(g("vgpr_count")), int(blk.split()[0])
        tot = vg                                     # gfx950: unified file, .vgpr_count already includes the accumulation registers
        waves = max(1, min(8, 512 // max(((tot + 7) // 8) * 8, 1)))
        out.append((dem, size.get(name, 0), tot, ag, int(g("sgpr_count")), int(g("vgpr_spill_count")), int(g("private_segment_fixed_size")),
                    int(g("group_segment_fixed_size")), waves))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    units = args or ge.UNITS
    print("%-70s %9s %5s %5s %5s %6s %8s %8s %6s" % ("kernel", "code B", "VGPR", "AGPR", "SGPR", "spill", "scratchB", "LDS B", "w/SIMD"))
    for u in units:
        print("# " + u)
        for r in sorted(stats(u, defs), key=lambda r: -r[1]):
            print("%-70s %9d %5d %5d %5d %6d %8d %8d %6d" % ((r[0][:70],) + r[1:]))


if __name__ == "__main__":
    main()
