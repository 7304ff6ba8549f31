#!/usr/bin/env python3
"""Kernel-variant experiments: build/libhk_<name>.so = the default objects with hk_ga4.hip (the quad kernels) recompiled
with extra -D flags.  Run a variant with HK_LIB_PATH=build/libhk_<name>.so (it travels to the GPU box with gpurun).

  python tools/build_variant.py <name> [-DFOO=1 ...] [--unit hk_ga4.hip]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge


def main():
    name = sys.argv[1]
    head = sys.argv[2:sys.argv.index("--flags")] if "--flags" in sys.argv else sys.argv[2:]
    head = [a for a in head if a not in ("--no-backend-flags", "--no-record-flags")]      # --no-record-flags: without the guard variant's flags the product's unit was built with            # --no-backend-flags: without __graft_entry__.BACKEND_FLAGS (the A/B of those switches)
    defs = [a for a in head if a.startswith("-D") or a.startswith("-m") or a.startswith("-f") or a.startswith("-O")]
    if "--flags" in sys.argv:                       # everything after --flags goes to hipcc verbatim (e.g. --flags -mllvm -some-option=1)
        defs += sys.argv[sys.argv.index("--flags") + 1:]
    unit = "hk_ga4.hip"
    if "--unit" in sys.argv:
        unit = sys.argv[sys.argv.index("--unit") + 1]
    unit_list = sys.argv[sys.argv.index("--units") + 1].split(",") if "--units" in sys.argv else None      # --units a.hip,b.hip: several units with the flags
    ge.build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    units = list(ge.UNITS) if "--all" in head else (unit_list or [unit])       # --all: every unit with the flags (constants the host code shares with the kernels)
    objs, procs, listings = [], [], []
    import json
    try:
        record = json.load(open(os.path.join(ge.OBJ_DIR, "codegen_guard.json")))  # the product's units may carry a guard variant's flags: an A/B keeps them
    except (OSError, ValueError):
        record = {}
    for u in ge.UNITS:
        if u not in units:
            objs.append(os.path.join(ge.OBJ_DIR, u.replace(".hip", ".o")))
            continue
        o = os.path.join(ge.OBJ_DIR, "%s_%s.o" % (name, u.replace(".hip", "")))
        objs.append(o)
        listings.append(o[:-2] + ".s")
        base = ge.HIPCC_FLAGS + ([] if "--no-backend-flags" in sys.argv else ge._backend_flags(hipcc)) + ([] if "--no-record-flags" in sys.argv else record.get(u, {}).get("flags", []))
        procs.append(subprocess.Popen([hipcc] + base + defs + ["-c", os.path.join(ge.CSRC, u), "-o", o]))
        procs.append(subprocess.Popen([hipcc] + base + defs + ["--cuda-device-only", "-S", os.path.join(ge.CSRC, u), "-o", o[:-2] + ".s"], stderr=subprocess.DEVNULL))
    assert all(p.wait() == 0 for p in procs)
    lib = os.path.join(ROOT, "build", "libhk_%s.so" % name)
    objs.append(os.path.join(ge.OBJ_DIR, "hk_build_info.o"))          # hk_build_info(): the product's record (a variant is an experiment and says so in its name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs)
    # the code-generation guard (tools/check_spill_exec.py): a variant may be AFFECTED — it is an experiment, not the product — and says so
    import check_spill_exec as guard
    bad = []
    for lst in listings:
        for kname, insts in guard.device_asm(lst).items():
            bad += [{"listing": os.path.basename(lst), "kernel": kname, "instruction": ins} for _, ins in guard.check(insts)]
    json.dump({"library": os.path.basename(lib), "flags": defs, "spill_stores_ahead_of_exec_restore": bad}, open(lib[:-3] + ".guard.json", "w"), indent=1)
    print(lib, "(code-generation guard: %s)" % ("AFFECTED in %d place(s)" % len(bad) if bad else "clean"))


if __name__ == "__main__":
    main()
