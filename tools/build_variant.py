#!/usr/bin/env python3
"""Kernel-variant experiments: build/libhk_<name>.so = the default objects with hk_ga4.hip (the quad kernels) recompiled
with extra -D flags.  Run a variant with HK_LIB_PATH=build/libhk_<name>.so (it travels to the GPU box with gpurun).

  python tools/build_variant.py <name> [-DFOO=1 ...] [--unit hk_ga4.hip]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    name = sys.argv[1]
    head = sys.argv[2:sys.argv.index("--flags")] if "--flags" in sys.argv else sys.argv[2:]
    defs = [a for a in head if a.startswith("-D") or a.startswith("-m") or a.startswith("-f") or a.startswith("-O")]
    if "--flags" in sys.argv:                       # everything after --flags goes to hipcc verbatim (e.g. --flags -mllvm -some-option=1)
        defs += sys.argv[sys.argv.index("--flags") + 1:]
    unit = "hk_ga4.hip"
    if "--unit" in sys.argv:
        unit = sys.argv[sys.argv.index("--unit") + 1]
    ge.build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if "--all" in head:                             # every unit with the flags (constants the host code shares with the kernels)
        objs, procs = [], []
        for u in ge.UNITS:
            o = os.path.join(ge.OBJ_DIR, "%s_%s.o" % (name, u.replace(".hip", "")))
            objs.append(o)
            procs.append(subprocess.Popen([hipcc] + ge.HIPCC_FLAGS + defs + ["-c", os.path.join(ge.CSRC, u), "-o", o]))
        assert all(p.wait() == 0 for p in procs)
        lib = os.path.join(ROOT, "build", "libhk_%s.so" % name)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs)
        print(lib)
        return
    obj = os.path.join(ge.OBJ_DIR, "%s_%s.o" % (name, unit.replace(".hip", "")))
    subprocess.check_call([hipcc] + ge.HIPCC_FLAGS + defs + ["-c", os.path.join(ge.CSRC, unit), "-o", obj])
    objs = [obj if u == unit else os.path.join(ge.OBJ_DIR, u.replace(".hip", ".o")) for u in ge.UNITS]
    lib = os.path.join(ROOT, "build", "libhk_%s.so" % name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
