#!/usr/bin/env python3
"""Static guard against a code-generation fault of this toolchain's AMDGPU back end (found in round 3).

In the join block of a divergent if / else the back end restores EXEC with `s_or_b64 exec, exec, s[..]`.  When the register allocator
spills a VGPR tuple that is live ACROSS the branch into that block, it can place the `scratch_store ... ; Folded Spill` BEFORE that
instruction: the store then runs for the lanes of the else side only, the other lanes keep whatever an earlier iteration left in the
slot, and the reload (under the full mask) hands them a stale value.  Nothing in the source can cause or prevent it; it moves with any
change of register pressure (round 2's "loop-form sensitivity").  This tool disassembles the device code of a library (or reads a .s
listing) and reports every spill store that sits in a block ahead of that block's `s_or_b64 exec, exec, ...`.

  python tools/check_spill_exec.py [hierarchicalkarting_amd/libhk.so | file.s ...]      exit status 1 when a kernel is affected"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def device_asm(path):
    """-> {kernel name: [instruction text, ...]} from a .s listing or from the gfx950 code objects bundled in a shared library"""
    if path.endswith(".s"):
        text = open(path).read()
    else:
        text = ""
        with tempfile.TemporaryDirectory() as td:
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + path], capture_output=True)
            # a hipcc shared library keeps its code objects in the .hip_fatbin section: unbundle every gfx950 entry
            raw = os.path.join(td, "fat.bin")
            subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, raw])
            data = open(raw, "rb").read()
            k = 0
            for m in re.finditer(rb"\x7fELF", data):
                co = os.path.join(td, "co%d.elf" % k); k += 1
                open(co, "wb").write(data[m.start():])
                r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True)
                if "s_endpgm" in r.stdout:
                    text += r.stdout
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^(?:[0-9a-f]+ <)?([A-Za-z_][\w$.]*)>?:", line.strip()) if not line.startswith(("\t", " ")) or line.rstrip().endswith(">:") else None
        if m and not m.group(1).startswith((".L", "L")):
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is not None:
            cur.append(line.strip())
    return kernels


def check(insts):
    """spill stores / reloads between a LABEL and that block's `s_or_b64 exec, exec, ...` with no other change of EXEC in between: the
    block is the join of a divergent branch (the label is where the skipped side lands) and has not restored the mask yet.  (A store at
    the tail of a then-block also precedes an `s_or_b64 exec`, but follows the `s_and_saveexec` that opened the block: that is the
    side's own value of the slot, and fine.)"""
    # who jumps to each label
    preds = {}
    for ins in insts:
        m = re.match(r"^(s_cbranch_\w+|s_branch)\s+(\S+)", ins)
        if m:
            preds.setdefault(m.group(2), []).append(m.group(1))
    bad, pending, after_label, prev_op = [], [], False, ""
    for n, ins in enumerate(insts):
        op = ins.split()[0] if ins.split() else ""
        head = ins.split(";")[0].strip()                      # (listings put comments after labels)
        if not head:
            continue
        if head.endswith(":") or re.match(r"^[0-9a-f]+ <[^>]+>:$", head):
            # A label entered only by `s_cbranch_execnz` (and by falling out of the `s_cbranch_exec*` in front of it) opens a block under
            # the mask that was just narrowed: its own reloads are its own business.  Any other label is where control flow meets again.
            name = head[:-1]
            falls = not prev_op.startswith(("s_branch", "s_endpgm", "s_setpc"))
            guarded = all(b == "s_cbranch_execnz" for b in preds.get(name, [])) and (not falls or prev_op.startswith("s_cbranch_exec"))
            pending, after_label = [], not guarded
            continue
        prev_op = op
        if re.match(r"^s_or_b64\s+exec,\s*exec,", ins):
            bad += pending
            pending, after_label = [], False
            continue
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64") or op.endswith("_saveexec_b64") or re.match(r"^s_\w+_b64\s+exec\b", ins):
            pending, after_label = [], False
            continue
        if after_label and ((op.startswith(("scratch_store", "buffer_store")) and "Spill" in ins) or (op.startswith(("scratch_load", "buffer_load")) and "Reload" in ins)):
            pending.append((n, ins))          # (a compiler listing marks its spills and reloads; a reload there leaves the other lanes' register stale)
        # The second pattern (round 5, profiles/r05_c_variant_fault.txt): the same placement with a REGISTER COPY instead of a scratch store.  When the allocator
        # splits a live range around a region it saves the value with `v_mov_b32 vS, vX` (or into an AGPR), reuses vX inside, and copies it back afterwards;
        # put at the top of a join block, ahead of the EXEC restore, the save runs for the fall-through side's lanes only, while the clobber and the copy
        # back run under wider masks: the other lanes come out with whatever vS held (a finished kart's tele_total_time read 4.6e-41 in the Training kernel).
        elif after_label and COPIES and re.match(r"^(v_mov_b32(_e32)?|v_mov_b64(_e32)?|v_accvgpr_write_b32|v_accvgpr_read_b32|v_accvgpr_mov_b32)\s+[va](\d+|\[\d+:\d+\]),\s*[va](\d+|\[\d+:\d+\])\s*$", head):
            pending.append((n, ins + "   ; live-range copy"))
    return bad


COPIES = os.environ.get("HK_GUARD_NO_COPIES") is None      # HK_GUARD_NO_COPIES=1: round 3's pattern only (spill stores / reloads)


def main(argv):
    paths = argv or [os.path.join(ROOT, "hierarchicalkarting_amd", "libhk.so")]
    rc = 0
    for p in paths:
        ks = device_asm(p)
        n_spill = 0
        for name, insts in ks.items():
            n_spill += sum(1 for i in insts if i.startswith("scratch_store") and "Spill" in i)
            for n, ins in check(insts):
                print("%s: %s: instruction %d: %s   <- spill store ahead of the block's EXEC restore" % (os.path.basename(p), name[:90], n, ins[:90]))
                rc = 1
        print("%s: %d kernels, %d spill stores checked, %s" % (os.path.basename(p), len(ks), n_spill, "AFFECTED" if rc else "clean"))
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
