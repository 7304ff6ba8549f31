#!/usr/bin/env python3
"""sha256 over the sources libhk.so is built from (csrc/*, include/*): what a PMC summary under profiles/ was measured on.  bench.py prints
the committed summary's counters only while this hash still matches the tree it runs from (no stale bytes)."""
import hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha256()
    for d in ("hierarchicalkarting_amd/csrc", "include"):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            if f.endswith((".h", ".hip")):
                h.update(f.encode()); h.update(open(os.path.join(ROOT, d, f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
