#!/usr/bin/env python3
"""The trained actors the reference's experiment scenes run (BehaviorParameters.m_Model of the agents in
tests/golden/reference_experiments.json), as plain float32 arrays: tests/golden/reference_actors.npz.

Build container only: reads the .onnx DATA files under /root/reference/Assets/Karting/Prefabs/AI with the in-repo protobuf
reader (hierarchicalkarting_amd/onnx_read.py).  The fixture holds numbers only (weights, biases, the observation normaliser,
log sigma) under "<model file name>/<array>", so that the closed-loop races of tests/test_reference_logs.py run on any box."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hierarchicalkarting_amd.policy import Policy          # noqa: E402

MODELS = "/root/reference/Assets/Karting/Prefabs/AI"


def main():
    exps = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_experiments.json")))
    wanted = set()
    for e in exps:
        if not str(e.get("ExperimentName", "")).endswith(("2", "3")) or "E2E" in str(e.get("ExperimentName")):
            continue
        for a in e["agents"]:
            m = (a.get("behavior") or {}).get("model")
            if m and a.get("LowMode") == 0:
                wanted.add(m)
    out = {}
    for m in sorted(wanted):
        p = Policy.from_onnx(os.path.join(MODELS, m))
        out.update(p.arrays(m + "/"))
        print("%-50s in %d hidden %d layers %d branches %d" % (m, p.in_dim, p.hidden, len(p.W), p.n_branch))
    dst = os.path.join(ROOT, "tests", "golden", "reference_actors.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
