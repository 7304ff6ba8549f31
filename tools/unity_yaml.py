"""Minimal Unity YAML / prefab resolver used by tools/extract_track.py.

Runs ONLY in the build container (reads /root/reference scene/prefab DATA files);
its output (tests/golden/*.json, hierarchicalkarting_amd/data/*.json) is what is committed.
"""
import os, re, yaml, functools

HDR = re.compile(r"^--- !u!(\d+) &(-?\d+)( stripped)?\s*$")


class UFile:
    def __init__(self, path):
        self.path = path
        self.docs = {}      # fileID -> (classID, stripped, dict)
        cur = None
        buf = []
        with open(path, "r", encoding="utf-8", errors="replace") as f:
            for line in f:
                m = HDR.match(line)
                if m:
                    self._flush(cur, buf)
                    cur = (int(m.group(1)), int(m.group(2)), bool(m.group(3)))
                    buf = []
                elif cur is not None:
                    buf.append(line)
        self._flush(cur, buf)

    def _flush(self, cur, buf):
        if cur is None:
            return
        cid, fid, stripped = cur
        try:
            body = yaml.safe_load("".join(buf))
        except Exception:
            body = None
        if isinstance(body, dict) and len(body) == 1:
            kind, body = next(iter(body.items()))
        else:
            kind = None
        self.docs[fid] = (cid, stripped, kind, body)

    def by_kind(self, kind):
        return {k: v for k, v in self.docs.items() if v[2] == kind}


@functools.lru_cache(maxsize=None)
def load(path):
    return UFile(path)


def build_guid_index(root):
    idx = {}
    for d, _, files in os.walk(root):
        for fn in files:
            if fn.endswith(".meta"):
                p = os.path.join(d, fn)
                try:
                    with open(p, "r", errors="replace") as f:
                        for line in f:
                            if line.startswith("guid:"):
                                idx[line.split()[1]] = p[:-5]
                                break
                except OSError:
                    pass
    return idx
