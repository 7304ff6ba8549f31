#!/usr/bin/env python3
"""Regenerate the track-table DATA fixture from the reference's Unity scene / prefab / FBX assets.

Build-container-only (needs /root/reference). Output is pure data (numbers), committed as
hierarchicalkarting_amd/data/<track>.json and mirrored in tests/golden/.  Method = SURVEY.md App. A:
  * Unity YAML split on '--- !u!<class> &<id>'; prefab-instance overrides are m_Modifications
    (target fileID, propertyPath, value); nested-prefab object id = instanceFileID XOR sourceFileID.
  * walls = horizontal slice (y = 0.78 m, the sensor-ray height) of each track piece's FBX collision mesh,
    road-side faces only, in world coordinates.
Coordinates: Unity world (x, z) metres; yaw = Unity Y-rotation in degrees (0 = +z, 90 = +x).
"""
import sys, os, json, math, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
import unity_yaml as uy
import fbx_mesh

REF = "/root/reference"
ASSETS = os.path.join(REF, "Assets")
DPT_GUID_PREFIX = "f5f3f07a"
WAYPOINT_DPT_ID = 2712598285713173431      # DPT MonoBehaviour inside Waypoint.prefab
WAYPOINT_ROOT_T = 6850553478831941103      # root Transform inside Waypoint.prefab
MASK64 = (1 << 64) - 1


def sxor(a, b):
    """Unity nested-prefab id: 64-bit XOR, reinterpreted as signed."""
    r = (a & MASK64) ^ (b & MASK64)
    return r - (1 << 64) if r >= (1 << 63) else r


# ---------------- transforms (Unity conventions) ----------------
def q_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw*bx + ax*bw + ay*bz - az*by,
                     aw*by - ax*bz + ay*bw + az*bx,
                     aw*bz + ax*by - ay*bx + az*bw,
                     aw*bw - ax*bx - ay*by - az*bz])


def q_rot(q, v):
    x, y, z, w = q
    u = np.array([x, y, z]); v = np.asarray(v, float)
    return v + 2.0*np.cross(u, np.cross(u, v) + w*v)


class T:
    def __init__(self, p=(0, 0, 0), q=(0, 0, 0, 1), s=(1, 1, 1)):
        self.p = np.array(p, float); self.q = np.array(q, float); self.s = np.array(s, float)

    def __mul__(self, o):      # self = parent, o = local child
        return T(self.p + q_rot(self.q, self.s*o.p), q_mul(self.q, o.q), self.s*o.s)

    def point(self, v):
        return self.p + q_rot(self.q, self.s*np.asarray(v, float))

    def yaw_deg(self):
        f = q_rot(self.q, [0, 0, 1])
        return math.degrees(math.atan2(f[0], f[2])) % 360.0


def t_from_doc(d):
    lp, lr, ls = d["m_LocalPosition"], d["m_LocalRotation"], d["m_LocalScale"]
    return T((lp["x"], lp["y"], lp["z"]), (lr["x"], lr["y"], lr["z"], lr["w"]), (ls["x"], ls["y"], ls["z"]))


def apply_mods(t, mods, target_id):
    """override local pos/rot/scale of transform `target_id` from an m_Modifications list."""
    t = T(t.p.copy(), t.q.copy(), t.s.copy())
    for m in mods:
        if m["target"]["fileID"] != target_id:
            continue
        pp = m["propertyPath"]; 
        try:
            val = float(m["value"])
        except (TypeError, ValueError):
            continue
        if pp.startswith("m_LocalPosition."):
            t.p["xyz".index(pp[-1])] = val
        elif pp.startswith("m_LocalRotation."):
            t.q["xyzw".index(pp[-1])] = val
        elif pp.startswith("m_LocalScale."):
            t.s["xyz".index(pp[-1])] = val
    return t


def mods_props(mods, target_id):
    out = {}
    for m in mods:
        if m["target"]["fileID"] == target_id:
            out[m["propertyPath"]] = m["value"] if m.get("objectReference", {}).get("fileID", 0) == 0 else m["objectReference"]
    return out


class Resolver:
    def __init__(self, scene_path):
        self.guid = uy.build_guid_index(ASSETS)
        self.scene = uy.load(scene_path)

    def prefab(self, guid):
        return uy.load(self.guid[guid])

    # world transform of a plain (non-stripped) or stripped scene Transform id
    def scene_world(self, tid):
        if tid == 0:
            return T()
        cid, stripped, kind, d = self.scene.docs[tid]
        if not stripped:
            return self.scene_world(d["m_Father"]["fileID"]) * t_from_doc(d)
        # stripped transform: belongs to a prefab instance in the scene
        inst = self.scene.docs[d["m_PrefabInstance"]["fileID"]][3]
        src = d["m_CorrespondingSourceObject"]
        return self.instance_world(inst, src["fileID"])

    def instance_world(self, inst, src_tid):
        """world transform of transform `src_tid` (id in the instance's source prefab) for scene-level instance `inst`."""
        mods = inst["m_Modification"]["m_Modifications"]
        parent_w = self.scene_world(inst["m_Modification"]["m_TransformParent"]["fileID"])
        pf = self.prefab(inst["m_SourcePrefab"]["guid"])
        return parent_w * self.prefab_local_chain(pf, src_tid, mods)

    def prefab_local_chain(self, pf, tid, mods):
        """transform of `tid` relative to the prefab root's parent, with overrides `mods` (ids in pf space)."""
        cid, stripped, kind, d = pf.docs[tid]
        assert not stripped
        loc = apply_mods(t_from_doc(d), mods, tid)
        fa = d["m_Father"]["fileID"]
        if fa == 0:
            return loc
        return self.prefab_local_chain(pf, fa, mods) * loc


def slice_chains(fbx_path, y_m=0.78):
    """horizontal slice of the main mesh -> list of polylines [(x,z),...] in Unity piece-local metres."""
    r = fbx_mesh.meshes(fbx_path)
    # main mesh = geometry connected to the root model (largest vertex count)
    roots = {c[1] for c in r["conns"] if c[0] == "OO" and c[2] == 0}
    gids = [c[1] for c in r["conns"] if c[0] == "OO" and c[2] in roots and c[1] in r["geos"]]
    g = r["geos"][gids[0]]
    V = g["verts"]; y = y_m*100.0
    segs = []
    for poly in g["polys"]:
        for i in range(1, len(poly) - 1):
            tri = [V[poly[0]], V[poly[i]], V[poly[i+1]]]
            pts = []
            for a, b in ((0, 1), (1, 2), (2, 0)):
                ya, yb = tri[a][1] - y, tri[b][1] - y
                if (ya < 0) != (yb < 0):
                    t = ya/(ya - yb)
                    pts.append(tri[a] + t*(tri[b] - tri[a]))
            if len(pts) == 2:
                a, b = pts
                segs.append(((-a[0]/100.0, a[2]/100.0), (-b[0]/100.0, b[2]/100.0)))   # Unity x = -FBX x, cm -> m
    # chain by shared endpoints
    key = lambda p: (round(p[0], 4), round(p[1], 4))
    adj = {}
    for a, b in segs:
        if key(a) == key(b):
            continue
        adj.setdefault(key(a), []).append(b); adj.setdefault(key(b), []).append(a)
    used = set(); chains = []
    ends = [k for k, v in adj.items() if len(v) == 1]
    for start in ends + sorted(adj.keys()):
        if start in used:
            continue
        ch = [start]; used.add(start); cur = start
        while True:
            nxt = [key(p) for p in adj[cur] if key(p) not in used]
            if not nxt:
                break
            cur = nxt[0]; used.add(cur); ch.append(cur)
        if len(adj[start]) == 2 and start in [key(p) for p in adj[cur]] and len(ch) > 2:
            ch.append(start)                       # closed loop
        chains.append(ch)
    # split at sharp corners (> 45 deg): a wall outline loop = 2 long faces + 2 end caps
    split = []
    for ch in chains:
        closed = ch[0] == ch[-1] and len(ch) > 3
        pts = ch[:-1] if closed else ch
        n = len(pts)
        def corner(i):
            a = np.array(pts[(i-1) % n]); b = np.array(pts[i]); c = np.array(pts[(i+1) % n])
            u = b - a; v = c - b
            cs = np.dot(u, v)/(np.linalg.norm(u)*np.linalg.norm(v) + 1e-30)
            return cs < math.cos(math.radians(45))
        if closed:
            corners = [i for i in range(n) if corner(i)]
            if not corners:
                split.append(ch); continue
            for a_i, b_i in zip(corners, corners[1:] + [corners[0] + n]):
                split.append([pts[j % n] for j in range(a_i, b_i + 1)])
        else:
            cur = [pts[0]]
            for i in range(1, n - 1):
                cur.append(pts[i])
                if corner(i):
                    split.append(cur); cur = [pts[i]]
            cur.append(pts[-1]); split.append(cur)
    chains = split
    # drop collinear interior points
    out = []
    for ch in chains:
        pts = [ch[0]]
        for i in range(1, len(ch) - 1):
            a = np.array(pts[-1]); b = np.array(ch[i]); c = np.array(ch[i+1])
            cr = (b[0]-a[0])*(c[1]-b[1]) - (b[1]-a[1])*(c[0]-b[0])
            if abs(cr) > 2e-3*np.linalg.norm(b-a)*np.linalg.norm(c-b) + 1e-12:
                pts.append(ch[i])
        pts.append(ch[-1])
        out.append(pts)
    return out


def chain_len(ch):
    return sum(math.dist(ch[i], ch[i+1]) for i in range(len(ch)-1))


def _poly_dist(p, poly):
    best = 1e30
    for i in range(len(poly) - 1):
        ax, az = poly[i]; bx, bz = poly[i + 1]
        dx, dz = bx - ax, bz - az
        l2 = dx * dx + dz * dz
        t = 0.0 if l2 == 0 else max(0.0, min(1.0, ((p[0] - ax) * dx + (p[1] - az) * dz) / l2))
        best = min(best, math.hypot(ax + t * dx - p[0], az + t * dz - p[1]))
    return best


def _mean_dist(a, b):
    return sum(_poly_dist(p, b) for p in a) / len(a)


def road_side_walls(chains, inside_radius=None, width=None, left_turn=None):
    """pick the two road-facing wall polylines of a piece (piece-local), for ANY piece shape.  A wall outline at the slice
    height is a loop: two long faces 0.4 m apart plus two 0.4 m end caps.  Pair the long faces that belong to one wall
    (mean distance < 0.6 m); of each pair, the road-side face is the one nearer to the other wall."""
    long = [c for c in chains if chain_len(c) > 1.0]
    pairs, used = [], set()
    for i, a in enumerate(long):
        if i in used:
            continue
        best, bj = 1e30, -1
        for j, b in enumerate(long):
            if j == i or j in used:
                continue
            d = max(_mean_dist(a, b), _mean_dist(b, a))
            if d < best:
                best, bj = d, j
        if bj >= 0 and best < 0.6:
            pairs.append((i, bj)); used.add(i); used.add(bj)
    if len(pairs) != 2 or len(used) != len(long):
        raise RuntimeError("unexpected wall outline structure: %d long faces, %d pairs" % (len(long), len(pairs)))
    out = []
    for k, (i, j) in enumerate(pairs):
        oi, oj = pairs[1 - k]
        other = long[oi] + long[oj]
        di = sum(_poly_dist(p, long[oi]) + _poly_dist(p, long[oj]) for p in long[i]) / len(long[i])
        dj = sum(_poly_dist(p, long[oi]) + _poly_dist(p, long[oj]) for p in long[j]) / len(long[j])
        out.append(long[i] if di < dj else long[j])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="Karting/Scenes/Compete/CompeteAgents-Oval.unity")
    ap.add_argument("--env", type=int, default=1132641209, help="fileID of the RacingEnvController MonoBehaviour")
    ap.add_argument("--name", default="oval")
    ap.add_argument("--out", nargs="+", default=["hierarchicalkarting_amd/data/oval_track.json"])
    a = ap.parse_args()
    R = Resolver(os.path.join(ASSETS, a.scene))
    env = R.scene.docs[a.env][3]
    sections = []
    pieces = {}
    wp = uy.load(os.path.join(ASSETS, "Karting/Prefabs/Props/Waypoint.prefab"))
    wp_children = {}
    for tid, (cid, st, kind, d) in wp.docs.items():
        if kind == "Transform" and d["m_Father"]["fileID"] == WAYPOINT_ROOT_T:
            go = wp.docs[d["m_GameObject"]["fileID"]][3]
            wp_children[go["m_Name"]] = tid
    wp_dpt = wp.docs[WAYPOINT_DPT_ID][3]
    for si, ref in enumerate(env["Sections"]):
        cid, st, kind, d = R.scene.docs[ref["fileID"]]
        inst_id = d["m_PrefabInstance"]["fileID"]
        inst = R.scene.docs[inst_id][3]
        src_dpt = d["m_CorrespondingSourceObject"]["fileID"]          # id in piece prefab space (nested)
        pf_guid = inst["m_SourcePrefab"]["guid"]
        pf = R.prefab(pf_guid)
        # which nested Waypoint instance inside the piece prefab?
        w_inst_id = None
        for fid, (c2, s2, k2, d2) in pf.docs.items():
            if k2 == "PrefabInstance" and sxor(fid, WAYPOINT_DPT_ID) == src_dpt:
                w_inst_id = fid
        assert w_inst_id is not None, "nested waypoint not found"
        w_inst = pf.docs[w_inst_id][3]
        scene_mods = inst["m_Modification"]["m_Modifications"]
        piece_mods = w_inst["m_Modification"]["m_Modifications"]
        # piece root world
        root_tid = [fid for fid, (c2, s2, k2, d2) in pf.docs.items() if k2 == "Transform" and not s2 and d2["m_Father"]["fileID"] == 0][0]
        piece_w = R.instance_world(inst, root_tid)
        # waypoint root: local = Waypoint.prefab root, overridden by piece-prefab mods, then by scene mods (nested id)
        wroot = apply_mods(t_from_doc(wp.docs[WAYPOINT_ROOT_T][3]), piece_mods, WAYPOINT_ROOT_T)
        wroot = apply_mods(wroot, scene_mods, sxor(w_inst_id, WAYPOINT_ROOT_T))
        par = w_inst["m_Modification"]["m_TransformParent"]["fileID"]
        par_w = R.instance_world(inst, par)
        wp_w = par_w * wroot
        rec = {"index": si, "piece": os.path.basename(R.guid[pf_guid]).replace(".prefab", ""), "piece_instance": inst_id}
        for nm in ("Trigger", "Lane1", "Lane2", "Lane3", "Lane4"):
            ct = apply_mods(t_from_doc(wp.docs[wp_children[nm]][3]), piece_mods, wp_children[nm])
            ct = apply_mods(ct, scene_mods, sxor(w_inst_id, wp_children[nm]))
            w = wp_w * ct
            rec[nm] = {"x": float(w.p[0]), "y": float(w.p[1]), "z": float(w.p[2]), "yaw_deg": w.yaw_deg()}
        rec["waypoint"] = {"x": float(wp_w.p[0]), "y": float(wp_w.p[1]), "z": float(wp_w.p[2]), "yaw_deg": wp_w.yaw_deg()}
        # DPT scalar fields: Waypoint.prefab default <- piece prefab mods <- scene mods
        props = {k: wp_dpt.get(k) for k in ("trackInsideRadius", "trackLength", "trackWidth", "leftTurn", "turnDegrees", "optimalLane")}
        props.update({k: v for k, v in mods_props(piece_mods, WAYPOINT_DPT_ID).items() if k in props})
        props.update({k: v for k, v in mods_props(scene_mods, src_dpt).items() if k in props})
        for k in props:
            rec[k] = float(props[k]) if k not in ("leftTurn", "optimalLane") else int(float(props[k]))
        sections.append(rec)
        if inst_id not in pieces:
            pieces[inst_id] = (pf_guid, pf, piece_w, rec)
    # walls
    walls = []
    for inst_id, (pf_guid, pf, piece_w, rec) in pieces.items():
        mc = [d2 for fid, (c2, s2, k2, d2) in pf.docs.items() if k2 == "MeshCollider"]
        assert mc, "piece has no MeshCollider"
        fbx = R.guid[mc[0]["m_Mesh"]["guid"]]
        chains = slice_chains(fbx)
        for ch in road_side_walls(chains, rec["trackInsideRadius"], rec["trackWidth"], rec["leftTurn"]):
            pts = [piece_w.point([p[0], 0.0, p[1]]) for p in ch]
            walls.append({"piece_instance": inst_id, "points": [[float(p[0]), float(p[2])] for p in pts]})
    out = {
        "name": a.name,
        "source": {"scene": a.scene, "env_fileID": a.env, "method": "tools/extract_track.py (SURVEY App. A)"},
        "rules": {k: env[k] for k in ("MaxLaneChanges", "laps", "maxEpisodeSteps", "disableOnEnd", "sectionHorizon")},
        "sections": sections,
        "walls": walls,
    }
    for o in a.out:
        os.makedirs(os.path.dirname(o), exist_ok=True)
        with open(o, "w") as f:
            json.dump(out, f, indent=1)
    print("sections", len(sections), "pieces", len(pieces), "wall polylines", len(walls),
          "wall segments", sum(len(w["points"]) - 1 for w in walls))
    for s in sections:
        print("%2d %-28s R=%4.1f deg=%4.1f opt=%d trig=(%8.3f,%8.3f) yaw=%7.2f L1=(%7.2f,%7.2f) L4=(%7.2f,%7.2f)" % (
            s["index"], s["piece"], s["trackInsideRadius"], s["turnDegrees"], s["optimalLane"], s["Trigger"]["x"], s["Trigger"]["z"],
            s["Trigger"]["yaw_deg"], s["Lane1"]["x"], s["Lane1"]["z"], s["Lane4"]["x"], s["Lane4"]["z"]))


if __name__ == "__main__":
    main()
