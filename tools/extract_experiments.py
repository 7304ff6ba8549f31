#!/usr/bin/env python3
"""Resolve the reference's experiment set-ups from its Unity scenes (DATA, build container only).

For every RacingEnvController in Assets/Karting/Scenes/Compete/CompeteAgents-*.unity: ExperimentName, mode, rule parameters
(laps, maxEpisodeSteps, MaxLaneChanges ...), Agents[] in order with each agent's script (HierarchicalKartAgent /
EndToEndKartAgent), HighMode / LowMode, gameParams, sectionHorizon, the kart's baseStats overrides, team wiring
(teamAgents / otherAgents as indices into Agents[]) and the ML-Agents BehaviorParameters (m_Model guid -> .onnx file,
VectorObservationSize, NumStackedVectorObservations, TeamId, BehaviorType), plus DecisionRequester.DecisionPeriod.

Method = SURVEY.md App. A: scene documents split on '--- !u!<class> &<id>'; an agent in Agents[] is a stripped MonoBehaviour whose
m_PrefabInstance carries m_Modifications (target fileID in the source prefab, propertyPath, value | objectReference); values
not overridden come from the prefab document itself.

  python tools/extract_experiments.py            # prints a table
  python tools/extract_experiments.py --update   # writes tests/golden/reference_experiments.json (data; no reference text)
"""
import argparse, glob, json, os, sys
sys.path.insert(0, os.path.dirname(__file__))
import unity_yaml as uy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
ASSETS = os.path.join(REF, "Assets")
SCENES = os.path.join(ASSETS, "Karting", "Scenes", "Compete")
GUID_REC = "b83c52cfa72e3b04aaada8bc4075b0e8"
GUID_HKA = "e8765d53e046ef74898109def269b8cb"
GUID_BP = "5d1c4e0b1822b495aa52bc52839ecb30"      # Unity.MLAgents.Policies.BehaviorParameters
GUID_DR = "3a5c9d521e5ef4759a8246a07d52221e"      # DecisionRequester
HIGH = {0: "MCTS", 1: "Fixed", 2: "Random"}       # HKA:21-33 enum order is checked in main()
LOW = {0: "RL", 1: "MPC", 2: "LQR"}


def script_guid(doc):
    return (doc or {}).get("m_Script", {}).get("guid")


def set_path(d, path, value):
    """apply a Unity propertyPath ('a.b', 'arr.Array.data[3]', 'arr.Array.size') to a nested dict copy"""
    keys = path.replace(".Array.data[", "[").replace(".Array.size", ".__size__").split(".")
    cur = d
    for i, k in enumerate(keys):
        idx = None
        if "[" in k:
            k, idx = k[:-1].split("[")
            idx = int(idx)
        last = i == len(keys) - 1
        if k == "__size__":
            return          # sizes are implied by the data entries we keep
        if idx is None:
            if last:
                cur[k] = value
            else:
                if not isinstance(cur.get(k), dict):
                    cur[k] = {}
                cur = cur[k]
        else:
            if not isinstance(cur.get(k), list):
                cur[k] = []
            while len(cur[k]) <= idx:
                cur[k].append(None)
            if last:
                cur[k][idx] = value
            else:
                if not isinstance(cur[k][idx], dict):
                    cur[k][idx] = {}
                cur = cur[k][idx]


def mod_value(m):
    ref = m.get("objectReference") or {}
    if ref.get("fileID", 0) != 0 or ref.get("guid"):
        return dict(ref)
    return m.get("value")


def num(v):
    try:
        f = float(v)
        return int(f) if f == int(f) and "." not in str(v) else f
    except (TypeError, ValueError):
        return v


class SceneExperiments:
    def __init__(self, scene_path, guid_index):
        self.scene = uy.load(scene_path)
        self.guid = guid_index
        self.path = scene_path

    def prefab_doc(self, guid, file_id):
        pf = uy.load(self.guid[guid])
        ent = pf.docs.get(file_id)
        return pf, (dict(ent[3]) if ent and isinstance(ent[3], dict) else None)

    def component_of_instance(self, inst, prefab_guid, script_guid_wanted):
        """the (fileID, doc) of the component with that script inside the instance's source prefab, overrides applied"""
        pf = uy.load(self.guid[prefab_guid])
        for fid, (cid, stripped, kind, body) in pf.docs.items():
            if kind == "MonoBehaviour" and isinstance(body, dict) and (script_guid(body) or "").startswith(script_guid_wanted):
                doc = json.loads(json.dumps(body))
                for m in inst["m_Modification"]["m_Modifications"]:
                    if m["target"]["fileID"] == fid:
                        set_path(doc, m["propertyPath"], mod_value(m))
                return fid, doc
        return None, None

    def agent(self, agent_fid):
        cid, stripped, kind, d = self.scene.docs[agent_fid]
        out = {"scene_id": agent_fid}
        if not stripped:          # a kart placed in the scene without a prefab (not seen in the Compete scenes)
            out["unresolved"] = True
            return out
        inst_id = d["m_PrefabInstance"]["fileID"]
        inst = self.scene.docs[inst_id][3]
        src = d["m_CorrespondingSourceObject"]
        prefab_guid = inst["m_SourcePrefab"]["guid"]
        out["prefab"] = os.path.basename(self.guid.get(prefab_guid, "?"))
        sg = script_guid(d)
        out["script"] = os.path.basename(self.guid.get(sg, sg or "?"))
        pf, agent_doc = self.prefab_doc(src["guid"], src["fileID"])
        agent_doc = json.loads(json.dumps(agent_doc or {}))
        name = None
        sizes = {}
        for m in inst["m_Modification"]["m_Modifications"]:
            if m["target"]["fileID"] == src["fileID"]:
                set_path(agent_doc, m["propertyPath"], mod_value(m))
                if m["propertyPath"].endswith(".Array.size"):
                    sizes[m["propertyPath"][:-len(".Array.size")]] = int(m["value"])
            if m["propertyPath"] == "m_Name":
                name = m["value"]
        for k, n in sizes.items():          # element overrides beyond the serialized size are stale
            if isinstance(agent_doc.get(k), list):
                agent_doc[k] = agent_doc[k][:n]
        out["name"] = name
        keep = ("Mode", "LowMode", "HighMode", "sectionHorizon", "gameParams", "name", "teamAgents", "otherAgents",
                "is_active", "AgentSensorsMask", "SpeedReward", "TowardsCheckpointReward", "PassCheckpointReward",
                "WallHitPenalty", "OpponentHitPenalty", "HitByOpponentPenalty", "AccelerationReward", "ReversePenalty",
                "SwervingPenalty", "NotAtGoalPenalty", "LaneDifferenceRewardDivider", "VelocityDifferenceRewardDivider",
                "SlowMovingPenalty", "BeingBehindPenalty", "SectionWindow", "sectionWindow", "MCTSSimulatedTime")
        for k in list(agent_doc):
            if k in keep or k.lower().endswith(("reward", "penalty", "divider")):
                out[k] = agent_doc[k]
        out["sensors"] = self.sensors(agent_doc.get("Sensors") or [], src["guid"])
        _, bp = self.component_of_instance(inst, prefab_guid, GUID_BP)
        if bp:
            model = bp.get("m_Model") or {}
            mg = model.get("guid") if isinstance(model, dict) else None
            out["behavior"] = {
                "model": os.path.basename(self.guid[mg]) if mg in self.guid else None,
                "vector_observation_size": num((bp.get("m_BrainParameters") or {}).get("VectorObservationSize")),
                "stacked": num((bp.get("m_BrainParameters") or {}).get("NumStackedVectorObservations")),
                "team_id": num(bp.get("TeamId")), "behavior_type": num(bp.get("m_BehaviorType")),
                "behavior_name": bp.get("m_BehaviorName"),
            }
        _, dr = self.component_of_instance(inst, prefab_guid, GUID_DR)
        if dr:
            out["decision_period"] = num(dr.get("DecisionPeriod"))
            out["take_actions_between_decisions"] = num(dr.get("TakeActionsBetweenDecisions"))
        # ArcadeKart sits three prefabs deep (HierarchicalMLAgent -> Player -> BaseKartClassic); the scene addresses it through
        # the chained-XOR id, so its overrides are recognised by their property path instead
        out["baseStats_overrides"] = {m["propertyPath"][len("baseStats."):]: num(m["value"])
                                      for m in inst["m_Modification"]["m_Modifications"] if m["propertyPath"].startswith("baseStats.")}
        return out

    def sensors(self, sens, agent_prefab_guid):
        """KartAgent.Sensors[] (KA:20-26) as the SCENE holds them: each entry's Transform is a reference — the agent prefab's own, or a
        scene override (the Compete scenes re-point the entries: their order is NOT the prefab's) — to a child of the nested
        MLAgent_Sensors prefab; its local Y rotation is the ray's yaw."""
        import math
        pf = uy.load(self.guid[agent_prefab_guid])
        out = []
        for s in sens:
            s = s or {}
            ref = s.get("Transform") or {}
            fid = ref.get("fileID", 0)
            ent = self.scene.docs.get(fid)
            if ent and ent[1] and isinstance(ent[3], dict):          # a stripped Transform of the scene -> its id in the agent prefab
                fid = ent[3]["m_CorrespondingSourceObject"]["fileID"]
            yaw = None
            pent = pf.docs.get(fid)
            if pent and pent[1] and isinstance(pent[3], dict):       # stripped in the agent prefab -> the MLAgent_Sensors prefab's Transform
                so = pent[3]["m_CorrespondingSourceObject"]
                sp = uy.load(self.guid[so["guid"]])
                t = sp.docs.get(so["fileID"])
                if t and isinstance(t[3], dict):
                    q = t[3]["m_LocalRotation"]
                    x, y, z, w = (float(q[k]) for k in "xyzw")
                    yaw = round(math.degrees(math.atan2(2 * (x * z + w * y), 1 - 2 * (x * x + y * y))), 4)
            out.append({"yaw_deg": yaw, "RayDistance": num(s.get("RayDistance")),
                        "WallHitValidationDistance": num(s.get("WallHitValidationDistance")),
                        "AgentHitValidationDistance": num(s.get("AgentHitValidationDistance"))})
        return out

    def envs(self):
        res = []
        for fid, (cid, stripped, kind, body) in self.scene.docs.items():
            if kind != "MonoBehaviour" or not isinstance(body, dict) or script_guid(body) != GUID_REC or stripped:
                continue
            go = self.scene.docs.get(body.get("m_GameObject", {}).get("fileID"))
            env = {"scene": os.path.basename(self.path), "env_id": fid,
                   "game_object_active": (go[3].get("m_IsActive") if go and isinstance(go[3], dict) else None),
                   "enabled": body.get("m_Enabled")}
            for k, v in body.items():
                if k in ("Teams", "Agents", "Sections", "m_Script", "m_GameObject") or k.startswith("m_"):
                    continue
                if isinstance(v, (int, float, str)) or v is None:
                    env[k] = v
            ids = [a["fileID"] for a in body.get("Agents", [])]
            env["n_sections"] = len(body.get("Sections", []))
            agents = [self.agent(a) for a in ids]
            idx = {a: i for i, a in enumerate(ids)}

            def refs(lst):
                return [idx.get((r or {}).get("fileID"), -1) if isinstance(r, dict) else -1 for r in (lst or [])]
            for a in agents:
                a["teamAgents"] = refs(a.get("teamAgents"))
                a["otherAgents"] = refs(a.get("otherAgents"))
            env["agents"] = agents
            env["teams"] = [[idx.get(r["fileID"], -1) for r in t.get("Racers", [])] for t in body.get("Teams", [])]
            res.append(env)
        return res


def optimal_lanes(scene, env_id):
    """DiscretePositionTracker.optimalLane of the env's Sections[], in order (tools/extract_track.py resolves the prefab instances and the
    scene's overrides).  Every RacingEnvController of an "All" scene owns a copy of the track, and the copies differ in exactly this field:
    CompeteAgents-OvalAll's 1v1 set-ups keep to lanes 3 / 2 on the straights where CompeteAgents-Oval (the track fixture) says 4 / 3."""
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "t.json")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "extract_track.py"), "--scene", "Karting/Scenes/Compete/" + scene, "--env", str(env_id),
                        "--name", "t", "--out", out], check=True, stdout=subprocess.DEVNULL)
        return [int(x["optimalLane"]) for x in json.load(open(out))["sections"]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--update", action="store_true")
    ap.add_argument("--scenes", default="*")
    a = ap.parse_args()
    guid = uy.build_guid_index(ASSETS)
    allenv = []
    for p in sorted(glob.glob(os.path.join(SCENES, "CompeteAgents-%s.unity" % a.scenes))):
        allenv += SceneExperiments(p, guid).envs()
    for e in allenv:
        e["optimal_lanes"] = optimal_lanes(e["scene"], e["env_id"])
        print("%-34s %-36s active=%s laps=%s maxSteps=%s MaxLaneChanges=%s mode=%s sections=%d H=%s" % (
            e["scene"], e.get("ExperimentName"), e["game_object_active"], e.get("laps"), e.get("maxEpisodeSteps"),
            e.get("MaxLaneChanges"), e.get("mode"), e["n_sections"], e.get("sectionHorizon")))
        print("        optimal lanes " + "".join(str(x) for x in e["optimal_lanes"]))
        for i, ag in enumerate(e["agents"]):
            b = ag.get("behavior") or {}
            print("        sensors yaw %s ray %s wall %s agent %s" % tuple([x.get(k) for x in ag.get("sensors", [])] for k in ("yaw_deg", "RayDistance", "WallHitValidationDistance", "AgentHitValidationDistance")))
            print("    [%d] %-14s %-26s high=%s low=%s H=%s depth=%s team=%s others=%s model=%s obs=%sx%s teamId=%s" % (
                i, ag.get("name"), ag.get("script"), HIGH.get(ag.get("HighMode"), ag.get("HighMode")), LOW.get(ag.get("LowMode"), ag.get("LowMode")),
                ag.get("sectionHorizon"), (ag.get("gameParams") or {}).get("treeSearchDepth"), ag.get("teamAgents"), ag.get("otherAgents"),
                b.get("model"), b.get("vector_observation_size"), b.get("stacked"), b.get("team_id")))
    if a.update:
        out = os.path.join(ROOT, "tests", "golden", "reference_experiments.json")
        json.dump(allenv, open(out, "w"), indent=1, sort_keys=True)
        print("wrote", out)


if __name__ == "__main__":
    main()
