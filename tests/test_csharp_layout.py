"""The C# host binding (host/HkNative.cs) against the C ABI, without a C# compiler:
  1. every field of every struct of include/hk.h: gcc's offsetof / sizeof == the ctypes mirror (hierarchicalkarting_amd/_lib.py);
  2. every [StructLayout(Sequential)] struct of HkNative.cs, parsed from the source: same field names in the same order, same
     element types and array lengths as the ctypes mirror, and — laid out with the CLR's sequential rule (each field at its
     natural alignment, size rounded to the largest alignment) — the same offsets and size;
  3. every [DllImport] names a symbol hk.h declares, and every declared symbol is bound."""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hierarchicalkarting_amd import _lib  # noqa: E402

PAIRS = [("hk_kart_stats", "HkKartStats", _lib.KartStats), ("hk_section", "HkSection", _lib.Section), ("hk_wall_seg", "HkWallSeg", _lib.WallSeg),
         ("hk_reward_params", "HkRewardParams", _lib.RewardParams), ("hk_engine_params", "HkEngineParams", _lib.EngineParams),
         ("hk_config", "HkConfig", _lib.Config),
         ("hk_mcts_plan", "HkMctsPlan", _lib.MctsPlan), ("hk_mcts_state", "HkMctsState", _lib.MctsState),
         ("hk_agent_state", "HkAgentState", _lib.AgentState), ("hk_env_state", "HkEnvState", _lib.EnvState),
         ("hk_episode_result", "HkEpisodeResult", _lib.EpisodeResult), ("hk_lq_debug", "HkLqDebug", _lib.LqDebug),
         ("hk_policy_desc", "HkPolicyDesc", _lib.PolicyDesc)]
CS_TYPES = {"float": ("f", 4), "int": ("i", 4), "uint": ("u", 4), "byte": ("b", 1), "double": ("d", 8), "long": ("l", 8)}
CT_KIND = {C.c_float: ("f", 4), C.c_int32: ("i", 4), C.c_uint32: ("u", 4), C.c_uint8: ("b", 1), C.c_double: ("d", 8), C.c_int64: ("l", 8)}


def test_ctypes_mirror_equals_c_layout_field_by_field(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % os.path.join(ROOT, "include", "hk.h"), "int main(){"]
    want = []
    for cname, _, ct in PAIRS:
        lines.append('printf("%%zu\\n", sizeof(%s));' % cname)
        want.append(C.sizeof(ct))
        for fname, _ in ct._fields_:
            lines.append('printf("%%zu\\n", offsetof(%s, %s));' % (cname, fname))
            want.append(getattr(ct, fname).offset)
    lines.append("return 0;}")
    src = tmp_path / "off.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "off"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == want


def _flatten(ct):
    """ctypes field -> (kind, elem size, count) | ("struct", class) | ("ptr",)"""
    n = 1
    while hasattr(ct, "_length_") and hasattr(ct, "_type_") and not isinstance(ct._type_, str):
        n *= ct._length_
        ct = ct._type_
    if isinstance(ct, type) and issubclass(ct, C.Structure):
        return ("struct", ct, n)
    if isinstance(ct, type) and issubclass(ct, (C._Pointer,)):
        return ("ptr", None, n)
    return CT_KIND[ct] + (n,)


def _parse_cs():
    txt = open(os.path.join(ROOT, "host", "HkNative.cs")).read()
    txt = re.sub(r"//[^\n]*", "", txt)
    structs = {}
    for m in re.finditer(r"\[StructLayout\(LayoutKind\.Sequential\)\]\s*public\s+(?:unsafe\s+)?struct\s+(\w+)\s*\{(.*?)\n    \}", txt, flags=re.S):
        fields = []
        for f in re.finditer(r"public\s+(fixed\s+)?([\w]+)(\s*\*)?\s+(\w+)(?:\[(\d+)\])?\s*;", m.group(2)):
            fixed, typ, ptr, name, cnt = f.groups()
            fields.append((name, typ, bool(ptr), int(cnt) if cnt else 1))
        structs[m.group(1)] = fields
    imports = re.findall(r"\[DllImport\(Lib\)\]\s*public static extern\s+[\w\*]+\s+(hk_\w+)\s*\(", txt)
    return structs, imports


def test_csharp_structs_match_the_abi():
    structs, _ = _parse_cs()
    cs_name_of = {ct: cs for _, cs, ct in PAIRS}
    sizes = {}
    for cname, cs, ct in PAIRS:            # PAIRS is ordered so that nested structs come first
        assert cs in structs, "HkNative.cs lacks %s" % cs
        fields = list(structs[cs])
        off, align = 0, 1
        k = 0
        for fname, ftype in ct._fields_:
            kind, esz, cnt = _flatten(ftype)
            if kind == "ptr" and cnt > 1:          # const float* W[4] <-> W0 .. W3
                names = [fname + str(j) for j in range(cnt)]
            else:
                names = [fname]
            for j, nm in enumerate(names):
                assert k < len(fields), "%s: missing field %s" % (cs, nm)
                cs_name, cs_type, cs_ptr, cs_cnt = fields[k]
                k += 1
                assert cs_name == nm, "%s: field %d is %s, hk.h has %s" % (cs, k, cs_name, nm)
                if kind == "ptr":
                    assert cs_ptr, "%s.%s must be a pointer" % (cs, nm)
                    sz, al, total = 8, 8, 8
                elif kind == "struct":
                    assert cs_type == cs_name_of[esz] and not cs_ptr, "%s.%s: nested struct type" % (cs, nm)
                    sz, al = sizes[esz]
                    total = sz * cnt
                else:
                    assert not cs_ptr and CS_TYPES[cs_type] == (kind, esz), "%s.%s: C# %s vs C %s%d" % (cs, nm, cs_type, kind, esz * 8)
                    assert cs_cnt == cnt, "%s.%s: array length %d vs %d" % (cs, nm, cs_cnt, cnt)
                    sz, al, total = esz, esz, esz * cnt
                off = (off + al - 1) // al * al
                want_off = getattr(ct, fname).offset + (j * 8 if kind == "ptr" else 0)
                assert off == want_off, "%s.%s: sequential layout puts it at %d, C at %d" % (cs, nm, off, want_off)
                off += total
                align = max(align, al)
        assert k == len(fields), "%s has extra fields: %s" % (cs, fields[k:])
        size = (off + align - 1) // align * align
        assert size == C.sizeof(ct), "%s: size %d vs %d" % (cs, size, C.sizeof(ct))
        sizes[ct] = (size, align)


def test_dllimports_are_the_declared_symbols():
    _, imports = _parse_cs()
    assert len(imports) == len(set(imports))
    assert set(imports) == set(_lib.SYMBOLS), sorted(set(_lib.SYMBOLS) ^ set(imports))
    consts = open(os.path.join(ROOT, "host", "HkNative.cs")).read()
    assert "HK_ABI_VERSION = %d" % _lib.HK_ABI_VERSION in consts


def test_agent_class_forwards_the_mlagents_surface():
    """HkKartAgent must override the three ML-Agents entry points the reference's agents override (HKA:485, KA:440, KA:508)"""
    src = open(os.path.join(ROOT, "host", "HkKartAgent.cs")).read()
    for sig in ("public override void CollectObservations(VectorSensor sensor)", "public override void OnActionReceived(ActionBuffers actions)",
                "public override void Heuristic(in ActionBuffers actionsOut)", "class HkKartAgent : Agent"):
        assert sig in src, sig
    ctl = open(os.path.join(ROOT, "host", "HkRacingEnvController.cs")).read()
    used = set(re.findall(r"Hk\.(hk_\w+)\(", ctl))
    assert used <= set(_lib.SYMBOLS) and {"hk_create", "hk_reset", "hk_set_actions", "hk_step", "hk_get_observations", "hk_get_agent_state",
                                         "hk_get_episode_results", "hk_destroy"} <= used
