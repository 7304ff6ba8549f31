"""The committed Oval track table (data extracted by tools/extract_track.py from the reference's scene / prefab / FBX
DATA files) against the values SURVEY.md App. A lists, and its internal consistency."""
import math
import numpy as np
from hierarchicalkarting_amd.config import load_track, make_config


def test_oval_table_matches_survey():
    tr = load_track("oval")
    s = tr["sections"]
    assert len(s) == 24 and tr["rules"] == {"MaxLaneChanges": 3, "laps": 4, "maxEpisodeSteps": 6000, "disableOnEnd": 1, "sectionHorizon": 5}
    assert [x["optimalLane"] for x in s] == [4, 4, 4, 3, 1, 2, 3, 3, 1, 2, 4, 4, 4, 4, 4, 3, 1, 2, 3, 3, 1, 3, 4, 4]
    assert [x["trackInsideRadius"] for x in s] == [0, 0, 0, 0, 15, 15, 0, 0, 15, 15, 0, 0, 0, 0, 0, 0, 15, 15, 0, 0, 15, 15, 0, 0]
    assert abs(s[0]["Trigger"]["x"] - 15.870) < 1e-3 and abs(s[0]["Trigger"]["z"] + 2.103) < 1e-3
    assert abs(s[5]["Trigger"]["x"] - 11.118) < 1e-3 and abs(s[5]["Trigger"]["z"] - 50.636) < 1e-3 and abs(s[5]["Trigger"]["yaw_deg"] - 314.98) < 0.01
    assert abs(s[17]["Lane4"]["x"] + 42.00) < 0.01 and abs(s[17]["Lane4"]["z"] + 38.75) < 0.01
    for x in s:                     # Waypoint prefab: lanes at local x = -3.5, -1.25, 1.25, 3.5; trigger 0.407 ahead
        d = [math.hypot(x["Lane%d" % (l + 1)]["x"] - x["waypoint"]["x"], x["Lane%d" % (l + 1)]["z"] - x["waypoint"]["z"]) for l in range(4)]
        assert np.allclose(d, [3.5, 1.25, 1.25, 3.5], atol=1e-4)
        assert abs(math.hypot(x["Trigger"]["x"] - x["waypoint"]["x"], x["Trigger"]["z"] - x["waypoint"]["z"]) - 0.407) < 1e-4
        assert abs(x["Trigger"]["y"] - 0.75) < 1e-3


def test_walls():
    tr = load_track("oval")
    nseg = sum(len(w["points"]) - 1 for w in tr["walls"])
    assert nseg == 16 * 2 + 4 * 64                     # straight: two faces; large curve: two 33-vertex polylines
    # road-side faces are 4.6 m from the centre line on straights (FBX cross-section)
    b = make_config(1, 2)
    w = b.walls
    assert abs(abs(w[0].x0 - 15.88) - 4.6) < 1e-3
    # curve wall radius about the nominal turn centre bulges 15.40 -> 15.91 (Bezier sweep, not an arc)
    curves = [x for x in tr["walls"] if len(x["points"]) == 33][:2]          # the two faces of the first curve piece
    rr = sorted((min(r), max(r)) for r in ([math.hypot(p[0] - (15.88 - 20.0), p[1] - 37.0) for p in c["points"]] for c in curves))
    assert abs(rr[0][0] - 15.40) < 0.02 and abs(rr[0][1] - 15.91) < 0.02
    assert abs(rr[1][0] - 24.60) < 0.02 and abs(rr[1][1] - 25.11) < 0.02   # (SURVEY quotes 25.51: that is the far face, 0.4 m behind)


def test_complex_table():
    tr = load_track("complex")
    s = tr["sections"]
    assert len(s) == 41 and tr["rules"]["laps"] == 3 and tr["rules"]["MaxLaneChanges"] == 4
    kinds = [x["piece"].replace("ModularTrack", "") for x in s]
    # SURVEY App. A piece sequence: S,S,S,S,CLR x2,CLR x2,CMR x2,S,CSL,S,SCL x2,SCL x2,CSR,S,CML x2,CML x2,CLR x2,S,CLR x2,SCL x2,CSR,S x5,SCR x2,S,S,S
    want = (["Straight"] * 4 + ["CurveLargeRight"] * 4 + ["CurveMediumRight"] * 2 + ["Straight", "CurveSmallLeft", "Straight"] +
            ["SCurveLeft"] * 4 + ["CurveSmallRight", "Straight"] + ["CurveMediumLeft"] * 4 + ["CurveLargeRight"] * 2 + ["Straight"] +
            ["CurveLargeRight"] * 2 + ["SCurveLeft"] * 2 + ["CurveSmallRight"] + ["Straight"] * 5 + ["SCurveRight"] * 2 + ["Straight"] * 3)
    assert kinds == want
    assert len(tr["walls"]) == 2 * len({x["piece_instance"] for x in s})       # two road-side faces per piece
    for x in s:
        d = [math.hypot(x["Lane%d" % (l + 1)]["x"] - x["waypoint"]["x"], x["Lane%d" % (l + 1)]["z"] - x["waypoint"]["z"]) for l in range(4)]
        assert np.allclose(d, [3.5, 1.25, 1.25, 3.5], atol=1e-3)
