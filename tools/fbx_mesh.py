"""Binary FBX (v7xxx) reader: just enough to pull mesh vertices/polygons and model transforms.
Build-container-only helper for tools/extract_track.py (reads reference DATA assets)."""
import struct, zlib
import numpy as np


def _read_prop(b, o):
    t = chr(b[o]); o += 1
    if t == 'Y': return struct.unpack_from('<h', b, o)[0], o + 2
    if t == 'C': return bool(b[o]), o + 1
    if t == 'I': return struct.unpack_from('<i', b, o)[0], o + 4
    if t == 'F': return struct.unpack_from('<f', b, o)[0], o + 4
    if t == 'D': return struct.unpack_from('<d', b, o)[0], o + 8
    if t == 'L': return struct.unpack_from('<q', b, o)[0], o + 8
    if t in 'fdlib':
        n, enc, clen = struct.unpack_from('<III', b, o); o += 12
        raw = b[o:o + clen]; o += clen
        if enc == 1:
            raw = zlib.decompress(raw)
        dt = {'f': '<f4', 'd': '<f8', 'l': '<i8', 'i': '<i4', 'b': 'u1'}[t]
        return np.frombuffer(raw, dtype=dt, count=n).copy(), o
    if t in 'SR':
        n = struct.unpack_from('<I', b, o)[0]; o += 4
        v = b[o:o + n]; o += n
        return (v.decode('latin1') if t == 'S' else v), o
    raise ValueError("bad fbx prop type %r" % t)


def _read_node(b, o, v75):
    if v75:
        end, nprops, plen = struct.unpack_from('<QQQ', b, o); o += 24
    else:
        end, nprops, plen = struct.unpack_from('<III', b, o); o += 12
    nlen = b[o]; o += 1
    name = b[o:o + nlen].decode('latin1'); o += nlen
    if end == 0:
        return None, o
    props = []
    for _ in range(nprops):
        p, o = _read_prop(b, o)
        props.append(p)
    kids = []
    while o < end:
        k, o = _read_node(b, o, v75)
        if k is None:
            break
        kids.append(k)
    return {'name': name, 'props': props, 'kids': kids}, end


def parse(path):
    b = open(path, 'rb').read()
    assert b[:20] == b'Kaydara FBX Binary  ', "not binary FBX"
    ver = struct.unpack_from('<I', b, 23)[0]
    o = 27
    nodes = []
    while o < len(b) - 160:
        n, o = _read_node(b, o, ver >= 7500)
        if n is None:
            break
        nodes.append(n)
    return ver, nodes


def find(nodes, name):
    return [n for n in nodes if n['name'] == name]


def child(n, name):
    r = find(n['kids'], name)
    return r[0] if r else None


def meshes(path):
    """-> list of dict(name, verts[N,3] (fbx units), polys list-of-index-lists), plus models & connections."""
    ver, nodes = parse(path)
    objs = find(nodes, 'Objects')[0]
    geos = {}
    for g in find(objs['kids'], 'Geometry'):
        v = child(g, 'Vertices')
        pi = child(g, 'PolygonVertexIndex')
        if v is None or pi is None:
            continue
        verts = np.asarray(v['props'][0], dtype=np.float64).reshape(-1, 3)
        polys, cur = [], []
        for idx in pi['props'][0]:
            if idx < 0:
                cur.append(int(~idx)); polys.append(cur); cur = []
            else:
                cur.append(int(idx))
        geos[g['props'][0]] = dict(name=g['props'][1], verts=verts, polys=polys)
    models = {}
    for m in find(objs['kids'], 'Model'):
        p70 = child(m, 'Properties70')
        props = {}
        if p70:
            for p in p70['kids']:
                props[p['props'][0]] = p['props'][4:]
        models[m['props'][0]] = dict(name=m['props'][1], props=props)
    conns = []
    c = find(nodes, 'Connections')
    if c:
        for k in c[0]['kids']:
            conns.append(tuple(k['props'][:3]))
    gs = find(nodes, 'GlobalSettings')
    gprops = {}
    if gs:
        p70 = child(gs[0], 'Properties70')
        for p in p70['kids']:
            gprops[p['props'][0]] = p['props'][4:]
    return dict(version=ver, geos=geos, models=models, conns=conns, globals=gprops)


if __name__ == '__main__':
    import sys
    r = meshes(sys.argv[1])
    print('version', r['version'])
    print('globals', {k: v for k, v in r['globals'].items() if 'Axis' in k or 'Scale' in k})
    for gid, g in r['geos'].items():
        v = g['verts']
        print('geo', gid, repr(g['name']), v.shape, 'min', v.min(0), 'max', v.max(0), 'polys', len(g['polys']))
    for mid, m in r['models'].items():
        print('model', mid, repr(m['name']), {k: v for k, v in m['props'].items() if k.startswith('Lcl') or 'Rotation' in k})
    print('conns', r['conns'])
