"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/echoscene_oracle.py for the rules).

Marching cubes as the reference applies it after the sampling path: ``mcubes.marching_cubes(sdf_i, level)`` in
model/diff_utils/util_3d.py:214-217 (PyMCubes; third party, not in this image, no version pinned by the reference's
requirements.txt).  Restated from the published algorithm (Lorensen & Cline 1987; cube / edge numbering of P. Bourke's
"Polygonising a scalar field", which PyMCubes' marchingcubes.cpp uses):

  * corner m of a cell is inside when value < level; a grid edge with endpoints on different sides carries one vertex at
    x1 + (x2 - x1) * (level - f1) / (f2 - f1) (mc_isovalue_interpolation), shared by the cells around the edge;
  * inside a cell the vertices are connected face by face; a face with four vertices is resolved by cutting off its INSIDE
    corners; the resulting closed loops are oriented with their normal towards the inside side and fanned.

Parity status: PARTLY PINNED.  No output of PyMCubes itself can be recorded here.  What this oracle shares with it by
construction: the vertex set (one vertex per sign-changing edge, same interpolation formula), hence also the vertex
multiset, and the loop structure of every cell without a four-vertex face.  Not pinned: the diagonal chosen inside a loop
and the resolution of ambiguous faces of the third-party table -- quantities that do not depend on them up to O(h^2)
(area, volume) are what tests compare.

This implementation is deliberately NOT table driven (the product is): every cell is polygonised by tracing its loops
directly (memoised per case), written independently of echoscene_amd/mc_tables.py.
"""
import numpy as np

_CORNER = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
_EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
_FACE = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]
_EID = {}
for _e, (_a, _b) in enumerate(_EDGE):
    _EID[(_a, _b)] = _EID[(_b, _a)] = _e
_memo = {}


def _polygonise(case):
    """list of triangles (triples of cube-edge ids) of one case, by tracing the loops"""
    if case in _memo:
        return _memo[case]
    ins = [bool(case >> m & 1) for m in range(8)]
    link = {}
    for f in _FACE:
        es = [_EID[(f[i], f[(i + 1) % 4])] for i in range(4)]
        cut = [ins[f[i]] != ins[f[(i + 1) % 4]] for i in range(4)]
        k = sum(cut)
        if k == 2:
            a, b = [es[i] for i in range(4) if cut[i]]
            link.setdefault(a, []).append(b)
            link.setdefault(b, []).append(a)
        elif k == 4:
            for i in range(4):
                if ins[f[i]]:                       # the two face edges meeting at an inside corner
                    a, b = es[i], es[(i + 3) % 4]
                    link.setdefault(a, []).append(b)
                    link.setdefault(b, []).append(a)
    tris, done = [], set()
    for start in sorted(link):
        if start in done:
            continue
        ring, prev, cur = [start], -1, start
        done.add(start)
        while True:
            n0, n1 = link[cur]
            nxt = n0 if n0 != prev else n1
            if nxt == start and len(ring) > 2:
                break
            ring.append(nxt)
            done.add(nxt)
            prev, cur = cur, nxt
        # orientation (exact, local): ring[0] -> ring[1] lies on one cube face; seen from outside that face the surface
        # normal (towards the inside side) demands that the inside endpoint of ring[0] is on the LEFT of the segment
        e0, e1 = ring[0], ring[1]
        face = next(f for f in _FACE if e0 in [_EID[(f[i], f[(i + 1) % 4])] for i in range(4)]
                    and e1 in [_EID[(f[i], f[(i + 1) % 4])] for i in range(4)])
        fc = np.mean([_CORNER[c] for c in face], 0)
        n_out = fc - np.array([0.5, 0.5, 0.5])                       # outward normal of that face (not normalised)
        mid = lambda e: (np.array(_CORNER[_EDGE[e][0]], dtype=float) + np.array(_CORNER[_EDGE[e][1]], dtype=float)) / 2
        d = mid(e1) - mid(e0)
        a_in = _EDGE[e0][0] if ins[_EDGE[e0][0]] else _EDGE[e0][1]
        u = np.array(_CORNER[a_in], dtype=float) - mid(e0)
        if float(np.dot(np.cross(n_out, d), u)) < 0:
            ring = ring[::-1]
        # fan apex: the first ring position whose diagonals avoid joining two vertices of one cube face (such a diagonal
        # lies IN the face and would coincide with the neighbour cell's -> an edge shared by four triangles)
        def on_one_face(ea, eb):
            return any(ea in fe and eb in fe for fe in ([_EID[(f[i], f[(i + 1) % 4])] for i in range(4)] for f in _FACE))
        L = len(ring)
        cost = [sum(on_one_face(ring[p0], ring[(p0 + i) % L]) for i in range(2, L - 1)) for p0 in range(L)]
        p0 = cost.index(min(cost))
        ring = ring[p0:] + ring[:p0]
        tris += [(ring[0], ring[i], ring[i + 1]) for i in range(1, L - 1)]
    _memo[case] = tris
    return tris


def marching_cubes(vol, level):
    """vol [n0,n1,n2] -> (verts float64 [V,3] in index units, faces int64 [T,3]); vertices shared per grid edge."""
    vol = np.asarray(vol, dtype=np.float64)
    inside = vol < level
    n0, n1, n2 = vol.shape
    vid = {}
    verts = []

    def vertex(p, q):                               # p, q grid points (tuples), p < q lexicographically
        key = (p, q)
        if key not in vid:
            f1, f2 = vol[p], vol[q]
            t = (level - f1) / (f2 - f1)
            vid[key] = len(verts)
            verts.append([p[d] + (q[d] - p[d]) * t for d in range(3)])
        return vid[key]

    # only cells that straddle the level need work
    c = inside[:-1, :-1, :-1].astype(np.int32)
    tot = np.zeros_like(c)
    for (dx, dy, dz) in _CORNER:
        tot += inside[dx:n0 - 1 + dx, dy:n1 - 1 + dy, dz:n2 - 1 + dz]
    faces = []
    for (i, j, k) in np.argwhere((tot > 0) & (tot < 8)):
        case = 0
        for m, (dx, dy, dz) in enumerate(_CORNER):
            if inside[i + dx, j + dy, k + dz]:
                case |= 1 << m
        for tri in _polygonise(case):
            ids = []
            for e in tri:
                a, b = _EDGE[e]
                p = (i + _CORNER[a][0], j + _CORNER[a][1], k + _CORNER[a][2])
                q = (i + _CORNER[b][0], j + _CORNER[b][1], k + _CORNER[b][2])
                if q < p:
                    p, q = q, p
                ids.append(vertex(p, q))
            faces.append(ids)
    return np.asarray(verts, dtype=np.float64).reshape(-1, 3), np.asarray(faces, dtype=np.int64).reshape(-1, 3)


def mesh_invariants(verts, faces):
    """Quantities that do not depend on vertex / face order: counts, signed volume, area, Euler characteristic, and the
    number of edges that are not shared by exactly two triangles with opposite directions (0 = closed, consistently
    oriented surface)."""
    v, f = np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    vol = float(np.einsum('ij,ij->i', a, np.cross(b, c)).sum() / 6.0)
    area = float(np.linalg.norm(np.cross(b - a, c - a), axis=1).sum() / 2.0)
    he = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = he[:, 0] * (len(v) + 1) + he[:, 1]
    rev = he[:, 1] * (len(v) + 1) + he[:, 0]
    uk, cnt = np.unique(key, return_counts=True)
    bad = int((cnt != 1).sum()) + int((~np.isin(rev, uk)).sum())
    n_edges = len(np.unique(np.minimum(key, rev)))
    return dict(V=len(v), F=len(f), volume=vol, area=area, euler=len(v) - n_edges + len(f), open_or_inconsistent_edges=bad)
