"""Convex hulls with many vertices: the vertex graph behind the hill-climbing support function.

The reference's default hand collides every `plastic_collision` mesh of the menagerie Shadow Hand (forearm, wrist,
palm, thumb links, the `*_distal_pst` fingertips) through its convex hull
(/root/reference/robopianist/models/hands/shadow_hand.py:144-152, shadow_hand_constants.py:52-53; MuJoCo's
mjc_Convex).  A support query on such a hull is a maximisation over its vertices: hulls of up to `SCAN_MAX` vertices
are scanned (first maximum wins); larger ones are walked -- start at vertex 0, move to the neighbour with the largest
dot product while one is strictly larger than the current value (MuJoCo does the same for meshes with a vertex
graph).  The oracle (oracle/rp_oracle.c: geom_support) and the engine (csrc/rp_narrow.hpp: hull_support_wave) follow
the same walk over the same neighbour lists, so both find the same vertex.

`attach_graphs` rewrites the model's hull tables in place: a large hull keeps only the vertices ON its hull (in their
original order) and gets one row of `mesh_graph` per vertex: [degree, neighbour 0, neighbour 1, ...] (vertex indices
within the hull's own set, ascending, padded with -1)."""
from __future__ import annotations

import warnings

import numpy as np

SCAN_MAX = 32      # hulls up to this many vertices are scanned
GRAPH_ROW = 24     # ints per vertex: degree + up to 23 neighbours

GEOM_MESH = 7


def hull_graph(v: np.ndarray):
    """(hull vertices in their original order, graph rows [n][GRAPH_ROW]) of the point set v [n][3], or None when a
    vertex of the triangulated hull has more neighbours than a row holds (the caller then scans the set)."""
    v = np.asarray(v, float).reshape(-1, 3)
    try:
        from scipy.spatial import ConvexHull, QhullError
    except ImportError:   # (no scipy: the set is scanned; hulls beyond the scanned table's capacity then fail loudly there)
        warnings.warn("scipy is not importable: convex hulls with more than %d vertices are scanned, not walked" % SCAN_MAX)
        return None
    try:
        hull = ConvexHull(v)            # (triangulated facets)
    except QhullError as e:             # (degenerate / coplanar mesh: no graph, the set is scanned)
        warnings.warn("ConvexHull failed on a %d-vertex mesh (%s): the set is scanned, not walked"
                      % (len(v), str(e).strip().splitlines()[0] if str(e).strip() else "QhullError"))
        return None
    keep = np.sort(hull.vertices)
    remap = -np.ones(len(v), np.int64); remap[keep] = np.arange(len(keep))
    nbrs = [set() for _ in keep]
    for tri in hull.simplices:
        a, b, c = (int(remap[i]) for i in tri)
        nbrs[a].update((b, c)); nbrs[b].update((a, c)); nbrs[c].update((a, b))
    if max(len(s) for s in nbrs) > GRAPH_ROW - 1:
        return None
    rows = -np.ones((len(keep), GRAPH_ROW), np.int32)
    for i, s in enumerate(nbrs):
        ids = sorted(s)
        rows[i, 0] = len(ids); rows[i, 1:1 + len(ids)] = ids
    return v[keep], rows


def attach_graphs(m) -> None:
    """Adds `mesh_graph` [nmeshvert][GRAPH_ROW] and `geom_vertgraph` [ngeom] (1: walk the graph, 0: scan) to the
    model; hulls with more than SCAN_MAX vertices are reduced to the vertices on their hull."""
    ngeom = int(m["ngeom"])
    vertadr = np.asarray(m["geom_vertadr"], np.int32).copy(); vertnum = np.asarray(m["geom_vertnum"], np.int32).copy()
    mv = np.asarray(m["mesh_vert"], float).reshape(-1, 3)
    flag = np.zeros(ngeom, np.int32)
    out_v, out_g, cache = [], [], {}
    n_out = 0
    for g in range(ngeom):
        if int(m["geom_type"][g]) != GEOM_MESH or vertnum[g] <= 0:
            continue
        v = mv[vertadr[g]:vertadr[g] + vertnum[g]]
        rows = -np.ones((len(v), GRAPH_ROW), np.int32)
        if len(v) > SCAN_MAX:
            key = v.tobytes()
            if key not in cache:
                cache[key] = hull_graph(v)
            if cache[key] is not None:
                v, rows = cache[key]
                flag[g] = 1
        vertadr[g], vertnum[g] = n_out, len(v)
        out_v.append(v); out_g.append(rows); n_out += len(v)
    m["geom_vertadr"] = vertadr; m["geom_vertnum"] = vertnum
    m["mesh_vert"] = np.concatenate(out_v, 0) if out_v else np.zeros((0, 3))
    m["mesh_graph"] = np.concatenate(out_g, 0) if out_g else np.zeros((0, GRAPH_ROW), np.int32)
    m["geom_vertgraph"] = flag
    m["nmeshvert"] = n_out
