#!/usr/bin/env python3
"""Generates sdf-viewer_amd/csrc/mc_table.inc: the triangle table of the cube-by-cube isosurface extractor.

Nothing is copied from a published table: each of the 256 corner-sign cases is triangulated here from first
principles, so the conventions are this file's own.
  corner c          : x = c & 1, y = (c >> 1) & 1, z = (c >> 2) & 1; case bit c set <=> corner inside (d < 0)
  edge e = 4*a + s  : runs along axis a; s = u + 2*v with (u, v) the corner coordinates on the two other axes
                      in increasing axis order.  An edge belongs to the lattice point at its low end.
  faces             : on every face the crossing points are joined by segments directed from the crossing where
                      a counter-clockwise walk (seen from outside the cube) LEAVES the inside to the one where it
                      ENTERS it.  A face with four crossings (two diagonal inside corners) cuts each inside corner
                      off on its own -- a rule that depends on the face's signs only, so the two cells sharing a
                      face agree and the mesh is watertight.
  loops             : every crossing point has one incoming and one outgoing segment; following them gives closed
                      polygons, triangulated so that no diagonal lies in a cube face (triangulate_loop).  Triangles are counter-clockwise seen from the outside
                      (positive distance), checked below against the trilinear interpolant of the corner signs.
Run: python tools/gen_mc_table.py            (rewrites the .inc; the committed copy must match -- tests check)
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "sdf-viewer_amd", "csrc", "mc_table.inc")


def corner_xyz(c):
    return (c & 1, (c >> 1) & 1, (c >> 2) & 1)


def corner_id(xyz):
    return xyz[0] | (xyz[1] << 1) | (xyz[2] << 2)


def edge_id(c0, c1):
    a = [i for i in range(3) if corner_xyz(c0)[i] != corner_xyz(c1)[i]]
    assert len(a) == 1
    a = a[0]
    others = [i for i in range(3) if i != a]
    lo = corner_xyz(c0)
    return 4 * a + lo[others[0]] + 2 * lo[others[1]]


def edge_ends(e):
    a, s = divmod(e, 4)
    others = [i for i in range(3) if i != a]
    lo = [0, 0, 0]
    lo[others[0]], lo[others[1]] = s & 1, s >> 1
    hi = list(lo)
    hi[a] = 1
    return corner_id(lo), corner_id(hi)


def faces():
    """Each face as its 4 corners in counter-clockwise order seen from outside the cube."""
    out = []
    for axis in range(3):
        for side in (0, 1):
            u, v = [i for i in range(3) if i != axis]
            ring = []
            for (a, b) in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [0, 0, 0]
                p[axis], p[u], p[v] = side, a, b
                ring.append(p)
            # orientation: (p1-p0) x (p3-p0) must point along the outward normal
            n = np.cross(np.subtract(ring[1], ring[0]), np.subtract(ring[3], ring[0]))
            outward = np.zeros(3)
            outward[axis] = 1 if side else -1
            if np.dot(n, outward) < 0:
                ring.reverse()
            out.append([corner_id(p) for p in ring])
    return out


FACES = faces()


def case_loops(case):
    inside = [(case >> c) & 1 for c in range(8)]
    nxt = {}
    for ring in FACES:
        leaving, entering = [], []  # (position in ring, edge)
        for k in range(4):
            c0, c1 = ring[k], ring[(k + 1) % 4]
            if inside[c0] and not inside[c1]:
                leaving.append((k, edge_id(min(c0, c1), max(c0, c1))))
            elif not inside[c0] and inside[c1]:
                entering.append((k, edge_id(min(c0, c1), max(c0, c1))))
        if len(leaving) == 1:
            nxt[leaving[0][1]] = entering[0][1]
        elif len(leaving) == 2:
            # cut each inside corner off alone: the walk leaves the inside right after corner ring[k+... ]:
            # leaving crossing on edge k (c_k inside -> c_k+1 outside) pairs with the entering crossing on
            # edge k-1 (c_k-1 outside -> c_k inside): both touch the same inside corner c_k.
            for k, e in leaving:
                partner = [ee for kk, ee in entering if kk == (k - 1) % 4]
                assert len(partner) == 1
                nxt[e] = partner[0]
        else:
            assert not leaving and not entering
    loops, seen = [], set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop, e = [], start
        while e not in seen:
            seen.add(e)
            loop.append(e)
            e = nxt[e]
        assert e == start and len(loop) >= 3
        loops.append(loop)
    return loops


def faces_of_edge(e):
    """The two cube faces (axis, side) a cube edge lies in."""
    c0, c1 = edge_ends(e)
    p0, p1 = corner_xyz(c0), corner_xyz(c1)
    return {(ax, p0[ax]) for ax in range(3) if p0[ax] == p1[ax]}


def triangulations(n):
    """All triangulations of a convex n-gon with vertices 0..n-1, as lists of index triples."""
    def rec(lo, hi):  # polygon lo, lo+1, ..., hi
        if hi - lo < 2:
            return [[]]
        out = []
        for mid in range(lo + 1, hi):
            for left in rec(lo, mid):
                for right in rec(mid, hi):
                    out.append(left + [(lo, mid, hi)] + right)
        return out
    return rec(0, n - 1)


def triangulate_loop(loop):
    """Triangles (as edge-id triples, loop orientation kept) such that no diagonal joins two crossing points of one cube
    face: such a diagonal lies IN that face, where the neighbouring cell may lay the same segment -- four triangles on
    one mesh edge.  Every loop of every case admits such a triangulation (asserted)."""
    n = len(loop)
    for tris in triangulations(n):
        ok = True
        for a, b, c in tris:
            for u, v in ((a, b), (b, c), (c, a)):
                if (v - u) % n in (1, n - 1):
                    continue  # a side of the loop (a face segment), not a diagonal
                if faces_of_edge(loop[u]) & faces_of_edge(loop[v]):
                    ok = False
        if ok:
            return [(loop[a], loop[b], loop[c]) for a, b, c in tris]
    raise AssertionError(f"no face-free triangulation for loop {loop}")


def edge_mid(e):
    c0, c1 = edge_ends(e)
    return (np.array(corner_xyz(c0), float) + np.array(corner_xyz(c1), float)) / 2


def trilinear(case, p):
    val = 0.0
    for c in range(8):
        w = 1.0
        for i, bit in enumerate(corner_xyz(c)):
            w *= p[i] if bit else 1 - p[i]
        val += w * (-1.0 if (case >> c) & 1 else 1.0)
    return val


def build():
    table = []
    flip = None
    for case in range(256):
        tris = []
        for loop in case_loops(case):
            pts = [edge_mid(e) for e in loop]
            centre = np.mean(pts, axis=0)
            normal = sum(np.cross(pts[i] - centre, pts[(i + 1) % len(pts)] - centre) for i in range(len(pts)))
            normal = normal / np.linalg.norm(normal)
            up = trilinear(case, centre + 0.05 * normal) - trilinear(case, centre - 0.05 * normal)
            assert abs(up) > 1e-6, (case, loop)
            want_flip = up < 0
            if flip is None:
                flip = want_flip
            assert flip == want_flip, "segment direction rule must orient every loop the same way"
            if flip:
                loop = loop[::-1]
            tris.extend(triangulate_loop(loop))
        table.append(tris)
    return table


def render(table):
    width = 3 * max(len(t) for t in table)
    lines = ["// mc_table.inc -- GENERATED by tools/gen_mc_table.py (conventions documented there); do not edit.",
             "// The includer defines SDFV_MC_TABLE (e.g. `__constant__ const`): statically initialised device tables are",
             "// loaded with the code object on every device, no upload call needed.",
             f"constexpr int kMcMaxIndices = {width};",
             "SDFV_MC_TABLE unsigned char kMcTriCount[256] = {"]
    for r in range(0, 256, 32):
        lines.append("    " + ", ".join(str(len(t)) for t in table[r:r + 32]) + ",")
    lines.append("};")
    lines.append("// edge ids (4*axis + u + 2*v) of each triangle's corners, -1 padded")
    lines.append(f"SDFV_MC_TABLE signed char kMcTriEdges[256][{width}] = {{")
    for case, tris in enumerate(table):
        flat = [e for t in tris for e in t]
        flat += [-1] * (width - len(flat))
        lines.append("    {" + ", ".join(f"{e:2d}" for e in flat) + f"}},  // {case}")
    lines.append("};")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    t = build()
    with open(OUT, "w") as f:
        f.write(render(t))
    print("wrote", OUT, "max triangles per cell:", max(len(x) for x in t),
          "total triangles over all cases:", sum(len(x) for x in t))
