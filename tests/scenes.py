"""Small hand-made scenes for the tests (built with the package's scene builder)."""
import numpy as np


def _quad(p0, p1, p2, p3):
    return [list(p0) + list(p1) + list(p2), list(p0) + list(p2) + list(p3)]


def mirror_hall(vrt, refl_front=0.7, refl_back=0.5):
    """Four instances in front of the RTU test's fixed camera ((0,100,0) looking along +x): floor + side
    wall (diffuse), a mirror facing the camera, a mirror behind the camera facing the first one, and a
    blob between them.  Returns the scene buffers as numpy arrays with blas_node_t::reflectivity (@152)
    of the two mirrors set -- the reference's scene builder hard-codes 0 there (scene.cpp:96)."""
    blob = vrt.scene.procedural("blob", 2, 0, 3)
    bt = np.frombuffer(bytes(blob.buffers["tri"]), np.float32).reshape(-1, 3).copy()
    c = bt.mean(0)
    r = np.abs(bt - c).max()
    bt = ((bt - c) * np.float32(40.0 / r) + np.array([180.0, 90.0, 30.0], np.float32)).reshape(-1, 9).astype(np.float32)
    floor = np.array(_quad((-50, 0, -300), (600, 0, -300), (600, 0, 300), (-50, 0, 300)) +
                     _quad((-50, 0, 300), (600, 0, 300), (600, 260, 300), (-50, 260, 300)), np.float32)
    m1 = np.array(_quad((400, 10, -200), (400, 10, 200), (400, 230, 200), (400, 230, -200)), np.float32)
    m2 = np.array(_quad((-40, 10, -220), (-40, 240, -220), (-40, 240, 220), (-40, 10, 220)), np.float32)
    sc = vrt.scene.from_triangles([floor, m1, m2, bt])
    b = {k: np.frombuffer(bytes(v), np.uint8).copy() for k, v in sc.buffers.items()}
    rec = b["blas"].view(np.float32).reshape(-1, 40)
    rec[1, 38] = refl_front
    rec[2, 38] = refl_back
    return b


def chain_bvh4(vrt, k):
    """A chain-like BVH4 in the RTU test's formats, k internal levels deep under one identity instance: node i holds the rest of the
    chain (child 0) and three single-triangle leaves (children 1-3; the last node holds four).  Triangles are stacked along the view
    direction of the fixed camera ((0,100,0), looking along +x), DEEPER = NEARER and smaller, so that a ray through the middle of the
    frame enters the chain child first at every level and leaves three pending siblings behind: 3 k stack entries at the bottom.
    Quantisation as the format defines it (origin + ldexp(q, e), e = ceil(log2(extent / 255)), floor / ceil: conservative)."""
    base = vrt.scene.procedural("cornell")
    n_tri = 3 * k + 1
    tris = np.zeros((n_tri, 9), np.float32)
    for t in range(n_tri):
        level = t // 3                                   # triangles 3i, 3i+1, 3i+2 hang off node i; the last one off node k-1 as well
        x = 420.0 - 5.0 * level - 1.25 * (t % 3)
        s = 300.0 - 5.0 * level - 1.25 * (t % 3)
        tris[t] = [x, 100.0 - s, -s, x, 100.0 + s, -s, x, 100.0 + 0.4 * s, s]
    lo_t, hi_t = tris.reshape(-1, 3, 3).min(1), tris.reshape(-1, 3, 3).max(1)
    n_nodes = 1 + 4 * k
    nodes = np.zeros((n_nodes, 52), np.uint8)

    def subtree_box(i):                                  # everything under node i
        return lo_t[3 * i:].min(0), hi_t[3 * i:].max(0)

    def put(idx, boxes, left_first, leaf_data):
        lo = np.min([b[0] for b in boxes], 0).astype(np.float32)
        hi = np.max([b[1] for b in boxes], 0).astype(np.float32)
        ext = np.maximum(hi.astype(np.float64) - lo, 1e-6)
        e = np.ceil(np.log2(ext / 255.0)).astype(np.int64)
        n = nodes[idx]
        n[0:12] = lo.view(np.uint8)
        n[12:15] = e.astype(np.int8).view(np.uint8)
        n[15] = 0                                        # imask: BLAS node
        n[16:20] = np.array([left_first], np.uint32).view(np.uint8)
        n[20:24] = np.array([leaf_data], np.uint32).view(np.uint8)
        for c, (blo, bhi) in enumerate(boxes):
            q_lo = np.clip(np.floor((blo.astype(np.float64) - lo) / np.exp2(e)), 0, 255)
            q_hi = np.clip(np.ceil((bhi.astype(np.float64) - lo) / np.exp2(e)), 0, 255)
            n[24 + 7 * c] = 1
            n[25 + 7 * c: 28 + 7 * c] = q_lo.astype(np.uint8)
            n[28 + 7 * c: 31 + 7 * c] = q_hi.astype(np.uint8)

    def put_leaf(idx, t):
        n = nodes[idx]
        n[0:12] = lo_t[t].view(np.uint8)
        n[16:20] = np.array([t], np.uint32).view(np.uint8)
        n[20:24] = np.array([1], np.uint32).view(np.uint8)

    pos = 0
    for i in range(k):
        first = 1 + 4 * i                                # children of node i: first .. first + 3
        last = i == k - 1
        kids = [(lo_t[3 * i + 3], hi_t[3 * i + 3]) if last else subtree_box(i + 1)] + [(lo_t[3 * i + j], hi_t[3 * i + j]) for j in range(3)]
        put(pos, kids, first, 0)
        if last:
            put_leaf(first, 3 * i + 3)
        for j in range(3):
            put_leaf(first + 1 + j, 3 * i + j)
        pos = first
    ex = np.zeros((n_tri, 16), np.float32)
    ex[:, 0] = ex[:, 3] = ex[:, 6] = -1.0                # normals facing the camera; texId 0 (bytes 60-63) stays 0
    b = {kk: np.frombuffer(bytes(v), np.uint8).copy() for kk, v in base.buffers.items()}
    b["bvh"] = nodes.reshape(-1)
    b["tri"] = tris.view(np.uint8).reshape(-1)
    b["triEx"] = ex.view(np.uint8).reshape(-1)
    if "triIdx" in b:
        b["triIdx"] = np.arange(n_tri, dtype=np.uint32).view(np.uint8)
    assert b["blas"].size == 160 and b["blas"].view(np.uint32)[0] == 0 and b["tlas"].size == 52
    return vrt.scene.Scene(b, name="chain_bvh4_%d" % k)
