#!/usr/bin/env python3
"""CPU prototype behind the GPU builder's clustering step (csrc/bvh_builder.hip): for the same Morton-sorted triangles, the 4-wide
SAH cost (expected bytes per random ray: 52 B x area of every node record, leaves included, + 36 B x triangles x area of every
leaf, over the root's area) of
  * the binary radix tree (Karras 2012) collapsed to 4-wide by largest surface area -- what vxrt_bvh_build did up to round 3,
  * the PLOC tree (Meister & Bittner 2018: mutual nearest neighbours within `radius` positions of the Morton order, merged in
    rounds) collapsed the same way,
  * the package's CPU SAH tree (csrc/scene_builder.cpp), decoded from its node array.
No GPU, no oracle.  usage: python tools/ploc_prototype.py [--level 7] [--radius 8 16] [--leaf-max 2]"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vrt = importlib.import_module("vortex-raytracing_amd")


FORCE_DIV = 16


def area(lo, hi):
    d = hi - lo
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]


def morton_order(lo, hi):
    c = 0.5 * (lo + hi)
    cl, ch = c.min(0), c.max(0)
    ext = np.maximum(ch - cl, 1e-30)
    q = np.clip(((c - cl) / ext * 2097152.0), 0, 2097151).astype(np.uint64)
    e3 = (ch - cl).astype(np.float64).copy()
    used = [0, 0, 0]
    key = np.zeros(len(c), np.uint64)
    for _ in range(63):
        a = max((k for k in range(3) if used[k] < 21), key=lambda k: (e3[k], -k))
        key = (key << np.uint64(1)) | ((q[:, a] >> np.uint64(20 - used[a])) & np.uint64(1))
        used[a] += 1
        e3[a] *= 0.5
    return np.argsort(key, kind="stable"), key


def radix_tree(keys):
    """binary radix tree over sorted keys: returns left, right, count per internal node (ids 0..n-2; leaf j = n-1+j), root 0"""
    n = len(keys)
    left = np.zeros(n - 1, np.int64); right = np.zeros(n - 1, np.int64)
    nxt = 1
    stack = [(0, 0, n - 1)]   # (node id, first, last)
    ks = keys.tolist()
    while stack:
        i, f, l = stack.pop()
        a, b = ks[f], ks[l]
        if a == b:
            s = (f + l) >> 1
        else:
            bit = (a ^ b).bit_length() - 1
            mask = ~((1 << bit) - 1)
            pref = b & mask   # first key with that bit set and the common prefix
            s = int(np.searchsorted(keys[f:l + 1], np.uint64(pref), side="left")) + f - 1
        for side, (cf, cl) in enumerate(((f, s), (s + 1, l))):
            if cf == cl:
                cid = n - 1 + cf
            else:
                cid = nxt; nxt += 1
                stack.append((cid, cf, cl))
            (left if side == 0 else right)[i] = cid
    return left, right


def fit(left, right, lo, hi):
    """boxes + counts of a binary tree whose internal nodes were created parents-first (children have larger ids) or any order:
    resolved by a post-order walk from the root"""
    n = len(lo)
    blo = np.zeros((2 * n - 1, 3), np.float32); bhi = np.zeros((2 * n - 1, 3), np.float32); cnt = np.zeros(2 * n - 1, np.int64)
    blo[n - 1:] = lo; bhi[n - 1:] = hi; cnt[n - 1:] = 1
    order = []
    st = [0]
    L = left.tolist(); R = right.tolist()
    while st:
        i = st.pop()
        order.append(i)
        for c in (L[i], R[i]):
            if c < n - 1: st.append(c)
    for i in reversed(order):
        l, r = L[i], R[i]
        blo[i] = np.minimum(blo[l], blo[r]); bhi[i] = np.maximum(bhi[l], bhi[r]); cnt[i] = cnt[l] + cnt[r]
    return blo, bhi, cnt


def ploc(lo, hi, radius):
    """PLOC over Morton-sorted boxes: same node numbering as radix_tree (leaf j = n-1+j, internal ids handed out downwards so that
    the root is 0)"""
    n = len(lo)
    left = np.zeros(n - 1, np.int64); right = np.zeros(n - 1, np.int64)
    blo = np.zeros((2 * n - 1, 3), np.float32); bhi = np.zeros((2 * n - 1, 3), np.float32); cnt = np.zeros(2 * n - 1, np.int64)
    blo[n - 1:] = lo; bhi[n - 1:] = hi; cnt[n - 1:] = 1
    cid = np.arange(n - 1, 2 * n - 1)
    clo, chi = lo.copy(), hi.copy()
    next_id = n - 2
    it = 0
    forced = False
    while len(cid) > 1:
        m = len(cid)
        idx = np.arange(m)
        if forced:
            nn = idx ^ 1
            nn[nn >= m] = -1
        else:
            best = np.full(m, np.inf, np.float32); nn = np.full(m, -1, np.int64)
            for d in list(range(min(radius, m - 1), 0, -1)):     # j = i - d, ascending j
                a = area(np.minimum(clo[d:], clo[:-d]), np.maximum(chi[d:], chi[:-d]))
                upd = a < best[d:]
                best[d:][upd] = a[upd]; nn[d:][upd] = idx[:-d][upd]
            for d in range(1, min(radius, m - 1) + 1):           # j = i + d
                a = area(np.minimum(clo[d:], clo[:-d]), np.maximum(chi[d:], chi[:-d]))
                upd = a < best[:-d]
                best[:-d][upd] = a[upd]; nn[:-d][upd] = idx[d:][upd]
        ok = nn >= 0
        mutual = np.zeros(m, bool)
        mutual[ok] = nn[nn[ok]] == idx[ok]
        lead = mutual & (idx < nn)
        absorbed = mutual & (idx > nn)
        k = int(lead.sum())
        li = idx[lead]; rj = nn[lead]
        ids = next_id - np.arange(k)
        next_id -= k
        left[ids] = cid[li]; right[ids] = cid[rj]
        nlo = np.minimum(clo[li], clo[rj]); nhi = np.maximum(chi[li], chi[rj])
        blo[ids] = nlo; bhi[ids] = nhi; cnt[ids] = cnt[cid[li]] + cnt[cid[rj]]
        cid = cid.copy(); cid[li] = ids
        clo[li] = nlo; chi[li] = nhi
        keep = ~absorbed
        cid, clo, chi = cid[keep], clo[keep], chi[keep]
        forced = (not forced) and k * FORCE_DIV < m
        it += 1
    assert next_id == -1
    return left, right, blo, bhi, cnt, it


def collapse_cost(left, right, blo, bhi, cnt, leaf_max):
    n = (len(cnt) + 1) // 2
    ar = area(blo.astype(np.float64), bhi.astype(np.float64)).tolist()
    L = left.tolist(); R = right.tolist(); C = cnt.tolist()
    root = ar[0]
    internal = leaves = 0.0
    n_int = n_leaf = 0
    depth = 0
    level = [0]
    while level:
        nxt = []
        for b in level:
            if C[b] <= leaf_max:
                leaves += ar[b] * C[b]; internal += ar[b]; n_leaf += 1   # (a leaf is a node record too)
                continue
            internal += ar[b]; n_int += 1
            ch = [L[b], R[b]]
            for _ in range(2):
                pick = -1; best = -1.0
                for k, c in enumerate(ch):
                    if C[c] > leaf_max and ar[c] > best: best = ar[c]; pick = k
                if pick < 0: break
                c = ch[pick]
                ch[pick] = L[c]; ch.append(R[c])
            nxt.extend(ch)
        level = nxt
        depth += 1
    return {"bytes_per_ray_sah": round((52 * internal + 36 * leaves) / root, 2), "internal": n_int, "leaves": n_leaf, "levels": depth}


def sah_tree_cost(scene_blas_nodes, n_nodes):
    """the same figure for a tree in the node format (13 dwords): boxes decoded from the parents' quantised planes"""
    nd = np.asarray(scene_blas_nodes).view(np.uint32).reshape(-1, 13)[:n_nodes]
    raw = nd.view(np.uint8).reshape(len(nd), 52)
    org = nd[:, 0:3].view(np.float32).astype(np.float64)
    ex = raw[:, 12:15].view(np.int8).astype(np.float64)
    first = nd[:, 4]; count = nd[:, 5]
    ch = raw[:, 24:52].reshape(len(nd), 4, 7)
    internal = leaves = 0.0
    # root area: union of its children
    def child_boxes(i):
        s = np.exp2(ex[i])
        lo = org[i] + ch[i, :, 1:4] * s; hi = org[i] + ch[i, :, 4:7] * s
        return lo, hi, ch[i, :, 0] != 0
    lo, hi, on = child_boxes(0)
    root = float(area(lo[on].min(0), hi[on].max(0)))
    st = [(0, root)]
    n_int = n_leaf = 0
    while st:
        i, a = st.pop()
        if count[i] != 0:
            leaves += a * int(count[i]); internal += a; n_leaf += 1
            continue
        internal += a; n_int += 1
        lo, hi, on = child_boxes(i)
        ca = area(lo, hi)
        for k in range(4):
            if on[k]: st.append((int(first[i]) + int(on[:k].sum()), float(ca[k])))
    return {"bytes_per_ray_sah": round((52 * internal + 36 * leaves) / root, 2), "internal": n_int, "leaves": n_leaf}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=7)
    ap.add_argument("--radius", type=int, nargs="*", default=[8, 16])
    ap.add_argument("--leaf-max", type=int, default=2)
    a = ap.parse_args()
    sc = vrt.scene.procedural("atrium", a.level, 0, 3)
    tri = sc["tri"].view(np.float32).reshape(-1, 3, 3).copy()
    lo, hi = tri.min(1), tri.max(1)
    order, key = morton_order(lo, hi)
    lo, hi, key = lo[order], hi[order], key[order]
    print("triangles", len(tri))
    t0 = time.time()
    l, r = radix_tree(key)
    blo, bhi, cnt = fit(l, r, lo, hi)
    print("radix tree   ", collapse_cost(l, r, blo, bhi, cnt, a.leaf_max), "%.1f s" % (time.time() - t0))
    for rad in a.radius:
        t0 = time.time()
        l, r, blo, bhi, cnt, it = ploc(lo, hi, rad)
        print("PLOC r=%-3d   " % rad, collapse_cost(l, r, blo, bhi, cnt, a.leaf_max), "iterations", it, "%.1f s" % (time.time() - t0))
    print("CPU SAH tree ", sah_tree_cost(sc["bvh"], sc.n_bvh_nodes))


if __name__ == "__main__":
    main()


def optimal_collapse_cost(left, right, blo, bhi, cnt, leaf_max, node_bytes=52.0, tri_bytes=36.0):
    """SAH-optimal 4-wide collapse of a binary tree (the dynamic programme of Ylitie et al. 2017, for width 4): F[n][j] = least cost
    of covering the subtree of binary node n with at most j child slots of a wide node; a subtree of <= leaf_max triangles may
    become a leaf (a node record + its triangles)."""
    n = (len(cnt) + 1) // 2
    ar = area(blo.astype(np.float64), bhi.astype(np.float64)).tolist()
    L = left.tolist(); R = right.tolist(); C = cnt.tolist()
    INF = float("inf")
    F = [None] * (2 * n - 1)
    order = []
    st = [0]
    while st:
        i = st.pop()
        order.append(i)
        if i < n - 1:
            st.append(L[i]); st.append(R[i])
    for i in reversed(order):
        if i >= n - 1:
            c1 = ar[i] * (node_bytes + tri_bytes)
            F[i] = (INF, c1, c1, c1, c1)
            continue
        fl, fr = F[L[i]], F[R[i]]
        # forest of the two children in k slots
        G = [INF] * 5
        for k in range(2, 5):
            G[k] = min(fl[a] + fr[k - a] for a in range(1, k))
        c1 = ar[i] * node_bytes + G[4]
        if C[i] <= leaf_max:
            c1 = min(c1, ar[i] * (node_bytes + tri_bytes * C[i]))
        f = [INF, c1, 0, 0, 0]
        for j in range(2, 5):
            f[j] = min(f[j - 1], G[j])
        F[i] = tuple(f)
    return {"bytes_per_ray_sah_optimal_collapse": round(F[0][1] / ar[0], 2)}
