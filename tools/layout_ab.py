#!/usr/bin/env python3
"""Does the ORDER of the node records in memory matter to the frame?  The GPU builder emits level by level (breadth first), the CPU builder
depth first (children contiguous, then each child's subtree).  Both trees of the 1,048,576-triangle atrium in both orders, same frame.
Permuting the records (and renumbering the child bases) changes no ray's result: the pixel sums must agree pairwise."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
vrt = importlib.import_module("vortex-raytracing_amd")
import tree_quality as tq

NODE = 52


def parse(bvh):
    n = bvh.size // NODE
    b = bvh.reshape(n, NODE)
    w = b[:, :24].copy().view(np.uint32).reshape(n, 6)
    first, leafcnt = w[:, 4].astype(np.int64), w[:, 5]
    nc = (b[:, 24] != 0).astype(np.int64) + (b[:, 31] != 0) + (b[:, 38] != 0) + (b[:, 45] != 0)
    nc[leafcnt != 0] = 0
    return n, b, first, nc


def reorder(scene, how):
    """how = 'dfs' | 'bfs'; the BLAS is the scene's only one (its nodes start at record 0)"""
    bvh = np.asarray(scene["bvh"]).view(np.uint8).reshape(-1)
    n, b, first, nc = parse(bvh)
    new_of = np.full(n, -1, np.int64)
    new_first = np.zeros(n, np.int64)
    new_of[0] = 0
    nxt = 1
    if how == "bfs":
        frontier = np.array([0], np.int64)
        while len(frontier):
            inner = frontier[nc[frontier] > 0]
            cnt = nc[inner]
            base = nxt + np.concatenate([[0], np.cumsum(cnt)[:-1]]) if len(inner) else np.zeros(0, np.int64)
            new_first[inner] = base
            kids, news = [], []
            for k in range(4):
                m = cnt > k
                kids.append(first[inner[m]] + k); news.append(base[m] + k)
            kids, news = (np.concatenate(kids), np.concatenate(news)) if len(inner) else (np.zeros(0, np.int64), np.zeros(0, np.int64))
            o = np.argsort(news, kind="stable")
            kids, news = kids[o], news[o]
            new_of[kids] = news
            nxt += int(cnt.sum())
            frontier = kids
    else:
        st = [0]
        fl, ncl = first.tolist(), nc.tolist()
        no, nf = new_of.tolist(), new_first.tolist()
        while st:
            o = st.pop()
            c = ncl[o]
            if c:
                nf[o] = nxt
                f = fl[o]
                for k in range(c):
                    no[f + k] = nxt + k
                nxt += c
                for k in range(c - 1, -1, -1):
                    st.append(f + k)
        new_of, new_first = np.array(no, np.int64), np.array(nf, np.int64)
    used = new_of >= 0                      # (a GPU-built buffer is its capacity long: the records behind the tree stay behind)
    assert nxt == int(used.sum())
    out = np.zeros((nxt, NODE), np.uint8)
    out[new_of[used]] = b[used]
    w4 = out[:, 16:20].copy().view(np.uint32).reshape(nxt)
    inner = used & (nc > 0)
    w4[new_of[inner]] = new_first[inner].astype(np.uint32)
    out[:, 16:20] = w4.view(np.uint8).reshape(nxt, 4)
    d = {k: np.asarray(scene[k]) for k in ("tri", "triEx", "triIdx", "tlas", "blas", "bvh", "mat", "tex")}
    d["bvh"] = out.reshape(-1)
    return vrt.scene.Scene(d, name="%s-%s" % (getattr(scene, "name", "scene"), how))


def main():
    sc = vrt.scene.procedural("atrium", 8, 0, 3)
    tri = sc["tri"].view(np.float32).reshape(-1, 9)
    ex = sc["triEx"].reshape(-1, 64)
    ds = vrt.tracer.DeviceScene.build_on_gpu(tri, ex, sc["mat"], sc["tex"], "cuda:0", leaf_max=2)
    gpu = ds.to_host()
    ds.close()
    LIGHT = (300.0, 480.0, 60.0)
    for name, s in (("cpu tree, as built (depth first)", sc), ("cpu tree, breadth first", reorder(sc, "bfs")), ("cpu tree, depth first again", reorder(sc, "dfs")),
                    ("gpu tree, as built (breadth first)", gpu), ("gpu tree, depth first", reorder(gpu, "dfs")), ("gpu tree, breadth first again", reorder(gpu, "bfs"))):
        for rep in range(2):
            r = tq.gpu_rate(s, 1920, 1080, LIGHT, frames=60)
            print(json.dumps({"tree": name, "mrays_s_serial": r["mrays_s_serial"], "ms": r["ms_per_frame"], "node_fetches_per_ray": r["frame_node_fetches_per_ray"], "pixels_crc": r["pixels_crc"]}), flush=True)


if __name__ == "__main__":
    main()
