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
