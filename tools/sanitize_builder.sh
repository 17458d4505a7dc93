#!/bin/bash
# CPU sanitizer builds of the scene builder (threads: TSAN; memory / UB: ASAN + UBSAN) over scenes that take every parallel piece:
# 262,144 triangles (nodes shared by the threads, rounds up to 2 M), 2.1 M (the schedule above 2 M), the twin's BVH2.  No GPU.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=${TMPDIR:-/tmp}/vxs_san; mkdir -p "$T"
cat > "$T/run.py" <<PY
import ctypes as C
L = C.CDLL("$T/libvxrt_scene.so")
L.vxs_scene_create_procedural.restype = C.c_void_p
L.vxs_scene_create_procedural.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32]
for name, a, b, s in ((b"atrium", 7, 0, 3), (b"hairball", 4200, 250, 7)):
    print(name, bool(L.vxs_scene_create_procedural(name, a, b, s)), flush=True)
L.vxs_rc_scene_create_procedural.restype = C.c_void_p
L.vxs_rc_scene_create_procedural.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
print("twin", bool(L.vxs_rc_scene_create_procedural(b"atrium", 7, 0, 3, 1, None)))
PY
for mode in thread "address,undefined"; do
  g++ -O1 -g -fsanitize=$mode -ffp-contract=off -fPIC -std=c++17 -pthread -shared -o "$T/libvxrt_scene.so" "$ROOT/vortex-raytracing_amd/csrc/scene_builder.cpp" -lz
  if [ "$mode" = thread ]; then pre=$(g++ -print-file-name=libtsan.so); else pre="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)"; fi
  echo "== -fsanitize=$mode"
  LD_PRELOAD="$pre" ASAN_OPTIONS=detect_leaks=0 VXS_THREADS=6 python "$T/run.py"
done
