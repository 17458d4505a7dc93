// libvxrt_scene.so -- host-side producer of the buffers the ray-tracing hot path reads.
//
// Role of reference tests/regression/raytracing/{scene,bvh}.cpp: per mesh a 4-wide binned-SAH BVH
// (8 bins, greedy widening by best SAH gain, triangles reordered in place so leaves index them
// directly), power-of-two 8-bit child-box quantisation, an instance record per mesh and a TLAS.
// It is a from-scratch builder, not a restatement: the reference widens clusters using split
// children whose bounds are never initialised (bvh.cpp:79-86 -> findBestSplitPlane reads
// centroidMin/Max of L and R), so its tree shape is not reproducible; this builder computes those
// bounds and additionally makes quantisation provably conservative.  Output FORMATS are the
// reference's byte for byte (rt_types.h) - that is what the kernels and the oracle consume.
//
// Also holds the deterministic procedural scenes of SURVEY.md s8d (none of Sponza/bunny/hairball
// exist offline) and a minimal OBJ/MTL reader for the reference's own small assets.
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <utility>
#include <vector>
#include <zlib.h>
#include "rt_types.h"

namespace {

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 vmin(V3 a, V3 b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
inline V3 vmax(V3 a, V3 b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }
inline float comp(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 normalize(V3 v) { float l = std::sqrt(dot(v, v)); return l > 0 ? v * (1.0f / l) : V3{0, 1, 0}; }

constexpr float kBig = RT_LARGE_FLOAT;
constexpr int kMaxBins = 256;
static int kBins = 128;     // reference: 8 (bvh.cpp:8).  Under the reinsertion passes the outcome scatters with the bin count by +-1.5 % of the frame rate without
                            // a trend -- 12: 10.44, 16: 10.27, 24: 10.32, 32: 10.44, 40: 10.50, 48: 10.33, 64: 10.47, 96: 10.42, 128: 10.52-10.56, 256: 10.31-10.37
                            // Grays/s (profiles/r03_v_bins_ab.txt; 100-step runs) -- the optimiser ends in a different local optimum each time; 128 is
                            // the best measured.  (The greedy builder: 16 bins 6 % fewer node visits than 8, 24 another 1.1 %.)  VXS_BINS overrides
static float kLeafK = 1.0f;  // keep <= kLeafMax triangles in one leaf when the best split saves less than kLeafK node-areas
                             // (reference: always split, i.e. 0; VXS_LEAF_K overrides): -9 % node visits, +15 % triangle tests
static int kLeafMax = 4;
static int kThreads = 8;     // worker threads of the BLAS builder: min(16, hardware threads); VXS_THREADS overrides; the tree and its layout do not depend on it
static int kWiden = 0;      // 0: widen the cluster with the largest SAH gain (reference), 1: the one with the largest area
static int kCollapse = 1;   // 1: build the binary SAH tree to the bottom, optimise it by reinsertion (kOptimize passes), collapse it to 4-wide by the SAH
                            // dynamic programme; 0: widen greedily while building (the builder up to profiles/r03_s).  VXS_COLLAPSE overrides.
                            // Measured on the 1M-triangle frame (profiles/r03_t_cpu_collapse_ab.txt, r03_v_reinsertion_*.txt): the collapse alone
                            // changes nothing (9 % fewer nodes, the same bytes per ray and frame rate); with the reinsertion passes +7.8 %
static int kOptimize = 2;   // passes of insertion-based optimisation of the binary tree before the collapse (1 pass +6.9 %, 2 +7.8 %, 3 the same). VXS_OPTIMIZE
static int kVerbose = 0;      // VXS_VERBOSE
static int kChildOrder = 0;   // slots of a wide node: 0 as the binary tree hands them out, 1 largest surface area first, 2 smallest first.  VXS_CHILD_ORDER
static int kOptimizeLocal = -1;  // 0: every node over the whole tree, serially (13 s per million triangles; the quality the rounds are measured against);
                                 // otherwise (default): rounds, coarse to fine, in place and in parallel (build_collapsed).  VXS_OPTIMIZE_LOCAL
static double kOptimizeFraction = 1.0;   // share of the nodes, largest first, a pass takes. VXS_OPTIMIZE_FRACTION
constexpr float kNodeCost = 52.0f;   // bytes a node visit fetches: the unit of the collapse's cost
static float kTriCost = 36.0f;       // ... and a triangle test.  VXS_TRI_COST (measured flat between 18 and 72: profiles/r03_v_tri_cost_ab.txt)

static std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();
static void glap(const char* what) {   // VXS_VERBOSE: seconds since the scene's creation began
  if (kVerbose) std::fprintf(stderr, "[scene_builder] %8.3f s  | %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count(), what);
}

// jobs 0..n-1 handed to up to kThreads threads through a counter (the caller is one of them).  What a job computes must not depend on which
// thread runs it or when: every use below writes disjoint data per job, or reduces per-job results in job order afterwards.
template <class F>
static void par_jobs(size_t n, F&& f) {
  if (n == 0) return;
  const size_t nt = std::min<size_t>((size_t)std::max(1, kThreads), n);
  if (nt <= 1) { for (size_t j = 0; j < n; ++j) f(j); return; }
  std::atomic<size_t> next{0};
  auto work = [&]() { for (;;) { const size_t j = next.fetch_add(1); if (j >= n) break; f(j); } };
  std::vector<std::thread> pool;
  for (size_t i = 1; i < nt; ++i) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
}
constexpr size_t kChunk = 65536;   // elements per job of a chunked pass (fixed: nothing depends on the thread count)

struct Box {
  V3 lo{kBig, kBig, kBig}, hi{-kBig, -kBig, -kBig};
  void grow(V3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }
  void grow(const Box& b) { if (b.lo.x != kBig) { grow(b.lo); grow(b.hi); } }
  float half_area() const { V3 e = hi - lo; return e.x * e.y + e.y * e.z + e.z * e.x; }
};

struct WideNode {           // float-box node before quantisation (role of bvh_node_t, common.h:70-83)
  Box box, cbox;
  uint32_t leftFirst = 0, triCount = 0, childCount = 0;
};

// node of the binary SAH tree the 4-wide tree is collapsed from (kCollapse): f[j-1] = least cost -- expected bytes fetched per random
// ray, up to the root's area -- of covering the subtree with at most j child slots of a wide node; plan = the choices behind it
// (bits [1:0] [3:2] [5:4]: slots given to the left child when the two children share 2 / 3 / 4; [6] [7] [8]: stays one child when
// offered 2 / 3 / 4; [9]: leaf).  The dynamic programme of Ylitie, Karras & Laine 2017 ("Efficient incoherent ray traversal on GPUs
// through compressed wide BVHs", s3.1) for width 4; csrc/bvh_builder.hip runs the same one on the GPU.
// one background thread at a time for work nobody waits for (freeing a big array); joined before the next one starts and when the library
// is unloaded or the process exits, so no thread of this library outlives its code
struct Disposer {
  std::thread th;
  template <class F> void run(F&& f) { if (th.joinable()) th.join(); th = std::thread(std::forward<F>(f)); }
  ~Disposer() { if (th.joinable()) th.join(); }
};
static Disposer g_disposer;

struct BinNode {
  Box box;
  uint32_t first = 0, count = 0;
  uint32_t left = 0, right = 0;      // 0 = none: a leaf (node 0 is the root, never a child)
  float f[4] = {0, 0, 0, 0};
  uint32_t plan = 0;
};
constexpr uint32_t kPlanLeaf = 1u << 9;
inline uint32_t plan_a(uint32_t plan, uint32_t j) { return (plan >> (2u * (j - 2u))) & 3u; }
inline bool plan_self(uint32_t plan, uint32_t j) { return (plan >> (4u + j)) & 1u; }

struct Mesh {
  std::vector<rt_tri_t> tri;
  std::vector<rt_triex_t> triEx;
  std::vector<rt_material_t> mats;
  std::vector<std::vector<uint32_t>> textures;  // 0x00RRGGBB texels
  std::vector<std::pair<uint32_t, uint32_t>> tex_dims;
  float transform[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};

inline V3 tv(const float* p) { return {p[0], p[1], p[2]}; }

// ---------------------------------------------------------------------------------------------
// BLAS builder
// ---------------------------------------------------------------------------------------------
class BlasBuilder {
public:
  BlasBuilder(rt_tri_t* tri, rt_triex_t* triEx, uint32_t n, uint32_t width = RT_BVH_WIDTH) : tri_(tri), triEx_(triEx), n_(n), width_(width) {
    cent_.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
      V3 s = tv(tri[i].v0) + tv(tri[i].v1) + tv(tri[i].v2);
      cent_[i] = {s.x / 3, s.y / 3, s.z / 3};     // scene.cpp:87
    }
    if (kCollapse == 1 && (width_ == 4 || width_ == 2)) { build_collapsed(); return; }
    nodes_.reserve(2 * (size_t)n + 1);
    nodes_.emplace_back();
    nodes_[0].leftFirst = 0;
    nodes_[0].triCount = n;
    // Top of the tree serially; subtrees below `defer_below` triangles are built by worker threads into
    // private node arrays (their triangle ranges are disjoint, so the in-place reorder needs no locks)
    // and appended in the order the serial pass met them: the layout does not depend on the thread count.
    const uint32_t defer_below = std::max<uint32_t>(4096u, n / 256u);
    std::vector<std::pair<uint32_t, uint32_t>> deferred;   // (node index, depth)
    subdivide(nodes_, 0, 0, max_depth_, defer_below, &deferred);
    if (!deferred.empty()) {
      std::vector<std::vector<WideNode>> sub(deferred.size());
      std::vector<uint32_t> subdepth(deferred.size(), 0);
      std::vector<size_t> order(deferred.size());
      for (size_t i = 0; i < order.size(); ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
        return nodes_[deferred[x].first].triCount > nodes_[deferred[y].first].triCount; });   // big subtrees first
      std::atomic<size_t> next{0};
      auto work = [&]() {
        for (;;) {
          const size_t k = next.fetch_add(1);
          if (k >= order.size()) break;
          const size_t t = order[k];
          sub[t].reserve(2 * (size_t)nodes_[deferred[t].first].triCount + 1);
          sub[t].push_back(nodes_[deferred[t].first]);
          subdivide(sub[t], 0, deferred[t].second, subdepth[t], 0u, nullptr);
        }
      };
      const int nthreads = (int)std::min<size_t>((size_t)kThreads, deferred.size());
      std::vector<std::thread> pool;
      for (int i = 1; i < nthreads; ++i) pool.emplace_back(work);
      work();
      for (auto& th : pool) th.join();
      for (size_t t = 0; t < deferred.size(); ++t) {
        const uint32_t root = deferred[t].first;
        const uint32_t base = (uint32_t)nodes_.size();          // local index i >= 1 -> base + i - 1
        std::vector<WideNode>& L = sub[t];
        for (size_t i = 0; i < L.size(); ++i)
          if (L[i].triCount == 0 && L[i].childCount != 0) L[i].leftFirst += base - 1;
        nodes_[root] = L[0];
        nodes_.insert(nodes_.end(), L.begin() + 1, L.end());
        max_depth_ = std::max(max_depth_, subdepth[t]);
        std::vector<WideNode>().swap(L);
      }
    }
  }
  std::vector<WideNode> nodes_;
  uint32_t max_depth_ = 0;

private:
  struct Split { int axis = -1; int pos = 0; float cost = INFINITY; };

  // ---- binary SAH tree to the bottom + SAH-optimal collapse to 4-wide ----
  void build_collapsed() {
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {   // VXS_VERBOSE: seconds since the build of this mesh began
      if (kVerbose) std::fprintf(stderr, "[scene_builder] %8.3f s  %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), what);
    };
    std::vector<BinNode> bn;
    bn.reserve(2 * (size_t)n_ + 1);
    bn.emplace_back();
    bn[0].first = 0; bn[0].count = n_;
    // the top serially; subtrees below `defer_below` triangles by worker threads into private arrays (disjoint triangle ranges: the
    // in-place partition needs no locks), appended in the order the serial pass met them: nothing depends on the thread count
    const uint32_t defer_below = std::max<uint32_t>(4096u, n_ / 256u);
    // 0: every node over the whole tree, serially (13 s per million triangles); otherwise (default) rounds of reinsertion at several scales IN
    // PLACE and in parallel -- subtrees of n/8, n/64, n/256 triangles, each taken by one thread -- between an opening round and a closing pass
    // over the whole tree: what a round of subtrees cannot do (move a node across the border of its subtree) a coarser round can
    const int mode = kOptimizeLocal == 0 ? 0 : 2;
    std::vector<uint32_t> deferred;
    {
      // The top in two levels.  Nodes of kBigNode triangles and more one after the other, each pass over a node's triangles shared by the
      // threads (chunks of kChunk triangles; bins and boxes reduce exactly, the partition is the stable one, which has one outcome); the
      // nodes below them ("mid" roots) one thread each into private arrays, down to `defer_below`; spliced in the order the walk met them.
      std::vector<uint32_t> mid;
      par_nodes_ = true;
      build_binary(bn, 0, std::max(defer_below, kBigNode), &mid);
      par_nodes_ = false;
      std::vector<rt_tri_t>().swap(tri2_); std::vector<rt_triex_t>().swap(ex2_); std::vector<V3>().swap(cent2_);
      lap("top of the binary tree: nodes of 131,072 triangles and more");
      struct MidOut { std::vector<BinNode> nodes; std::vector<uint32_t> def; };
      std::vector<MidOut> mo(mid.size());
      par_jobs(mid.size(), [&](size_t t) {
        const BinNode& r = bn[mid[t]];
        if (r.count < defer_below) return;      // (a subtree of the last level as it is)
        mo[t].nodes.reserve(2 * (size_t)(r.count / defer_below + 1) * 2);
        mo[t].nodes.push_back(r);
        build_binary(mo[t].nodes, 0, defer_below, &mo[t].def);
      });
      for (size_t t = 0; t < mid.size(); ++t) {
        std::vector<BinNode>& L = mo[t].nodes;
        if (L.empty()) { deferred.push_back(mid[t]); continue; }
        const uint32_t base = (uint32_t)bn.size();          // local index i >= 1 -> base + i - 1
        for (BinNode& x : L) if (x.left) { x.left += base - 1; x.right += base - 1; }
        bn[mid[t]] = L[0];
        bn.insert(bn.end(), L.begin() + 1, L.end());
        for (uint32_t d : mo[t].def) deferred.push_back(base + d - 1);      // (d >= 1: a root is never deferred)
      }
    }
    const uint32_t n_top = (uint32_t)bn.size();
    lap("top of the binary tree");
    if (!deferred.empty()) {
      BinNode* arena = (BinNode*)std::malloc(2 * (size_t)n_ * sizeof(BinNode));
      if (!arena) throw std::bad_alloc();
      std::vector<NodeSlice> sub(deferred.size());
      std::vector<size_t> order(deferred.size());
      for (size_t i = 0; i < order.size(); ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return bn[deferred[x]].count > bn[deferred[y]].count; });   // big subtrees first
      par_jobs(order.size(), [&](size_t k) {
        const size_t t = order[k];
        sub[t].p = arena + 2 * (size_t)(bn[deferred[t]].first - bn[0].first);
        sub[t].emplace_back();
        sub[t][0] = bn[deferred[t]];
        build_binary(sub[t], 0, 0u, nullptr);
      });
      lap("subtrees built by the workers");
      // appended in the order the serial pass met them (local index i >= 1 -> base + i - 1), the copies shared by the threads
      std::vector<size_t> base(deferred.size() + 1, bn.size());
      for (size_t t = 0; t < deferred.size(); ++t) base[t + 1] = base[t] + sub[t].size() - 1;
      bn.resize(base.back());
      par_jobs(deferred.size(), [&](size_t t) {
        NodeSlice& L = sub[t];
        const uint32_t off = (uint32_t)base[t] - 1u;
        for (uint32_t i = 0; i < L.n; ++i) if (L.p[i].left) { L.p[i].left += off; L.p[i].right += off; }
        bn[deferred[t]] = L.p[0];
        std::copy(L.p + 1, L.p + L.n, bn.begin() + (ptrdiff_t)base[t]);
      });
      std::free(arena);
    }
    lap("binary SAH tree built");
    if (kOptimize > 0 && mode == 2 && !deferred.empty()) {
      std::vector<uint32_t> parent(bn.size(), 0u);
      par_jobs((bn.size() + kChunk - 1) / kChunk, [&](size_t c) {      // (every node has one parent: the writes are disjoint)
        const size_t e = std::min(bn.size(), (c + 1) * kChunk);
        for (size_t i = c * kChunk; i < e; ++i) if (bn[i].left) { parent[bn[i].left] = (uint32_t)i; parent[bn[i].right] = (uint32_t)i; }
      });
      // rounds, coarse to fine (as the serial pass takes its nodes largest first): subtrees of at most n / divisor triangles, `passes` passes
      // over the largest `fraction` of their nodes.  The first round is the one serial piece: the whole tree, its few thousand largest
      // nodes -- the moves that cross every later border; with it the frame traces as fast as on the serially optimised tree
      // (profiles/r04_r_builder_ab*.txt: 0.5 % of the nodes do as well as 5 %), without it 1.5 % slower.
      // VXS_ROUNDS = "divisor:passes:fraction,..." overrides.
      struct Round { uint32_t divisor; int passes; double fraction; size_t max_nodes; };
      // up to 2 M triangles: the largest hundredth of the nodes over the whole tree (serial), then n/8, n/64, n/256; above (a 10 M-triangle scene has
      // 15 M nodes: every full round costs seconds), the same first round and kOptimize passes inside subtrees of ~4,096 triangles
      const uint32_t big = std::max<uint32_t>(1u, n_ / 1048576u);
      std::vector<Round> rounds = {{1u, 1, 0.01, 0}, {8u, 1, 1.0, 0}, {64u, 1, 1.0, 0}, {256u, 1, 1.0, 0}};
      if (n_ > 2000000u) rounds = {{1u, 1, 1.0, 8192}, {256u * big, kOptimize, 1.0, 0}};
      if (const char* e = std::getenv("VXS_ROUNDS")) {
        rounds.clear();
        for (const char* q = e; *q;) {
          Round r{1u, 1, 1.0, 0};
          int used = 0;
          if (std::sscanf(q, "%u:%d:%lf%n", &r.divisor, &r.passes, &r.fraction, &used) < 3 || r.divisor == 0u) break;
          rounds.push_back(r);
          q += used;
          if (*q == ',') ++q;
        }
      }
      for (const Round& rd : rounds) {
        const uint32_t thresh = std::max<uint32_t>(n_ / rd.divisor, 2u);
        // the subtrees of this round: the highest nodes of at most `thresh` triangles, largest first
        std::vector<uint32_t> roots, st{0u};
        while (!st.empty()) {
          const uint32_t i = st.back(); st.pop_back();
          if (!bn[i].left) continue;
          if (bn[i].count <= thresh) { roots.push_back(i); continue; }
          st.push_back(bn[i].right); st.push_back(bn[i].left);
        }
        std::stable_sort(roots.begin(), roots.end(), [&](uint32_t x, uint32_t y) { return bn[x].count > bn[y].count; });
        std::atomic<size_t> next{0};
        auto work = [&]() {
          for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= roots.size()) break;
            reinsertion_passes(bn, parent, roots[k], 0xffffffffu, rd.passes, rd.fraction, rd.max_nodes);
          }
        };
        const int nthreads = (int)std::min<size_t>((size_t)kThreads, roots.size());
        std::vector<std::thread> pool;
        for (int i = 1; i < nthreads; ++i) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
        if (kVerbose) { char msg[96]; std::snprintf(msg, sizeof msg, "round n/%u (%zu subtrees, %d passes over %.2f of their nodes)", rd.divisor, roots.size(), rd.passes, rd.fraction); lap(msg); }
        recount(bn);     // (the next, finer round picks its subtrees by triangle count: inside this round's subtrees the counts have moved with the nodes)
      }
      // ... and, after the fine rounds, the nodes of the serially built top once more over the whole tree: measured, this closing pass is worth
      // as much as the opening round (75.1 M -> 73.6 M node steps in the headline frame; profiles/r04_r_builder_ab4.txt)
      reinsertion_passes(bn, parent, 0u, n_top, kOptimize, 1.0);
      lap("pass over the top");
      leaf_order(bn);
      lap("triangles in leaf order");
    } else
    if (kOptimize > 0) { optimize_by_reinsertion(bn, 0xffffffffu); lap("reinsertion"); }
    // the dynamic programme, children before parents: the subtrees below the cut one job each, then the nodes above it
    auto programme = [&](BinNode& x) {
      const float ar = x.box.half_area();
      if (!x.left) {
        const float c = ar * ((width_ == 2 ? 32.0f : kNodeCost) + kTriCost * (float)x.count);
        x.f[0] = x.f[1] = x.f[2] = x.f[3] = c;
        x.plan = kPlanLeaf;
        return;
      }
      const float* A = bn[x.left].f; const float* B = bn[x.right].f;
      if (width_ == 2) {   // the raycast twin's BVH2: no collapse, only the leaf-or-node choice (32-byte nodes)
        const float c_node = ar * 32.0f + A[0] + B[0];
        const float c_leaf = (int)x.count <= kLeafMax ? ar * (32.0f + kTriCost * (float)x.count) : INFINITY;
        const bool leaf = c_leaf <= c_node;
        x.f[0] = x.f[1] = x.f[2] = x.f[3] = leaf ? c_leaf : c_node;
        x.plan = 1u | (7u << 6) | (leaf ? kPlanLeaf : 0u);
        return;
      }
      const float g2 = A[0] + B[0];
      float g3 = A[0] + B[1]; uint32_t a3 = 1;
      if (A[1] + B[0] < g3) { g3 = A[1] + B[0]; a3 = 2; }
      float g4 = A[0] + B[2]; uint32_t a4 = 1;
      if (A[1] + B[1] < g4) { g4 = A[1] + B[1]; a4 = 2; }
      if (A[2] + B[0] < g4) { g4 = A[2] + B[0]; a4 = 3; }
      const float c_node = ar * kNodeCost + g4;
      const float c_leaf = (int)x.count <= kLeafMax ? ar * (kNodeCost + kTriCost * (float)x.count) : INFINITY;
      const bool leaf = c_leaf <= c_node;
      x.f[0] = leaf ? c_leaf : c_node;
      const bool s2 = x.f[0] <= g2, s3 = x.f[0] <= g3, s4 = x.f[0] <= g4;
      x.f[1] = s2 ? x.f[0] : g2; x.f[2] = s3 ? x.f[0] : g3; x.f[3] = s4 ? x.f[0] : g4;
      x.plan = 1u | (a3 << 2) | (a4 << 4) | ((uint32_t)s2 << 6) | ((uint32_t)s3 << 7) | ((uint32_t)s4 << 8) | (leaf ? kPlanLeaf : 0u);
    };
    {
      std::vector<uint32_t> top, roots;
      cut_tree(bn, top, roots);
      par_jobs(roots.size(), [&](size_t r) {
        std::vector<uint32_t> st{roots[r]}, post;
        while (!st.empty()) { const uint32_t i = st.back(); st.pop_back(); post.push_back(i); if (bn[i].left) { st.push_back(bn[i].left); st.push_back(bn[i].right); } }
        for (size_t pi = post.size(); pi-- > 0;) programme(bn[post[pi]]);
      });
      for (size_t pi = top.size(); pi-- > 0;) programme(bn[top[pi]]);
    }
    lap("collapse programme");
    // emit, depth first, children contiguous and after their parent
    nodes_.reserve(bn.size());
    nodes_.emplace_back();
    struct Todo { uint32_t wide, bin, depth; };
    std::vector<Todo> st{{0u, 0u, 0u}};
    while (!st.empty()) {
      const Todo t = st.back(); st.pop_back();
      max_depth_ = std::max(max_depth_, t.depth);
      const BinNode& x = bn[t.bin];
      nodes_[t.wide].box = x.box;
      if (x.plan & kPlanLeaf) { nodes_[t.wide].leftFirst = x.first; nodes_[t.wide].triCount = x.count; nodes_[t.wide].childCount = 0; continue; }
      uint32_t c[4] = {x.left, x.right, 0, 0}, slots[4] = {width_ == 4 ? plan_a(x.plan, 4u) : 1u, 0, 0, 0}, nc = 2;
      slots[1] = width_ - slots[0];
      for (;;) {   // a child offered j > 1 slots hands them to its own children unless it stays one child
        int pick = -1;
        for (uint32_t k = 0; k < nc && pick < 0; ++k)
          if (slots[k] > 1u && bn[c[k]].left && !plan_self(bn[c[k]].plan, slots[k])) pick = (int)k;
        if (pick < 0) break;
        const BinNode& y = bn[c[pick]];
        const uint32_t j = slots[pick], ja = plan_a(y.plan, j);
        c[pick] = y.left; slots[pick] = ja;
        c[nc] = y.right; slots[nc] = j - ja;
        ++nc;
      }
      // slot order = the order in which the frame's occlusion rays (any-hit, unordered: rt_kernels.hip) visit the children; closest-hit
      // rays sort by distance, for them the slot order decides ties only.  Measurement knob (VXS_CHILD_ORDER; profiles/
      // r05_g_gpu_reinsertion.txt sections 6 and 8): 1 / 2 largest / smallest surface area first, 3 / 4 by centre along this node's
      // widest axis.  The default stays "as built" -- left = below the split plane, mostly what 3 makes exact: on this builder's tree
      // 3 measured the same as 0 at two lights and 21 % fewer occlusion-ray node fetches at the third; the GPU builder, whose
      // clustering leaves no such order, uses 3
      if (kChildOrder != 0 && width_ == 4) {
        float key[4];
        // 3 / 4: by the child's centre along this node's widest axis, ascending / descending
        const V3 ext = x.box.hi - x.box.lo;
        const int ax = ext.x >= ext.y && ext.x >= ext.z ? 0 : (ext.y >= ext.z ? 1 : 2);
        for (uint32_t k = 0; k < nc; ++k) {
          const Box& cb = bn[c[k]].box;
          const float cen = ax == 0 ? cb.lo.x + cb.hi.x : (ax == 1 ? cb.lo.y + cb.hi.y : cb.lo.z + cb.hi.z);
          key[k] = kChildOrder == 1 ? -cb.half_area() : (kChildOrder == 2 ? cb.half_area() : (kChildOrder == 3 ? cen : -cen));
        }
        for (uint32_t i = 1; i < nc; ++i)        // (insertion sort, stable: equal areas keep the order the binary tree gave them)
          for (uint32_t j = i; j > 0 && key[j] < key[j - 1]; --j) { std::swap(key[j], key[j - 1]); std::swap(c[j], c[j - 1]); }
      }
      const uint32_t first = (uint32_t)nodes_.size();
      for (uint32_t k = 0; k < nc; ++k) nodes_.emplace_back();
      nodes_[t.wide].triCount = 0; nodes_[t.wide].leftFirst = first; nodes_[t.wide].childCount = nc;
      for (uint32_t k = nc; k-- > 0;) st.push_back({first + k, c[k], t.depth + 1});
    }
    lap("wide nodes emitted");
    // (returning a gigabyte of touched pages to the system costs most of a second at 10 M triangles: not on the caller's time)
    if (bn.size() > (1u << 20)) { auto* gone = new std::vector<BinNode>(std::move(bn)); g_disposer.run([gone]() { delete gone; }); }
  }

  // Insertion-based optimisation of the binary tree (Bittner, Hapala & Havran, "Fast insertion-based optimization of bounding volume
  // hierarchies", 2013, in the subtree-reinsertion form of Meister & Bittner 2018): a node is cut out together with its parent (its
  // sibling takes the parent's place), the tree is searched best-first for the position where putting it back costs the least
  // surface area -- the area of the new common parent plus what every ancestor's box grows by -- and the freed parent node is reused
  // there.  The position it came from is among the candidates, so a step never raises the cost.  Nodes are taken largest area
  // first, kOptimize passes.  Leaves stop being ranges of the triangle array: the array is put into the final tree's leaf order
  // afterwards.
  // `limit`: only nodes with an index below it are moved (the pass over the serially built top after the subtrees were optimised by
  // their workers); the search always covers the whole tree under node 0.
  void optimize_by_reinsertion(std::vector<BinNode>& bn, uint32_t limit = 0xffffffffu) {
    const uint32_t N = (uint32_t)bn.size();
    if (N < 8) return;
    std::vector<uint32_t> parent(N, 0u);
    for (uint32_t i = 0; i < N; ++i) if (bn[i].left) { parent[bn[i].left] = i; parent[bn[i].right] = i; }
    reinsertion_passes(bn, parent, 0u, limit, kOptimize, kOptimizeFraction);
    leaf_order(bn);
  }

  // kOptimize passes of reinsertion INSIDE the subtree under `root`, in place: only nodes of that subtree move, only positions inside it
  // are searched, only its nodes' boxes and links (and their entries of `parent`) are written -- so disjoint subtrees can be optimised by
  // different threads at the same time on the shared arrays, and the result does not depend on which thread took which.  `limit`: only
  // nodes with an index below it are moved.  `fraction`: share of the movable nodes, largest area first, a pass takes (at most `max_nodes`
  // of them, 0 = no cap).
  void reinsertion_passes(std::vector<BinNode>& bn, std::vector<uint32_t>& parent, uint32_t root, uint32_t limit, int passes, double fraction, size_t max_nodes = 0) {
    auto unite = [](const Box& a, const Box& b) { Box r; r.lo = vmin(a.lo, b.lo); r.hi = vmax(a.hi, b.hi); return r; };
    auto same = [](const Box& a, const Box& b) { return a.lo.x == b.lo.x && a.lo.y == b.lo.y && a.lo.z == b.lo.z && a.hi.x == b.hi.x && a.hi.y == b.hi.y && a.hi.z == b.hi.z; };
    auto refit_up = [&](uint32_t i) {   // boxes of i and its ancestors (up to the subtree's root) from their children, until one does not change
      for (;;) {
        const Box b = unite(bn[bn[i].left].box, bn[bn[i].right].box);
        if (same(b, bn[i].box)) break;
        bn[i].box = b;
        if (i == root) break;
        i = parent[i];
      }
    };
    if (!bn[root].left) return;
    struct Cand { float ci; uint32_t node; bool operator<(const Cand& o) const { return ci > o.ci; } };   // (min-heap on the induced cost)
    std::vector<Cand> heap;
    std::vector<uint32_t> order, dfs;
    auto report = [&](int pass) {   // VXS_VERBOSE: surface-area cost of the binary tree (internal nodes' areas over the root's)
      if (kVerbose < 2 || root != 0u) return;     // (a walk over the whole tree per report: VXS_VERBOSE=2)
      double a = 0;
      dfs.assign(1, root);
      while (!dfs.empty()) { const uint32_t i = dfs.back(); dfs.pop_back(); if (bn[i].left) { a += bn[i].box.half_area(); dfs.push_back(bn[i].left); dfs.push_back(bn[i].right); } }
      std::fprintf(stderr, "[scene_builder] reinsertion pass %d: internal area / root area = %.3f\n", pass, a / bn[root].box.half_area());
    };
    report(0);
    for (int pass = 0; pass < passes; ++pass) {
      if (pass) report(pass);
      // the movable nodes of the subtree: not the root, not its two children (the root keeps its place), index below `limit`; in index
      // order first, so that the stable sort by area gives the same sequence whatever order the walk met them in
      order.clear();
      if (max_nodes && fraction >= 1.0 && limit == 0xffffffffu) {
        // the `max_nodes` largest nodes of a big subtree without visiting all of it: a child's box lies inside its parent's, so areas fall
        // along every path and the largest nodes are found from the root down, largest first (ties: index, as the sort below breaks them)
        struct A { float area; uint32_t node; };
        auto below = [](const A& x, const A& y) { return x.area < y.area || (x.area == y.area && x.node > y.node); };
        std::vector<A> hp;
        auto put = [&](uint32_t i) { hp.push_back({bn[i].box.half_area(), i}); std::push_heap(hp.begin(), hp.end(), below); };
        put(root);
        while (!hp.empty() && order.size() < max_nodes) {
          std::pop_heap(hp.begin(), hp.end(), below);
          const uint32_t i = hp.back().node; hp.pop_back();
          if (i != root && parent[i] != root) order.push_back(i);
          if (bn[i].left) { put(bn[i].left); put(bn[i].right); }
        }
      } else
      if (root == 0u) {     // the whole tree: every node is in it, no walk needed (the pass over the top takes a few thousand of 15 M nodes)
        const uint32_t hi = std::min<uint32_t>(limit, (uint32_t)bn.size());
        for (uint32_t i = 1; i < hi; ++i) if (parent[i] != root) order.push_back(i);
      } else {
        dfs.assign(1, root);
        while (!dfs.empty()) {
          const uint32_t i = dfs.back(); dfs.pop_back();
          if (i != root && parent[i] != root && i < limit) order.push_back(i);
          if (bn[i].left) { dfs.push_back(bn[i].left); dfs.push_back(bn[i].right); }
        }
      }
      // largest area first, ties by index (a total order: the same sequence whatever order the walk met the nodes in).  A pass over a small
      // share of a big subtree does not pay for sorting all of it: the share is selected first (nth_element), then sorted.
      auto larger = [&](uint32_t a, uint32_t b) { const float x = bn[a].box.half_area(), y = bn[b].box.half_area(); return x > y || (x == y && a < b); };
      size_t take = (size_t)((double)order.size() * fraction);
      if (max_nodes && take > max_nodes) take = max_nodes;
      if (take < order.size()) { std::nth_element(order.begin(), order.begin() + (ptrdiff_t)take, order.end(), larger); order.resize(take); }
      std::sort(order.begin(), order.end(), larger);
      for (size_t oi = 0; oi < take; ++oi) {
        const uint32_t X = order[oi];
        const uint32_t P = parent[X];
        if (P == root) continue;          // (moved under the root by an earlier step of this pass)
        const uint32_t G = parent[P];
        const uint32_t S = bn[P].left == X ? bn[P].right : bn[P].left;
        // cut X and P out: S takes P's place under G
        if (bn[G].left == P) bn[G].left = S; else bn[G].right = S;
        parent[S] = G;
        refit_up(G);
        // best position: least (area of the common parent + growth of every ancestor)
        const Box xb = bn[X].box;
        const float ax = xb.half_area();
        float best_cost = INFINITY; uint32_t best = S;
        heap.clear();
        heap.push_back({0.0f, bn[root].left}); std::push_heap(heap.begin(), heap.end());
        heap.push_back({0.0f, bn[root].right}); std::push_heap(heap.begin(), heap.end());
        // (X lay inside the root's box and still does: the root itself grows by nothing; the root is not a candidate, it keeps its place)
        while (!heap.empty()) {
          std::pop_heap(heap.begin(), heap.end());
          const Cand c = heap.back(); heap.pop_back();
          if (c.ci + ax >= best_cost) break;
          const BinNode& t = bn[c.node];
          const float direct = unite(t.box, xb).half_area();
          const float total = c.ci + direct;
          if (total < best_cost) { best_cost = total; best = c.node; }
          const float ci = total - t.box.half_area();
          if (t.left && ci + ax < best_cost) {
            heap.push_back({ci, t.left}); std::push_heap(heap.begin(), heap.end());
            heap.push_back({ci, t.right}); std::push_heap(heap.begin(), heap.end());
          }
        }
        // P goes where `best` was, over best and X
        const uint32_t T = best, Q = parent[T];
        if (bn[Q].left == T) bn[Q].left = P; else bn[Q].right = P;
        parent[P] = Q;
        bn[P].left = T; bn[P].right = X;
        parent[T] = P; parent[X] = P;
        bn[P].box = unite(bn[T].box, xb);
        refit_up(Q);
      }
    }
    report(passes);
  }

  // a worker's node array: a slice of one allocation shared by all subtrees (a subtree over triangles [first, first + count) has fewer than
  // 2 * count nodes and owns the slots [2 * first, 2 * (first + count)): disjoint, no allocation inside the parallel region -- hundreds of
  // megabyte-sized vectors allocated and freed by 16 threads spend more time in mmap / munmap than in the build)
  struct NodeSlice {
    BinNode* p; uint32_t n = 0;
    uint32_t size() const { return n; }
    void emplace_back() { new (p + n) BinNode(); ++n; }
    BinNode& operator[](size_t i) { return p[i]; }
    const BinNode& operator[](size_t i) const { return p[i]; }
  };

  // The tree cut at depth kCutDepth for the passes that walk all of it: `top` = the internal nodes above the cut in pre-order (parents before
  // children), `roots` = the nodes at the cut and the leaves above it, left to right.  The subtrees under the roots are disjoint: one job each.
  static constexpr uint32_t kCutDepth = 12;
  static void cut_tree(const std::vector<BinNode>& bn, std::vector<uint32_t>& top, std::vector<uint32_t>& roots) {
    top.clear(); roots.clear();
    std::vector<std::pair<uint32_t, uint32_t>> st{{0u, 0u}};
    while (!st.empty()) {
      const auto [i, d] = st.back(); st.pop_back();
      if (bn[i].left && d < kCutDepth) { top.push_back(i); st.push_back({bn[i].right, d + 1}); st.push_back({bn[i].left, d + 1}); }
      else roots.push_back(i);
    }
  }

  // triangle counts of the internal nodes of a re-linked tree, from the leaves up
  void recount(std::vector<BinNode>& bn) {
    std::vector<uint32_t> top, roots;
    cut_tree(bn, top, roots);
    par_jobs(roots.size(), [&](size_t r) {
      std::vector<uint32_t> st{roots[r]}, post;
      while (!st.empty()) { const uint32_t i = st.back(); st.pop_back(); post.push_back(i); if (bn[i].left) { st.push_back(bn[i].left); st.push_back(bn[i].right); } }
      for (size_t pi = post.size(); pi-- > 0;) { BinNode& x = bn[post[pi]]; if (x.left) x.count = bn[x.left].count + bn[x.right].count; }
    });
    for (size_t pi = top.size(); pi-- > 0;) { BinNode& x = bn[top[pi]]; x.count = bn[x.left].count + bn[x.right].count; }
  }

  // triangles into the leaf order of the (re-linked) tree, left before right; ranges and counts from the leaves up
  void leaf_order(std::vector<BinNode>& bn) {
    const uint32_t r0 = bn[0].first, rn = bn[0].count;   // (the root's range: a worker's subtree owns a slice of the array)
    recount(bn);
    std::vector<rt_tri_t> tri2(rn);
    std::vector<rt_triex_t> ex2(triEx_ ? rn : 0);
    std::vector<uint32_t> top, roots;
    cut_tree(bn, top, roots);
    // where each node's triangles go: the root's range starts at r0, a left child's where its parent's does, a right child's behind the left's.
    // `first` of a LEAF still names where its triangles are now: the new place of a root is kept aside until its subtree is walked.
    std::vector<uint32_t> dst(roots.size());
    {
      size_t ri = 0;
      struct E { uint32_t node, first, depth; };    // (the walk of cut_tree: it meets the roots in the same order)
      std::vector<E> es{{0u, r0, 0u}};
      while (!es.empty()) {
        const E e = es.back(); es.pop_back();
        if (bn[e.node].left && e.depth < kCutDepth) {
          const uint32_t l = bn[e.node].left, r = bn[e.node].right;
          bn[e.node].first = e.first;
          es.push_back({r, e.first + bn[l].count, e.depth + 1}); es.push_back({l, e.first, e.depth + 1});
        } else dst[ri++] = e.first;
      }
    }
    par_jobs(roots.size(), [&](size_t r) {
      std::vector<std::pair<uint32_t, uint32_t>> st{{roots[r], dst[r]}};
      while (!st.empty()) {
        const auto [i, f] = st.back(); st.pop_back();
        BinNode& x = bn[i];
        if (x.left) { const uint32_t l = x.left, rr = x.right; x.first = f; st.push_back({rr, f + bn[l].count}); st.push_back({l, f}); continue; }
        const uint32_t at = f - r0;
        for (uint32_t k = 0; k < x.count; ++k) { tri2[at + k] = tri_[x.first + k]; if (triEx_) ex2[at + k] = triEx_[x.first + k]; }
        x.first = f;
      }
    });
    const size_t nch = ((size_t)rn + kChunk - 1) / kChunk;
    par_jobs(nch, [&](size_t c) {
      const size_t b0 = c * kChunk, b1 = std::min<size_t>(rn, b0 + kChunk);
      std::memcpy(tri_ + r0 + b0, tri2.data() + b0, (b1 - b0) * sizeof(rt_tri_t));
      if (triEx_) std::memcpy(triEx_ + r0 + b0, ex2.data() + b0, (b1 - b0) * sizeof(rt_triex_t));
    });
  }

  // binary binned-SAH tree under node `root` of `bn` (its range set), split until single triangles or no split separates anything.
  // Two passes over a node's triangles per level: one that fills the bins of all three axes, one that partitions them and takes the two
  // children's boxes (triangles and centroids) on the way -- the same boxes a pass of their own would find (min / max are exact).
  template <class Nodes>
  void build_binary(Nodes& bn, uint32_t root, uint32_t defer_below, std::vector<uint32_t>* deferred) {
    struct Item { uint32_t node; Box cbox; };
    std::vector<Item> st;
    {
      WideNode w;
      w.leftFirst = bn[root].first; w.triCount = bn[root].count;
      bounds(w);
      bn[root].box = w.box;
      st.push_back({root, w.cbox});
    }
    while (!st.empty()) {
      const Item it = st.back(); st.pop_back();
      const uint32_t i = it.node;
      WideNode w;
      w.leftFirst = bn[i].first; w.triCount = bn[i].count; w.box = bn[i].box; w.cbox = it.cbox;
      if (w.triCount <= 1) continue;
      if (deferred && i != root && w.triCount < defer_below) { deferred->push_back(i); continue; }
      const bool shared = par_nodes_ && w.triCount >= kBigNode;      // (a property of the node, not of the thread count)
      const Split s = w.triCount <= kSmallNode ? best_split_small(w) : shared ? best_split_shared(w) : best_split_fused(w);
      if (s.cost == INFINITY) continue;
      Box lb, lcb, rb, rcb;
      const uint32_t lc = shared ? partition_shared(w, s, lb, lcb, rb, rcb) : partition_bounds(w, s, lb, lcb, rb, rcb), rc = w.triCount - lc;
      if (lc == 0 || rc == 0) continue;
      const uint32_t l = (uint32_t)bn.size();
      bn.emplace_back(); bn.emplace_back();
      bn[l].first = w.leftFirst; bn[l].count = lc; bn[l].box = lb;
      bn[l + 1].first = w.leftFirst + lc; bn[l + 1].count = rc; bn[l + 1].box = rb;
      bn[i].left = l; bn[i].right = l + 1;
      st.push_back({l + 1, rcb}); st.push_back({l, lcb});
    }
  }

  // ---- the same two passes for a node of kBigNode triangles and more, shared by the threads ----
  static constexpr uint32_t kBigNode = 131072;
  bool par_nodes_ = false;

  Split best_split_shared(const WideNode& nd) const {
    float lo[3], scale[3]; bool on[3];
    for (int a = 0; a < 3; ++a) {
      lo[a] = comp(nd.cbox.lo, a);
      const float hi = comp(nd.cbox.hi, a);
      on[a] = lo[a] != hi;
      scale[a] = on[a] ? kBins / (hi - lo[a]) : 0.0f;
    }
    struct Bins { Box bb[3][kMaxBins]; int cnt[3][kMaxBins]; };
    const size_t nch = ((size_t)nd.triCount + kChunk - 1) / kChunk;
    std::vector<Bins> part(nch);
    par_jobs(nch, [&](size_t c) {
      Bins& B = part[c];
      for (int a = 0; a < 3; ++a) for (int i = 0; i < kBins; ++i) B.cnt[a][i] = 0;
      const uint32_t e = (uint32_t)std::min<size_t>(nd.triCount, (c + 1) * kChunk);
      for (uint32_t i = (uint32_t)(c * kChunk); i < e; ++i) {
        const rt_tri_t& t = tri_[nd.leftFirst + i];
        Box tb; tb.grow(tv(t.v0)); tb.grow(tv(t.v1)); tb.grow(tv(t.v2));
        const V3 cc = cent_[nd.leftFirst + i];
        for (int a = 0; a < 3; ++a) {
          if (!on[a]) continue;
          const int b = bin_of(comp(cc, a), lo[a], scale[a]);
          B.cnt[a][b]++;
          B.bb[a][b].lo = vmin(B.bb[a][b].lo, tb.lo); B.bb[a][b].hi = vmax(B.bb[a][b].hi, tb.hi);
        }
      }
    });
    Bins& T = part[0];      // (sums of integers and unions of boxes: the same totals in any order)
    for (size_t c = 1; c < nch; ++c)
      for (int a = 0; a < 3; ++a) for (int i = 0; i < kBins; ++i) { T.cnt[a][i] += part[c].cnt[a][i]; T.bb[a][i].grow(part[c].bb[a][i]); }
    Split best;
    for (int a = 0; a < 3; ++a) if (on[a]) sweep(T.bb[a], T.cnt[a], a, best);
    return best;
  }

  // Stable partition (left side in its order, then right side in its order: one outcome, however the work is cut): per chunk the size and the
  // boxes of its two sides, a prefix over the chunks, then every triangle -- 36 + 64 + 12 bytes -- moved once through a copy of the node's range.
  uint32_t partition_shared(const WideNode& nd, const Split& s, Box& lb, Box& lcb, Box& rb, Box& rcb) {
    const float lo = comp(nd.cbox.lo, s.axis), hi = comp(nd.cbox.hi, s.axis);
    const float scale = kBins / (hi - lo);
    const uint32_t n = nd.triCount, f = nd.leftFirst;
    const size_t nch = ((size_t)n + kChunk - 1) / kChunk;
    struct Side { uint32_t nl = 0; Box lb, lcb, rb, rcb; };
    std::vector<Side> part(nch);
    par_jobs(nch, [&](size_t c) {
      Side& S = part[c];
      const uint32_t e = (uint32_t)std::min<size_t>(n, (c + 1) * kChunk);
      for (uint32_t i = (uint32_t)(c * kChunk); i < e; ++i) {
        const rt_tri_t& t = tri_[f + i];
        const V3 cc = cent_[f + i];
        if (bin_of(comp(cc, s.axis), lo, scale) < s.pos) { S.lb.grow(tv(t.v0)); S.lb.grow(tv(t.v1)); S.lb.grow(tv(t.v2)); S.lcb.grow(cc); ++S.nl; }
        else { S.rb.grow(tv(t.v0)); S.rb.grow(tv(t.v1)); S.rb.grow(tv(t.v2)); S.rcb.grow(cc); }
      }
    });
    std::vector<uint32_t> loff(nch + 1, 0), roff(nch + 1, 0);
    for (size_t c = 0; c < nch; ++c) {
      const uint32_t cn = (uint32_t)(std::min<size_t>(n, (c + 1) * kChunk) - c * kChunk);
      loff[c + 1] = loff[c] + part[c].nl; roff[c + 1] = roff[c] + (cn - part[c].nl);
      lb.grow(part[c].lb); lcb.grow(part[c].lcb); rb.grow(part[c].rb); rcb.grow(part[c].rcb);
    }
    const uint32_t nl = loff[nch];
    if (nl == 0 || nl == n) return nl;
    if (tri2_.size() < n) { tri2_.resize(n); if (triEx_) ex2_.resize(n); cent2_.resize(n); }
    par_jobs(nch, [&](size_t c) {
      uint32_t l = loff[c], r = nl + roff[c];
      const uint32_t e = (uint32_t)std::min<size_t>(n, (c + 1) * kChunk);
      for (uint32_t i = (uint32_t)(c * kChunk); i < e; ++i) {
        const uint32_t to = bin_of(comp(cent_[f + i], s.axis), lo, scale) < s.pos ? l++ : r++;
        tri2_[to] = tri_[f + i]; cent2_[to] = cent_[f + i];
        if (triEx_) ex2_[to] = triEx_[f + i];
      }
    });
    par_jobs(nch, [&](size_t c) {
      const size_t b0 = c * kChunk, b1 = std::min<size_t>(n, b0 + kChunk);
      std::memcpy(tri_ + f + b0, tri2_.data() + b0, (b1 - b0) * sizeof(rt_tri_t));
      std::memcpy(&cent_[f + b0], cent2_.data() + b0, (b1 - b0) * sizeof(V3));
      if (triEx_) std::memcpy(triEx_ + f + b0, ex2_.data() + b0, (b1 - b0) * sizeof(rt_triex_t));
    });
    return nl;
  }
  std::vector<rt_tri_t> tri2_; std::vector<rt_triex_t> ex2_; std::vector<V3> cent2_;

  // the bins of the three axes in one pass over the node's triangles (what best_split fills axis by axis)
  Split best_split_fused(const WideNode& nd) const {
    float lo[3], scale[3]; bool on[3];
    for (int a = 0; a < 3; ++a) {
      lo[a] = comp(nd.cbox.lo, a);
      const float hi = comp(nd.cbox.hi, a);
      on[a] = lo[a] != hi;
      scale[a] = on[a] ? kBins / (hi - lo[a]) : 0.0f;
    }
    Box bb[3][kMaxBins]; int cnt[3][kMaxBins];
    for (int a = 0; a < 3; ++a) for (int i = 0; i < kBins; ++i) cnt[a][i] = 0;
    for (uint32_t i = 0; i < nd.triCount; ++i) {
      const rt_tri_t& t = tri_[nd.leftFirst + i];
      Box tb; tb.grow(tv(t.v0)); tb.grow(tv(t.v1)); tb.grow(tv(t.v2));
      const V3 c = cent_[nd.leftFirst + i];
      for (int a = 0; a < 3; ++a) {
        if (!on[a]) continue;
        const int b = bin_of(comp(c, a), lo[a], scale[a]);
        cnt[a][b]++;
        bb[a][b].lo = vmin(bb[a][b].lo, tb.lo); bb[a][b].hi = vmax(bb[a][b].hi, tb.hi);
      }
    }
    Split best;
    for (int a = 0; a < 3; ++a) if (on[a]) sweep(bb[a], cnt[a], a, best);
    return best;
  }

  // bvh.cpp:111-133 (the same two-pointer partition as `partition`), taking the boxes of the two sides as it decides each triangle
  uint32_t partition_bounds(const WideNode& nd, const Split& s, Box& lb, Box& lcb, Box& rb, Box& rcb) {
    const float lo = comp(nd.cbox.lo, s.axis), hi = comp(nd.cbox.hi, s.axis);
    const float scale = kBins / (hi - lo);
    int64_t i = 0, j = (int64_t)nd.triCount - 1;
    while (i <= j) {
      const uint32_t a = nd.leftFirst + (uint32_t)i;
      const rt_tri_t& t = tri_[a];
      const V3 c = cent_[a];
      if (bin_of(comp(c, s.axis), lo, scale) < s.pos) {
        lb.grow(tv(t.v0)); lb.grow(tv(t.v1)); lb.grow(tv(t.v2)); lcb.grow(c);
        ++i;
      } else {
        rb.grow(tv(t.v0)); rb.grow(tv(t.v1)); rb.grow(tv(t.v2)); rcb.grow(c);
        const uint32_t b = nd.leftFirst + (uint32_t)j;
        std::swap(tri_[a], tri_[b]);
        if (triEx_) std::swap(triEx_[a], triEx_[b]);
        std::swap(cent_[a], cent_[b]);
        --j;
      }
    }
    return (uint32_t)i;
  }

  void bounds(WideNode& nd) const {
    nd.box = Box(); nd.cbox = Box();
    for (uint32_t i = 0; i < nd.triCount; ++i) {
      const rt_tri_t& t = tri_[nd.leftFirst + i];
      nd.box.grow(tv(t.v0)); nd.box.grow(tv(t.v1)); nd.box.grow(tv(t.v2));
      nd.cbox.grow(cent_[nd.leftFirst + i]);
    }
  }

  int bin_of(float c, float lo, float scale) const {
    int b = (int)((c - lo) * scale);
    return b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
  }

  // Nodes of up to kSmallNode triangles: the same binned SAH, evaluated over the OCCUPIED bins only.  With fewer triangles than bins most
  // bins are empty; across a run of empty bins neither side of the plane changes, so the cost is constant there and only the first plane of
  // the run -- the one right after an occupied bin -- can be the first strict minimum the sweep below would report.  Same boxes (unions are
  // exact), same counts, same products, same order of comparison: the same split, at a cost proportional to the triangles instead of the bins
  // (the bottom of the tree is where most nodes are: 128 bins x 3 axes per 2-triangle node was 80 % of the build).
  static constexpr uint32_t kSmallNode = 64;
  Split best_split_small(const WideNode& nd) const {
    const uint32_t n = nd.triCount;
    Box tb[kSmallNode];
    for (uint32_t k = 0; k < n; ++k) {
      const rt_tri_t& t = tri_[nd.leftFirst + k];
      tb[k].grow(tv(t.v0)); tb[k].grow(tv(t.v1)); tb[k].grow(tv(t.v2));
    }
    Split best;
    for (int a = 0; a < 3; ++a) {
      const float lo = comp(nd.cbox.lo, a), hi = comp(nd.cbox.hi, a);
      if (lo == hi) continue;
      const float scale = kBins / (hi - lo);
      int bin[kSmallNode]; uint8_t ord[kSmallNode];
      for (uint32_t k = 0; k < n; ++k) {
        bin[k] = bin_of(comp(cent_[nd.leftFirst + k], a), lo, scale);
        uint32_t j = k;                                   // insertion sort of the triangle slots by bin
        while (j > 0 && bin[ord[j - 1]] > bin[k]) { ord[j] = ord[j - 1]; --j; }
        ord[j] = (uint8_t)k;
      }
      int gb[kSmallNode], gc[kSmallNode]; Box gx[kSmallNode]; uint32_t m = 0;      // occupied bins, ascending
      for (uint32_t k = 0; k < n; ++k) {
        const uint32_t t = ord[k];
        if (m == 0 || gb[m - 1] != bin[t]) { gb[m] = bin[t]; gc[m] = 0; gx[m] = Box(); ++m; }
        gc[m - 1]++; gx[m - 1].grow(tb[t]);
      }
      float ra[kSmallNode];
      { Box rb; int rs = 0; for (uint32_t j = m; j-- > 1;) { rs += gc[j]; rb.grow(gx[j]); ra[j - 1] = rs * rb.half_area(); } }
      Box lb; int ls = 0;
      for (uint32_t j = 0; j + 1 < m; ++j) {
        ls += gc[j]; lb.grow(gx[j]);
        const float c = ls * lb.half_area() + ra[j];
        if (c < best.cost) { best.axis = a; best.pos = gb[j] + 1; best.cost = c; }
      }
    }
    return best;
  }

  Split best_split(const WideNode& nd) const {   // binned SAH, 7 planes per axis (bvh.cpp:135-191)
    if (nd.triCount <= kSmallNode) return best_split_small(nd);
    Split best;
    for (int a = 0; a < 3; ++a) {
      const float lo = comp(nd.cbox.lo, a), hi = comp(nd.cbox.hi, a);
      if (lo == hi) continue;
      const float scale = kBins / (hi - lo);
      Box bb[kMaxBins]; int cnt[kMaxBins] = {0};
      for (uint32_t i = 0; i < nd.triCount; ++i) {
        const rt_tri_t& t = tri_[nd.leftFirst + i];
        const int b = bin_of(comp(cent_[nd.leftFirst + i], a), lo, scale);
        cnt[b]++;
        bb[b].grow(tv(t.v0)); bb[b].grow(tv(t.v1)); bb[b].grow(tv(t.v2));
      }
      sweep(bb, cnt, a, best);
    }
    return best;
  }

  // the planes of one axis, left to right: first strict minimum of (triangles x area) left + right
  void sweep(const Box* bb, const int* cnt, int a, Split& best) const {
    float la[kMaxBins], ra[kMaxBins];
    Box lb, rb; int ls = 0, rs = 0;
    for (int i = 0; i < kBins - 1; ++i) {
      ls += cnt[i]; lb.grow(bb[i]);
      la[i] = ls > 0 ? ls * lb.half_area() : INFINITY;
      rs += cnt[kBins - 1 - i]; rb.grow(bb[kBins - 1 - i]);
      ra[kBins - 2 - i] = rs > 0 ? rs * rb.half_area() : INFINITY;
    }
    for (int i = 0; i < kBins - 1; ++i) {
      const float c = la[i] + ra[i];
      if (c < best.cost) { best.axis = a; best.pos = i + 1; best.cost = c; }
    }
  }

  uint32_t partition(const WideNode& nd, const Split& s) {   // bvh.cpp:111-133
    const float lo = comp(nd.cbox.lo, s.axis), hi = comp(nd.cbox.hi, s.axis);
    const float scale = kBins / (hi - lo);
    int64_t i = 0, j = (int64_t)nd.triCount - 1;
    while (i <= j) {
      const uint32_t a = nd.leftFirst + (uint32_t)i;
      if (bin_of(comp(cent_[a], s.axis), lo, scale) < s.pos) {
        ++i;
      } else {
        const uint32_t b = nd.leftFirst + (uint32_t)j;
        std::swap(tri_[a], tri_[b]);
        if (triEx_) std::swap(triEx_[a], triEx_[b]);
        std::swap(cent_[a], cent_[b]);
        --j;
      }
    }
    return (uint32_t)i;
  }

  void subdivide(std::vector<WideNode>& nodes, uint32_t idx, uint32_t depth, uint32_t& maxd, uint32_t defer_below,
                 std::vector<std::pair<uint32_t, uint32_t>>* deferred) {
    maxd = std::max(maxd, depth);
    bounds(nodes[idx]);
    if (nodes[idx].triCount <= 1) return;
    if (deferred && depth > 0 && nodes[idx].triCount < defer_below) { deferred->push_back({idx, depth}); return; }
    std::vector<WideNode> cl;
    cl.push_back(nodes[idx]);
    while (cl.size() < width_) {
      Split bs; float bestDelta = 0.f; int bi = -1;
      for (int i = 0; i < (int)cl.size(); ++i) {
        if (cl[i].triCount <= 1) continue;
        Split s = best_split(cl[i]);
        if (s.cost == INFINITY) continue;
        // node cost as the reference prices it: surfaceArea * triCount with surfaceArea = 2*half
        // and split cost in half-areas (common.h:81-83 vs bvh.h:24-27) - kept, it biases to leaves
        float delta = 2.0f * cl[i].box.half_area() * cl[i].triCount - s.cost;
        if (delta <= 0.f) continue;
        // SAH leaf termination (extension): a leaf of a few triangles is cheaper to intersect than
        // another 4-wide node when the split barely separates them
        if (kLeafK > 0.f && (int)cl[i].triCount <= kLeafMax &&
            s.cost >= cl[i].box.half_area() * ((float)cl[i].triCount - kLeafK)) continue;
        if (kWiden == 1) delta = cl[i].box.half_area();
        if (delta > bestDelta) { bestDelta = delta; bs = s; bi = i; }
      }
      if (bi < 0) break;
      const uint32_t lc = partition(cl[bi], bs);
      const uint32_t rc = cl[bi].triCount - lc;
      if (lc == 0 || rc == 0) break;
      WideNode L, R;
      L.leftFirst = cl[bi].leftFirst; L.triCount = lc;
      R.leftFirst = cl[bi].leftFirst + lc; R.triCount = rc;
      bounds(L); bounds(R);
      cl[bi] = L;
      cl.push_back(R);
    }
    if (cl.size() == 1) return;   // leaf with several triangles
    const uint32_t first = (uint32_t)nodes.size();
    for (size_t i = 0; i < cl.size(); ++i) {
      nodes.emplace_back();
      nodes.back().leftFirst = cl[i].leftFirst;
      nodes.back().triCount = cl[i].triCount;
    }
    const uint32_t cc = (uint32_t)cl.size();
    for (uint32_t i = 0; i < cc; ++i) subdivide(nodes, first + i, depth + 1, maxd, defer_below, deferred);
    nodes[idx].triCount = 0;
    nodes[idx].leftFirst = first;
    nodes[idx].childCount = cc;
  }

  rt_tri_t* tri_;
  rt_triex_t* triEx_;
  uint32_t n_;
  uint32_t width_;
  std::vector<V3> cent_;
};

// ---------------------------------------------------------------------------------------------
// Quantisation (format of bvh.cpp:215-264; decode = origin + ldexp(q, e), rt_traversal.cpp:61-67)
// ---------------------------------------------------------------------------------------------
int8_t pick_exp(float extent) {
  if (!(extent > 0.0f)) return 0;   // reference: log2(0) -> -inf -> int8 cast (UB) gave 0 on x86-64 g++ 11
  float e = std::ceil(std::log2(extent / 255.0f));
  if (!(e >= -126.0f)) e = -126.0f;
  if (e > 126.0f) e = 126.0f;
  return (int8_t)e;
}

// returns false if some coordinate needs q > 255 at this exponent
bool quant_axis(float origin, int8_t e, float cmin, float cmax, uint8_t* qlo, uint8_t* qhi) {
  const float s = std::exp2f((float)e);
  float fl = std::floor((cmin - origin) / s), fh = std::ceil((cmax - origin) / s);
  if (fl < 0) fl = 0;
  if (fh < fl) fh = fl;
  if (fh > 255.0f) return false;
  int lo = (int)fl, hi = (int)fh;
  while (lo > 0 && origin + std::ldexp((float)lo, e) > cmin) --lo;     // conservative after the decode's rounding
  while (hi < 255 && origin + std::ldexp((float)hi, e) < cmax) ++hi;
  if (origin + std::ldexp((float)hi, e) < cmax) return false;
  *qlo = (uint8_t)lo; *qhi = (uint8_t)hi;
  return true;
}

template <class ChildBox>
void quantize_node(rt_qnode_t& q, const Box& box, uint32_t nchild, ChildBox child_box) {
  q.origin[0] = box.lo.x; q.origin[1] = box.lo.y; q.origin[2] = box.lo.z;
  int8_t e[3] = {pick_exp(box.hi.x - box.lo.x), pick_exp(box.hi.y - box.lo.y), pick_exp(box.hi.z - box.lo.z)};
  for (int a = 0; a < 3; ++a) {
    for (;;) {
      bool ok = true;
      for (uint32_t k = 0; k < nchild && ok; ++k) {
        Box cb = child_box(k);
        uint8_t l, h;
        ok = quant_axis(q.origin[a], e[a], comp(cb.lo, a), comp(cb.hi, a), &l, &h);
        if (ok) { q.children[k].qaabb[a] = l; q.children[k].qaabb[3 + a] = h; }
      }
      if (ok || e[a] >= 126) break;
      ++e[a];
    }
  }
  q.ex = e[0]; q.ey = e[1]; q.ez = e[2];
  for (uint32_t k = 0; k < RT_BVH_WIDTH; ++k) {
    if (k < nchild) q.children[k].meta = 1;
    else std::memset(&q.children[k], 0, sizeof(rt_child_t));
  }
}

// ---------------------------------------------------------------------------------------------
// 4x4 helpers (row-major, like geometry.h mat4_t)
// ---------------------------------------------------------------------------------------------
bool invert4(const float* m, float* out) {
  double a[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = m[i * 4 + j]; a[i][4 + j] = i == j; }
  for (int c = 0; c < 4; ++c) {
    int p = c;
    for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
    if (a[p][c] == 0) return false;
    for (int j = 0; j < 8; ++j) std::swap(a[c][j], a[p][j]);
    const double d = a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] /= d;
    for (int r = 0; r < 4; ++r) if (r != c) { const double f = a[r][c]; for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j]; }
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[i * 4 + j] = (float)a[i][4 + j];
  return true;
}
V3 xform_point(const float* m, V3 p) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}

// ---------------------------------------------------------------------------------------------
// Scene = what Tracer::init/setup uploads (tracer.cpp:124-161, 217-241)
// ---------------------------------------------------------------------------------------------
struct Scene {
  std::vector<rt_qnode_t> tlas;
  std::vector<rt_blas_t> blas;
  std::vector<rt_qnode_t> bvh;
  std::vector<rt_tri_t> tri;
  std::vector<rt_triex_t> triEx;
  std::vector<rt_material_t> mat;
  std::vector<uint8_t> tex;
  std::vector<uint32_t> triIdx;
  uint32_t max_depth = 0, n_leaves = 0, max_leaf = 0;
  float bounds[6] = {0};
};

struct TlasItem { Box box; uint32_t blasIdx; };

void build_tlas_rec(Scene& sc, std::vector<TlasItem>& items, uint32_t begin, uint32_t end, uint32_t nodeIdx, uint32_t depth, uint32_t* maxd) {
  *maxd = std::max(*maxd, depth);
  rt_qnode_t& q0 = sc.tlas[nodeIdx];
  std::memset(&q0, 0, sizeof q0);
  q0.imask = 1;
  Box box;
  for (uint32_t i = begin; i < end; ++i) box.grow(items[i].box);
  if (end - begin == 1) {   // TLAS leaf (bvh.cpp:325-328): leafData = blasIdx
    q0.origin[0] = box.lo.x; q0.origin[1] = box.lo.y; q0.origin[2] = box.lo.z;
    q0.ex = pick_exp(box.hi.x - box.lo.x); q0.ey = pick_exp(box.hi.y - box.lo.y); q0.ez = pick_exp(box.hi.z - box.lo.z);
    q0.leftFirst = 0;
    q0.leafData = items[begin].blasIdx;
    return;
  }
  // split into up to 4 groups: median cuts along the widest axis of the centroids
  std::vector<std::pair<uint32_t, uint32_t>> groups{{begin, end}};
  while (groups.size() < RT_BVH_WIDTH) {
    int gi = -1; uint32_t best = 1;
    for (size_t g = 0; g < groups.size(); ++g) { uint32_t c = groups[g].second - groups[g].first; if (c > best) { best = c; gi = (int)g; } }
    if (gi < 0) break;
    auto [b, e] = groups[gi];
    Box cb;
    for (uint32_t i = b; i < e; ++i) cb.grow((items[i].box.lo + items[i].box.hi) * 0.5f);
    V3 ext = cb.hi - cb.lo;
    int ax = ext.x >= ext.y ? (ext.x >= ext.z ? 0 : 2) : (ext.y >= ext.z ? 1 : 2);
    std::stable_sort(items.begin() + b, items.begin() + e, [ax](const TlasItem& l, const TlasItem& r) {
      return comp(l.box.lo, ax) + comp(l.box.hi, ax) < comp(r.box.lo, ax) + comp(r.box.hi, ax);
    });
    uint32_t mid = b + (e - b) / 2;
    groups[gi] = {b, mid};
    groups.push_back({mid, e});
  }
  const uint32_t first = (uint32_t)sc.tlas.size();
  const uint32_t nc = (uint32_t)groups.size();
  sc.tlas.resize(sc.tlas.size() + nc);
  std::vector<Box> cbx(nc);
  for (uint32_t k = 0; k < nc; ++k) for (uint32_t i = groups[k].first; i < groups[k].second; ++i) cbx[k].grow(items[i].box);
  {
    rt_qnode_t& q = sc.tlas[nodeIdx];
    quantize_node(q, box, nc, [&](uint32_t k) { return cbx[k]; });
    q.leftFirst = first;
    q.leafData = 0xffffffffu;   // internal (bvh.cpp:417)
  }
  for (uint32_t k = 0; k < nc; ++k) build_tlas_rec(sc, items, groups[k].first, groups[k].second, first + k, depth + 1, maxd);
}

Scene* build_scene(std::vector<Mesh>& meshes) {
  glap("meshes generated / loaded");
  auto sc = new Scene();
  size_t ntri = 0, nmat = 0;
  for (auto& m : meshes) { ntri += m.tri.size(); nmat += m.mats.size(); }
  sc->tri.reserve(ntri); sc->triEx.reserve(ntri); sc->mat.reserve(nmat);
  sc->blas.resize(meshes.size());
  std::vector<TlasItem> items;
  uint32_t tri_off = 0, bvh_off = 0, mat_off = 0;
  uint32_t blas_depth = 0;
  Box world;
  for (size_t mi = 0; mi < meshes.size(); ++mi) {
    Mesh& m = meshes[mi];
    // textures first so materials can point at them (scene.cpp:61-80)
    std::vector<uint64_t> tex_offsets;
    for (size_t t = 0; t < m.textures.size(); ++t) {
      tex_offsets.push_back(sc->tex.size());
      const uint8_t* p = (const uint8_t*)m.textures[t].data();
      sc->tex.insert(sc->tex.end(), p, p + m.textures[t].size() * 4);
    }
    for (auto mat : m.mats) {
      if (mat.diffuse_tex_id >= 0 && (size_t)mat.diffuse_tex_id < tex_offsets.size()) {
        mat.tex_offset = tex_offsets[mat.diffuse_tex_id];
        mat.tex_width = m.tex_dims[mat.diffuse_tex_id].first;
        mat.tex_height = m.tex_dims[mat.diffuse_tex_id].second;
      } else {
        mat.diffuse_tex_id = -1;
      }
      sc->mat.push_back(mat);
    }
    const uint32_t n = (uint32_t)m.tri.size();
    sc->tri.insert(sc->tri.end(), m.tri.begin(), m.tri.end());
    sc->triEx.insert(sc->triEx.end(), m.triEx.begin(), m.triEx.end());
    for (uint32_t j = 0; j < n; ++j) sc->triEx[tri_off + j].texId += mat_off;   // scene.cpp:56-58

    glap("triangles copied into the scene");
    BlasBuilder bb(sc->tri.data() + tri_off, sc->triEx.data() + tri_off, n);
    glap("BLAS built");
    blas_depth = std::max(blas_depth, bb.max_depth_);
    const uint32_t nn = (uint32_t)bb.nodes_.size();
    sc->bvh.resize(bvh_off + nn);
    {
      const size_t nch = ((size_t)nn + kChunk - 1) / kChunk;
      std::vector<uint32_t> leaves(nch, 0u), big(nch, 0u);
      par_jobs(nch, [&](size_t c) {
        const uint32_t e = (uint32_t)std::min<size_t>(nn, (c + 1) * kChunk);
        for (uint32_t i = (uint32_t)(c * kChunk); i < e; ++i) {
          const WideNode& w = bb.nodes_[i];
          rt_qnode_t& q = sc->bvh[bvh_off + i];
          std::memset(&q, 0, sizeof q);
          q.imask = 0;
          if (w.triCount == 0) {
            quantize_node(q, w.box, w.childCount, [&](uint32_t k) { return bb.nodes_[w.leftFirst + k].box; });
            q.leftFirst = w.leftFirst;      // relative to this BLAS's first node (rt_traversal.cpp:92,119)
            q.leafData = 0;
          } else {
            q.origin[0] = w.box.lo.x; q.origin[1] = w.box.lo.y; q.origin[2] = w.box.lo.z;
            q.ex = pick_exp(w.box.hi.x - w.box.lo.x); q.ey = pick_exp(w.box.hi.y - w.box.lo.y); q.ez = pick_exp(w.box.hi.z - w.box.lo.z);
            q.leftFirst = w.leftFirst + tri_off;   // bvh.cpp:260
            q.leafData = w.triCount;
            leaves[c]++;
            big[c] = std::max(big[c], w.triCount);
          }
        }
      });
      for (size_t c = 0; c < nch; ++c) { sc->n_leaves += leaves[c]; sc->max_leaf = std::max(sc->max_leaf, big[c]); }
    }
    glap("nodes quantised");
    rt_blas_t& b = sc->blas[mi];
    std::memset(&b, 0, sizeof b);
    b.bvh_offset = bvh_off;
    std::memcpy(b.transform, m.transform, sizeof b.transform);
    if (!invert4(m.transform, b.invTransform)) { delete sc; return nullptr; }
    b.mat_offset = mat_off;
    b.reflectivity = 0.0f;   // scene.cpp:96
    // world-space bounds of the instance (bvh.cpp:295-304)
    TlasItem it; it.blasIdx = (uint32_t)mi;
    const Box& rb = bb.nodes_[0].box;
    for (int c = 0; c < 8; ++c) {
      V3 p{c & 1 ? rb.hi.x : rb.lo.x, c & 2 ? rb.hi.y : rb.lo.y, c & 4 ? rb.hi.z : rb.lo.z};
      it.box.grow(xform_point(m.transform, p));
    }
    world.grow(it.box);
    items.push_back(it);
    tri_off += n; bvh_off += nn; mat_off += (uint32_t)m.mats.size();
  }
  sc->triIdx.resize(ntri);
  for (size_t i = 0; i < ntri; ++i) sc->triIdx[i] = (uint32_t)i;
  if (sc->mat.empty()) {   // shading always dereferences mat[texId] (closest.cpp:55): give it a default
    rt_material_t d{};
    d.diffuse[0] = d.diffuse[1] = d.diffuse[2] = 0.8f;
    d.diffuse_tex_id = -1;
    sc->mat.push_back(d);
  }
  if (sc->tex.empty()) sc->tex.resize(4, 0);
  sc->tlas.resize(1);
  uint32_t tlas_depth = 0;
  build_tlas_rec(*sc, items, 0, (uint32_t)items.size(), 0, 0, &tlas_depth);
  sc->max_depth = tlas_depth + blas_depth;   // TLAS leaf and BLAS root share a level (rt_traversal.cpp:109-121)
  sc->bounds[0] = world.lo.x; sc->bounds[1] = world.lo.y; sc->bounds[2] = world.lo.z;
  sc->bounds[3] = world.hi.x; sc->bounds[4] = world.hi.y; sc->bounds[5] = world.hi.z;
  glap("scene assembled");
  return sc;
}

// ---------------------------------------------------------------------------------------------
// Scenes in the formats of the software twin (tests/regression/raycast/common.h): binary BVH of 32-byte
// nodes with adjacent children, triIdx indirection (identity here: triangles are reordered in place),
// 160-byte instance records carrying the texture, 32-byte TLAS nodes with 16-bit child indices.  Same SAH
// builder with width 2; the reference's own builder is raycast/bvh.cpp (not reproduced node for node).
// ---------------------------------------------------------------------------------------------
#pragma pack(push, 1)
struct rc_bvh_node_t { float aabbMin[3]; uint32_t leftFirst; float aabbMax[3]; uint32_t triCount; };
struct rc_tlas_node_t { float aabbMin[3]; uint32_t leftRight; float aabbMax[3]; uint32_t blasIdx; };
struct rc_blas_t { float transform[16]; float invTransform[16]; uint32_t bvh_offset, _pad0; uint64_t tex_offset; uint32_t tex_width, tex_height; float reflectivity; uint32_t _pad1; };
struct rc_triex_t { float N0[3], N1[3], N2[3]; float uv0[2], uv1[2], uv2[2]; };
#pragma pack(pop)
static_assert(sizeof(rc_bvh_node_t) == 32 && sizeof(rc_tlas_node_t) == 32 && sizeof(rc_blas_t) == 160 && sizeof(rc_triex_t) == 60, "raycast layouts");

struct RcScene {
  std::vector<rc_tlas_node_t> tlas;
  std::vector<rc_blas_t> blas;
  std::vector<rc_bvh_node_t> bvh;
  std::vector<rt_tri_t> tri;
  std::vector<rc_triex_t> triEx;
  std::vector<uint32_t> triIdx;
  std::vector<uint8_t> tex;
  uint32_t tlas_root = 0, max_depth = 0;
  float bounds[6] = {0};
};

static uint32_t rc_tlas_rec(RcScene& sc, std::vector<TlasItem>& items, uint32_t begin, uint32_t end) {
  if (end - begin == 1) {
    rc_tlas_node_t n{};
    const Box& b = items[begin].box;
    n.aabbMin[0] = b.lo.x; n.aabbMin[1] = b.lo.y; n.aabbMin[2] = b.lo.z;
    n.aabbMax[0] = b.hi.x; n.aabbMax[1] = b.hi.y; n.aabbMax[2] = b.hi.z;
    n.leftRight = 0; n.blasIdx = items[begin].blasIdx;
    sc.tlas.push_back(n);
    return (uint32_t)sc.tlas.size() - 1;
  }
  Box cb;
  for (uint32_t i = begin; i < end; ++i) cb.grow((items[i].box.lo + items[i].box.hi) * 0.5f);
  const V3 ext = cb.hi - cb.lo;
  const int axis = ext.x >= ext.y && ext.x >= ext.z ? 0 : (ext.y >= ext.z ? 1 : 2);
  std::sort(items.begin() + begin, items.begin() + end, [&](const TlasItem& a, const TlasItem& b) {
    return comp(a.box.lo + a.box.hi, axis) < comp(b.box.lo + b.box.hi, axis); });
  const uint32_t mid = begin + (end - begin) / 2;
  const uint32_t l = rc_tlas_rec(sc, items, begin, mid), r = rc_tlas_rec(sc, items, mid, end);
  rc_tlas_node_t n{};
  Box b; 
  b.grow(V3{sc.tlas[l].aabbMin[0], sc.tlas[l].aabbMin[1], sc.tlas[l].aabbMin[2]}); b.grow(V3{sc.tlas[l].aabbMax[0], sc.tlas[l].aabbMax[1], sc.tlas[l].aabbMax[2]});
  b.grow(V3{sc.tlas[r].aabbMin[0], sc.tlas[r].aabbMin[1], sc.tlas[r].aabbMin[2]}); b.grow(V3{sc.tlas[r].aabbMax[0], sc.tlas[r].aabbMax[1], sc.tlas[r].aabbMax[2]});
  n.aabbMin[0] = b.lo.x; n.aabbMin[1] = b.lo.y; n.aabbMin[2] = b.lo.z;
  n.aabbMax[0] = b.hi.x; n.aabbMax[1] = b.hi.y; n.aabbMax[2] = b.hi.z;
  n.leftRight = (r << 16) | l;    // common.h:72-74 (16-bit indices)
  n.blasIdx = 0;
  sc.tlas.push_back(n);
  return (uint32_t)sc.tlas.size() - 1;
}

RcScene* build_rc_scene(std::vector<Mesh>& meshes, const float* reflectivity) {
  auto sc = new RcScene();
  if (meshes.size() > 0x7fff) { delete sc; return nullptr; }   // TLAS child indices are 16 bit
  std::vector<TlasItem> items;
  uint32_t tri_off = 0, bvh_off = 0;
  Box world;
  for (size_t mi = 0; mi < meshes.size(); ++mi) {
    Mesh& m = meshes[mi];
    const uint32_t n = (uint32_t)m.tri.size();
    sc->tri.insert(sc->tri.end(), m.tri.begin(), m.tri.end());
    std::vector<rt_triex_t> ex(m.triEx);
    BlasBuilder bb(sc->tri.data() + tri_off, ex.data(), n, 2);
    for (uint32_t j = 0; j < n; ++j) {
      rc_triex_t e;
      std::memcpy(e.N0, ex[j].N0, sizeof e.N0); std::memcpy(e.N1, ex[j].N1, sizeof e.N1); std::memcpy(e.N2, ex[j].N2, sizeof e.N2);
      std::memcpy(e.uv0, ex[j].uv0, sizeof e.uv0); std::memcpy(e.uv1, ex[j].uv1, sizeof e.uv1); std::memcpy(e.uv2, ex[j].uv2, sizeof e.uv2);
      sc->triEx.push_back(e);
    }
    sc->max_depth = std::max(sc->max_depth, bb.max_depth_);
    const uint32_t nn = (uint32_t)bb.nodes_.size();
    sc->bvh.resize(bvh_off + nn);
    for (uint32_t i = 0; i < nn; ++i) {
      const WideNode& w = bb.nodes_[i];
      rc_bvh_node_t& q = sc->bvh[bvh_off + i];
      q.aabbMin[0] = w.box.lo.x; q.aabbMin[1] = w.box.lo.y; q.aabbMin[2] = w.box.lo.z;
      q.aabbMax[0] = w.box.hi.x; q.aabbMax[1] = w.box.hi.y; q.aabbMax[2] = w.box.hi.z;
      if (w.triCount == 0) { q.leftFirst = w.leftFirst; q.triCount = 0; }               // children left, left + 1 (render.h:103-104)
      else { q.leftFirst = w.leftFirst + tri_off; q.triCount = w.triCount; }            // index into triIdx
    }
    rc_blas_t b{};
    std::memcpy(b.transform, m.transform, sizeof b.transform);
    if (!invert4(m.transform, b.invTransform)) { delete sc; return nullptr; }
    b.bvh_offset = bvh_off;
    b.tex_offset = sc->tex.size();
    if (!m.textures.empty()) {
      b.tex_width = m.tex_dims[0].first; b.tex_height = m.tex_dims[0].second;
      const uint8_t* p = (const uint8_t*)m.textures[0].data();
      sc->tex.insert(sc->tex.end(), p, p + m.textures[0].size() * 4);
    } else {
      const uint32_t px[4] = {0xC8C8C8u, 0xB4B4B4u, 0xB4B4B4u, 0xC8C8C8u};
      b.tex_width = 2; b.tex_height = 2;
      sc->tex.insert(sc->tex.end(), (const uint8_t*)px, (const uint8_t*)px + sizeof px);
    }
    b.reflectivity = reflectivity ? reflectivity[mi] : 0.0f;
    sc->blas.push_back(b);
    TlasItem it; it.blasIdx = (uint32_t)mi;
    const Box& rb = bb.nodes_[0].box;
    for (int c = 0; c < 8; ++c) {
      V3 p{c & 1 ? rb.hi.x : rb.lo.x, c & 2 ? rb.hi.y : rb.lo.y, c & 4 ? rb.hi.z : rb.lo.z};
      it.box.grow(xform_point(m.transform, p));
    }
    world.grow(it.box);
    items.push_back(it);
    tri_off += n; bvh_off += nn;
  }
  sc->triIdx.resize(tri_off);
  for (uint32_t i = 0; i < tri_off; ++i) sc->triIdx[i] = i;
  sc->tlas.emplace_back();            // index 0 is never a child: leftRight == 0 means "leaf" (common.h:68)
  std::memset(&sc->tlas[0], 0, sizeof(rc_tlas_node_t));
  sc->tlas_root = rc_tlas_rec(*sc, items, 0, (uint32_t)items.size());
  sc->bounds[0] = world.lo.x; sc->bounds[1] = world.lo.y; sc->bounds[2] = world.lo.z;
  sc->bounds[3] = world.hi.x; sc->bounds[4] = world.hi.y; sc->bounds[5] = world.hi.z;
  return sc;
}

// ---------------------------------------------------------------------------------------------
// deterministic procedural content
// ---------------------------------------------------------------------------------------------
struct Rng {   // Marsaglia xorshift32 seeded through WangHash, as raytracing/common.h:129-147
  uint32_t s;
  explicit Rng(uint32_t seed) {
    uint32_t x = seed;
    x = (x ^ 61) ^ (x >> 16); x *= 9; x = x ^ (x >> 4); x *= 0x27d4eb2d; x = x ^ (x >> 15);
    s = x ? x : 1;
  }
  uint32_t u32() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
  float f() { return u32() * 2.3283064365387e-10f; }
  float range(float a, float b) { return a + (b - a) * f(); }
};

rt_material_t make_mat(float r, float g, float b, int tex) {
  rt_material_t m{};
  m.ambient[0] = m.ambient[1] = m.ambient[2] = 0.1f;
  m.diffuse[0] = r; m.diffuse[1] = g; m.diffuse[2] = b;
  m.shininess = 1; m.ior = 1; m.dissolve = 1;
  m.diffuse_tex_id = tex;
  return m;
}

void add_tri(Mesh& m, V3 a, V3 b, V3 c, V3 na, V3 nb, V3 nc, float ua, float va, float ub, float vb, float uc, float vc, uint32_t mat) {
  rt_tri_t t; rt_triex_t e;
  t.v0[0] = a.x; t.v0[1] = a.y; t.v0[2] = a.z;
  t.v1[0] = b.x; t.v1[1] = b.y; t.v1[2] = b.z;
  t.v2[0] = c.x; t.v2[1] = c.y; t.v2[2] = c.z;
  e.N0[0] = na.x; e.N0[1] = na.y; e.N0[2] = na.z;
  e.N1[0] = nb.x; e.N1[1] = nb.y; e.N1[2] = nb.z;
  e.N2[0] = nc.x; e.N2[1] = nc.y; e.N2[2] = nc.z;
  e.uv0[0] = ua; e.uv0[1] = va; e.uv1[0] = ub; e.uv1[1] = vb; e.uv2[0] = uc; e.uv2[1] = vc;
  e.texId = mat;
  m.tri.push_back(t); m.triEx.push_back(e);
}

// parametric sheet: nu x nv quads of P(u,v), u,v in [0,1]; normals by finite differences
template <class F>
void add_sheet(Mesh& m, uint32_t nu, uint32_t nv, uint32_t mat, float uv_scale, F P) {
  std::vector<V3> p((size_t)(nu + 1) * (nv + 1)), nrm(p.size());
  for (uint32_t j = 0; j <= nv; ++j) for (uint32_t i = 0; i <= nu; ++i) p[(size_t)j * (nu + 1) + i] = P((float)i / nu, (float)j / nv);
  const float h = 1e-3f;
  for (uint32_t j = 0; j <= nv; ++j) for (uint32_t i = 0; i <= nu; ++i) {
    float u = (float)i / nu, v = (float)j / nv;
    V3 du = P(u + h, v) - P(u - h, v), dv = P(u, v + h) - P(u, v - h);
    nrm[(size_t)j * (nu + 1) + i] = normalize(cross(du, dv));
  }
  for (uint32_t j = 0; j < nv; ++j) for (uint32_t i = 0; i < nu; ++i) {
    size_t a = (size_t)j * (nu + 1) + i, b = a + 1, c = a + nu + 1, d = c + 1;
    float u0 = uv_scale * i / nu, u1 = uv_scale * (i + 1) / nu, v0 = uv_scale * j / nv, v1 = uv_scale * (j + 1) / nv;
    add_tri(m, p[a], p[b], p[d], nrm[a], nrm[b], nrm[d], u0, v0, u1, v0, u1, v1, mat);
    add_tri(m, p[a], p[d], p[c], nrm[a], nrm[d], nrm[c], u0, v0, u1, v1, u0, v1, mat);
  }
}

void add_texture(Mesh& m, uint32_t w, uint32_t h, uint32_t kind, uint32_t seed) {
  std::vector<uint32_t> px((size_t)w * h);
  Rng rng(seed);
  uint32_t base[3] = {128 + rng.u32() % 100, 128 + rng.u32() % 100, 128 + rng.u32() % 100};
  for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) {
    uint32_t k;
    switch (kind % 4) {
    case 0: k = ((x / 16) ^ (y / 16)) & 1 ? 255 : 140; break;                       // checker
    case 1: k = (y % 32 < 3 || (x + (y / 32 % 2) * 32) % 64 < 3) ? 90 : 230; break;  // bricks
    case 2: k = 150 + (uint32_t)(100.0 * (0.5 + 0.5 * std::sin(x * 0.11 + 3.0 * std::sin(y * 0.07)))); break;  // marble
    default: k = 120 + ((x * 2654435761u ^ y * 40503u) >> 27) * 4; break;           // noise
    }
    uint32_t r = base[0] * k / 255, g = base[1] * k / 255, b = base[2] * k / 255;
    px[(size_t)y * w + x] = (r << 16) | (g << 8) | b;
  }
  m.textures.push_back(std::move(px));
  m.tex_dims.push_back({w, h});
}

inline float bump(float x, float y, float amp) {   // cheap smooth relief so big surfaces are not flat
  return amp * (std::sin(x * 0.045f) * std::sin(y * 0.06f) + 0.5f * std::sin(x * 0.13f + 1.3f) * std::sin(y * 0.17f + 0.7f));
}

// small rigid rotation so that nothing in the big scenes is exactly axis aligned (axis-aligned
// zero-thickness boxes trip the reference traverser's 2^32-iteration spin)
void tilt(Mesh& m, float ay, float ax) {
  const float cy = std::cos(ay), sy = std::sin(ay), cx = std::cos(ax), sx = std::sin(ax);
  auto rot = [&](float* p, float px, float py, float pz) {
    float x = p[0] - px, y = p[1] - py, z = p[2] - pz;
    float x1 = cy * x + sy * z, z1 = -sy * x + cy * z;
    float y2 = cx * y - sx * z1, z2 = sx * y + cx * z1;
    p[0] = x1 + px; p[1] = y2 + py; p[2] = z2 + pz;
  };
  for (auto& t : m.tri) { rot(t.v0, 0, 100, 0); rot(t.v1, 0, 100, 0); rot(t.v2, 0, 100, 0); }
  for (auto& e : m.triEx) { rot(e.N0, 0, 0, 0); rot(e.N1, 0, 0, 0); rot(e.N2, 0, 0, 0); }
}

// config 1: 12-triangle Cornell box in front of the RTU kernel's fixed camera (0,100,0)->+x
Mesh make_cornell() {
  Mesh m;
  m.mats = {make_mat(0.73f, 0.73f, 0.73f, -1), make_mat(0.65f, 0.05f, 0.05f, -1), make_mat(0.12f, 0.45f, 0.15f, -1)};
  const float x0 = 60, x1 = 260, y0 = 0, y1 = 200, z0 = -100, z1 = 100;
  auto quad = [&](V3 a, V3 b, V3 c, V3 d, uint32_t mat) {
    V3 n = normalize(cross(b - a, c - a));
    add_tri(m, a, b, c, n, n, n, 0, 0, 1, 0, 1, 1, mat);
    add_tri(m, a, c, d, n, n, n, 0, 0, 1, 1, 0, 1, mat);
  };
  quad({x0, y0, z0}, {x0, y0, z1}, {x1, y0, z1}, {x1, y0, z0}, 0);   // floor
  quad({x0, y1, z0}, {x1, y1, z0}, {x1, y1, z1}, {x0, y1, z1}, 0);   // ceiling
  quad({x1, y0, z0}, {x1, y0, z1}, {x1, y1, z1}, {x1, y1, z0}, 0);   // back wall
  quad({x0, y0, z0}, {x1, y0, z0}, {x1, y1, z0}, {x0, y1, z0}, 1);   // left (red)
  quad({x0, y0, z1}, {x0, y1, z1}, {x1, y1, z1}, {x1, y0, z1}, 2);   // right (green)
  quad({150, 60, -40}, {150, 60, 30}, {210, 60, 30}, {210, 60, -40}, 0);  // top face of the short box
  return m;
}

// config 2: "bunny-class" blob: subdivided icosahedron with radial harmonics.  cx: distance of its centre along the fixed
// camera's axis (kernel.cpp:28-39: eye (0,100,0) looking along +x, 45 degrees to the top and bottom frame edges); 220 leaves
// it small in the frame (14 % of a square frame), 135 ("bunny") makes it fill the view as BASELINE config 2 means it
Mesh make_blob(uint32_t subdiv, uint32_t seed, float cx = 220.0f) {
  Mesh m;
  m.mats = {make_mat(0.8f, 0.7f, 0.6f, 0)};
  add_texture(m, 256, 256, 2, seed);
  const float t = (1.0f + std::sqrt(5.0f)) / 2.0f;
  std::vector<V3> v = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t}, {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
  for (auto& p : v) p = normalize(p);
  std::vector<uint32_t> f = {0, 11, 5, 0, 5, 1, 0, 1, 7, 0, 7, 10, 0, 10, 11, 1, 5, 9, 5, 11, 4, 11, 10, 2, 10, 7, 6, 7, 1, 8,
                             3, 9, 4, 3, 4, 2, 3, 2, 6, 3, 6, 8, 3, 8, 9, 4, 9, 5, 2, 4, 11, 6, 2, 10, 8, 6, 7, 9, 8, 1};
  for (uint32_t s = 0; s < subdiv; ++s) {
    std::map<uint64_t, uint32_t> mid;
    auto midpoint = [&](uint32_t a, uint32_t b) {
      uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
      auto it = mid.find(key);
      if (it != mid.end()) return it->second;
      v.push_back(normalize((v[a] + v[b]) * 0.5f));
      return mid[key] = (uint32_t)v.size() - 1;
    };
    std::vector<uint32_t> nf;
    nf.reserve(f.size() * 4);
    for (size_t i = 0; i < f.size(); i += 3) {
      uint32_t a = f[i], b = f[i + 1], c = f[i + 2];
      uint32_t ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
      uint32_t q[12] = {a, ab, ca, b, bc, ab, c, ca, bc, ab, bc, ca};
      nf.insert(nf.end(), q, q + 12);
    }
    f.swap(nf);
  }
  Rng rng(seed);
  float ph[6];
  for (float& x : ph) x = rng.range(0.f, 6.2831853f);
  auto radius = [&](V3 d) {
    return 80.0f * (1.0f + 0.15f * std::sin(5 * d.x + ph[0]) * std::sin(4 * d.y + ph[1]) + 0.10f * std::sin(9 * d.z + ph[2]) * std::sin(7 * d.x + ph[3]) +
                    0.05f * std::sin(17 * d.y + ph[4]) * std::sin(13 * d.z + ph[5]));
  };
  const V3 c{cx, 100, 0};
  std::vector<V3> p(v.size());
  for (size_t i = 0; i < v.size(); ++i) p[i] = c + v[i] * radius(v[i]);
  std::vector<V3> nrm(v.size(), V3{0, 0, 0});
  for (size_t i = 0; i < f.size(); i += 3) {
    V3 n = cross(p[f[i + 1]] - p[f[i]], p[f[i + 2]] - p[f[i]]);
    nrm[f[i]] = nrm[f[i]] + n; nrm[f[i + 1]] = nrm[f[i + 1]] + n; nrm[f[i + 2]] = nrm[f[i + 2]] + n;
  }
  for (auto& n : nrm) n = normalize(n);
  auto uvof = [&](V3 d, float* u, float* w) { *u = 0.5f + std::atan2(d.z, d.x) / 6.2831853f; *w = 0.5f - std::asin(std::max(-1.f, std::min(1.f, d.y))) / 3.14159265f; };
  for (size_t i = 0; i < f.size(); i += 3) {
    float u0, v0, u1, v1, u2, v2;
    uvof(v[f[i]], &u0, &v0); uvof(v[f[i + 1]], &u1, &v1); uvof(v[f[i + 2]], &u2, &v2);
    add_tri(m, p[f[i]], p[f[i + 1]], p[f[i + 2]], nrm[f[i]], nrm[f[i + 1]], nrm[f[i + 2]], u0 * 4, v0 * 4, u1 * 4, v1 * 4, u2 * 4, v2 * 4, 0);
  }
  return m;
}

// config 3/4: "Sponza-class" atrium.  `level` scales the tessellation: triangle count = 2^(2*level+4)
// (level 8 -> 1,048,576).  Hall x in [-600,1400], y in [0,600], z in [-400,400]; camera (0,100,0).
Mesh make_atrium(uint32_t level, uint32_t seed) {
  Mesh m;
  Rng rng(seed);
  for (int i = 0; i < 16; ++i) m.mats.push_back(make_mat(rng.range(0.4f, 0.9f), rng.range(0.4f, 0.9f), rng.range(0.4f, 0.9f), i < 8 ? i : -1));
  for (uint32_t i = 0; i < 8; ++i) add_texture(m, 256, 256, i, seed * 31 + i);
  const uint32_t g = 1u << level;          // 256 at level 8
  const uint32_t gh = g / 2, gq = g / 8 ? g / 8 : 1;
  const float X0 = -600, X1 = 1400, Y0 = 0, Y1 = 600, Z0 = -400, Z1 = 400;
  // floor / ceiling: g x g quads each
  add_sheet(m, g, g, 0, 16, [&](float u, float v) { float x = X0 + (X1 - X0) * u, z = Z0 + (Z1 - Z0) * v; return V3{x, Y0 + bump(x, z, 1.5f), z}; });
  add_sheet(m, g, g, 1, 16, [&](float u, float v) { float x = X0 + (X1 - X0) * u, z = Z1 - (Z1 - Z0) * v; return V3{x, Y1 + bump(x, z, 4.0f), z}; });
  // four walls: g x g/2 quads each
  add_sheet(m, g, gh, 2, 12, [&](float u, float v) { float x = X0 + (X1 - X0) * u, y = Y0 + (Y1 - Y0) * v; return V3{x, y, Z0 + bump(x, y, 3.0f)}; });
  add_sheet(m, g, gh, 3, 12, [&](float u, float v) { float x = X1 - (X1 - X0) * u, y = Y0 + (Y1 - Y0) * v; return V3{x, y, Z1 + bump(x, y, 3.0f)}; });
  add_sheet(m, g, gh, 4, 8, [&](float u, float v) { float z = Z1 - (Z1 - Z0) * u, y = Y0 + (Y1 - Y0) * v; return V3{X0 + bump(z, y, 3.0f), y, z}; });
  add_sheet(m, g, gh, 5, 8, [&](float u, float v) { float z = Z0 + (Z1 - Z0) * u, y = Y0 + (Y1 - Y0) * v; return V3{X1 + bump(z, y, 3.0f), y, z}; });
  // 32 fluted columns in two rows: (g/2) x (g/8) quads each
  for (int c = 0; c < 32; ++c) {
    const float cx = X0 + 100 + (c / 2) * ((X1 - X0 - 200) / 15.0f), cz = (c & 1) ? 220.0f : -220.0f;
    const float r0 = 28.0f + 4.0f * rng.f();
    add_sheet(m, gh, gq, 6 + (c % 4), 4, [&](float u, float v) {
      float a = 6.2831853f * u, y = Y0 + (Y1 - Y0) * 0.75f * v;
      float r = r0 * (1.0f + 0.06f * std::cos(16 * a)) * (1.0f - 0.15f * v + 0.2f * std::exp(-40 * v) + 0.25f * std::exp(-40 * (1 - v)));
      return V3{cx + r * std::cos(a), y, cz - r * std::sin(a)};
    });
  }
  // 16 hanging drapes / arches across the nave: (g/2) x (g/4) quads each
  for (int d = 0; d < 16; ++d) {
    const float dx = X0 + 160 + d * ((X1 - X0 - 320) / 15.0f);
    const float ph = rng.range(0.f, 6.28f), amp = rng.range(10.f, 25.f);
    add_sheet(m, gh, g / 4 ? g / 4 : 1, 10 + (d % 6), 6, [&](float u, float v) {
      float z = -200.0f + 400.0f * u;
      float arch = 450.0f + 100.0f * std::sin(3.14159265f * u);
      float y = arch - 160.0f * v;
      return V3{dx + amp * std::sin(10 * u + ph) * (0.3f + v) + 6.0f * std::sin(14 * v + ph), y, z};
    });
  }
  tilt(m, 0.0617f, 0.0291f);
  return m;
}

// config 5: hairball - `strands` splines of `segs` thin triangles pairs inside a sphere
// cx: distance of the centre from the camera; 260 = small in the frame, 150 ("hairball_fill") = the ball spans the 16:9 view
Mesh make_hairball(uint32_t strands, uint32_t segs, uint32_t seed, float cx = 260.0f) {
  Mesh m;
  m.mats = {make_mat(0.85f, 0.75f, 0.55f, -1)};
  Rng rng(seed);
  const V3 c{cx, 100, 0};
  const float R = 120.0f;
  m.tri.reserve((size_t)strands * segs * 2);
  m.triEx.reserve((size_t)strands * segs * 2);
  for (uint32_t s = 0; s < strands; ++s) {
    V3 d = normalize(V3{rng.range(-1, 1), rng.range(-1, 1), rng.range(-1, 1)});
    V3 side = normalize(cross(d, V3{0.3f, 1.0f, 0.2f}));
    V3 w1 = normalize(V3{rng.range(-1, 1), rng.range(-1, 1), rng.range(-1, 1)});
    const float f1 = rng.range(2.f, 9.f), a1 = rng.range(4.f, 18.f), wdt = 0.35f;
    V3 prev{0, 0, 0}; bool have = false;
    for (uint32_t k = 0; k <= segs; ++k) {
      float t = (float)k / segs;
      V3 p = c + d * (R * (0.15f + 0.85f * t)) + w1 * (a1 * std::sin(f1 * t * 6.28f) * t);
      if (have) {
        V3 n = normalize(cross(p - prev, side));
        add_tri(m, prev - side * wdt, prev + side * wdt, p + side * wdt, n, n, n, 0, 0, 1, 0, 1, 1, 0);
        add_tri(m, prev - side * wdt, p + side * wdt, p - side * wdt, n, n, n, 0, 0, 1, 1, 0, 1, 0);
      }
      prev = p; have = true;
    }
  }
  return m;
}

// ---------------------------------------------------------------------------------------------
// Image ingest (role of surface.cpp:28-55, which calls stb_image with 3 forced channels and packs
// (r << 16) + (g << 8) + b): PNG (8/16-bit, grey / grey+alpha / RGB / RGBA / palette, non-interlaced and
// Adam7, zlib inflate) and binary/ASCII PPM/PGM.  Written from the PNG specification, no stb code.
// ---------------------------------------------------------------------------------------------
static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo the scanline filters of one (sub)image in place: rows of `stride` bytes, each preceded by its filter byte
static bool png_unfilter(uint8_t* d, size_t rows, size_t stride, size_t bpp, uint8_t* out) {
  std::vector<uint8_t> zero(stride, 0);
  const uint8_t* prev = zero.data();
  for (size_t y = 0; y < rows; ++y) {
    const uint8_t f = d[y * (stride + 1)];
    const uint8_t* in = d + y * (stride + 1) + 1;
    uint8_t* o = out + y * stride;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? o[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int v;
      switch (f) {
      case 0: v = in[i]; break;
      case 1: v = in[i] + a; break;
      case 2: v = in[i] + b; break;
      case 3: v = in[i] + ((a + b) >> 1); break;
      case 4: v = in[i] + paeth(a, b, c); break;
      default: return false;
      }
      o[i] = (uint8_t)v;
    }
    prev = o;
  }
  return true;
}

static bool load_png(const std::vector<uint8_t>& f, uint32_t& W, uint32_t& H, std::vector<uint32_t>& px) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (f.size() < 8 + 25 || std::memcmp(f.data(), sig, 8) != 0) return false;
  uint32_t depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte;
  size_t pos = 8;
  bool have_hdr = false, end = false;
  while (!end && pos + 12 <= f.size()) {
    const uint32_t len = be32(&f[pos]);
    const char* type = (const char*)&f[pos + 4];
    if (pos + 12 + (size_t)len > f.size()) return false;
    const uint8_t* data = &f[pos + 8];
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) return false;
      W = be32(data); H = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
      if (data[10] != 0 || data[11] != 0 || interlace > 1) return false;
      have_hdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
    else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!std::memcmp(type, "IEND", 4)) end = true;
    pos += 12 + (size_t)len;
  }
  if (!have_hdr || W == 0 || H == 0 || (uint64_t)W * H > (1ull << 28)) return false;
  uint32_t ch;
  switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: return false; }
  if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return false;
  if (ctype == 3 && depth == 16) return false;
  const size_t bits = (size_t)ch * depth, bpp = std::max<size_t>(1, bits / 8);
  auto stride_of = [&](uint32_t w) { return ((size_t)w * bits + 7) / 8; };
  // sub-images: one for non-interlaced, the seven Adam7 passes otherwise
  struct Pass { uint32_t x0, y0, dx, dy; };
  static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  std::vector<Pass> passes;
  if (interlace) passes.assign(adam7, adam7 + 7); else passes.push_back({0, 0, 1, 1});
  size_t raw = 0;
  for (const Pass& ps : passes) {
    const uint32_t pw = (W - ps.x0 + ps.dx - 1) / ps.dx, ph = (H - ps.y0 + ps.dy - 1) / ps.dy;
    if (ps.x0 < W && ps.y0 < H && pw && ph) raw += (stride_of(pw) + 1) * ph;
  }
  std::vector<uint8_t> buf(raw);
  uLongf got = (uLongf)raw;
  if (uncompress(buf.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw) return false;
  px.assign((size_t)W * H, 0u);
  auto sample = [&](const uint8_t* row, uint32_t x, uint32_t c) -> uint32_t {   // channel c of pixel x, scaled to 8 bit like stb (16 -> high byte)
    if (depth == 8) return row[(size_t)x * ch + c];
    if (depth == 16) return row[((size_t)x * ch + c) * 2];
    const uint32_t per = 8 / depth, v = (row[x / per] >> ((per - 1 - x % per) * depth)) & ((1u << depth) - 1u);
    return ctype == 3 ? v : v * 255u / ((1u << depth) - 1u);
  };
  size_t off = 0;
  for (const Pass& ps : passes) {
    if (ps.x0 >= W || ps.y0 >= H) continue;
    const uint32_t pw = (W - ps.x0 + ps.dx - 1) / ps.dx, ph = (H - ps.y0 + ps.dy - 1) / ps.dy;
    if (!pw || !ph) continue;
    const size_t st = stride_of(pw);
    std::vector<uint8_t> img(st * ph);
    if (!png_unfilter(buf.data() + off, ph, st, bpp, img.data())) return false;
    off += (st + 1) * ph;
    for (uint32_t y = 0; y < ph; ++y) {
      const uint8_t* row = img.data() + (size_t)y * st;
      for (uint32_t x = 0; x < pw; ++x) {
        uint32_t r, g, b;
        if (ctype == 3) {
          const uint32_t i = sample(row, x, 0);
          if ((size_t)i * 3 + 2 >= plte.size()) return false;
          r = plte[i * 3]; g = plte[i * 3 + 1]; b = plte[i * 3 + 2];
        } else if (ctype == 0 || ctype == 4) { r = g = b = sample(row, x, 0); }
        else { r = sample(row, x, 0); g = sample(row, x, 1); b = sample(row, x, 2); }
        px[(size_t)(ps.y0 + y * ps.dy) * W + ps.x0 + x * ps.dx] = (r << 16) + (g << 8) + b;   // surface.cpp:47
      }
    }
  }
  return true;
}

static bool load_pnm(const std::vector<uint8_t>& f, uint32_t& W, uint32_t& H, std::vector<uint32_t>& px) {
  if (f.size() < 7 || f[0] != 'P') return false;
  const int kind = f[1] - '0';
  if (kind != 2 && kind != 3 && kind != 5 && kind != 6) return false;
  size_t pos = 2;
  auto next_int = [&](uint32_t& v) -> bool {
    for (;;) {
      while (pos < f.size() && std::isspace(f[pos])) ++pos;
      if (pos < f.size() && f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') ++pos; continue; }
      break;
    }
    if (pos >= f.size() || !std::isdigit(f[pos])) return false;
    uint64_t a = 0;
    while (pos < f.size() && std::isdigit(f[pos])) { a = a * 10 + (f[pos++] - '0'); if (a > 0xffffffffull) return false; }
    v = (uint32_t)a;
    return true;
  };
  uint32_t maxv = 0;
  if (!next_int(W) || !next_int(H) || !next_int(maxv) || W == 0 || H == 0 || maxv == 0 || maxv > 65535 || (uint64_t)W * H > (1ull << 28)) return false;
  const uint32_t ch = (kind == 3 || kind == 6) ? 3 : 1;
  px.assign((size_t)W * H, 0u);
  auto scale = [&](uint32_t v) { return maxv == 255 ? v : v * 255u / maxv; };
  if (kind == 5 || kind == 6) {
    ++pos;   // the single whitespace after maxval
    const size_t bps = maxv > 255 ? 2 : 1;
    if (pos + (size_t)W * H * ch * bps > f.size()) return false;
    for (size_t i = 0; i < (size_t)W * H; ++i) {
      uint32_t c[3];
      for (uint32_t k = 0; k < ch; ++k) { const uint8_t* q = &f[pos + (i * ch + k) * bps]; c[k] = scale(bps == 2 ? ((uint32_t)q[0] << 8) | q[1] : q[0]); }
      if (ch == 1) c[1] = c[2] = c[0];
      px[i] = (c[0] << 16) + (c[1] << 8) + c[2];
    }
  } else {
    for (size_t i = 0; i < (size_t)W * H; ++i) {
      uint32_t c[3];
      for (uint32_t k = 0; k < ch; ++k) { if (!next_int(c[k])) return false; c[k] = scale(std::min(c[k], maxv)); }
      if (ch == 1) c[1] = c[2] = c[0];
      px[i] = (c[0] << 16) + (c[1] << 8) + c[2];
    }
  }
  return true;
}

// file -> 0x00RRGGBB texels, row-major, top row first
static bool load_image(const char* path, uint32_t& W, uint32_t& H, std::vector<uint32_t>& px) {
  std::ifstream in(path, std::ios::binary);
  if (!in) return false;
  std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  return load_png(f, W, H, px) || load_pnm(f, W, H, px);
}

// OBJ + MTL reader: v / vn / vt / f (fan-triangulated) / usemtl / mtllib with Kd, Ka and map_Kd (PNG, PPM/PGM)
bool load_obj(const char* path, Mesh& m) {
  std::ifstream in(path);
  if (!in) return false;
  std::vector<V3> P, N; std::vector<std::pair<float, float>> T;
  std::map<std::string, uint32_t> matid;
  std::map<std::string, int> texid;
  uint32_t cur = 0;
  std::string line, dir(path);
  auto slash = dir.find_last_of('/');
  dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string k; ss >> k;
    if (k == "v") { V3 p; ss >> p.x >> p.y >> p.z; P.push_back(p); }
    else if (k == "vn") { V3 p; ss >> p.x >> p.y >> p.z; N.push_back(p); }
    else if (k == "vt") { float u = 0, v = 0; ss >> u >> v; T.push_back({u, v}); }
    else if (k == "mtllib") {
      std::string f; ss >> f;
      std::ifstream mi(dir + f);
      std::string ml, name;
      while (std::getline(mi, ml)) {
        std::istringstream ms(ml); std::string mk; ms >> mk;
        if (mk == "newmtl") { ms >> name; matid[name] = (uint32_t)m.mats.size(); m.mats.push_back(make_mat(0.8f, 0.8f, 0.8f, -1)); }
        else if (mk == "Kd" && !m.mats.empty()) { ms >> m.mats.back().diffuse[0] >> m.mats.back().diffuse[1] >> m.mats.back().diffuse[2]; }
        else if (mk == "Ka" && !m.mats.empty()) { ms >> m.mats.back().ambient[0] >> m.mats.back().ambient[1] >> m.mats.back().ambient[2]; }
        else if (mk == "map_Kd" && !m.mats.empty()) {
          // mesh.cpp:130-293 keeps one texture per distinct file; a file that cannot be read leaves the material untextured
          std::string tf, tok;
          while (ms >> tok) tf = tok;          // options (-s, -o ...) precede the file name
          std::replace(tf.begin(), tf.end(), '\\', '/');
          auto it = texid.find(tf);
          if (it == texid.end()) {
            uint32_t tw = 0, th = 0; std::vector<uint32_t> px;
            if (load_image((dir + tf).c_str(), tw, th, px)) {
              it = texid.emplace(tf, (int)m.textures.size()).first;
              m.textures.push_back(std::move(px));
              m.tex_dims.push_back({tw, th});
            } else {
              it = texid.emplace(tf, -1).first;
              std::fprintf(stderr, "[vxs] cannot read texture %s\n", (dir + tf).c_str());
            }
          }
          m.mats.back().diffuse_tex_id = it->second;
        }
      }
    } else if (k == "usemtl") { std::string n; ss >> n; auto it = matid.find(n); cur = it == matid.end() ? 0 : it->second; }
    else if (k == "f") {
      struct Ix { int p, t, n; };
      std::vector<Ix> ix; std::string tok;
      while (ss >> tok) {
        Ix i{0, 0, 0};
        if (std::sscanf(tok.c_str(), "%d/%d/%d", &i.p, &i.t, &i.n) == 3) {}
        else if (std::sscanf(tok.c_str(), "%d//%d", &i.p, &i.n) == 2) { i.t = 0; }
        else if (std::sscanf(tok.c_str(), "%d/%d", &i.p, &i.t) == 2) { i.n = 0; }
        else { std::sscanf(tok.c_str(), "%d", &i.p); }
        ix.push_back(i);
      }
      auto P_ = [&](int i) { return i > 0 ? P[i - 1] : P[P.size() + i]; };
      auto N_ = [&](int i) { return i == 0 ? V3{0, 0, 0} : (i > 0 ? N[i - 1] : N[N.size() + i]); };
      auto T_ = [&](int i) { return i == 0 ? std::pair<float, float>{0, 0} : (i > 0 ? T[i - 1] : T[T.size() + i]); };
      for (size_t j = 1; j + 1 < ix.size(); ++j) {
        auto t0 = T_(ix[0].t), t1 = T_(ix[j].t), t2 = T_(ix[j + 1].t);
        add_tri(m, P_(ix[0].p), P_(ix[j].p), P_(ix[j + 1].p), N_(ix[0].n), N_(ix[j].n), N_(ix[j + 1].n),
                t0.first, t0.second, t1.first, t1.second, t2.first, t2.second, cur);
      }
    }
  }
  return !m.tri.empty();
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

// name: "cornell" | "blob" (a = icosphere subdivisions) | "atrium" (a = level, 8 -> 1,048,576 tris)
//       | "hairball" (a = strands, b = segments per strand) | "bunny" / "hairball_fill": the same two objects placed so that
//       they fill the fixed camera's view (BASELINE configs 2 and 5 mean a framed object, not one in the distance)
static void read_knobs() {
  if (const char* e = std::getenv("VXS_BINS")) { int v = std::atoi(e); if (v >= 2 && v <= kMaxBins) kBins = v; }
  if (const char* e = std::getenv("VXS_WIDEN")) kWiden = std::atoi(e);
  if (const char* e = std::getenv("VXS_COLLAPSE")) kCollapse = std::atoi(e);
  if (const char* e = std::getenv("VXS_OPTIMIZE")) kOptimize = std::atoi(e);
  if (const char* e = std::getenv("VXS_VERBOSE")) kVerbose = std::atoi(e);
  if (const char* e = std::getenv("VXS_TRI_COST")) kTriCost = (float)std::atof(e);
  if (const char* e = std::getenv("VXS_OPTIMIZE_LOCAL")) kOptimizeLocal = std::atoi(e);
  if (const char* e = std::getenv("VXS_OPTIMIZE_FRACTION")) kOptimizeFraction = std::atof(e);
  { unsigned hc = std::thread::hardware_concurrency(); kThreads = (int)std::min<unsigned>(hc ? hc : 1u, 16u); }
  { const unsigned hc = std::thread::hardware_concurrency(); kThreads = (int)std::min(16u, std::max(1u, hc)); }   // (a GPU box gives a process 16 cores per GPU)
  if (const char* e = std::getenv("VXS_THREADS")) { int v = std::atoi(e); if (v >= 1 && v <= 64) kThreads = v; }
  if (const char* e = std::getenv("VXS_LEAF_K")) kLeafK = (float)std::atof(e);
  if (const char* e = std::getenv("VXS_LEAF_MAX")) kLeafMax = std::atoi(e);
  if (const char* e = std::getenv("VXS_CHILD_ORDER")) kChildOrder = std::atoi(e);
}

// decode an image file the way the scene ingest does (PNG, PPM/PGM -> 0x00RRGGBB); out may be NULL to query the size
int vxs_image_load(const char* path, uint32_t* w, uint32_t* h, uint32_t* out, uint64_t cap_pixels) {
  uint32_t W = 0, H = 0; std::vector<uint32_t> px;
  if (!path || !w || !h || !load_image(path, W, H, px)) return -1;
  *w = W; *h = H;
  if (out) {
    if (cap_pixels < px.size()) return -1;
    std::memcpy(out, px.data(), px.size() * 4);
  }
  return 0;
}

void* vxs_scene_create_procedural(const char* name, uint32_t a, uint32_t b, uint32_t seed) {
  read_knobs();
  g_t0 = std::chrono::steady_clock::now();
  std::vector<Mesh> meshes(1);
  std::string n(name ? name : "");
  if (n == "cornell") meshes[0] = make_cornell();
  else if (n == "blob") meshes[0] = make_blob(a, seed);
  else if (n == "atrium") meshes[0] = make_atrium(a, seed);
  else if (n == "hairball") meshes[0] = make_hairball(a, b, seed);
  else if (n == "bunny") meshes[0] = make_blob(a, seed, 135.0f);                 // the blob framed to fill the view
  else if (n == "hairball_fill") meshes[0] = make_hairball(a, b, seed, 150.0f);  // the hairball framed to fill the view
  else return nullptr;
  return build_scene(meshes);
}

// general entry: n_meshes meshes; mesh i has ntris[i] triangles (9 floats each) starting at
// tris + 9*sum(ntris[0..i)), optional triEx (64 B each, same order), optional per-mesh 4x4
// row-major transforms, optional materials (88 B each; per-mesh counts nmats[i], texId is
// mesh-local as in mesh.cpp) and one shared texture blob is not supported here (use procedural).
void* vxs_scene_create_from_tris(uint32_t n_meshes, const uint32_t* ntris, const float* tris, const void* triEx,
                                 const float* transforms, const uint32_t* nmats, const void* mats) {
  if (!n_meshes || !ntris || !tris) return nullptr;
  std::vector<Mesh> meshes(n_meshes);
  size_t off = 0, moff = 0;
  for (uint32_t i = 0; i < n_meshes; ++i) {
    Mesh& m = meshes[i];
    if (ntris[i] == 0) return nullptr;
    m.tri.resize(ntris[i]);
    std::memcpy(m.tri.data(), tris + off * 9, (size_t)ntris[i] * sizeof(rt_tri_t));
    m.triEx.resize(ntris[i]);
    if (triEx) std::memcpy(m.triEx.data(), (const rt_triex_t*)triEx + off, (size_t)ntris[i] * sizeof(rt_triex_t));
    else {
      for (uint32_t j = 0; j < ntris[i]; ++j) {
        rt_triex_t e{};
        V3 nn = normalize(cross(tv(m.tri[j].v1) - tv(m.tri[j].v0), tv(m.tri[j].v2) - tv(m.tri[j].v0)));
        for (float* p : {e.N0, e.N1, e.N2}) { p[0] = nn.x; p[1] = nn.y; p[2] = nn.z; }
        e.uv1[0] = 1; e.uv2[1] = 1;
        m.triEx[j] = e;
      }
    }
    if (transforms) std::memcpy(m.transform, transforms + 16 * i, sizeof m.transform);
    if (nmats && mats) {
      m.mats.resize(nmats[i]);
      std::memcpy(m.mats.data(), (const rt_material_t*)mats + moff, (size_t)nmats[i] * sizeof(rt_material_t));
      for (auto& mm : m.mats) mm.diffuse_tex_id = -1;
      moff += nmats[i];
    } else {
      m.mats = {make_mat(0.8f, 0.8f, 0.8f, -1)};
    }
    off += ntris[i];
  }
  return build_scene(meshes);
}

// instances: like the reference's `-n mesh_count` (tracer.cpp:92-98): the same model n times, placed
// on a circle around Y (scene.cpp:214-250) when n > 1
void* vxs_scene_load_obj(const char* path, uint32_t instances) {
  if (!path || instances == 0) return nullptr;
  Mesh base;
  if (!load_obj(path, base)) return nullptr;
  if (base.mats.empty()) base.mats = {make_mat(0.8f, 0.8f, 0.8f, -1)};
  std::vector<Mesh> meshes(instances, base);
  if (instances > 1) {
    Box bb;
    for (auto& t : base.tri) { bb.grow(tv(t.v0)); bb.grow(tv(t.v1)); bb.grow(tv(t.v2)); }
    const float dx = bb.hi.x - bb.lo.x, dz = bb.hi.z - bb.lo.z;
    const float radius = 0.5f * std::sqrt(dx * dx + dz * dz);
    const float step = 2.0f * 3.14159265358979f / (float)instances;
    const float R = (2 * radius) / (2.0f * std::sin(step / 2.0f));
    for (uint32_t i = 0; i < instances; ++i) {
      meshes[i].transform[3] = R * std::cos(step * i);
      meshes[i].transform[11] = R * std::sin(step * i);
    }
  }
  return build_scene(meshes);
}

void vxs_scene_destroy(void* h) { delete (Scene*)h; }

// software-twin scenes (raycast formats).  name / a / b / seed as vxs_scene_create_procedural; `copies` instances of the
// mesh side by side along z, reflectivity[i] per instance (may be NULL)
void* vxs_rc_scene_create_procedural(const char* name, uint32_t a, uint32_t b, uint32_t seed, uint32_t copies, const float* reflectivity) {
  read_knobs();
  std::string n(name ? name : "");
  Mesh m0;
  if (n == "cornell") m0 = make_cornell();
  else if (n == "blob") m0 = make_blob(a, seed);
  else if (n == "atrium") m0 = make_atrium(a, seed);
  else if (n == "hairball") m0 = make_hairball(a, b, seed);
  else if (n == "bunny") m0 = make_blob(a, seed, 135.0f);
  else if (n == "hairball_fill") m0 = make_hairball(a, b, seed, 150.0f);
  else return nullptr;
  if (copies == 0) copies = 1;
  Box mb;
  for (const rt_tri_t& t : m0.tri) { mb.grow(tv(t.v0)); mb.grow(tv(t.v1)); mb.grow(tv(t.v2)); }
  std::vector<Mesh> meshes(copies, m0);
  for (uint32_t i = 0; i < copies; ++i) {
    if (m0.textures.size() > 1) {   // vary the instance texture
      meshes[i].textures[0] = m0.textures[i % m0.textures.size()];
      meshes[i].tex_dims[0] = m0.tex_dims[i % m0.tex_dims.size()];
    }
    meshes[i].transform[11] = ((float)i - 0.5f * (float)(copies - 1)) * (mb.hi.z - mb.lo.z) * 1.15f;   // row-major: translation in column 3
  }
  return build_rc_scene(meshes, reflectivity);
}
void vxs_rc_scene_destroy(void* h) { delete (RcScene*)h; }
// which: 0 tlas 1 blas 2 bvh 3 tri 4 triEx 5 triIdx 6 tex ; returns bytes
uint64_t vxs_rc_scene_buffer(void* h, int which, const void** ptr) {
  auto s = (RcScene*)h;
  if (!s || !ptr) return 0;
  switch (which) {
  case 0: *ptr = s->tlas.data(); return s->tlas.size() * sizeof(rc_tlas_node_t);
  case 1: *ptr = s->blas.data(); return s->blas.size() * sizeof(rc_blas_t);
  case 2: *ptr = s->bvh.data(); return s->bvh.size() * sizeof(rc_bvh_node_t);
  case 3: *ptr = s->tri.data(); return s->tri.size() * sizeof(rt_tri_t);
  case 4: *ptr = s->triEx.data(); return s->triEx.size() * sizeof(rc_triex_t);
  case 5: *ptr = s->triIdx.data(); return s->triIdx.size() * sizeof(uint32_t);
  case 6: *ptr = s->tex.data(); return s->tex.size();
  }
  *ptr = nullptr;
  return 0;
}
void vxs_rc_scene_info(void* h, uint32_t* out2, float* bounds6) {
  auto s = (RcScene*)h;
  if (out2) { out2[0] = s->tlas_root; out2[1] = s->max_depth; }
  if (bounds6) std::memcpy(bounds6, s->bounds, sizeof s->bounds);
}

// which: 0 tlas 1 blas 2 bvh 3 tri 4 triEx 5 mat 6 tex 7 triIdx ; returns bytes
uint64_t vxs_scene_buffer(void* h, int which, const void** ptr) {
  auto s = (Scene*)h;
  if (!s || !ptr) return 0;
  switch (which) {
  case 0: *ptr = s->tlas.data(); return s->tlas.size() * sizeof(rt_qnode_t);
  case 1: *ptr = s->blas.data(); return s->blas.size() * sizeof(rt_blas_t);
  case 2: *ptr = s->bvh.data(); return s->bvh.size() * sizeof(rt_qnode_t);
  case 3: *ptr = s->tri.data(); return s->tri.size() * sizeof(rt_tri_t);
  case 4: *ptr = s->triEx.data(); return s->triEx.size() * sizeof(rt_triex_t);
  case 5: *ptr = s->mat.data(); return s->mat.size() * sizeof(rt_material_t);
  case 6: *ptr = s->tex.data(); return s->tex.size();
  case 7: *ptr = s->triIdx.data(); return s->triIdx.size() * sizeof(uint32_t);
  }
  *ptr = nullptr;
  return 0;
}

// out: [0] max_depth (levels, as the reference's trail counts them) [1] leaves [2] max tris/leaf
//      [3] bvh nodes [4] tlas nodes [5] tris ; bounds6: world AABB
void vxs_scene_info(void* h, uint32_t* out6, float* bounds6) {
  auto s = (Scene*)h;
  if (out6) {
    out6[0] = s->max_depth; out6[1] = s->n_leaves; out6[2] = s->max_leaf;
    out6[3] = (uint32_t)s->bvh.size(); out6[4] = (uint32_t)s->tlas.size(); out6[5] = (uint32_t)s->tri.size();
  }
  if (bounds6) std::memcpy(bounds6, s->bounds, sizeof s->bounds);
}

}  // extern "C"
