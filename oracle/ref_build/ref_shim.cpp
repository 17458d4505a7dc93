// TEST INFRASTRUCTURE ONLY (oracle/_ref): thin driver around the *reference's own*
// BVHTraverser, compiled from /root/reference/sim/simx/rt_traversal.cpp where it lies.
// Nothing here is shipped or timed as product; see oracle/README.md.
//
// The only project symbol rt_traversal.cpp needs is RTUnit::dcache_read
// (sim/simx/rt_unit.cpp:42-44 forwards it to Core::dcache_read).  We back it by a flat
// byte image, fill DCRs 0x6..0x9 the way tests/regression/raytracing/tracer.cpp:252-256
// does, and run the accept loop of rt_unit.cpp:98-116,199-202 (any-hit shader always
// commits ACCEPT: shaders/anyhit.cpp:34).
#include "rt_traversal.h"
#include "rt_unit.h"
#include <cstring>
#include <cstdint>

using namespace vortex;

static thread_local const uint8_t* g_image = nullptr;
static thread_local uint64_t g_image_size = 0;
static thread_local uint64_t g_oob = 0;

void RTUnit::dcache_read(void* data, uint64_t addr, uint32_t size) {
  if (addr + size > g_image_size) {  // reference would read simulated RAM; flag it instead
    memset(data, 0, size);
    ++g_oob;
    return;
  }
  memcpy(data, g_image + addr, size);
}

extern "C" {

struct vxref_hit_t {
  float dist, bx, by, bz;
  uint32_t blasIdx, triIdx;
};

struct vxref_stats_t {
  uint64_t node_reads, inst_reads, tri_reads, bytes, accepts, oob;
};

// rays: n x 6 floats (o.xyz, d.xyz). any_hit_first!=0: stop at the first accepted
// candidate (COMMIT_ACCEPT followed by COMMIT_TERM) -- used for occlusion goldens.
int vxref_trace(const uint8_t* image, uint64_t image_size,
                uint32_t tlas_off, uint32_t blas_off, uint32_t bvh_off, uint32_t tri_off,
                const float* rays, uint64_t n, vxref_hit_t* out, vxref_stats_t* stats,
                int any_hit_first) {
  g_image = image;
  g_image_size = image_size;
  g_oob = 0;
  DCRS dcrs;
  dcrs.base_dcrs.write(VX_DCR_BASE_RTX_TLAS_PTR, tlas_off);
  dcrs.base_dcrs.write(VX_DCR_BASE_RTX_BLAS_PTR, blas_off);
  dcrs.base_dcrs.write(VX_DCR_BASE_RTX_BVH_PTR, bvh_off);
  dcrs.base_dcrs.write(VX_DCR_BASE_RTX_TRI_PTR, tri_off);
  BVHTraverser trav(nullptr, dcrs);
  vxref_stats_t st{};
  for (uint64_t i = 0; i < n; ++i) {
    Ray r{rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2],
          rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
    Hit h;
    h.pending_dist = 0; h.bx = h.by = h.bz = 0; h.blasIdx = 0; h.triIdx = 0;
    TraversalTrail trail{};
    TraversalStack stack;
    per_thread_info ti;
    for (;;) {
      bool done = trav.traverse(r, h, trail, stack, ti);
      for (auto& a : ti.RT_mem_accesses) {
        st.bytes += a.size;
        if (a.type == TransactionType::BVH_INTERNAL_NODE) st.node_reads++;
        else if (a.type == TransactionType::BVH_INSTANCE_LEAF) st.inst_reads++;
        else st.tri_reads++;
      }
      ti.clear_mem_accesses();
      if (done) break;
      h.dist = h.pending_dist;  // COMMIT_ACCEPT
      st.accepts++;
      if (any_hit_first) break;
    }
    out[i] = {h.dist, h.bx, h.by, h.bz, h.blasIdx, h.triIdx};
  }
  st.oob = g_oob;
  if (stats) *stats = st;
  return 0;
}

}  // extern "C"
