/*
 * include/vortex_hip.h -- C ABI of the MI355X-native ray-tracing hot path.
 *
 * Two levels, both exported by libvortex-hip.so (plain pointers and sizes, no torch types):
 *
 *  (1) DROP-IN DRIVER BOUNDARY.  `vx_dev_init(callbacks_t*)` is the one symbol the reference's
 *      dispatcher resolves with dlsym after dlopen("libvortex-$VORTEX_DRIVER.so")
 *      (reference runtime/stub/vortex.cpp:58-82).  It fills the 16 entry points declared in
 *      reference runtime/common/callbacks.h:23-72; each one replaces the simx implementation in
 *      reference runtime/simx/vortex.cpp + runtime/common/callbacks.inc (lines cited per member).
 *      With VORTEX_DRIVER=hip the unmodified host code of tests/regression/raytracing keeps
 *      calling vx_mem_alloc / vx_copy_to_dev / vx_dcr_write / vx_start / vx_ready_wait.
 *
 *  (2) DIRECT LAUNCH API `vxrt_*` for callers that already own device memory and a HIP stream
 *      (bench.py, tests, embedding in another runtime).  `start` of level (1) is implemented on
 *      top of it.  All pointers in vxrt_scene_t / rays / hits / dst are DEVICE pointers; `stream`
 *      is a hipStream_t passed as void* (NULL = default stream).  Launches are asynchronous.
 *
 * Error convention of the reference (callbacks.inc): 0 = success, non-zero (-1) = failure; no
 * exceptions cross this ABI.
 */
#ifndef VORTEX_HIP_H
#define VORTEX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * (1) driver boundary -- mirrors reference runtime/include/vortex.h:27-28 and
 *     runtime/common/callbacks.h:23-72 (same member order; the struct is filled positionally
 *     by the backend and read by the dispatcher, so the order IS the ABI).
 * ---------------------------------------------------------------------------------------- */
typedef void* vx_device_h;
typedef void* vx_buffer_h;

typedef struct {
  int (*dev_open)(vx_device_h* hdevice);                                   /* callbacks.inc:24-36 / simx vortex.cpp:49-78 */
  int (*dev_close)(vx_device_h hdevice);                                   /* callbacks.inc:38-45 */
  int (*dev_caps)(vx_device_h hdevice, uint32_t caps_id, uint64_t* value); /* callbacks.inc:47-58 / vortex.cpp:80-135 */
  int (*mem_alloc)(vx_device_h hdevice, uint64_t size, int flags, vx_buffer_h* hbuffer);   /* callbacks.inc:60-78 / vortex.cpp:209-232 */
  int (*mem_reserve)(vx_device_h hdevice, uint64_t address, uint64_t size, int flags, vx_buffer_h* hbuffer); /* :80-97 / :234-249 */
  int (*mem_free)(vx_buffer_h hbuffer);                                    /* callbacks.inc:99-108 */
  int (*mem_access)(vx_buffer_h hbuffer, uint64_t offset, uint64_t size, int flags);       /* :110-119 */
  int (*mem_address)(vx_buffer_h hbuffer, uint64_t* address);              /* :121-128 */
  int (*mem_info)(vx_device_h hdevice, uint64_t* mem_free, uint64_t* mem_used);            /* :130-146 */
  int (*copy_to_dev)(vx_buffer_h hbuffer, const void* host_ptr, uint64_t dst_offset, uint64_t size);   /* :148-158 / vortex.cpp:276-304 */
  int (*copy_from_dev)(void* host_ptr, vx_buffer_h hbuffer, uint64_t src_offset, uint64_t size);       /* :160-170 / vortex.cpp:306-327 */
  int (*start)(vx_device_h hdevice, vx_buffer_h hkernel, vx_buffer_h harguments);          /* :172-180 / vortex.cpp:329-348 */
  int (*ready_wait)(vx_device_h hdevice, uint64_t timeout);                /* :182-188 / vortex.cpp:350-364 */
  int (*dcr_read)(vx_device_h hdevice, uint32_t addr, uint32_t* value);    /* :190-201 / vortex.cpp:375-377 */
  int (*dcr_write)(vx_device_h hdevice, uint32_t addr, uint32_t value);    /* :203-209 / vortex.cpp:366-373 */
  int (*mpm_query)(vx_device_h hdevice, uint32_t addr, uint32_t core_id, uint64_t* value); /* :211-222 / vortex.cpp:379-391 */
} callbacks_t;

/* The plug-in entry (callbacks.inc:20). */
int vx_dev_init(callbacks_t* callbacks);

/* caps ids (vortex.h:31-45) */
#define VX_CAPS_VERSION 0x0
#define VX_CAPS_NUM_THREADS 0x1
#define VX_CAPS_NUM_WARPS 0x2
#define VX_CAPS_NUM_CORES 0x3
#define VX_CAPS_CACHE_LINE_SIZE 0x4
#define VX_CAPS_GLOBAL_MEM_SIZE 0x5
#define VX_CAPS_LOCAL_MEM_SIZE 0x6
#define VX_CAPS_ISA_FLAGS 0x7
#define VX_CAPS_NUM_MEM_BANKS 0x8
#define VX_CAPS_MEM_BANK_SIZE 0x9
#define VX_CAPS_NUM_CLUSTERS 0xA
#define VX_CAPS_SOCKET_SIZE 0xB
#define VX_CAPS_ISSUE_WIDTH 0xC
#define VX_CAPS_CLOCK_RATE 0xD
#define VX_CAPS_PEAK_MEM_BW 0xE

#define VX_MEM_READ 0x1
#define VX_MEM_WRITE 0x2
#define VX_MEM_READ_WRITE 0x3
#define VX_MAX_TIMEOUT (24 * 60 * 60 * 1000)

/* DCR ids the RTU path uses (hw/VX_types.toml:16-19; written by tracer.cpp:252-256). */
#define VX_DCR_BASE_STARTUP_ADDR0 0x001
#define VX_DCR_BASE_STARTUP_ADDR1 0x002
#define VX_DCR_BASE_STARTUP_ARG0 0x003
#define VX_DCR_BASE_STARTUP_ARG1 0x004
#define VX_DCR_BASE_MPM_CLASS 0x005
#define VX_DCR_BASE_RTX_TLAS_PTR 0x006
#define VX_DCR_BASE_RTX_BLAS_PTR 0x007
#define VX_DCR_BASE_RTX_BVH_PTR 0x008
#define VX_DCR_BASE_RTX_TRI_PTR 0x009
/* Backend extension (no reference counterpart): framebuffer row window rendered by this device,
 * for sharding one frame over several GPUs/processes.  [begin,end); end==0 means dst_height. */
#define VX_DCR_HIP_ROW_BEGIN 0x7F0
#define VX_DCR_HIP_ROW_END 0x7F1
/* Backend extension: 0 = closest-hit only (reference behaviour), 1 = +1 shadow ray toward light_pos. */
#define VX_DCR_HIP_SHADOW_RAYS 0x7F2
/* Backend extension: tile-row stride of the row window.  0/1 = the contiguous rows [ROW_BEGIN, ROW_END); n > 1 = this device
 * renders the 8-row tile rows phase, phase + n, phase + 2n, ... of the frame with phase = ROW_BEGIN / 8 (< n): the split of one
 * frame over n GPUs that balances them (see vxrt_render_interleaved). */
#define VX_DCR_HIP_ROW_STRIDE 0x7F3
/* Backend extension: 1 = REFERENCE-QUIRKS traversal for the RTU test's frames (default 0 = the canonical algorithm).  The frame's
 * camera rays are then traced by vxrt_trace_reference_quirks on a flat image of the device's address space -- every allocation at
 * its address, as the simulator's RAM holds it -- with the RTX DCR values as base pointers, so that the stale base_ptr of
 * rt_traversal.cpp:91-92 reads here what it reads there.  Slow (one thread per ray, the address space copied when an allocation
 * changed); closest-hit frames only (no shadow-ray extension, no row stride). */
#define VX_DCR_HIP_REFERENCE_QUIRKS 0x7F4

/* CSRs answered by mpm_query (read by vx_dump_perf at vx_dev_close: stub/perf.cpp:195-227). */
#define VX_CSR_MPM_BASE 0xB00
#define VX_CSR_MCYCLE 0xB00
#define VX_CSR_MINSTRET 0xB02

/* kernel_arg_t of the RTU test (tests/regression/raytracing/common.h:164-195): 216 bytes.  This
 * is what the host uploads with vx_upload_bytes and what `start` decodes. */
typedef struct {
  uint32_t dst_width;
  uint32_t dst_height;
  uint64_t dst_addr;
  uint64_t tri_addr;
  uint64_t triEx_addr;
  uint64_t triIdx_addr;
  uint64_t mat_addr;
  uint64_t tex_addr;
  uint64_t bvh_addr;
  uint64_t qBvh_addr;
  uint64_t blas_addr;
  uint64_t tlas_addr;
  uint32_t tlas_root;
  float camera_pos[3];
  float camera_forward[3];
  float camera_right[3];
  float camera_up[3];
  float viewplane[2];
  uint32_t samples_per_pixel;
  uint32_t max_depth;
  float light_pos[3];
  float light_color[3];
  float ambient_color[3];
  float background_color[3];
  uint64_t sbt_addr;
} vx_rt_kernel_arg_t;

/* ------------------------------------------------------------------------------------------
 * (2) direct launch API
 * ---------------------------------------------------------------------------------------- */

/* Device-resident scene in the reference's own buffer formats:
 *   tlas, bvh : bvh_quantized_node_t[]  52 B   (common.h:52-67 == sim rt_traversal.h:14-33)
 *   blas      : blas_node_t[]          160 B   (common.h:86-99)
 *   tri       : tri_t[]                 36 B   (geometry.h:1401-1405)
 *   triEx     : tri_ex_t[]              64 B   (common.h:39-43)
 *   mat       : material_info_t[]       88 B   (common.h:20-36)
 *   tex       : 0x00RRGGBB u32 texels, concatenated (surface.cpp:28-55)
 * Counts are element counts and are used for host-side shape checks before every launch. */
typedef struct {
  const void* tlas;
  const void* blas;
  const void* bvh;
  const void* tri;
  const void* triEx;
  const void* mat;
  const void* tex;
  uint32_t n_tlas_nodes;
  uint32_t n_blas;
  uint32_t n_bvh_nodes;
  uint32_t n_tris;
  uint32_t n_mats;
  uint32_t reserved;
  uint64_t tex_bytes;
} vxrt_scene_t;

/* Hit record: the 6 RTU hit attributes (VX_RT_HIT_DIST..TRI_IDX, hw/VX_types.toml:270-285;
 * sim rt_traversal.h:48-52).  dist == 1e30f means miss. */
typedef struct {
  float dist, bx, by, bz;
  uint32_t blasIdx, triIdx;
} vxrt_hit_t;

typedef struct {
  float ambient[3], light_color[3], light_pos[3], background[3];
  uint32_t max_depth;
} vxrt_shade_params_t;

#define VXRT_MODE_CLOSEST 0 /* reference semantics: global closest hit, reference tie order */
#define VXRT_MODE_ANY 1     /* occlusion: stop at first accepted candidate (extension) */

/* Device-side acceleration layout built ONCE per scene from the reference-format buffers above
 * (compact 64-byte nodes with inlined leaf / instance descriptors, edge-form triangles).  The build
 * validates every index the traversal can follow and fails (-1) on a malformed tree instead of
 * letting a kernel fault; so are the indices shading follows (every triangle's texId < n_mats; a textured material's
 * texels inside [tex, tex + tex_bytes) with non-zero dimensions).  With RT_TOP_NODES > 0 (build flag) the first internal
 * nodes in breadth-first order are also laid out as an image each workgroup stages in LDS.
 * The vxrt_scene_t buffers must stay alive and unchanged while the accel
 * is in use (shading reads blas/triEx/mat/tex from them).  Synchronous with respect to `stream`. */
typedef struct vxrt_accel vxrt_accel_t;
int vxrt_accel_build(const vxrt_scene_t* scene, void* stream, vxrt_accel_t** out);
int vxrt_accel_destroy(vxrt_accel_t* accel);
uint64_t vxrt_accel_bytes(const vxrt_accel_t* accel);
/* What the build found (diagnostic): which = 0 -> internal levels on the longest root-to-leaf path, TLAS and BLAS together, counted up
 * to 17; 1 -> 1 if the scene is at most 16 levels deep and its timed launches keep 48-entry traversal stacks (deeper scenes: the
 * reference's 32 levels, 96 entries + the LDS part); 2 -> 1 if the TLAS root is a single identity instance; 3 -> 1 if the scene
 * takes the ldexp decode / generic slab form. */
int vxrt_accel_info(const vxrt_accel_t* accel, uint32_t which, uint64_t* value);

/* Number of frames (vxrt_render / vxrt_trace calls) this accel keeps in flight, 1..8, default 1.
 * Each in-flight frame has its own hit-record buffer, deferred-ray list and side stream; calls take
 * them round robin, and a call that reuses a context is ordered behind that context's previous
 * call by an event.  With n > 1 and the calls issued on n different streams, the draining tail of
 * one persistent traversal launch overlaps the head of the next frame's (the reference renders one
 * frame per vx_start, tracer.cpp:272-281; this is the knob a frame loop around it would use).
 * Waits for the device to go idle.  Results do not depend on n. */
int vxrt_accel_frames_in_flight(vxrt_accel_t* accel, uint32_t n);

/* Render rows [y0,y1) of the RTU test's frame: camera ray (kernel.cpp:28-39) -> closest hit ->
 * closest/miss shade -> RGB8 pack -> dst[x + y*W] (kernel.cpp:95-106).  `dst` points at pixel
 * (0,0) of the full W x H frame.  shadow != 0 adds one occlusion ray per hit (extension).
 * rays_traced (device u64, may be NULL) is atomically incremented by the number of rays traced.
 * Two launches on `stream`: the persistent traversal kernel leaves 24-byte hit records (tile-major, one
 * contiguous 1,536-byte block per 8x8 tile, written once) in a buffer owned by the frame context, the shading
 * kernel turns them into pixels.
 * hits (optional): the pixel's closest-hit record in pixel order; with shadow != 0, bit 31 of blasIdx is set
 * when the pixel's occlusion ray was blocked (mask it off to compare with a closest-hit record). */
int vxrt_render(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                const vxrt_shade_params_t* params, int shadow, uint32_t* dst,
                vxrt_hit_t* hits /* optional, W*H */, float* colors /* optional, 3*W*H */,
                unsigned long long* rays_traced, void* stream);

/* One rank's share of a frame split over `stride` devices: the 8-row tile rows phase, phase + stride, ... (phase < stride).
 * Same contract as vxrt_render otherwise (dst points at pixel (0,0) of the full frame and only this share's rows are
 * written).  Interleaving balances the ranks: the cost of a tile varies 4x over the frame, mostly with image height. */
int vxrt_render_interleaved(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t phase, uint32_t stride,
                            const vxrt_shade_params_t* params, int shadow, uint32_t* dst, vxrt_hit_t* hits,
                            float* colors, unsigned long long* rays_traced, void* stream);

/* BLAS construction on the GPU, in the reference's formats.  Replaces, for one mesh, BVH::build + the 4-wide collapse + the
 * quantiser of tests/regression/raytracing/bvh.cpp:30-264 (host code run at scene load in the reference; csrc/scene_builder.cpp
 * is the CPU counterpart here).  Morton order, PLOC clustering (mutual nearest neighbours by surface area of the union), 4-wide
 * collapse by the SAH dynamic programme; see csrc/bvh_builder.hip.
 *   tri      device, n_tris x 36 B (tri_t); REORDERED IN PLACE so that a leaf is a range (bvh.cpp:126-128)
 *   triEx    device, n_tris x 64 B (tri_ex_t), reordered alongside; may be NULL
 *   tri_offset  added to every leaf's leftFirst (index of the mesh's first triangle in the scene's buffer, bvh.cpp:260)
 *   leaf_max    largest leaf, 1..15 (0 = 2); a subtree of <= leaf_max triangles becomes a leaf where the surface-area cost of that
 *               is lower than a node over it (2 / 3 / 4 measured equal within 1 % on the 1M-triangle scene: profiles/r03_t_builder_cost_ab.txt)
 *   nodes    device, node_capacity x 52 B (bvh_quantized_node_t), node_capacity >= 2 * n_tris - 1 (the reference allocates
 *            2 * numTris, scene.cpp:40); node 0 is the root, children follow their parent
 * Synchronises `stream` a few times (the cluster count between groups of clustering rounds, the counts at the end).  Returns 0; -1 on bad arguments, allocation failure or a box that cannot
 * be quantised; -2 if the tree is deeper than the 32 levels the reference's trail supports (use the SAH builder). */
typedef struct vxrt_bvh_info {
  uint32_t n_nodes, n_leaves, max_leaf, max_depth;
  float bounds[6];        /* of the mesh: lo xyz, hi xyz */
} vxrt_bvh_info_t;
int vxrt_bvh_build(void* tri, void* triEx, uint32_t n_tris, uint32_t tri_offset, uint32_t leaf_max,
                   void* nodes, uint32_t node_capacity, vxrt_bvh_info_t* info, void* stream);
/* TLAS over instances on the GPU (reference: BVH::buildTLAS, bvh.cpp:266-421): instance_boxes = device, n_instances x 6 floats
 * (world-space lo xyz, hi xyz of instance i = blas record i); nodes = device, node_capacity >= 2 * n_instances - 1 entries of 52 B;
 * node 0 is the root, a leaf carries leafData = i, internal nodes UINT32_MAX, imask = 1.  Same return codes as vxrt_bvh_build. */
int vxrt_tlas_build(const float* instance_boxes, uint32_t n_instances, void* nodes, uint32_t node_capacity,
                    vxrt_bvh_info_t* info, void* stream);
/* The builder keeps one grow-only scratch allocation per process (about 170 B per triangle of the largest build) so that a
 * mesh rebuilt every frame allocates nothing; this returns it to the device. */
void vxrt_bvh_release_scratch(void);

/* n_frames (1..VXRT_MAX_BATCH) consecutive frames of the same share in ONE set of launches: frame f is lit and shaded with
 * params[f] (an array of n_frames entries: e.g. a moving light) and written to dst + f * dst_frame_stride (in pixels; each frame
 * buffer addressed like vxrt_render_interleaved's).  A rank's share of a frame split N ways is small against the machine and takes
 * as long as its slowest tile however small it is; a sequence of frames traced side by side keeps the GPU full, and the assembly
 * of N shares becomes one collective per batch (bench.py --gpus N).  No optional outputs; scenes without reflective instances. */
#define VXRT_MAX_BATCH 32
int vxrt_render_interleaved_batch(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t phase, uint32_t stride, uint32_t n_frames,
                                  const vxrt_shade_params_t* params, int shadow, uint32_t* dst, uint64_t dst_frame_stride,
                                  unsigned long long* rays_traced, void* stream);

/* diagnostic: the traversal launch of vxrt_render_interleaved_batch in the counting build with vxrt_render_wave_log's per-wavefront
 * log ([15] = 100 MHz clock at which the wavefront found every queue shard empty); pixels are not produced.  tools/wave_balance_batch.py */
int vxrt_render_interleaved_batch_wave_log(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t phase, uint32_t stride, uint32_t n_frames,
                                           const vxrt_shade_params_t* params, int shadow, uint32_t* dst, uint64_t dst_frame_stride,
                                           unsigned long long* counters, unsigned long long* wave_log, void* stream);

/* Image assembly of a frame split by interleaved tile rows (north_star: "RCCL gather only for final image assembly"), the two ends of
 * the wire.  Pixels are 0x00RRGGBB (common.h:149-154): the top byte is always zero, so a share travels as 3 bytes per pixel.
 *   vxrt_wire_pack    rank `rank` of `world`: the tile rows rank, rank + world, ... (tile_rows_per_rank of them, 8 rows each; rows
 *                     past the frame's end are whatever the padded frame buffer holds) of n_frames frames -- frame f at
 *                     frames + f * frame_stride, `width` pixels per row -- packed into `wire`:
 *                     [n_frames][tile_rows_per_rank * 8][width][3] bytes (b, g, r).  width must be a multiple of 4.
 *   vxrt_wire_unpack  rank 0: the `world` shares, rank r's at wire_all + r * wire_stride_bytes, expanded and interleaved into
 *                     n_frames frames of tile_rows_per_rank * world * 8 rows (the padded height the shares were rendered into).
 * One launch each, asynchronous on `stream`; they replace a strided extraction copy and an interleaving copy of 4-byte pixels. */
int vxrt_wire_pack(const uint32_t* frames, uint64_t frame_stride, uint32_t width, uint32_t tile_rows_per_rank, uint32_t world, uint32_t rank,
                   uint32_t n_frames, uint8_t* wire, void* stream);
int vxrt_wire_unpack(const uint8_t* wire_all, uint64_t wire_stride_bytes, uint32_t width, uint32_t tile_rows_per_rank, uint32_t world,
                     uint32_t n_frames, uint32_t* frames, uint64_t frame_stride, void* stream);

/* The same for a contiguous band of rows [y0, y1) (0 <= y0 <= y1 <= height; any row, tiles are counted from y0): what a rank
 * renders when the frame is split into bands whose heights are balanced by cost (bench.py --shard bands).  dst addresses every
 * frame as a FULL frame does -- pixel (x, y) of frame f at dst[f * dst_frame_stride + x + y * width] -- and only rows [y0, y1)
 * are written, so a caller that keeps only its band passes (band buffer - y0 * width) and dst_frame_stride = (y1 - y0) * width:
 * the band then lies contiguous in memory, ready to be sent, and rank 0 receives every band straight into its place in the
 * final image (no extraction and no interleaving copy: the grid being sharded is kernel.cpp:128-133). */
int vxrt_render_rows_batch(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, uint32_t n_frames,
                           const vxrt_shade_params_t* params, int shadow, uint32_t* dst, uint64_t dst_frame_stride,
                           unsigned long long* rays_traced, void* stream);

/* The same for whole frames (one rank): n_frames frames of width x height, frame f lit and shaded with params[f], written to
 * dst + f * dst_frame_stride.  Every wavefront then works through n_frames times as many tiles per launch, so a launch's ramp and
 * tail weigh less: +8 % at 1920x1080 with 5 frames per set of launches. */
int vxrt_render_batch(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t n_frames, const vxrt_shade_params_t* params, int shadow,
                      uint32_t* dst, uint64_t dst_frame_stride, unsigned long long* rays_traced, void* stream);

/* vxrt_render with the fetch counters compiled in (diagnostic build of the same kernel, never
 * timed): counters = device u64[7]: rays, node fetches, instance fetches, triangle fetches,
 * shaded hits, textured hits, pixels written.  Counts are what the reference logs per ray in
 * RT_mem_accesses (rt_traversal.cpp:54,116,148,158) without its restart re-reads. */
int vxrt_render_stats(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                      const vxrt_shade_params_t* params, int shadow, uint32_t* dst,
                      unsigned long long* counters, void* stream);

/* vxrt_render_stats for the traversal the TIMED kernel performs (same counters): a frame's occlusion rays visit children in
 * slot order instead of near-to-far (the result is a boolean) and idle lanes test a leaf's second triangle, so its node /
 * triangle fetch counts differ from the reference-order ones; bench.py reports both. */
int vxrt_render_stats_timed(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                            const vxrt_shade_params_t* params, int shadow, uint32_t* dst,
                            unsigned long long* counters, void* stream);

/* vxrt_trace with the fetch counters compiled in (slower; never a timed path).  counters: device u64[8], of which
 * [0..3] = rays, node fetches, instance fetches, triangle fetches as the reference accounts them (rt_traversal.cpp:
 * 54,116,148,158, without restart re-reads) -- the inputs of the algorithmic bytes per ray. */
int vxrt_trace_stats(vxrt_accel_t* accel, const float* rays, uint64_t n, const float* tmax,
                     vxrt_hit_t* hits, int mode, unsigned long long* counters, void* stream);

/* Ambient-occlusion frame (extension for BASELINE config 5, "16 spp Monte-Carlo AO"; the reference has no
 * such pass, only its RNG is used: common.h:129-147).  Per pixel with a primary hit: spp cosine-weighted
 * occlusion rays about the shading normal (tmax = radius, any-hit), pixel = Lambert colour of the hit
 * (closest.cpp:57-127, else arm) x unoccluded / spp.  The sampling recipe is fixed in
 * oracle/rt_oracle.c:orc_ao_ray and uses IEEE add/mul/div/sqrt only, so the frame is reproducible bit
 * for bit.  unoccluded (optional): device u32 per pixel.  rays_traced counts primary + occlusion rays. */
typedef struct vxrt_ao_params {
  uint32_t spp;      /* occlusion rays per hit, 1..4096 */
  float radius;      /* tmax of the occlusion rays, > 0 */
  uint32_t seed;     /* user seed mixed into the per-sample WangHash seed */
  uint32_t reserved;
} vxrt_ao_params_t;
int vxrt_render_ao(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                   const vxrt_shade_params_t* params, const vxrt_ao_params_t* ao, uint32_t* dst, float* colors,
                   uint32_t* unoccluded, unsigned long long* rays_traced, void* stream);

/* One diffuse bounce (extension for BASELINE config 3, "1 bounce diffuse"; absent from the reference): per pixel with
 * a primary hit one cosine-weighted ray about the shading normal -- the vxrt_render_ao recipe with spp = 1, sample 0,
 * no tmax -- traced for its closest hit; pixel = Lambert colour of the primary hit + albedo * (Lambert colour of the
 * bounce hit | background).  Defined by oracle/rt_oracle.c:orc_render_gi, reproducible bit for bit. */
int vxrt_render_diffuse_bounce(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                               const vxrt_shade_params_t* params, uint32_t seed, uint32_t* dst, float* colors,
                               unsigned long long* rays_traced, void* stream);

/* ---- software twin: the reference's raycast test (tests/regression/raycast) ----
 * Buffers in the reference's formats (raycast/common.h): tlas_node_t 32 B, blas_node_t 160 B (transform,
 * invTransform, bvh_offset@128, tex_offset@136, tex_width@144, tex_height@148, reflectivity@152),
 * bvh_node_t 32 B, tri_t 36 B, tri_ex_t 60 B, triIdx u32, 0x00RRGGBB texels.  Device pointers. */
typedef struct vxrc_scene {
  const void* tlas; const void* blas; const void* bvh; const void* tri; const void* triEx; const void* triIdx; const void* tex;
  uint32_t n_tlas_nodes, n_blas, n_bvh_nodes, n_tris, n_tri_idx, tlas_root;
  uint64_t tex_bytes;
} vxrc_scene_t;
typedef struct vxrc_params {     /* the fields of raycast/common.h:126-150 kernel_arg_t that are not addresses */
  float camera_pos[3], camera_forward[3], camera_right[3], camera_up[3], viewplane[2];
  uint32_t samples_per_pixel, max_depth;
  float light_pos[3], light_color[3], ambient_color[3], background_color[3];
} vxrc_params_t;
/* Acceleration layout of a raycast scene, built once per scene (compact 64-byte BVH2 nodes holding both children's boxes and
 * complete child descriptors; triangles in edge form in leaf order, the triIdx indirection resolved).  The build validates every
 * index the BVH walk follows (children inside their instance's node range and after their parent, leaf ranges inside triIdx,
 * triIdx entries inside tri, bvh_offset of every instance) and fails (-1) on a malformed tree; TLAS indices, instance indices
 * and texture extents are checked where the kernel follows them (status bit 2).  The scene's buffers must stay alive and
 * unchanged while the layout is in use. */
typedef struct vxrc_accel vxrc_accel_t;
int vxrc_accel_build(const vxrc_scene_t* scene, void* stream, vxrc_accel_t** out);
int vxrc_accel_destroy(vxrc_accel_t* accel);
/* What the build decided (diagnostic): which = 0 -> 1 if the walk takes two BVH2 levels per fetch (wide nodes), 0 if it keeps the
 * reference's two-wide walk (a box that is not the union of its children's, an unbounded box, or a tree deeper than 42 internal
 * levels, whose wide walk could need more than the reference's 64 stack entries); which = 1 -> internal nodes on the longest
 * root-to-leaf path, counted up to 43 (0 when the wide layout was not requested). */
int vxrc_accel_info(const vxrc_accel_t* accel, uint32_t which, uint64_t* value);
/* vxrc_render on a prebuilt layout (asynchronous on `stream`; the layout keeps four frame contexts: frames issued round robin on up to
 * four streams overlap). */
int vxrc_render_accel(vxrc_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                      const vxrc_params_t* params, uint32_t* dst, float* colors, void* stream);

/* One-shot form: builds the layout, renders, frees it (synchronises the device).
 * Rows [y0,y1) of the frame kernel.cpp:9-33 renders: GenerateRay (render.h:192-211) -> Trace (:213-275) summed
 * over the samples -> RGB32FtoRGB8 -> dst[x + y*W].  colors (optional): f32 rgb per pixel before packing.
 * A malformed BVH fails the call (-1); a TLAS / instance index or texture extent outside its buffer, a stack deeper than the
 * reference's BVH_STACK_SIZE (64, undefined behaviour there) or a runaway TLAS walk sets vxrt_status bits 2 / 0 / 1. */
int vxrc_render(const vxrc_scene_t* scene, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                const vxrc_params_t* params, uint32_t* dst, float* colors, void* stream);

/* Diagnostic variant of vxrt_render_stats_timed: additionally logs 16 u64 per wavefront of the main traversal launch into
 * wave_log (device u64[16 * 4 * 8 * 256]): [0] first and [1] last 100 MHz clock, [2] rays started, [3] loop iterations,
 * [4] runs of the node body and [5] lanes active in them, [6] runs of the leaf body and [7] lanes active in them,
 * [8] lane node steps served from the LDS top-of-tree image, [9] node-body runs in which every node lane was at the same
 * node, [10] shader clocks inside the node body, [11] inside the instance + leaf part, [12] of the whole wavefront,
 * [13] inside the fetch section (job queue, ray generation), [14] inside the finish section (records, occlusion-ray
 * start), [15] reserved -- to study load balance and lane occupancy of the launch (tools/wave_balance.py). */
int vxrt_render_wave_log(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                         const vxrt_shade_params_t* params, int shadow, uint32_t* dst,
                         unsigned long long* counters, unsigned long long* wave_log, void* stream);

/* Trace n rays (6 floats each: origin, direction) read from HBM, write n hit records.
 * tmax: optional per-ray upper bound (NULL = 1e30); a bound above 1e30 is taken as 1e30 (the reference's hit.dist never
 * exceeds it: rt_traversal.h:7,44-52). */
int vxrt_trace(vxrt_accel_t* accel, const float* rays, uint64_t n, const float* tmax,
               vxrt_hit_t* hits, int mode, void* stream);

/* Closest-hit / miss shader over n (ray, hit record) pairs -- the hit records vxrt_trace wrote for those rays: f32 colour (3 per
 * ray, optional) and packed RGB8 (optional) as closest.cpp:57-127 (no secondary ray) / miss.cpp:9-14 / common.h:149-154 compute
 * them.  The records must come from this accel (their blasIdx / triIdx are followed unchecked). */
int vxrt_shade_rays(vxrt_accel_t* accel, const float* rays, const vxrt_hit_t* hits, uint64_t n, const vxrt_shade_params_t* params,
                    float* colors, uint32_t* rgb8, void* stream);

/* Camera rays of rows [y0, y1) of the RTU test's frame (kernel.cpp:28-39) as a ray buffer: 6 floats per ray, ray of pixel (x, y) at
 * index x + (y - y0) * width. */
int vxrt_camera_rays(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, float* rays, void* stream);

/* Opt-in REFERENCE-QUIRKS traversal.  vxrt_trace / vxrt_render implement the canonical algorithm: the reference RTU's result
 * wherever the RTU addresses its own data.  With a TLAS deeper than one level it does not: children of a TLAS internal node popped
 * from the short stack are addressed relative to the last BLAS's base_ptr (sim/simx/rt_traversal.cpp:91-92 after :119-120), it reads
 * unrelated memory and loses hits that exist.  What it returns then depends on the memory layout, so this entry point works on what
 * the simulator works on: ONE flat memory image addressed with 32-bit offsets (its RAM) and the four base pointers of the RTX DCRs
 * 0x6..0x9, and restates BVHTraverser::traverse literally (trail[32], 5-entry short stack, restart, re-descent after every
 * accepted candidate, libstdc++ min/max).  One thread per ray; a compatibility mode, not a fast path.  Reads outside the image
 * return zeros; a path deeper than 32 levels sets status bit 0 (undefined behaviour in the reference).
 * mode: VXRT_MODE_CLOSEST = fixed point of the accept loop, VXRT_MODE_ANY = first accepted candidate.  tmax (optional) as vxrt_trace. */
int vxrt_trace_reference_quirks(const void* image, uint64_t image_size, uint32_t tlas_off, uint32_t blas_off, uint32_t bvh_off, uint32_t tri_off,
                                const float* rays, uint64_t n, const float* tmax, vxrt_hit_t* hits, int mode, void* stream);

/* Diagnostic (tests): copies the first n_dwords (<= 800) of the control block of frame context `ctx` to `out` after synchronising
 * `stream`: [0] number of rays the main launch handed to the EXACT launch, [32 + 32 k] queue shard k (k = 0..7) of the main launch,
 * [288 + 32 k] of the EXACT launch over the deferred list, [544 + 32 k] of the a-priori EXACT launch.  A vxrt_trace call leaves the
 * block as its launches left it (the next call clears it). */
int vxrt_debug_read_control(vxrt_accel_t* accel, uint32_t ctx, uint32_t* out, uint32_t n_dwords, void* stream);
/* Diagnostic (tools/xcd_tail.py): from now on every main traversal launch on this layout -- the TIMED kernels included -- leaves per
 * wavefront (index = workgroup * 4 + wavefront of the workgroup) [0] the constant 100 MHz clock at its end and [1] rays it started |
 * physical XCD << 56 in `log` (device memory, 2 u64 per wavefront, room for 8,192 wavefronts); NULL switches it off.  One store per
 * wavefront, at its end.  The pointer is captured by the launches enqueued after the call (no synchronisation): the caller keeps the
 * memory alive until they have run. */
int vxrt_debug_end_log(vxrt_accel_t* accel, unsigned long long* log);
/* Diagnostic (tools/trace_phases.py): vxrt_trace_stats launches keep vxrt_render_wave_log's 16 u64 per wavefront in `log` (device
 * memory, 16 x 8,192 u64) from now on; NULL switches it off. */
int vxrt_debug_trace_wave_log(vxrt_accel_t* accel, unsigned long long* log);
/* diagnostic (tools/tile_tail.py): what frame context `ctx` learned for sets of `batch` frames -- per-tile cost (loop iterations of the
 * wavefront that traced it in the last launch; start clocks, durations and steal distances behind them after a wave-log launch) and the
 * tile order derived from it.  Returns the capacity in tiles. */
int vxrt_debug_read_lpt(vxrt_accel_t* accel, uint32_t ctx, uint32_t batch, uint32_t* cost, uint32_t cost_cap, uint32_t* order, uint32_t order_cap,
                        void* stream);

/* Status word of the launches on this device since the last call: 0 = ok, bit0 = traversal stack overflow
 * (tree deeper than the 32 levels the reference's own trail supports; the twin: deeper than BVH_STACK_SIZE),
 * bit1 = iteration limit, bit2 = the twin's kernel met an index outside its buffers.  Synchronises `stream`;
 * a non-zero word is cleared by the call (read-and-clear), so one failed run does not fail the next. */
int vxrt_status(void* stream, uint32_t* status);
/* Raw device pointer behind a vx_buffer_h of the hip backend (for zero-copy hand-off to RCCL). */
int vx_hip_buffer_device_ptr(vx_buffer_h hbuffer, void** dev_ptr);

/* Host-side counters of a hip-backend device: which 0 = acceleration layouts built by vx_start so far (one per scene upload,
 * not one per run), 1 = hipMalloc calls made for buffers (buffers up to 4 KB share slabs), 2 = runs split over more than one GPU,
 * 3 = GPUs behind this device, 4 / 5 = joined runs whose MCYCLE came from the device's clock / from the host's, 6 = the last joined run on
 * the host's clock (vx_start -> the stream seen drained), microseconds, 7 = runs whose shares were gathered through RCCL.
 *
 * vx_mpm_query(MCYCLE) = the last run's duration x the shader clock.  The duration is taken on the DEVICE, on its constant 100 MHz clock:
 * vx_start launches a one-thread kernel on a stream of its own that stores the clock (it starts with the run's first launch and delays
 * nothing on the run's stream), the run's last kernel -- the one that sends rays and status back -- reads the clock again.  A host that
 * does other work between vx_start and vx_ready_wait does not lengthen it.  Should the stamp not have landed when the run's last kernel
 * reads it (never observed) the run reports the host's clock (vx_start -> the moment the stream was seen drained); stats 4 / 5 count both.
 *
 * More than one GPU behind ONE vx_device (the unmodified reference host, which opens one device: tracer.cpp:78):
 *   VORTEX_HIP_DEVICES=0,1,2,3   the first index holds the address space (every vx_mem_* / vx_copy_* call), the others keep a copy of the
 * scene's seven buffers (refreshed when one is uploaded again) and their own acceleration layout.  vx_start of a whole frame traces tile
 * rows k, k+n, ... on the k-th listed device and copies them into the first device's output buffer, behind which the run's last kernel
 * waits: vx_ready_wait, vx_copy_from_dev and vx_mpm_query (MINSTRET = rays of all shares) behave as with one GPU.  A run the host restricted
 * itself (DCR 0x7F0-0x7F3), a reference-quirks run and the software twin's kernel stay on the first device.
 *   VORTEX_HIP_GATHER=rccl       the shares travel through RCCL instead of peer copies (north_star: "RCCL gather over xGMI only for final
 * image assembly", from the C host): one communicator per distinct listed GPU (ncclCommInitAll, librccl.so.1 loaded with dlopen at
 * vx_dev_open), per run every share packed into contiguous bytes on its own GPU, ONE group of ncclSend / ncclRecv pairs, the first device
 * placing the received shares into the output buffer's rows.  Default ("copy"): hipMemcpy2DAsync between the devices.  stat 7 counts
 * the runs gathered through RCCL. */
int vx_hip_device_stat(vx_device_h hdevice, uint32_t which, uint64_t* value);

const char* vxrt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VORTEX_HIP_H */
