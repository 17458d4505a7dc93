// Calibration loops for the VALU roofline (tools/calibrate_valu.py): streams of INDEPENDENT vector instructions
// of a known count, launched at a chosen number of wavefronts per SIMD.  Every wavefront stamps the shader clock (s_memtime)
// and the 100 MHz wall clock (s_memrealtime) around its loop, so the cycles one wave64 instruction occupies its SIMD follow
// from counts and clocks alone -- no assumption about the frequency the chip holds under this load.  Run under the same
// rocprofv3 --pmc pass as the traversal kernel, it also shows what SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES read
// for a VALU pipe whose utilisation is known.  No reference counterpart.  MEASUREMENT ONLY: built into lib/libvxrt_calib.so, which
// tools/calibrate_valu.py and bench.py's clock probe load; the product library (libvortex-hip.so) does not contain it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CAL_CHAINS 16   // independent accumulators per lane: no instruction waits for the one before it
#define CAL_UNROLL 8    // loop body = CAL_UNROLL * CAL_CHAINS instructions, so loop control is < 3 % of the stream

enum { OP_FMA = 0, OP_PK_FMA = 1, OP_ADD = 2, OP_CNDMASK = 3, OP_CVT_UBYTE = 4, OP_MAX3 = 5, OP_RCP = 6, OP_CNDMASK_SGPR = 7, OP_MOV = 8,
       OP_CMP_VCC = 9, OP_CMP_SGPR = 10, OP_MUL = 11, OP_MAX = 12, OP_AND = 13, OP_CNDMASK_CONST = 14, OP_MIN = 15, OP_XOR = 16, OP_BFI = 17, OP_SUB = 18,
       OP_LSHL = 19, OP_ADD_U32 = 20, OP_MED3 = 21, OP_FMAC = 22, OP_MUL_LO = 23, OP_PERM = 24, OP_CNDMASK_VCC_SET = 25, OP_ASHR = 26, OP_MIN_U32 = 27,
       OP_LSHL_OR = 28, OP_AND_OR = 29, OP_PK_MUL = 30, OP_PK_ADD = 31, OP_CNDMASK_E64_VCC = 32, OP_CNDMASK_VCC_ALT_ADD = 33, OP_CNDMASK_VCC_DISTINCT = 34,
       OP_CNDMASK_VCC_1OF8 = 35, OP_CVT_F32_F16 = 36, OP_FMA_MIX = 37, OP_CVT_F32_U32 = 38, OP_CVT_UBYTE0 = 39,
       OP_LSHR = 40, OP_BFE = 41, OP_ADD3 = 42, OP_MAD_U24 = 43, OP_CVT_SDWA = 44, OP_CMP_I32 = 45, OP_SUB_U32 = 46, OP_MIN_I32 = 47, OP_MIX_ADD_MAX = 48, OP_MIX_FMA_CVT = 49, OP_MIX_ADD_CND64 = 50,
       OP_MIX_ADD_CMP = 51, OP_MIX_3ADD_1MAX = 52, OP_MIX_ADD_RCP = 53, OP_COUNT = 54 };

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void vxcal_kernel(float* __restrict__ out, unsigned long long* __restrict__ clocks, uint32_t n_iter, float m, float c) {
  float a[CAL_CHAINS], b[CAL_CHAINS];
#pragma unroll
  for (int k = 0; k < CAL_CHAINS; ++k) { a[k] = (float)(threadIdx.x + k) * 1e-3f + 1.0f; b[k] = 0.f; }
  unsigned long long mask = __ballot(a[0] > 1.03f), mask2 = 0;
  if (OP == OP_CNDMASK_VCC_SET) asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (uint32_t i = 0; i < n_iter; ++i) {
#pragma unroll
    for (int u = 0; u < CAL_UNROLL; ++u) {
#pragma unroll
      for (int k = 0; k < CAL_CHAINS; ++k) {
        if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if (OP == OP_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c) : );
        if (OP == OP_CVT_UBYTE) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[k]));
        if (OP == OP_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
        if (OP == OP_CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "s"(mask));
        if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(c));
        if (OP == OP_CMP_VCC) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(c) : "vcc");
        if (OP == OP_CMP_SGPR) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mask2) : "v"(a[k]), "v"(c));
        if (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        if (OP == OP_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        if (OP == OP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        if (OP == OP_MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        if (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        if (OP == OP_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if (OP == OP_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        if (OP == OP_LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[k]));
        if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        if (OP == OP_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if (OP == OP_FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if (OP == OP_CNDMASK_VCC_SET) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c));
        if (OP == OP_ASHR) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[k]));
        if (OP == OP_MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        if (OP == OP_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(m));
        if (OP == OP_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if (OP == OP_CNDMASK_E64_VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c));
        if (OP == OP_CNDMASK_VCC_ALT_ADD) { if (k & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }
        if (OP == OP_CNDMASK_VCC_DISTINCT) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(b[k]) : "v"(a[k]), "v"(c));
        if (OP == OP_CNDMASK_VCC_1OF8) { if ((k & 7) == 0) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }
        if (OP == OP_CVT_F32_F16) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[k]));
        if (OP == OP_FMA_MIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[k]) : "v"(m), "v"(c));
        if (OP == OP_CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[k]));
        if (OP == OP_CVT_UBYTE0) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[k]));
        if (OP == OP_LSHR) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(a[k]));
        if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[k]));
        if (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if (OP == OP_MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if (OP == OP_CVT_SDWA) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "+v"(a[k]));
        if (OP == OP_CMP_I32) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(a[k]), "v"(c) : "vcc");
        if (OP == OP_SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        if (OP == OP_MIN_I32) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        if (OP == OP_MIX_ADD_MAX) { if (k & 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }
        if (OP == OP_MIX_FMA_CVT) { if (k & 1) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[k])); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c)); }
        if (OP == OP_MIX_ADD_CND64) { if (k & 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "s"(mask)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }
        if (OP == OP_MIX_ADD_CMP) { if (k & 1) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mask2) : "v"(a[k]), "v"(c)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }
        if (OP == OP_MIX_3ADD_1MAX) { if ((k & 3) == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }
        if (OP == OP_MIX_ADD_RCP) { if ((k & 3) == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k])); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }
        if (OP == OP_CNDMASK_CONST) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "s"(0x5555555555555555ull));
      }
      if (OP == OP_PK_MUL || OP == OP_PK_ADD) {
#pragma unroll
        for (int rep2 = 0; rep2 < 2; ++rep2) {
#pragma unroll
          for (int k = 0; k < CAL_CHAINS; k += 2) {
            float2_t v = {a[k], a[k + 1]};
            const float2_t mm = {m, m};
            if (OP == OP_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(mm));
            else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(mm));
            a[k] = v.x; a[k + 1] = v.y;
          }
        }
      }
      if (OP == OP_PK_FMA) {
#pragma unroll
        for (int k = 0; k < CAL_CHAINS; k += 2) {   // half as many instructions for the same number of lane operations ...
          float2_t v = {a[k], a[k + 1]};
          const float2_t mm = {m, m}, cc = {c, c};
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(cc));
          a[k] = v.x; a[k + 1] = v.y;
        }
#pragma unroll
        for (int k = 0; k < CAL_CHAINS; k += 2) {   // ... issued twice, so every OP executes CAL_CHAINS instructions per unroll step
          float2_t v = {a[k], a[k + 1]};
          const float2_t mm = {m, m}, cc = {c, c};
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(cc));
          a[k] = v.x; a[k + 1] = v.y;
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = (float)(mask2 & 1ull);
#pragma unroll
  for (int k = 0; k < CAL_CHAINS; ++k) s += a[k] + b[k];
  out[(size_t)blockIdx.x * 256u + threadIdx.x] = s;
  if ((threadIdx.x & 63u) == 0u) {
    unsigned long long* w = clocks + 2ull * ((size_t)blockIdx.x * 4u + (threadIdx.x >> 6));
    w[0] = t1 - t0; w[1] = r1 - r0;
  }
}

__global__ __launch_bounds__(64) void vxcal_clock_probe_kernel(uint32_t ticks, unsigned long long* __restrict__ out) {
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  unsigned long long r1 = r0;
  float x = 1.0f;
  for (uint32_t i = 0; i < 4000000u && r1 - r0 < ticks; ++i) {     // (bounded: ~4 M iterations at most)
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
    r1 = wall_clock64();
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  r1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
  if (x == 123.456f) out[2] = 1;   // (keeps the chain alive)
}

extern "C" {
// blocks of 256 threads (4 wavefronts, one per SIMD of a CU); vector instructions per wavefront = CAL_CHAINS * CAL_UNROLL * n_iter.
// clocks: device u64[2 * 4 * blocks] = per wavefront {shader cycles, 100 MHz ticks} around the loop.
int vxcal_valu_loop(int op, uint32_t blocks, uint32_t n_iter, float* out, unsigned long long* clocks, void* stream) {
  if (!blocks || !out || !clocks) return -1;
  hipStream_t s = (hipStream_t)stream;
  const float m = 1.0000001f, c = 0.25f;
  switch (op) {
    case 0: hipLaunchKernelGGL(vxcal_kernel<0>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 1: hipLaunchKernelGGL(vxcal_kernel<1>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 2: hipLaunchKernelGGL(vxcal_kernel<2>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 3: hipLaunchKernelGGL(vxcal_kernel<3>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 4: hipLaunchKernelGGL(vxcal_kernel<4>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 5: hipLaunchKernelGGL(vxcal_kernel<5>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 6: hipLaunchKernelGGL(vxcal_kernel<6>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 7: hipLaunchKernelGGL(vxcal_kernel<7>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 8: hipLaunchKernelGGL(vxcal_kernel<8>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 9: hipLaunchKernelGGL(vxcal_kernel<9>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 10: hipLaunchKernelGGL(vxcal_kernel<10>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 11: hipLaunchKernelGGL(vxcal_kernel<11>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 12: hipLaunchKernelGGL(vxcal_kernel<12>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 13: hipLaunchKernelGGL(vxcal_kernel<13>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 14: hipLaunchKernelGGL(vxcal_kernel<14>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 15: hipLaunchKernelGGL(vxcal_kernel<15>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 16: hipLaunchKernelGGL(vxcal_kernel<16>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 17: hipLaunchKernelGGL(vxcal_kernel<17>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 18: hipLaunchKernelGGL(vxcal_kernel<18>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 19: hipLaunchKernelGGL(vxcal_kernel<19>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 20: hipLaunchKernelGGL(vxcal_kernel<20>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 21: hipLaunchKernelGGL(vxcal_kernel<21>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 22: hipLaunchKernelGGL(vxcal_kernel<22>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 23: hipLaunchKernelGGL(vxcal_kernel<23>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 24: hipLaunchKernelGGL(vxcal_kernel<24>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 25: hipLaunchKernelGGL(vxcal_kernel<25>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 26: hipLaunchKernelGGL(vxcal_kernel<26>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 27: hipLaunchKernelGGL(vxcal_kernel<27>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 28: hipLaunchKernelGGL(vxcal_kernel<28>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 29: hipLaunchKernelGGL(vxcal_kernel<29>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 30: hipLaunchKernelGGL(vxcal_kernel<30>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 31: hipLaunchKernelGGL(vxcal_kernel<31>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 32: hipLaunchKernelGGL(vxcal_kernel<32>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 33: hipLaunchKernelGGL(vxcal_kernel<33>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 34: hipLaunchKernelGGL(vxcal_kernel<34>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 35: hipLaunchKernelGGL(vxcal_kernel<35>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 36: hipLaunchKernelGGL(vxcal_kernel<36>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 37: hipLaunchKernelGGL(vxcal_kernel<37>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 38: hipLaunchKernelGGL(vxcal_kernel<38>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 39: hipLaunchKernelGGL(vxcal_kernel<39>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 40: hipLaunchKernelGGL(vxcal_kernel<40>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 41: hipLaunchKernelGGL(vxcal_kernel<41>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 42: hipLaunchKernelGGL(vxcal_kernel<42>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 43: hipLaunchKernelGGL(vxcal_kernel<43>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 44: hipLaunchKernelGGL(vxcal_kernel<44>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 45: hipLaunchKernelGGL(vxcal_kernel<45>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 46: hipLaunchKernelGGL(vxcal_kernel<46>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 47: hipLaunchKernelGGL(vxcal_kernel<47>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 48: hipLaunchKernelGGL(vxcal_kernel<48>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 49: hipLaunchKernelGGL(vxcal_kernel<49>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 50: hipLaunchKernelGGL(vxcal_kernel<50>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 51: hipLaunchKernelGGL(vxcal_kernel<51>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 52: hipLaunchKernelGGL(vxcal_kernel<52>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    case 53: hipLaunchKernelGGL(vxcal_kernel<53>, dim3(blocks), dim3(256), 0, s, out, clocks, n_iter, m, c); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
uint32_t vxcal_instr_per_iter(void) { return CAL_CHAINS * CAL_UNROLL; }

// XCD probe (tools/xcd_probe.py): is one XCD slower than another, and at what -- issuing VALU instructions, waiting for memory, or device-scope
// atomics?  One wavefront per workgroup, as many workgroups as asked for (the dispatcher deals them over the XCDs); every wavefront stamps the
// shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime) around three loops of fixed work:
//   (1) n_alu x 64 dependent v_fma_f32                                   -> VALU issue: 100 MHz ticks per loop give the XCD's real clock
//   (2) n_chase dependent loads through `chase` (a random cycle over a buffer far larger than the L2), every lane its own chain -> memory latency
//   (3) n_atom dependent device-scope atomic adds on ONE word (lane 0)    -> round trip to wherever that word's line lives
// out, 8 u64 per wavefront: [0] physical XCD | HW_ID << 8, [1] 100 MHz clock at the start, [2..3] loop 1 in 100 MHz ticks / shader clocks,
// [4..5] loop 2, [6..7] loop 3.
__global__ __launch_bounds__(64) void vxcal_xcd_probe_kernel(uint32_t n_alu, const uint32_t* __restrict__ chase, uint32_t chase_len, uint32_t n_chase,
                                                             uint32_t* __restrict__ atom, uint32_t n_atom, unsigned long long* __restrict__ out, float* __restrict__ sink) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20), hwid = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  unsigned long long r[4], c[4];
  float a = (float)lane * 1e-3f + 1.0f;
  const float m = 1.0000001f, k = 0.25f;
  r[0] = wall_clock64(); c[0] = __builtin_readcyclecounter();
  for (uint32_t i = 0; i < n_alu; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));
  }
  r[1] = wall_clock64(); c[1] = __builtin_readcyclecounter();
  uint32_t idx = (blockIdx.x * 64u + lane) * 2654435761u % chase_len;
  for (uint32_t i = 0; i < n_chase; ++i) idx = __builtin_nontemporal_load(chase + idx);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  r[2] = wall_clock64(); c[2] = __builtin_readcyclecounter();
  uint32_t v = idx & 1u;
  if (lane == 0) for (uint32_t i = 0; i < n_atom; ++i) v = atomicAdd(atom, (v & 1u) + 1u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  r[3] = wall_clock64(); c[3] = __builtin_readcyclecounter();
  if (a == 123.456f || idx == 0xFFFFFFFFu || v == 0xFFFFFFFFu) sink[0] = a;   // (keeps the loops alive)
  if (lane == 0) {
    unsigned long long* w = out + 8ull * blockIdx.x;
    w[0] = (unsigned long long)xcc | ((unsigned long long)hwid << 8); w[1] = r[0];
    for (int j = 0; j < 3; ++j) { w[2 + 2 * j] = r[j + 1] - r[j]; w[3 + 2 * j] = c[j + 1] - c[j]; }
  }
}

// Clock probe: ONE wavefront reads the shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime), spins for `ticks`
// of the latter (a bounded loop: it also ends after max_iter iterations) and reads both again.  out[0] = shader cycles, out[1] =
// 100 MHz ticks: shader clock in GHz = out[0] / out[1] / 10.  Launched right before and right after a timed region, on the stream
// that carries it, it says what clock the chip held there (the power management reacts over milliseconds, the probe takes ~30 us).
int vxcal_xcd_probe(uint32_t n_waves, uint32_t n_alu, const uint32_t* chase, uint32_t chase_len, uint32_t n_chase, uint32_t* atom, uint32_t n_atom,
                    unsigned long long* out, float* sink, void* stream) {
  if (!n_waves || n_waves > (1u << 20) || !chase || !chase_len || !atom || !out || !sink) return -1;
  hipLaunchKernelGGL(vxcal_xcd_probe_kernel, dim3(n_waves), dim3(64), 0, (hipStream_t)stream, n_alu, chase, chase_len, n_chase, atom, n_atom, out, sink);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int vxcal_clock_probe(uint32_t ticks, unsigned long long* out, void* stream) {
  if (!out || ticks == 0 || ticks > 100000u) return -1;
  hipLaunchKernelGGL(vxcal_clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
}
