// tools/valu_ceiling.hip -- what the leased MI355X issues per second, per instruction class, at the
// occupancy the step kernel runs at.  The step kernel (k_step_tiled) is bound by instruction issue,
// not by HBM; this gives the ceiling its instruction counts are priced against:
// streams of 32 independent instructions of ONE class per loop trip (the classes the step kernel's inner
// loops are made of), each at 1, 2, 4 and 8 waves per SIMD, plus two mixes (the time-to-collision loop's
// and the filter's) and two LDS forms.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_ceiling tools/valu_ceiling.hip
//   run:   tools/valu_ceiling > gpurun_out/valu_ceiling.json          (tools/rocprof_passes.sh does both)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNROLL = 32;  // independent instructions per loop trip

// name, instructions per asm statement, asm text (%0 = a[k] read-write VGPR, %1 / %2 = loop-invariant VGPRs;
// s8 / s[8:9] = loop-invariant SGPRs, vcc defined before the loop)
#define CLASSES(X)                                                                                   \
  X(v_fma_f32, 1, "v_fma_f32 %0, %0, %1, %2")                                                         \
  X(v_fmac_f32, 1, "v_fmac_f32 %0, %1, %2")                                                           \
  X(v_mul_f32, 1, "v_mul_f32 %0, %0, %1")                                                             \
  X(v_add_f32, 1, "v_add_f32 %0, %0, %1")                                                             \
  X(v_sub_f32_neg_modifier, 1, "v_sub_f32_e64 %0, -%0, %1")                                           \
  X(v_min_f32, 1, "v_min_f32 %0, %0, %1")                                                             \
  X(v_add_u32, 1, "v_add_u32 %0, %0, %1")                                                             \
  X(v_sub_u32, 1, "v_sub_u32 %0, %0, %1")                                                             \
  X(v_and_b32, 1, "v_and_b32 %0, %0, %1")                                                             \
  X(v_lshlrev_b32, 1, "v_lshlrev_b32 %0, 3, %0")                                                      \
  X(v_lshrrev_b32, 1, "v_lshrrev_b32 %0, 31, %0")                                                     \
  X(v_mov_b32, 1, "v_mov_b32 %0, %1")                                                                 \
  X(v_min_u32, 1, "v_min_u32 %0, %0, %1")                                                             \
  X(v_add_f32_sgpr_operand, 1, "v_add_f32 %0, s8, %0")                                                \
  X(v_mul_f32_sgpr_operand, 1, "v_mul_f32 %0, s8, %0")                                                \
  X(v_fma_f32_sgpr_operand, 1, "v_fma_f32 %0, %0, s8, %1")                                            \
  X(v_mul_f32_literal, 1, "v_mul_f32 %0, 0x3fb8aa3b, %0")                                             \
  X(v_mul_f32_inline_const, 1, "v_mul_f32 %0, 2.0, %0")                                               \
  X(v_add_u32_sgpr_operand, 1, "v_add_u32 %0, s8, %0")                                                \
  X(v_cvt_f32_i32, 1, "v_cvt_f32_i32 %0, %0")                                                         \
  X(v_cvt_i32_f32, 1, "v_cvt_i32_f32 %0, %0")                                                         \
  X(v_floor_f32, 1, "v_floor_f32 %0, %0")                                                             \
  X(v_cmp_lt_f32_vcc, 1, "v_cmp_lt_f32 vcc, %0, %1")                                                  \
  X(v_cmp_lt_f32_e64_sgpr, 1, "v_cmp_lt_f32_e64 s[10:11], %0, %1")                                    \
  X(v_cmp_lt_u32_vcc, 1, "v_cmp_lt_u32 vcc, %0, %1")                                                  \
  X(v_cndmask_b32_e64_sgpr_mask, 1, "v_cndmask_b32_e64 %0, %0, %1, s[8:9]")                           \
  X(pair_v_cmp_then_v_cndmask, 2, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")          \
  X(v_lshl_add_u32, 1, "v_lshl_add_u32 %0, %0, 3, %1")                                                \
  X(v_add3_u32, 1, "v_add3_u32 %0, %0, %1, %2")                                                       \
  X(v_and_or_b32, 1, "v_and_or_b32 %0, %0, %1, %2")                                                   \
  X(v_mad_u32_u24, 1, "v_mad_u32_u24 %0, %0, %1, %2")                                                 \
  X(v_alignbit_b32, 1, "v_alignbit_b32 %0, %0, %1, 31")                                               \
  X(v_bfe_u32, 1, "v_bfe_u32 %0, %0, 4, 8")                                                           \
  X(v_max3_f32, 1, "v_max3_f32 %0, %0, %1, %2")                                                       \
  X(v_min3_f32, 1, "v_min3_f32 %0, %0, %1, %2")                                                       \
  X(v_med3_f32, 1, "v_med3_f32 %0, %0, %1, %2")                                                       \
  X(v_perm_b32, 1, "v_perm_b32 %0, %0, %1, %2")                                                       \
  X(v_ffbh_u32, 1, "v_ffbh_u32 %0, %0")                                                               \
  X(v_bcnt_u32_b32, 1, "v_bcnt_u32_b32 %0, %0, %1")                                                   \
  X(v_pk_sub_i16, 1, "v_pk_sub_i16 %0, %0, %1")                                                       \
  X(v_dot2c_i32_i16, 1, "v_dot2c_i32_i16 %0, %1, %2")                                                 \
  X(v_mad_i32_i16, 1, "v_mad_i32_i16 %0, %1, %2, %0")                                                 \
  X(v_mul_lo_u32, 1, "v_mul_lo_u32 %0, %0, %1")                                                       \
  X(v_sqrt_f32, 1, "v_sqrt_f32 %0, %0")                                                               \
  X(v_rcp_f32, 1, "v_rcp_f32 %0, %0")                                                                 \
  X(v_exp_f32, 1, "v_exp_f32 %0, %0")                                                                 \
  X(v_pk_fma_f32_two_fmas_each, 1, "v_pk_fma_f32 %3, %3, %4, %4")                                     \
  X(v_mov_b32_dpp_row_shr, 1, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")            \
  X(v_readlane_b32, 1, "v_readlane_b32 s12, %0, 5")                                                   \
  X(s_add_u32_beside_nothing, 1, "s_add_u32 s12, s12, s8")

#define DECL_ENUM(name, n, text) K_##name,
enum Kind { CLASSES(DECL_ENUM) K_FMA_DEP, K_MIX_TTC, K_MIX_FILTER, K_LDS_READ_B64, K_LDS_READ_B128, K_LDS_WRITE16, KINDS };
#define DECL_NAME(name, n, text) #name,
static const char* kind_name[KINDS] = {CLASSES(DECL_NAME) "v_fma_f32_dependent_chain", "mix_ttc_6fma_1sqrt_1rcp",
                                       "mix_filter_4sub_4cvt_2mul_2fma_2cmp", "ds_read_b64", "ds_read_b128", "ds_write_b16"};
#define DECL_COUNT(name, n, text) n,
static const int kind_insts[KINDS] = {CLASSES(DECL_COUNT) 1, 1, 1, 1, 1, 1};

template <int KIND>
__global__ void __launch_bounds__(256) k_issue(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  float a[UNROLL];
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) a[k] = seed + (float)(threadIdx.x + k);
  float b = seed * 0.5f + 1.0f, c = seed * 0.25f;
  lds[threadIdx.x] = seed;
  lds[threadIdx.x + 256] = seed;
  __syncthreads();
  unsigned addr = (threadIdx.x * 16u) & 4095u;
  asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_mov_b64 s[8:9], vcc\n s_mov_b32 s12, 0" : : "v"(a[0]), "v"(b) : "vcc", "s8", "s9", "s10", "s11", "s12");
  for (int it = 0; it < iters; ++it) {
    switch (KIND) {
#define DECL_CASE(name, n, text)                                                                                              \
      case K_##name:                                                                                                           \
        _Pragma("unroll") for (int k = 0; k < UNROLL; k += 2) {                                                                \
          asm volatile(text : "+v"(a[k]) : "v"(b), "v"(c), "v"(*reinterpret_cast<double*>(&a[k])), "v"(*reinterpret_cast<double*>(&a[(k + 2) % UNROLL])) : "vcc", "s10", "s11", "s12"); \
          if (K_##name != K_v_pk_fma_f32_two_fmas_each)                                                                        \
            asm volatile(text : "+v"(a[k + 1]) : "v"(b), "v"(c), "v"(*reinterpret_cast<double*>(&a[k])), "v"(*reinterpret_cast<double*>(&a[(k + 2) % UNROLL])) : "vcc", "s10", "s11", "s12"); \
        }                                                                                                                      \
        break;
      CLASSES(DECL_CASE)
      case K_FMA_DEP:
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
        break;
      case K_MIX_TTC:  // the time-to-collision mix: 8 instructions, 2 of them quarter rate
#pragma unroll
        for (int k = 0; k < UNROLL; k += 8) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k + 1]) : "v"(b), "v"(c));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k + 2]) : "v"(b), "v"(c));
          asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k + 3]));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k + 4]) : "v"(b), "v"(c));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k + 5]) : "v"(b), "v"(c));
          asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k + 6]));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k + 7]) : "v"(b), "v"(c));
        }
        break;
      case K_MIX_FILTER:  // the distance filter's mix per pair of candidates (14 of its instructions; 32 here = 2.3 pairs)
#pragma unroll
        for (int k = 0; k < UNROLL; k += 16) {
          asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
          asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k + 1]) : "v"(b));
          asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k + 2]) : "v"(b));
          asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k + 3]) : "v"(b));
          asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[k + 4]));
          asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[k + 5]));
          asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[k + 6]));
          asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[k + 7]));
          asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[k + 8]));
          asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[k + 9]));
          asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(a[k + 10]) : "v"(b));
          asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(a[k + 11]) : "v"(b));
          asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k + 12]), "v"(b) : "vcc");
          asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k + 13]), "v"(b) : "vcc");
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k + 14]) : "v"(b));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k + 15]) : "v"(b));
        }
        break;
      case K_LDS_READ_B64:
#pragma unroll
        for (int k = 0; k < UNROLL; k += 2)
          asm volatile("ds_read_b64 %0, %1" : "=v"(*reinterpret_cast<double*>(&a[k])) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)");
        break;
      case K_LDS_READ_B128:
#pragma unroll
        for (int k = 0; k < UNROLL; k += 4)
          asm volatile("ds_read_b128 %0, %1" : "=v"(*reinterpret_cast<float4*>(&a[k])) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)");
        break;
      case K_LDS_WRITE16:
#pragma unroll
        // (2 bytes per lane, the step kernel's list rows.  7.6e10 /s = 8 clocks per wave64 and CU at this stride AND at
        // the 16-byte stride of the reads above: the rate of sub-dword LDS writes, not a bank conflict)
        for (int k = 0; k < UNROLL; ++k) asm volatile("ds_write_b16 %0, %1" : : "v"(addr >> 3), "v"(a[k]));
        asm volatile("s_waitcnt lgkmcnt(0)");
        break;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) s += a[k];
  if (s == 12345.678f) out[threadIdx.x] = s + lds[threadIdx.x];  // keeps everything live, never taken
}

template <int KIND>
double run(int waves_per_simd, int iters, float* out, int n_cu) {
  // 256 threads = 4 waves = one per SIMD of a CU; waves_per_simd workgroups per CU
  const int blocks = n_cu * waves_per_simd;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters / 4, 1.0f);  // warm-up
  CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  double per_trip = UNROLL;
  if (KIND == K_v_pk_fma_f32_two_fmas_each || KIND == K_LDS_READ_B64) per_trip = UNROLL / 2;
  if (KIND == K_LDS_READ_B128) per_trip = UNROLL / 4;
  if (KIND < K_FMA_DEP) per_trip *= kind_insts[KIND];
  const double wave_insts = (double)blocks * 4.0 * (double)iters * per_trip;
  return wave_insts / (best * 1e-3);
}

template <int KIND>
void all(float* out, int n_cu, double* fma4) {
  const int occ[4] = {1, 2, 4, 8};
  const bool slow = KIND == K_v_sqrt_f32 || KIND == K_v_rcp_f32 || KIND == K_v_exp_f32 || KIND >= K_LDS_READ_B64;
  const int iters = slow ? 6000 : 20000;
  printf("  \"%s\": {", kind_name[KIND]);
  for (int o = 0; o < 4; ++o) {
    const double r = run<KIND>(occ[o], iters, out, n_cu);
    if (KIND == K_v_fma_f32 && occ[o] == 4) *fma4 = r;
    printf("\"waves_per_simd_%d\": %.4g%s", occ[o], r, o < 3 ? ", " : "");
  }
  printf("}%s\n", KIND + 1 < KINDS ? "," : "");
  fflush(stdout);
  if constexpr (KIND + 1 < KINDS) all<KIND + 1>(out, n_cu, fma4);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  float* out;
  CHECK(hipMalloc(&out, 4096));
  printf("{\"device\": \"%s\", \"compute_units\": %d, \"simds\": %d, \"clock_mhz_reported\": %d,\n", prop.gcnArchName, n_cu, n_cu * 4,
         prop.clockRate / 1000);
  printf(" \"unit\": \"wave64 instructions per second, whole chip\", \"per_class\": {\n");
  double fma4 = 0;
  all<0>(out, n_cu, &fma4);
  printf(" },\n \"fma_wave_insts_per_s\": %.4g,\n", fma4);
  printf(" \"fma_cycles_per_inst_per_simd_at_2400mhz\": %.3f,\n", 2.4e9 * n_cu * 4 / fma4);
  printf(" \"note\": \"32 independent instructions of one class per loop trip; fma_wave_insts_per_s (4 waves per SIMD) is the "
         "ceiling bench.py prices k_step_tiled's VALU count against (roofline.valu_issue_frac)\"}\n");
  return 0;
}
