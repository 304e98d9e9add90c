// valu_rate.hip -- issue rate of single VALU instructions on gfx950, per SIMD, with 1 / 2 / 4 / 8 waves per SIMD.
// Measurement aid: the block kernels are bound by VALU issue, so which instruction classes
// run at 2, 4 or 8 cycles per wave64 instruction decides what is worth removing from them.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate scripts/microbench/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define CHECK(c)                                                                     \
  do {                                                                               \
    hipError_t e_ = (c);                                                             \
    if (e_ != hipSuccess) {                                                          \
      std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);     \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

// 8 independent accumulators, 8 x 8 = 64 instructions per loop trip (+ 2 scalar loop instructions)
#define KERNEL32(NAME, ASM)                                                                                     \
  __global__ void NAME(float* out, int iters) {                                                                 \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    const float b = 1.0001f, c = 0.5f;                                                                          \
    for (int i = 0; i < iters; i++) {                                                                           \
      REP8(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                 \
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)         \
                        : "v"(b), "v"(c)                                                                        \
                        : "vcc");)                                                                              \
    }                                                                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                          \
  }
#define KERNEL64(NAME, ASM)                                                                                     \
  __global__ void NAME(float* out, int iters) {                                                                 \
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    const double b = 1.0001, c = 0.5;                                                                           \
    for (int i = 0; i < iters; i++) {                                                                           \
      REP8(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                 \
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)         \
                        : "v"(b), "v"(c)                                                                        \
                        : "vcc");)                                                                              \
    }                                                                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                 \
  }

#define A_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_ADD32(i) "v_add_f32 %" #i ", %" #i ", %9\n"
#define A_MUL32(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_FMAC32(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define A_MOV32(i) "v_mov_b32 %" #i ", %8\n"
#define A_CND32(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CND32S(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define A_CND32X(i) "v_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define A_SUB32(i) "v_sub_f32 %" #i ", %" #i ", %9\n"
#define A_FMANEG(i) "v_fma_f32 %" #i ", -%" #i ", %8, -%9\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_MIN32(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define A_CVT(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define A_PERM(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 2, 5\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_RCP32(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_LOG32(i) "v_log_f32 %" #i ", %" #i "\n"
#define A_SQRT32(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define A_CMP32(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define A_MAX32(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define A_DPP(i) "v_add_f32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define A_ADD64(i) "v_add_f64 %" #i ", %" #i ", %9\n"
#define A_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define A_MOV64(i) "v_mov_b64 %" #i ", %8\n"
#define A_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define A_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define A_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %9\n"
#define A_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 2, %8\n"

KERNEL32(k_fma32, A_FMA32)
KERNEL32(k_add32, A_ADD32)
KERNEL32(k_mul32, A_MUL32)
KERNEL32(k_fmac32, A_FMAC32)
KERNEL32(k_mov32, A_MOV32)
KERNEL32(k_cnd32, A_CND32)
KERNEL32(k_addu, A_ADDU)
KERNEL32(k_cnd32x, A_CND32X)
KERNEL32(k_sub32, A_SUB32)
KERNEL32(k_fmaneg, A_FMANEG)
KERNEL32(k_xor, A_XOR)
KERNEL32(k_min32, A_MIN32)
KERNEL32(k_cvt, A_CVT)
KERNEL32(k_perm, A_PERM)
// the select condition in an SGPR pair that the kernel sets itself (s[10:11] is clobbered and written first)
__global__ void k_cnd32s(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("s_mov_b64 s[10:11], 0x5555\n" A_CND32S(0) A_CND32S(1) A_CND32S(2) A_CND32S(3) A_CND32S(4) A_CND32S(5) A_CND32S(6) A_CND32S(7)
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                      : "v"(b), "v"(c)
                      : "s10", "s11");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// VOP3 encoding naming vcc explicitly, vcc written by the scalar unit
__global__ void k_cnd32e(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0001f, c = 0.5f;
#define A_CND32E(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("s_mov_b64 vcc, 0x5555\n" A_CND32E(0) A_CND32E(1) A_CND32E(2) A_CND32E(3) A_CND32E(4) A_CND32E(5) A_CND32E(6) A_CND32E(7)
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                      : "v"(b), "v"(c)
                      : "vcc");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// VOP2 encoding (implicit vcc), vcc written by the scalar unit
__global__ void k_cnd32v(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("s_mov_b64 vcc, 0x5555\n" A_CND32(0) A_CND32(1) A_CND32(2) A_CND32(3) A_CND32(4) A_CND32(5) A_CND32(6) A_CND32(7)
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                      : "v"(b), "v"(c)
                      : "vcc");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// condition in an SGPR pair written by a VALU compare (what compiled code mostly does)
__global__ void k_cnd32q(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %9\n" A_CND32S(0) A_CND32S(1) A_CND32S(2) A_CND32S(3) A_CND32S(4) A_CND32S(5) A_CND32S(6) A_CND32S(7)
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                      : "v"(b), "v"(c)
                      : "s10", "s11");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// vcc written by a compare inside the block (one v_cmp per 8 selects)
__global__ void k_cnd32c(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("v_cmp_lt_f32 vcc, %8, %9\n" A_CND32(0) A_CND32(1) A_CND32(2) A_CND32(3) A_CND32(4) A_CND32(5) A_CND32(6) A_CND32(7)
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                      : "v"(b), "v"(c)
                      : "vcc");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
KERNEL32(k_lshl, A_LSHL)
KERNEL32(k_and, A_AND)
KERNEL32(k_lshladd, A_LSHLADD)
KERNEL32(k_bfe, A_BFE)
KERNEL32(k_mullo, A_MULLO)
KERNEL32(k_rcp32, A_RCP32)
KERNEL32(k_log32, A_LOG32)
KERNEL32(k_sqrt32, A_SQRT32)
KERNEL32(k_cmp32, A_CMP32)
KERNEL32(k_max32, A_MAX32)
KERNEL32(k_dpp, A_DPP)
KERNEL64(k_fma64, A_FMA64)
KERNEL64(k_add64, A_ADD64)
KERNEL64(k_mul64, A_MUL64)
KERNEL64(k_mov64, A_MOV64)
KERNEL64(k_rcp64, A_RCP64)
KERNEL64(k_pkfma, A_PKFMA)
KERNEL64(k_pkmul, A_PKMUL)
KERNEL64(k_pkadd, A_PKADD)
KERNEL64(k_lshladd64, A_LSHLADD64)

struct Case {
  const char* name;
  void (*fn)(float*, int);
};

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int    cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;   // kHz -> GHz (the nominal peak engine clock; the chip may run below it)
  std::printf("device: %s, %d CUs, nominal %.2f GHz\n", prop.name, cus, ghz);
  float* out = nullptr;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * 8 * 4096));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const Case cases[] = {{"v_fma_f32", k_fma32}, {"v_add_f32", k_add32}, {"v_mul_f32", k_mul32}, {"v_fmac_f32", k_fmac32},
                        {"v_mov_b32", k_mov32}, {"v_cndmask_b32", k_cnd32}, {"cndmask sgpr cond", k_cnd32s}, {"cndmask after cmp", k_cnd32c}, {"cnd e64 vcc (salu)", k_cnd32e}, {"cnd e32 vcc (salu)", k_cnd32v},
                        {"cnd sgpr after cmp", k_cnd32q}, {"cndmask dst!=src", k_cnd32x}, {"v_sub_f32", k_sub32},
                        {"v_fma_f32 neg mods", k_fmaneg}, {"v_xor_b32", k_xor}, {"v_min_f32", k_min32}, {"v_cvt_f32_i32", k_cvt},
                        {"v_mov_b32 dpp", k_perm}, {"v_add_u32", k_addu}, {"v_lshlrev_b32", k_lshl},
                        {"v_and_b32", k_and}, {"v_lshl_add_u32", k_lshladd}, {"v_bfe_u32", k_bfe}, {"v_mul_lo_u32", k_mullo},
                        {"v_cmp_lt_f32", k_cmp32}, {"v_max_f32", k_max32}, {"v_add_f32 dpp", k_dpp}, {"v_rcp_f32", k_rcp32},
                        {"v_log_f32", k_log32}, {"v_sqrt_f32", k_sqrt32}, {"v_pk_fma_f32", k_pkfma}, {"v_pk_mul_f32", k_pkmul},
                        {"v_pk_add_f32", k_pkadd}, {"v_fma_f64", k_fma64}, {"v_add_f64", k_add64}, {"v_mul_f64", k_mul64},
                        {"v_mov_b64", k_mov64}, {"v_rcp_f64", k_rcp64}, {"v_lshl_add_u64", k_lshladd64}};
  const int iters = 2000;   // x 64 instructions
  std::printf("cycles per wave64 instruction per SIMD at the nominal clock (lower bound on the true rate if the clock is lower)\n");
  std::printf("%-18s %10s %10s %10s %10s\n", "instruction", "1 wave", "2 waves", "4 waves", "8 waves");
  for (const Case& c : cases) {
    std::printf("%-18s", c.name);
    for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD: one workgroup of 256 * wps lanes per CU ...
      const int  threads = wps <= 4 ? 256 * wps : 1024;   // ... or two of 1024
      const dim3 grid(cus * (wps <= 4 ? 1 : 2)), block(threads);
      hipLaunchKernelGGL(c.fn, grid, block, 0, 0, out, 10);   // warm-up
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(c.fn, grid, block, 0, 0, out, iters);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double instr_per_simd = 64.0 * iters * wps;
      std::printf(" %10.2f", ms * 1e-3 * ghz * 1e9 / instr_per_simd);
    }
    std::printf("\n");
  }
  CHECK(hipFree(out));
  return 0;
}
