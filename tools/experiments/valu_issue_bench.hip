// tools/experiments/valu_issue_bench.hip -- what is the VALU issue peak of an MI355X for the integer program the rollout
// kernel runs?  (VERDICT r3 "weak" #2: bench.py priced k_rollout_queue against 1,024 SIMDs x 2.4 GHz / 4 = 614 G
// wave-instructions/s; the microarchitecture guide says a wave64 VALU instruction takes 2 cycles on a SIMD-32 once more than
// one wave feeds the SIMD, 4 for a lone wave.)
//
// Every CU gets W workgroups of 256 threads (one wave per SIMD each; dynamic LDS sized so that exactly W fit a CU), W = 1, 2,
// 4, 8 waves per SIMD.  Each wave runs `iters` x 256 instructions of ONE kind, either as independent streams (8 registers
// round-robin) or as one dependent chain, optionally with only the low `LANES` lanes of the wave active (the engine runs with
// ~18 of 64).  Reported per configuration: wave-instructions/s over the whole chip (HIP events), shader cycles per
// instruction per SIMD (s_memtime / instructions of the SIMD's W waves), the in-kernel clock (s_memtime / s_memrealtime).
// With `--one OP DEP W LANES` a single configuration is launched (for `rocprofv3 --pmc`, tools/gpu_valu_issue.sh).
// Build: hipcc --offload-arch=gfx950 -O3 -o prof_build/valu_issue_bench tools/experiments/valu_issue_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <utility>
#include <vector>

enum Op { OP_AND, OP_SHL, OP_BFE, OP_CNDMASK, OP_ADD, OP_MUL_LO, OP_LANE, OP_ADD3, OP_CMP, OP_SALU, OP_MIX, OP_VS,
          // encoding study (which instructions issue in 2 cycles, which in 4)
          OP_ADD_E64, OP_AND_LIT, OP_XOR, OP_MOV, OP_SHL_V, OP_LSHR, OP_CND_E32, OP_CMP_E32, OP_SUB, OP_MIN, OP_MAD24, OP_LSHL_OR, OP_AND_OR,
          OP_MUL24, OP_ADD_BFE, OP_ADD_SHL, OP_ADDC, OP_FFSS, OP_F4S4, OP_F6S2, OP_F7S1, OP_CND_VCC64, OP_CMP_CND32, OP_CMP_CND64, OP_PK_ADD, OP_ADD_F32, OP_PK_ADD_MIX, OP_COUNT };
static const char *op_name[OP_COUNT] = {"v_and_b32", "v_lshlrev_b32", "v_bfe_u32", "v_cndmask_b32", "v_add_u32", "v_mul_lo_u32",
                                        "v_readlane+v_writelane", "v_add3_u32", "v_cmp_lt_u32", "s_add_u32",
                                        "engine mix (and/shl/bfe/cndmask/add/cmp/and/add3)", "v_add_u32+s_add_u32 alternating",
                                        "v_add_u32_e64 (VOP3 encoding)", "v_and_b32 + 32-bit literal", "v_xor_b32", "v_mov_b32", "v_lshlrev_b32 (VGPR amount)",
                                        "v_lshrrev_b32", "v_cndmask_b32_e32 (vcc)", "v_cmp_lt_u32_e32 (vcc)", "v_sub_u32", "v_min_u32", "v_mad_u32_u24",
                                        "v_lshl_or_b32", "v_and_or_b32", "v_mul_u32_u24", "v_add_u32 / v_bfe_u32 alternating",
                                        "v_add_u32 / v_lshlrev_b32 alternating", "v_addc_co_u32 (vcc in/out)",
                                        "pattern add add bfe bfe", "pattern 4 x add, 4 x bfe", "pattern 6 x add, 2 x bfe", "pattern 7 x add, 1 x bfe",
                                        "v_cndmask_b32_e64 (vcc as the mask)", "v_cmp_lt_u32_e32 vcc + v_cndmask_b32_e32 vcc pairs", "v_cmp_lt_u32_e64 s[] + v_cndmask_b32_e64 s[] pairs",
                                        "v_pk_add_f32 (two floats per lane)", "v_add_f32", "v_pk_add_f32 / v_bfe_u32 alternating"};

// One asm statement holds the whole 256-instruction body (`.rept 32` over 8 instructions): hipcc pads every boundary between
// two dependent asm statements with s_nop (it cannot see inside them), which would put a nop behind every instruction of a
// dependent chain.  Operands: %0-%7 the eight stream registers, %8 an SGPR stream, %9 a 64-bit SGPR result (v_cmp), %10 a vector
// constant, %11 a 64-bit SGPR mask (v_cndmask), %12 its low half (v_writelane's data).
#define I_AND(X) "v_and_b32 " X ", %10, " X "\n"
#define I_SHL(X) "v_lshlrev_b32 " X ", 1, " X "\n"
#define I_BFE(X) "v_bfe_u32 " X ", " X ", 1, 31\n"
#define I_CND(X) "v_cndmask_b32_e64 " X ", " X ", %10, %11\n"
#define I_ADD(X) "v_add_u32 " X ", %10, " X "\n"
#define I_MUL(X) "v_mul_lo_u32 " X ", " X ", %10\n"
#define I_ADD3(X) "v_add3_u32 " X ", " X ", %10, 1\n"
#define I_CMP(X) "v_cmp_lt_u32_e64 %9, " X ", %10\n"
#define I_RDL(X) "v_readlane_b32 %8, " X ", 3\n"
#define I_WRL(X) "v_writelane_b32 " X ", %12, 5\n"
#define I_SADD(X) "s_add_u32 %8, %8, 3\n"
#define I_ADD64(X) "v_add_u32_e64 " X ", %10, " X "\n"
#define I_ANDL(X) "v_and_b32 " X ", 0x7fff1234, " X "\n"
#define I_XOR(X) "v_xor_b32 " X ", %10, " X "\n"
#define I_MOV(X) "v_mov_b32 " X ", %10\n"
#define I_SHLV(X) "v_lshlrev_b32 " X ", %10, " X "\n"
#define I_LSHR(X) "v_lshrrev_b32 " X ", 1, " X "\n"
#define I_CND32(X) "v_cndmask_b32 " X ", " X ", %10, vcc\n"
#define I_CMP32(X) "v_cmp_lt_u32 vcc, " X ", %10\n"
#define I_SUB(X) "v_sub_u32 " X ", " X ", %10\n"
#define I_MIN(X) "v_min_u32 " X ", %10, " X "\n"
#define I_MAD24(X) "v_mad_u32_u24 " X ", " X ", %10, " X "\n"
#define I_LSHLOR(X) "v_lshl_or_b32 " X ", " X ", 1, %10\n"
#define I_ANDOR(X) "v_and_or_b32 " X ", " X ", %10, 1\n"
#define I_MUL24(X) "v_mul_u32_u24 " X ", %10, " X "\n"
#define I_ADDC(X) "v_addc_co_u32 " X ", vcc, %10, " X ", vcc\n"
#define I_CNDV64(X) "v_cndmask_b32_e64 " X ", " X ", %10, vcc\n"
#define I_CNDS64(X) "v_cndmask_b32_e64 " X ", " X ", %10, %9\n"
#define I_ADDF(X) "v_add_f32 " X ", %10, " X "\n"
#define IND8(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")
#define DEP8(I) I("%0") I("%0") I("%0") I("%0") I("%0") I("%0") I("%0") I("%0")
#define MIX8(A, B, C, D, E, F, G, H) I_AND(A) I_SHL(B) I_BFE(C) I_CND(D) I_ADD(E) I_CMP(F) I_AND(G) I_ADD3(H)
#define BODY(TEXT) asm volatile(".rept 32\n" TEXT ".endr\n" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+s"(sreg), "+s"(cmp_out) : "v"(c), "s"(m), "s"(mlo) : "scc", "vcc")
constexpr int UNROLL = 256;
template <int OP, bool DEP>
__device__ __forceinline__ void issue_all(uint32_t (&r)[8], uint32_t &sreg, uint32_t c, uint64_t m) {
  const uint32_t mlo = (uint32_t)m;
  uint64_t cmp_out = m ^ 1;
  if constexpr (OP == OP_AND) { if constexpr (DEP) BODY(DEP8(I_AND)); else BODY(IND8(I_AND)); }
  else if constexpr (OP == OP_SHL) { if constexpr (DEP) BODY(DEP8(I_SHL)); else BODY(IND8(I_SHL)); }
  else if constexpr (OP == OP_BFE) { if constexpr (DEP) BODY(DEP8(I_BFE)); else BODY(IND8(I_BFE)); }
  else if constexpr (OP == OP_CNDMASK) { if constexpr (DEP) BODY(DEP8(I_CND)); else BODY(IND8(I_CND)); }
  else if constexpr (OP == OP_ADD) { if constexpr (DEP) BODY(DEP8(I_ADD)); else BODY(IND8(I_ADD)); }
  else if constexpr (OP == OP_MUL_LO) { if constexpr (DEP) BODY(DEP8(I_MUL)); else BODY(IND8(I_MUL)); }
  else if constexpr (OP == OP_ADD3) { if constexpr (DEP) BODY(DEP8(I_ADD3)); else BODY(IND8(I_ADD3)); }
  else if constexpr (OP == OP_CMP) BODY(IND8(I_CMP));
  else if constexpr (OP == OP_SALU) BODY(DEP8(I_SADD));
  // (readlane / writelane alternate on different registers; nothing reads an SGPR a VALU instruction has just written)
  else if constexpr (OP == OP_LANE) BODY(I_RDL("%0") I_WRL("%1") I_RDL("%2") I_WRL("%3") I_RDL("%4") I_WRL("%5") I_RDL("%6") I_WRL("%7"));
  else if constexpr (OP == OP_MIX) { if constexpr (DEP) BODY(MIX8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0")); else BODY(MIX8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7")); }
  else if constexpr (OP == OP_ADD_E64) { if constexpr (DEP) BODY(DEP8(I_ADD64)); else BODY(IND8(I_ADD64)); }
  else if constexpr (OP == OP_AND_LIT) { if constexpr (DEP) BODY(DEP8(I_ANDL)); else BODY(IND8(I_ANDL)); }
  else if constexpr (OP == OP_XOR) { if constexpr (DEP) BODY(DEP8(I_XOR)); else BODY(IND8(I_XOR)); }
  else if constexpr (OP == OP_MOV) BODY(IND8(I_MOV));
  else if constexpr (OP == OP_SHL_V) { if constexpr (DEP) BODY(DEP8(I_SHLV)); else BODY(IND8(I_SHLV)); }
  else if constexpr (OP == OP_LSHR) { if constexpr (DEP) BODY(DEP8(I_LSHR)); else BODY(IND8(I_LSHR)); }
  else if constexpr (OP == OP_CND_E32) { asm volatile("s_mov_b64 vcc, %0" :: "s"(m) : "vcc"); if constexpr (DEP) BODY(DEP8(I_CND32)); else BODY(IND8(I_CND32)); }
  else if constexpr (OP == OP_CMP_E32) BODY(IND8(I_CMP32));
  else if constexpr (OP == OP_SUB) { if constexpr (DEP) BODY(DEP8(I_SUB)); else BODY(IND8(I_SUB)); }
  else if constexpr (OP == OP_MIN) { if constexpr (DEP) BODY(DEP8(I_MIN)); else BODY(IND8(I_MIN)); }
  else if constexpr (OP == OP_MAD24) { if constexpr (DEP) BODY(DEP8(I_MAD24)); else BODY(IND8(I_MAD24)); }
  else if constexpr (OP == OP_LSHL_OR) { if constexpr (DEP) BODY(DEP8(I_LSHLOR)); else BODY(IND8(I_LSHLOR)); }
  else if constexpr (OP == OP_AND_OR) { if constexpr (DEP) BODY(DEP8(I_ANDOR)); else BODY(IND8(I_ANDOR)); }
  else if constexpr (OP == OP_MUL24) { if constexpr (DEP) BODY(DEP8(I_MUL24)); else BODY(IND8(I_MUL24)); }
  else if constexpr (OP == OP_ADDC) { if constexpr (DEP) BODY(DEP8(I_ADDC)); else BODY(IND8(I_ADDC)); }
  else if constexpr (OP == OP_ADD_BFE) BODY(I_ADD("%0") I_BFE("%1") I_ADD("%2") I_BFE("%3") I_ADD("%4") I_BFE("%5") I_ADD("%6") I_BFE("%7"));
  else if constexpr (OP == OP_ADD_SHL) BODY(I_ADD("%0") I_SHL("%1") I_ADD("%2") I_SHL("%3") I_ADD("%4") I_SHL("%5") I_ADD("%6") I_SHL("%7"));
  else if constexpr (OP == OP_CND_VCC64) { asm volatile("s_mov_b64 vcc, %0" :: "s"(m) : "vcc"); BODY(IND8(I_CNDV64)); }
  else if constexpr (OP == OP_CMP_CND32) BODY(I_CMP32("%0") I_CND32("%1") I_CMP32("%2") I_CND32("%3") I_CMP32("%4") I_CND32("%5") I_CMP32("%6") I_CND32("%7"));
  else if constexpr (OP == OP_CMP_CND64) BODY(I_CMP("%0") I_CNDS64("%1") I_CMP("%2") I_CNDS64("%3") I_CMP("%4") I_CNDS64("%5") I_CMP("%6") I_CNDS64("%7"));
  else if constexpr (OP == OP_ADD_F32) BODY(IND8(I_ADDF));
  else if constexpr (OP == OP_PK_ADD || OP == OP_PK_ADD_MIX) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {__uint_as_float(r[0]), __uint_as_float(r[1])}, p1 = {__uint_as_float(r[2]), __uint_as_float(r[3])}, p2 = {__uint_as_float(r[4]), __uint_as_float(r[5])},
       p3 = {__uint_as_float(r[6]), __uint_as_float(r[7])}, cc = {1.0f, 2.0f};
    if constexpr (OP == OP_PK_ADD)
      asm volatile(".rept 32\nv_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n"
                   "v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n.endr\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
    else
      asm volatile(".rept 32\nv_pk_add_f32 %0, %0, %4\nv_bfe_u32 %5, %5, 1, 31\nv_pk_add_f32 %1, %1, %4\nv_bfe_u32 %5, %5, 1, 31\n"
                   "v_pk_add_f32 %2, %2, %4\nv_bfe_u32 %5, %5, 1, 31\nv_pk_add_f32 %3, %3, %4\nv_bfe_u32 %5, %5, 1, 31\n.endr\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc), "v"(c));
    r[0] = __float_as_uint(p0.x + p1.x + p2.x + p3.x); r[1] = __float_as_uint(p0.y + p1.y + p2.y + p3.y);
  }
  else if constexpr (OP == OP_FFSS) BODY(I_ADD("%0") I_ADD("%1") I_BFE("%2") I_BFE("%3") I_ADD("%4") I_ADD("%5") I_BFE("%6") I_BFE("%7"));
  else if constexpr (OP == OP_F4S4) BODY(I_ADD("%0") I_ADD("%1") I_ADD("%2") I_ADD("%3") I_BFE("%4") I_BFE("%5") I_BFE("%6") I_BFE("%7"));
  else if constexpr (OP == OP_F6S2) BODY(I_ADD("%0") I_ADD("%1") I_ADD("%2") I_ADD("%3") I_ADD("%4") I_ADD("%5") I_BFE("%6") I_BFE("%7"));
  else if constexpr (OP == OP_F7S1) BODY(I_ADD("%0") I_ADD("%1") I_ADD("%2") I_ADD("%3") I_ADD("%4") I_ADD("%5") I_ADD("%6") I_BFE("%7"));
  else if constexpr (OP == OP_VS) BODY(I_ADD("%0") I_SADD("") I_ADD("%1") I_SADD("") I_ADD("%2") I_SADD("") I_ADD("%3") I_SADD(""));
}

// W is a template parameter only so that the kernel NAME in a rocprofv3 trace carries the configuration
template <int OP, bool DEP, int W, int LANES>
__global__ __launch_bounds__(256) void k_issue(uint32_t *out, unsigned long long *stamps, int iters, uint64_t m) {
  extern __shared__ uint32_t pad[];
  uint32_t r[8], sreg = (uint32_t)m;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i * 40503u + 1;
  const uint32_t c = threadIdx.x * 97u + 0x55aa55u;
  if (threadIdx.x == 9999) pad[0] = 1; // (keeps the dynamic LDS allocation in the descriptor)
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) < LANES) {
    for (int it = 0; it < iters; ++it) {
      issue_all<OP, DEP>(r, sreg, c, m);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  uint32_t s = sreg;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= r[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    stamps[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 0] = t1 - t0;
    stamps[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = w1 - w0;
  }
}

struct Row { int op, dep, w, lanes; double ginst, cyc_per_inst_simd, clock_ghz, ms; };

static uint32_t *g_out;
static unsigned long long *g_stamps;
static int g_cus = 256;

template <int OP, bool DEP, int W, int LANES>
static Row run(int iters) {
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)
  const int blocks = g_cus * W;
  // exactly W workgroups per CU: 160 KiB / W of LDS each (minus a little, so that W fit and W + 1 do not)
  const int lds = W == 1 ? 96 * 1024 : (160 * 1024) / W - 1024;
  CK(hipFuncSetAttribute((const void *)k_issue<OP, DEP, W, LANES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t m = 0x5555aaaa3333ccccull;
  hipLaunchKernelGGL((k_issue<OP, DEP, W, LANES>), dim3(blocks), dim3(256), lds, 0, g_out, g_stamps, iters / 8 + 1, m); // warm-up
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_issue<OP, DEP, W, LANES>), dim3(blocks), dim3(256), lds, 0, g_out, g_stamps, iters, m);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * blocks * 4);
  CK(hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, wall = 0;
  for (int i = 0; i < blocks * 4; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
  cyc /= blocks * 4; wall /= blocks * 4;
  Row r;
  r.op = OP; r.dep = DEP; r.w = W; r.lanes = LANES; r.ms = ms;
  const double inst = (double)blocks * 4 * (double)iters * UNROLL;
  r.ginst = inst / (ms * 1e-3) / 1e9;
  r.cyc_per_inst_simd = cyc / ((double)iters * UNROLL) / W; // a SIMD's W waves run side by side for `cyc` cycles
  r.clock_ghz = cyc / wall * 0.1;                          // s_memrealtime ticks at 100 MHz
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return r;
}

static void print_row(const Row &r, bool last) {
  printf("  {\"op\": \"%s\", \"chain\": \"%s\", \"waves_per_simd\": %d, \"active_lanes\": %d, \"g_wave_inst_per_s\": %.1f, "
         "\"cycles_per_inst_per_simd\": %.3f, \"clock_ghz\": %.3f, \"ms\": %.3f}%s\n",
         op_name[r.op], r.dep ? "dependent" : "independent", r.w, r.lanes, r.ginst, r.cyc_per_inst_simd, r.clock_ghz, r.ms, last ? "" : ",");
  fflush(stdout);
}

template <int OP, bool DEP, int LANES>
static void sweep_w(std::vector<Row> &rows, int iters) {
  rows.push_back(run<OP, DEP, 1, LANES>(iters));
  rows.push_back(run<OP, DEP, 2, LANES>(iters));
  rows.push_back(run<OP, DEP, 4, LANES>(iters));
  rows.push_back(run<OP, DEP, 8, LANES>(iters));
}
template <int OP>
static void sweep(std::vector<Row> &rows, int iters) {
  sweep_w<OP, false, 64>(rows, iters);
  sweep_w<OP, true, 64>(rows, iters);
}

template <int OP, bool DEP, int LANES>
static bool one_w(int w, int iters, Row &r) {
  switch (w) {
    case 1: r = run<OP, DEP, 1, LANES>(iters); return true;
    case 2: r = run<OP, DEP, 2, LANES>(iters); return true;
    case 4: r = run<OP, DEP, 4, LANES>(iters); return true;
    case 8: r = run<OP, DEP, 8, LANES>(iters); return true;
  }
  return false;
}
template <int OP>
static bool one_op(int dep, int w, int lanes, int iters, Row &r) {
  if (lanes == 64) return dep ? one_w<OP, true, 64>(w, iters, r) : one_w<OP, false, 64>(w, iters, r);
  if (lanes == 32) return dep ? one_w<OP, true, 32>(w, iters, r) : one_w<OP, false, 32>(w, iters, r);
  if (lanes == 16) return dep ? one_w<OP, true, 16>(w, iters, r) : one_w<OP, false, 16>(w, iters, r);
  return false;
}

int main(int argc, char **argv) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
  g_cus = prop.multiProcessorCount;
  if (hipMalloc(&g_out, (size_t)g_cus * 8 * 256 * 4) != hipSuccess || hipMalloc(&g_stamps, (size_t)g_cus * 8 * 4 * 2 * 8) != hipSuccess) return 1;
  const int iters = 400;
  if (argc >= 6 && !strcmp(argv[1], "--one")) {
    const int op = atoi(argv[2]), dep = atoi(argv[3]), w = atoi(argv[4]), lanes = atoi(argv[5]);
    Row r; bool ok = false;
    switch (op) {
      case OP_AND: ok = one_op<OP_AND>(dep, w, lanes, iters, r); break;
      case OP_ADD: ok = one_op<OP_ADD>(dep, w, lanes, iters, r); break;
      case OP_MUL_LO: ok = one_op<OP_MUL_LO>(dep, w, lanes, iters, r); break;
      case OP_LANE: ok = one_op<OP_LANE>(dep, w, lanes, iters, r); break;
      case OP_SALU: ok = one_op<OP_SALU>(dep, w, lanes, iters, r); break;
      case OP_MIX: ok = one_op<OP_MIX>(dep, w, lanes, iters, r); break;
      case OP_VS: ok = one_op<OP_VS>(dep, w, lanes, iters, r); break;
    }
    if (!ok) { fprintf(stderr, "unsupported --one configuration\n"); return 2; }
    print_row(r, true);
    return 0;
  }
  std::vector<Row> rows;
  sweep<OP_AND>(rows, iters);
  sweep<OP_SHL>(rows, iters);
  sweep<OP_BFE>(rows, iters);
  sweep<OP_CNDMASK>(rows, iters);
  sweep<OP_ADD>(rows, iters);
  sweep<OP_MUL_LO>(rows, iters);
  sweep_w<OP_LANE, false, 64>(rows, iters);
  sweep<OP_ADD3>(rows, iters);
  sweep_w<OP_CMP, false, 64>(rows, iters);
  sweep_w<OP_SALU, true, 64>(rows, iters);
  sweep<OP_MIX>(rows, iters);
  sweep_w<OP_VS, false, 64>(rows, iters);
  // encoding study, 4 and 8 waves per SIMD
#define STUDY(OPX) rows.push_back(run<OPX, false, 4, 64>(iters)); rows.push_back(run<OPX, false, 8, 64>(iters));
  STUDY(OP_ADD_E64) STUDY(OP_AND_LIT) STUDY(OP_XOR) STUDY(OP_MOV) STUDY(OP_SHL_V) STUDY(OP_LSHR) STUDY(OP_CND_E32) STUDY(OP_CMP_E32) STUDY(OP_SUB)
  STUDY(OP_MIN) STUDY(OP_MAD24) STUDY(OP_LSHL_OR) STUDY(OP_AND_OR) STUDY(OP_MUL24) STUDY(OP_ADD_BFE) STUDY(OP_ADD_SHL) STUDY(OP_ADDC) STUDY(OP_PK_ADD) STUDY(OP_ADD_F32) STUDY(OP_PK_ADD_MIX) STUDY(OP_CND_VCC64) STUDY(OP_CMP_CND32) STUDY(OP_CMP_CND64) STUDY(OP_FFSS) STUDY(OP_F4S4) STUDY(OP_F6S2) STUDY(OP_F7S1)
  // partial EXEC: does a wave with its upper half (or three quarters) masked off issue faster?
  sweep_w<OP_ADD, false, 32>(rows, iters);
  sweep_w<OP_ADD, false, 16>(rows, iters);
  sweep_w<OP_MIX, false, 32>(rows, iters);
  sweep_w<OP_MIX, false, 16>(rows, iters);
  printf("{\"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"unroll\": %d, \"iters\": %d, \"rows\": [\n", prop.gcnArchName, g_cus, g_cus * 4, UNROLL, iters);
  for (size_t i = 0; i < rows.size(); ++i) print_row(rows[i], i + 1 == rows.size());
  printf("]}\n");
  return 0;
}
