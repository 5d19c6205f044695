// issue_cost.hip — measured ISSUE cost (cycles per 64-lane instruction on one SIMD) of the vector instructions the IOD
// kernels are made of, on the device at hand. bench.py's `roofline.issue` prices every VALU instruction at 4 cycles (the
// fp64 FMA rate behind the 78.6 TFLOP/s peak); the hardware counters split the fp64 instructions into ADD / MUL / FMA /
// TRANS only, so the question this answers is what a TRANS (v_rcp_f64, v_rsq_f64, v_sqrt_f64) and the division helpers
// really cost — i.e. how far below "all slots used" the kernels' 0.7 – 0.8 issue fraction actually is.
//
// Method: one workgroup on one CU; every wave runs `kReps` passes over a block of 64 INDEPENDENT instructions of one
// kind (8 accumulators x 8, inline asm so that the compiler neither removes nor reorders them) between two reads of the
// shader clock (s_memtime). 1 wave (64 threads: the issue cost seen by a lone wave, latency included) and 16 waves
// (4 per SIMD: the throughput cost — latency hidden by the other waves). cycles / instruction = (max t1 - min t0) of a
// SIMD's waves / (instructions issued on that SIMD).
//
// build + run (GPU box):  hipcc -O2 --offload-arch=gfx950 tools/issue_cost.hip -o /tmp/issue_cost && /tmp/issue_cost > out.json
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

constexpr int kReps = 512;   // passes over the 64-instruction block
constexpr int kBlock = 64;   // instructions per pass

// The 64-instruction block is ONE asm statement (8 double accumulators %0-%7, 8 32-bit ones %8-%15, x = %16, y = %17,
// i = %18): with one statement per instruction the compiler separates dependent inline-asm instructions by `s_nop 0`
// (4 cycles each), which is not what the kernels' compiled code looks like. I(d, e) = one instruction on accumulator
// d (double) / e (32-bit). Values are meaningless (timing only); dependent instructions are 8 apart except in *_chain.
#define GROUP8(I) I(0, 8) I(1, 9) I(2, 10) I(3, 11) I(4, 12) I(5, 13) I(6, 14) I(7, 15)
#define BLOCK64(I) GROUP8(I) GROUP8(I) GROUP8(I) GROUP8(I) GROUP8(I) GROUP8(I) GROUP8(I) GROUP8(I)

#define I_v_fma_f64(d, e) "v_fma_f64 %" #d ", %" #d ", %16, %17\n"
#define I_v_add_f64(d, e) "v_add_f64 %" #d ", %" #d ", %17\n"
#define I_v_mul_f64(d, e) "v_mul_f64 %" #d ", %" #d ", %16\n"
#define I_v_rcp_f64(d, e) "v_rcp_f64 %" #d ", %" #d "\n"
#define I_v_rsq_f64(d, e) "v_rsq_f64 %" #d ", %" #d "\n"
#define I_v_sqrt_f64(d, e) "v_sqrt_f64 %" #d ", %" #d "\n"
#define I_v_div_scale_f64(d, e) "v_div_scale_f64 %" #d ", vcc, %" #d ", %16, %" #d "\n"
#define I_v_div_fmas_f64(d, e) "v_div_fmas_f64 %" #d ", %" #d ", %16, %17\n"
#define I_v_div_fixup_f64(d, e) "v_div_fixup_f64 %" #d ", %" #d ", %16, %17\n"
#define I_v_ldexp_f64(d, e) "v_ldexp_f64 %" #d ", %" #d ", %18\n"
#define I_v_max_f64(d, e) "v_max_f64 %" #d ", %" #d ", %16\n"
#define I_v_rndne_f64(d, e) "v_rndne_f64 %" #d ", %" #d "\n"
#define I_v_fract_f64(d, e) "v_fract_f64 %" #d ", %" #d "\n"
#define I_v_frexp_mant_f64(d, e) "v_frexp_mant_f64 %" #d ", %" #d "\n"
#define I_v_cmp_lt_f64(d, e) "v_cmp_lt_f64 vcc, %" #d ", %16\n"
#define I_v_cmp_class_f64(d, e) "v_cmp_class_f64 vcc, %" #d ", %18\n"
#define I_v_cvt_i32_f64(d, e) "v_cvt_i32_f64 %" #e ", %" #d "\n"
#define I_v_cvt_f64_i32(d, e) "v_cvt_f64_i32 %" #d ", %" #e "\n"
#define I_v_frexp_exp_i32_f64(d, e) "v_frexp_exp_i32_f64 %" #e ", %" #d "\n"
#define I_v_cndmask_b32(d, e) "v_cndmask_b32 %" #e ", %" #e ", %18, vcc\n"
// the mask from an SGPR pair (VOP3 form), and the usual select of a double: one compare, two v_cndmask (counted as ONE unit)
#define I_v_cndmask_b32_sgpr(d, e) "v_cndmask_b32 %" #e ", %" #e ", %18, s[10:11]\n"
#define I_select_f64(d, e) "v_cmp_lt_f64 vcc, %" #d ", %16\n v_cndmask_b32 %" #e ", %" #e ", %18, vcc\n v_cndmask_b32 %" #e ", %18, %" #e ", vcc\n"
#define I_select_f64_sgpr(d, e) "v_cmp_lt_f64 s[10:11], %" #d ", %16\n v_cndmask_b32 %" #e ", %" #e ", %18, s[10:11]\n v_cndmask_b32 %" #e ", %18, %" #e ", s[10:11]\n"
// v_cndmask reading VCC some instructions after the compare that wrote it (the compiler's usual form, _e32): one compare
// per GROUP8 position 0, selects behind it; and selects alternating with fp64 arithmetic
#define I_cmp_then_7_cndmask(d, e) I_cmp7_##d(e)
#define I_cmp7_0(e) "v_cmp_lt_f64 vcc, %0, %16\n"
#define I_cmp7_1(e) "v_cndmask_b32 %" #e ", %" #e ", %18, vcc\n"
#define I_cmp7_2(e) I_cmp7_1(e)
#define I_cmp7_3(e) I_cmp7_1(e)
#define I_cmp7_4(e) I_cmp7_1(e)
#define I_cmp7_5(e) I_cmp7_1(e)
#define I_cmp7_6(e) I_cmp7_1(e)
#define I_cmp7_7(e) I_cmp7_1(e)
#define I_fma_cndmask_alternating(d, e) "v_fma_f64 %" #d ", %" #d ", %16, %17\n v_cndmask_b32 %" #e ", %" #e ", %18, vcc\n"
#define I_fma_cndmask_sgpr_alternating(d, e) "v_fma_f64 %" #d ", %" #d ", %16, %17\n v_cndmask_b32 %" #e ", %" #e ", %18, s[10:11]\n"
#define I_v_cndmask_b32_e64_vcc(d, e) "v_cndmask_b32_e64 %" #e ", %" #e ", %18, vcc\n"
#define I_cnd_cnd_fma(d, e) "v_cndmask_b32 %" #e ", %" #e ", %18, vcc\n v_cndmask_b32 %" #e ", %18, %" #e ", vcc\n v_fma_f64 %" #d ", %" #d ", %16, %17\n"
#define I_cnd_cnd_4fma(d, e) "v_cndmask_b32 %" #e ", %" #e ", %18, vcc\n v_cndmask_b32 %" #e ", %18, %" #e ", vcc\n v_fma_f64 %" #d ", %" #d ", %16, %17\n v_fma_f64 %" #d ", %" #d ", %16, %17\n v_fma_f64 %" #d ", %" #d ", %16, %17\n v_fma_f64 %" #d ", %" #d ", %16, %17\n"
#define I_cnd_mov_alternating(d, e) "v_cndmask_b32 %" #e ", %" #e ", %18, vcc\n v_mov_b32 %" #e ", %18\n"
#define I_cnd64_cnd64_fma(d, e) "v_cndmask_b32_e64 %" #e ", %" #e ", %18, vcc\n v_cndmask_b32_e64 %" #e ", %18, %" #e ", vcc\n v_fma_f64 %" #d ", %" #d ", %16, %17\n"
#define I_v_cmp_lt_f64_sgpr(d, e) "v_cmp_lt_f64 s[10:11], %" #d ", %16\n"
#define I_v_fma_f64_sgpr_operand(d, e) "v_fma_f64 %" #d ", %" #d ", s[12:13], %17\n"
#define I_v_readfirstlane_b32(d, e) "v_readfirstlane_b32 s10, %" #e "\n"
#define I_v_mov_b32(d, e) "v_mov_b32 %" #e ", %18\n"
#define I_v_add_u32(d, e) "v_add_u32 %" #e ", %" #e ", %18\n"
#define I_v_and_b32(d, e) "v_and_b32 %" #e ", %" #e ", %18\n"
#define I_v_lshlrev_b64(d, e) "v_lshlrev_b64 %" #d ", 1, %" #d "\n"
#define I_v_mul_lo_u32(d, e) "v_mul_lo_u32 %" #e ", %" #e ", %18\n"
#define I_v_mul_hi_u32(d, e) "v_mul_hi_u32 %" #e ", %" #e ", %18\n"
#define I_v_mad_u64_u32(d, e) "v_mad_u64_u32 %" #d ", vcc, %18, %18, %" #d "\n"
#define I_v_mov_b32_dpp(d, e) "v_mov_b32_dpp %" #e ", %18 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_v_fma_f32(d, e) "v_fma_f32 %" #e ", %" #e ", %18, %18\n"
#define I_v_rcp_f32(d, e) "v_rcp_f32 %" #e ", %" #e "\n"
// every instruction depends on the one before it: issue cost + result latency, what a lone wave pays on a chain
#define I_v_fma_f64_chain(d, e) "v_fma_f64 %0, %0, %16, %17\n"
#define I_v_add_f64_chain(d, e) "v_add_f64 %0, %0, %17\n"
#define I_v_rcp_f64_chain(d, e) "v_rcp_f64 %0, %0\n"
// the compiler's own fp64 division (contraction off, as the library is built): v_div_scale x2, v_rcp, 5 fma, v_mul,
// v_div_fmas, v_div_fixup — cycles per DIVISION, for the weight of a `/` in the kernels
#define I_division(d, e)                                                                                     \
  "v_div_scale_f64 v[100:101], s[10:11], %16, %16, %" #d "\n v_div_scale_f64 v[102:103], vcc, %" #d ", %16, %" #d "\n"  \
  "v_rcp_f64 v[104:105], v[100:101]\n v_fma_f64 v[106:107], -v[100:101], v[104:105], 1.0\n"                 \
  "v_fma_f64 v[104:105], v[104:105], v[106:107], v[104:105]\n v_fma_f64 v[106:107], -v[100:101], v[104:105], 1.0\n" \
  "v_fma_f64 v[104:105], v[104:105], v[106:107], v[104:105]\n v_mul_f64 v[106:107], v[102:103], v[104:105]\n" \
  "v_fma_f64 v[100:101], -v[100:101], v[106:107], v[102:103]\n s_nop 1\n"                                    \
  "v_div_fmas_f64 v[100:101], v[100:101], v[104:105], v[106:107]\n v_div_fixup_f64 %" #d ", v[100:101], %16, %" #d "\n"

#define KERNEL(NAME)                                                                                    \
  template <int kBlocks>                                                                                \
  __global__ void __launch_bounds__(1024) k_##NAME(unsigned long long* t, double* sink, double seed) { \
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,    \
           a6 = a0 + 6, a7 = a0 + 7;                                                                   \
    int j0 = (int)threadIdx.x, j1 = j0 + 1, j2 = j0 + 2, j3 = j0 + 3, j4 = j0 + 4, j5 = j0 + 5, j6 = j0 + 6, \
        j7 = j0 + 7;                                                                                   \
    double x = 1.0000001, y = 1e-9;                                                                    \
    int i = 0x3f800001;                                                                                \
    asm volatile("" : "+v"(x), "+v"(y), "+v"(i));                                                      \
    __syncthreads();                                                                                   \
    unsigned long long t0, t1;                                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                         \
    _Pragma("unroll 1") for (int r = 0; r < kReps; ++r) {                                              \
      _Pragma("unroll") for (int b = 0; b < kBlocks; ++b)                                              \
      asm volatile(BLOCK64(I_##NAME)                                                                   \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),   \
                     "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7)    \
                   : "v"(x), "v"(y), "v"(i)                                                            \
                   : "vcc", "s10", "s11", "s12", "s13", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"); \
    }                                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                         \
    if ((threadIdx.x & 63) == 0) {                                                                     \
      t[2 * (threadIdx.x >> 6)] = t0;                                                                  \
      t[2 * (threadIdx.x >> 6) + 1] = t1;                                                              \
    }                                                                                                  \
    sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (j0 + j1 + j2 + j3 + j4 + j5 + j6 + j7); \
  }

#define ALL(X)                                                                                                   \
  X(v_fma_f64) X(v_add_f64) X(v_mul_f64) X(v_rcp_f64) X(v_rsq_f64) X(v_sqrt_f64) X(v_div_scale_f64)              \
  X(v_div_fmas_f64) X(v_div_fixup_f64) X(division) X(v_ldexp_f64) X(v_max_f64) X(v_rndne_f64) X(v_fract_f64)     \
  X(v_frexp_mant_f64) X(v_cmp_lt_f64) X(v_cmp_class_f64) X(v_cvt_i32_f64) X(v_cvt_f64_i32)                       \
  X(v_frexp_exp_i32_f64) X(v_cndmask_b32) X(v_cndmask_b32_sgpr) X(select_f64) X(select_f64_sgpr)               \
  X(v_cndmask_b32_e64_vcc) X(cnd_cnd_fma) X(cnd_cnd_4fma) X(cnd_mov_alternating) X(cnd64_cnd64_fma)            \
  X(cmp_then_7_cndmask) X(fma_cndmask_alternating) X(fma_cndmask_sgpr_alternating)                              \
  X(v_cmp_lt_f64_sgpr) X(v_fma_f64_sgpr_operand) X(v_readfirstlane_b32) X(v_mov_b32) X(v_add_u32) X(v_and_b32) X(v_lshlrev_b64)               \
  X(v_mul_lo_u32) X(v_mul_hi_u32) X(v_mad_u64_u32) X(v_mov_b32_dpp) X(v_fma_f32) X(v_rcp_f32)                    \
  X(v_fma_f64_chain) X(v_add_f64_chain) X(v_rcp_f64_chain)

ALL(KERNEL)

typedef void (*kern_t)(unsigned long long*, double*, double);
struct Entry {
  const char* name;
  kern_t fn1, fn2;  // one / two 64-instruction blocks per loop pass
};
#define E(NAME) {#NAME, k_##NAME<1>, k_##NAME<2>},
static const Entry kEntries[] = {ALL(E)};

// Cycles of one SIMD from the earliest start to the latest end of its waves (fastest of five launches). Waves of one
// workgroup are spread evenly over the CU's four SIMDs, so with `waves` >= 4 each SIMD issues waves / 4 of them.
static double span(kern_t fn, int waves, unsigned long long* d_t, double* d_sink) {
  std::vector<unsigned long long> t(2 * waves);
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {  // the first launch loads the code
    hipLaunchKernelGGL(fn, dim3(1), dim3(64 * waves), 0, 0, d_t, d_sink, 3.0 + rep);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(t.data(), d_t, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost));
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < waves; ++w) {
      if (t[2 * w] < lo) lo = t[2 * w];
      if (t[2 * w + 1] > hi) hi = t[2 * w + 1];
    }
    if ((double)(hi - lo) < best) best = (double)(hi - lo);
  }
  return best;
}
// {average incl. the loop's own scalar instructions and branch, marginal = (two blocks per pass - one block per pass) / block}:
// the marginal figure is the instruction's own cost
static void cost(const Entry& e, int waves, unsigned long long* d_t, double* d_sink, double& avg, double& marginal) {
  const int simds = waves < 4 ? waves : 4;
  const double units_per_simd = (double)kReps * kBlock * waves / simds;
  const double t1 = span(e.fn1, waves, d_t, d_sink), t2 = span(e.fn2, waves, d_t, d_sink);
  avg = t1 / units_per_simd;
  marginal = (t2 - t1) / units_per_simd;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  unsigned long long* d_t;
  double* d_sink;
  CHECK(hipMalloc(&d_t, sizeof(unsigned long long) * 64));
  CHECK(hipMalloc(&d_sink, sizeof(double) * 1024));
  printf("{\n \"gcn_arch\": \"%s\",\n", prop.gcnArchName);
  printf(" \"method\": \"one workgroup on one CU; every wave runs %d loop passes over one or two blocks of %d instructions of one kind "
         "(8 accumulators, dependent instructions 8 apart) between two s_memtime stamps; [average with one block per pass, "
         "marginal = (two blocks - one block) / %d] in shader cycles per instruction (per unit for division / select_*) per SIMD, "
         "fastest of 5 launches\",\n", kReps, kBlock, kBlock);
  printf(" \"cycles_per_instruction\": {\n");
  const int n = (int)(sizeof(kEntries) / sizeof(kEntries[0]));
  for (int k = 0; k < n; ++k) {
    double a1, m1, a2, m2, a3, m3, a4, m4;
    cost(kEntries[k], 1, d_t, d_sink, a1, m1);
    cost(kEntries[k], 8, d_t, d_sink, a2, m2);
    cost(kEntries[k], 12, d_t, d_sink, a3, m3);
    cost(kEntries[k], 16, d_t, d_sink, a4, m4);
    printf("  \"%s\": {\"one_wave\": [%.3f, %.3f], \"two_waves_per_simd\": [%.3f, %.3f], \"three_waves_per_simd\": [%.3f, %.3f], "
           "\"four_waves_per_simd\": [%.3f, %.3f]}%s\n",
           kEntries[k].name, a1, m1, a2, m2, a3, m3, a4, m4, k + 1 < n ? "," : "");
    fflush(stdout);
  }
  printf(" }\n}\n");
  CHECK(hipFree(d_t));
  CHECK(hipFree(d_sink));
  return 0;
}
