// Diagnostic (round 6): which VOP3P operand-select forms lose their lo half in lanes 48-63 next to which co-running kernel?
// Follow-up of pk_probe.hip (the (mean, rstd) chain of the LayerNorm'd-residual epilogue): ONE packed instruction per victim
// kernel, every op_sel / op_sel_hi placement, streamed over a large array; each run beside an aggressor on another stream is
// compared bit for bit with the idle run.  Aggressors: library GEMMs (rocBLAS: fp16 / f32) and hand-made MFMA loops of several
// shapes (instruction variant, broadcast modifiers, accumulators in AGPRs), an LDS-heavy kernel, a high-priority wave kernel.
//   hipcc --offload-arch=gfx950 -O3 -DWITH_ROCBLAS diag/pk_opsel_probe.hip -lrocblas -o pk_opsel_probe && ./pk_opsel_probe [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef WITH_ROCBLAS
#include <rocblas/rocblas.h>
#endif

typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VICTIMS(X)                                                                              \
  X(0, "v_pk_mul_f32 (no modifiers)", "v_pk_mul_f32 %0, %1, %2")                                  \
  X(1, "v_pk_mul_f32 op_sel:[0,1]", "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]")                       \
  X(2, "v_pk_mul_f32 op_sel:[1,0]", "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]")                       \
  X(3, "v_pk_mul_f32 op_sel:[1,1]", "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]")                       \
  X(4, "v_pk_mul_f32 op_sel_hi:[1,0]", "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]")                 \
  X(5, "v_pk_mul_f32 op_sel_hi:[0,1]", "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")                 \
  X(6, "v_pk_mul_f32 op_sel_hi:[0,0]", "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,0]")                 \
  X(7, "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]") \
  X(8, "v_pk_add_f32 op_sel:[0,1]", "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]")                       \
  X(9, "v_pk_add_f32 op_sel:[1,0]", "v_pk_add_f32 %0, %1, %2 op_sel:[1,0]")                       \
  X(10, "v_pk_add_f32 op_sel_hi:[1,0] neg", "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]") \
  X(11, "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]")              \
  X(12, "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]")              \
  X(13, "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]")              \
  X(14, "v_pk_fma_f32 (no modifiers)", "v_pk_fma_f32 %0, %1, %2, %3")                             \
  X(15, "v_pk_mov_b32 op_sel:[1,0]", "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]")                      \
  X(16, "v_pk_mov_b32 op_sel:[0,1]", "v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]")                      \
  X(17, "v_pk_mul_f32 op_sel:[0,1], s_nop 4 before", "s_nop 4\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1]") \
  X(18, "v_pk_mul_f32 op_sel:[0,1], s_nop 4 after", "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 4") \
  /* census: every other modifier-bearing packed-f32 form the shipped library contains (llvm-objdump of libltr_hip.so, round 6) */ \
  X(19, "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]")         \
  X(20, "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]")         \
  X(21, "v_pk_fma_f32 op_sel_hi:[1,0,1] neg:[1,0,0]", "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]") \
  X(22, "v_pk_fma_f32 neg:[0,0,1]", "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]") \
  X(23, "v_pk_fma_f32 neg:[1,0,0]", "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]") \
  X(24, "v_pk_add_f32 neg:[0,1]", "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")             \
  X(25, "v_pk_add_f32 v, v, 1.0 op_sel_hi:[1,0]", "v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0]")      \
  X(26, "v_pk_mul_f32 v, s, v op_sel_hi:[0,1]", "v_pk_mul_f32 %0, %4, %2 op_sel_hi:[0,1]")         \
  X(27, "v_pk_mul_f32 v, v, s op_sel_hi:[1,0]", "v_pk_mul_f32 %0, %1, %4 op_sel_hi:[1,0]")         \
  X(28, "v_pk_fma_f32 v, v, s, v op_sel_hi:[1,0,1]", "v_pk_fma_f32 %0, %1, %4, %3 op_sel_hi:[1,0,1]") \
  /* siblings of the hazardous form the lint also rejects */                                       \
  X(29, "v_pk_fma_f32 op_sel:[0,1,1]", "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1]")               \
  X(30, "v_pk_fma_f32 op_sel:[1,1,0]", "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0]")               \
  X(31, "v_pk_mul_f32 v, v, s op_sel:[0,1]", "v_pk_mul_f32 %0, %1, %4 op_sel:[0,1]")
constexpr int N_VICTIMS = 32;

template <int V>
__global__ void __launch_bounds__(256) victim(const v2f* __restrict__ a, const v2f* __restrict__ b, const v2f* __restrict__ c, v2f* __restrict__ o,
                                              size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    v2f x = a[i], y = b[i], z = c[i & 4095], d = {0.f, 0.f};
    const v2f sc = {1.25f, -0.75f};                 // (uniform: an SGPR pair operand)
#define X(ID, NAME, ASM) if (V == ID) asm volatile(ASM : "=&v"(d) : "v"(x), "v"(y), "v"(z), "s"(sc));
    VICTIMS(X)
#undef X
    o[i] = d;
  }
}
// 32-bit-operand forms (packed f16, mixed precision, dot): the same question for op_sel on 16-bit halves
#define VICTIMS32(X)                                                                             \
  X(100, "v_pk_mul_f16 op_sel:[0,1]", "v_pk_mul_f16 %0, %1, %2 op_sel:[0,1]")                     \
  X(101, "v_pk_mul_f16 op_sel:[1,0]", "v_pk_mul_f16 %0, %1, %2 op_sel:[1,0]")                     \
  X(102, "v_pk_add_f16 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]") \
  X(103, "v_pk_fma_f16 op_sel:[0,1,0]", "v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0]")             \
  X(104, "v_fma_mix_f32 op_sel:[0,1,0] op_sel_hi:[1,1,0]", "v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]") \
  X(105, "v_dot2c_f32_f16 (no modifiers)", "v_mov_b32 %0, %3\n\tv_dot2c_f32_f16 %0, %1, %2")        \
  X(106, "v_pk_mul_lo_u16 op_sel:[0,1]", "v_pk_mul_lo_u16 %0, %1, %2 op_sel:[0,1]")               \
  X(107, "v_mul_f32 (VOP3, no modifiers)", "v_mul_f32_e64 %0, %1, %2")                          \
  X(108, "v_cvt_f32_f16_sdwa src0_sel:WORD_1", "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1") \
  X(109, "v_cvt_f16_f32_sdwa dst_sel:WORD_1", "v_mov_b32 %0, %2\n\ts_nop 1\n\tv_cvt_f16_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD") \
  X(110, "v_fma_mixlo_f16 -v op_sel_hi:[0,0,1]", "v_mov_b32 %0, 0\n\ts_nop 1\n\tv_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]") \
  X(111, "v_fma_mixhi_f16 -v op_sel_hi:[0,0,1]", "v_mov_b32 %0, 0\n\ts_nop 1\n\tv_fma_mixhi_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]") \
  X(112, "v_fma_mix_f32 op_sel_hi:[0,1,0]", "v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]")     \
  X(113, "v_max3_f32 v, |v|, |v|", "v_max3_f32 %0, %1, |%2|, |%3|")
constexpr int N_VICTIMS32 = 14;
template <int V>
__global__ void __launch_bounds__(256) victim32(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ o,
                                                size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float x = a[i], y = b[i], z = c[i & 8191], d = 0.f;
#define X(ID, NAME, ASM) if (V == ID) asm volatile(ASM : "=&v"(d) : "v"(x), "v"(y), "v"(z));
    VICTIMS32(X)
#undef X
    o[i] = d;
  }
}
static const char* victim_name(int v) {
#define X(ID, NAME, ASM) if (v == ID) return NAME;
  VICTIMS32(X)
#undef X
#define X(ID, NAME, ASM) if (v == ID) return NAME;
  VICTIMS(X)
#undef X
  return "?";
}
static void launch_victim(int v, const v2f* a, const v2f* b, const v2f* c, v2f* o, size_t n, hipStream_t s) {
#define X(ID, NAME, ASM) if (v == ID) victim32<ID><<<2048, 256, 0, s>>>((const float*)a, (const float*)b, (const float*)c, (float*)o, 2 * n);
  VICTIMS32(X)
#undef X
#define X(ID, NAME, ASM) if (v == ID) victim<ID><<<2048, 256, 0, s>>>(a, b, c, o, n);
  VICTIMS(X)
#undef X
}

// ---- aggressors
template <int KIND>
__global__ void __launch_bounds__(256) aggr_mfma(float* sink, int iters) {
  f16x8 a8, b8; f16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8[i] = (_Float16)(0.5f - i * 0.01f); }
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
  f32x4 c4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16 c16[2];
  for (int i = 0; i < 16; ++i) { c16[0][i] = 0.f; c16[1][i] = 0.f; }
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) for (int k = 0; k < 4; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c4[k], 0, 0, 0);
    if (KIND == 1) for (int k = 0; k < 2; ++k) c16[k] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c16[k], 0, 0, 0);
    if (KIND == 2) for (int k = 0; k < 2; ++k) c16[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c16[k], 0, 0, 0);
    if (KIND == 3) for (int k = 0; k < 4; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c4[k], 0, 0, 0);
    if (KIND == 4) for (int k = 0; k < 2; ++k) c16[k] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c16[k], 1, 1, 0);      // cbsz 1 abid 1
    if (KIND == 5) for (int k = 0; k < 4; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c4[k], 2, 3, 0);       // cbsz 2 abid 3
    if (KIND == 7) {  // gfx950 bf16 forms
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      bf16x8 p, q;
      for (int i = 0; i < 8; ++i) { p[i] = (__bf16)(float)a8[i]; q[i] = (__bf16)(float)b8[i]; }
      for (int k = 0; k < 2; ++k) c16[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, c16[k], 0, 0, 0);
      for (int k = 0; k < 2; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, c4[k], 0, 0, 0);
    }
    if (KIND == 9 || KIND == 10) {  // bf16, one shape at a time
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      bf16x8 p, q;
      for (int i = 0; i < 8; ++i) { p[i] = (__bf16)(float)a8[i]; q[i] = (__bf16)(float)b8[i]; }
      if (KIND == 9) for (int k = 0; k < 2; ++k) c16[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, c16[k], 0, 0, 0);
      else for (int k = 0; k < 4; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, c4[k], 0, 0, 0);
    }
    if (KIND == 11) {  // f16, both shapes back to back
      for (int k = 0; k < 2; ++k) c16[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c16[k], 0, 0, 0);
      for (int k = 0; k < 2; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c4[k], 0, 0, 0);
    }
    if (KIND == 12) {  // the bf16 CONVERSIONS of KIND 7 without any MFMA (what else that aggressor executes)
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      bf16x8 p;
      for (int i = 0; i < 8; ++i) p[i] = (__bf16)((float)a8[i] + (float)it);
      for (int i = 0; i < 4; ++i) c4[0][i] += (float)p[i] + (float)p[i + 4];
    }
    if (KIND == 8) {  // 32x32x16 f16 fed from LDS-like register traffic: four accumulators, VALU between
      for (int k = 0; k < 2; ++k) c16[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c16[k], 0, 0, 0);
      a8[it & 7] += (_Float16)0.001f; b8[(it + 3) & 7] -= (_Float16)0.001f;
    }
    if (KIND == 6)    // accumulators in AGPRs
      asm volatile("v_mfma_f32_32x32x8_f16 a[0:15], %0, %1, a[0:15]\n\tv_mfma_f32_32x32x8_f16 a[16:31], %0, %1, a[16:31]" ::"v"(a4), "v"(b4)
                   : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20",
                     "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
  }
  float v = c4[0][0] + c4[1][1] + c4[2][2] + c4[3][3] + c16[0][5] + c16[1][7];
  if (v == 1.2345e-30f) sink[0] = v;
}
__global__ void __launch_bounds__(256) aggr_lds(float* sink, int iters) {
  extern __shared__ float sm[];
  for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = i;
  __syncthreads();
  float4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const float4 v = *reinterpret_cast<const float4*>(sm + (((threadIdx.x * 4 + it * 132) & 16383) & ~3));
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    if ((it & 63) == 0) { __builtin_amdgcn_s_setprio(3); __syncthreads(); __builtin_amdgcn_s_setprio(0); }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) sink[0] = acc.x;
}
// many VGPRs + packed-f16 / dot instructions (what a GEMM's conversion epilogue uses)
__global__ void __launch_bounds__(256) aggr_pkf16(float* sink, int iters) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 a = {(_Float16)1.001f, (_Float16)0.999f}, b = {(_Float16)0.5f, (_Float16)0.25f}, c = {(_Float16)0.f, (_Float16)0.f};
  for (int it = 0; it < iters; ++it) asm volatile("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\tv_pk_mul_f16 %0, %0, %1 op_sel:[1,0]" : "+v"(c) : "v"(a), "v"(b));
  if ((float)c[0] == 1.2345e-30f) sink[0] = (float)c[1];
}

// pk_opsel_probe [rounds] [quick | hand]     quick: the known-bad form and its safe twins beside the library GEMMs only; hand: only
// part 2 (the known-bad form beside the hand-made aggressors, 4 x rounds runs each).
// Exit code: 0 = every form computed the same beside every aggressor, 1 = only forms the ISA lint rejects differed (the expected
// state of this hardware), 2 = a form the library is allowed to contain differed (the lint's rule is no longer sufficient).
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 4;
  const bool quick = argc > 2 && !strcmp(argv[2], "quick");
  const bool hand = argc > 2 && !strcmp(argv[2], "hand");
  long bad_known = 0, bad_other = 0;
  const size_t n = (size_t)16 << 20;            // 16 M v2f per array (128 MB each)
  std::vector<v2f> ha(n), hb(n), hc(4096);
  srand(11);
  auto fr = [] { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; };
  for (auto& v : ha) v = v2f{fr() * 2, fr() * 2};
  for (auto& v : hb) v = v2f{1.f + fr() * 0.5f, 3.f + fr()};
  for (auto& v : hc) v = v2f{fr(), 10.f + fr()};
  v2f *da, *db, *dc, *dout; float* sink;
  (void)hipMalloc(&da, n * 8); (void)hipMalloc(&db, n * 8); (void)hipMalloc(&dc, 4096 * 8); (void)hipMalloc(&dout, n * 8); (void)hipMalloc(&sink, 64);
  (void)hipMemcpy(da, ha.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb.data(), n * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dc, hc.data(), 4096 * 8, hipMemcpyHostToDevice);
  hipStream_t sa, sb;
  (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  (void)hipFuncSetAttribute((const void*)aggr_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
#ifdef WITH_ROCBLAS
  rocblas_handle rh; rocblas_create_handle(&rh); rocblas_set_stream(rh, sb);
  const int G = 2048;
  void *gz, *gc_, *gf;
  (void)hipMalloc(&gz, (size_t)G * G * 2); (void)hipMalloc(&gc_, (size_t)G * G * 4); (void)hipMalloc(&gf, (size_t)G * G * 4);
  (void)hipMemset(gz, 0, (size_t)G * G * 2); (void)hipMemset(gf, 0, (size_t)G * G * 4);
#endif
  const char* aggr_names[] = {"library fp16 GEMM", "library f32 GEMM", "library bf16 GEMM", "mfma 16x16x32 f16", "mfma 32x32x8 f16", "mfma 32x32x16 f16", "mfma 16x16x16 f16",
                              "mfma 32x32x8 cbsz 1 abid 1", "mfma 16x16x16 cbsz 2 abid 3", "mfma 32x32x8, AGPR accumulators", "LDS + s_setprio + barriers",
                              "packed-f16 VALU with op_sel", "mfma 32x32x16 + 16x16x32 bf16", "mfma 32x32x16 f16 + VALU, 8192 blocks", "mfma 32x32x16 f16, 8192 blocks",
                              "mfma 32x32x16 bf16 only", "mfma 16x16x32 bf16 only", "mfma 32x32x16 + 16x16x32 f16", "bf16 conversions, no MFMA"};
  const int n_aggr = 19;
  auto aggress = [&](int ag) {
    const float one = 1.f, nul = 0.f;
    (void)one; (void)nul;
#ifdef WITH_ROCBLAS
    if (ag == 0) for (int k = 0; k < 30; ++k)
      rocblas_gemm_ex(rh, rocblas_operation_none, rocblas_operation_none, G, G, G, &one, gz, rocblas_datatype_f16_r, G, gz, rocblas_datatype_f16_r, G, &nul,
                      gc_, rocblas_datatype_f16_r, G, gc_, rocblas_datatype_f16_r, G, rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    if (ag == 1) for (int k = 0; k < 8; ++k)
      rocblas_gemm_ex(rh, rocblas_operation_none, rocblas_operation_none, G, G, G, &one, gf, rocblas_datatype_f32_r, G, gf, rocblas_datatype_f32_r, G, &nul,
                      gc_, rocblas_datatype_f32_r, G, gc_, rocblas_datatype_f32_r, G, rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
#endif
#ifdef WITH_ROCBLAS
    if (ag == 2) for (int k = 0; k < 30; ++k)
      rocblas_gemm_ex(rh, rocblas_operation_none, rocblas_operation_none, G, G, G, &one, gz, rocblas_datatype_bf16_r, G, gz, rocblas_datatype_bf16_r, G, &nul,
                      gc_, rocblas_datatype_bf16_r, G, gc_, rocblas_datatype_bf16_r, G, rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
#endif
    if (ag == 3) aggr_mfma<0><<<2048, 256, 0, sb>>>(sink, 20000);
    if (ag == 4) aggr_mfma<1><<<2048, 256, 0, sb>>>(sink, 10000);
    if (ag == 5) aggr_mfma<2><<<2048, 256, 0, sb>>>(sink, 10000);
    if (ag == 6) aggr_mfma<3><<<2048, 256, 0, sb>>>(sink, 20000);
    if (ag == 7) aggr_mfma<4><<<2048, 256, 0, sb>>>(sink, 10000);
    if (ag == 8) aggr_mfma<5><<<2048, 256, 0, sb>>>(sink, 20000);
    if (ag == 9) aggr_mfma<6><<<2048, 256, 0, sb>>>(sink, 10000);
    if (ag == 10) aggr_lds<<<512, 256, 65536, sb>>>(sink, 60000);
    if (ag == 11) aggr_pkf16<<<2048, 256, 0, sb>>>(sink, 200000);
    if (ag == 12) aggr_mfma<7><<<2048, 256, 0, sb>>>(sink, 6000);
    if (ag == 13) aggr_mfma<8><<<8192, 256, 0, sb>>>(sink, 2500);
    if (ag == 14) aggr_mfma<2><<<8192, 256, 0, sb>>>(sink, 2500);
    if (ag == 15) aggr_mfma<9><<<2048, 256, 0, sb>>>(sink, 8000);
    if (ag == 16) aggr_mfma<10><<<2048, 256, 0, sb>>>(sink, 12000);
    if (ag == 17) aggr_mfma<11><<<2048, 256, 0, sb>>>(sink, 6000);
    if (ag == 18) aggr_mfma<12><<<2048, 256, 0, sb>>>(sink, 20000);
  };
  std::vector<v2f> ref(n), got(n);
  auto run = [&](int v, int ag, long* by_half, long* by_quarter) -> long {
    (void)hipDeviceSynchronize();
    if (ag >= 0) aggress(ag);
    (void)hipMemsetAsync(dout, 0xff, n * 8, sa);
    launch_victim(v, da, db, dc, dout, n, sa);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(got.data(), dout, n * 8, hipMemcpyDeviceToHost);
    long e = 0;
    if (memcmp(got.data(), ref.data(), n * 8))
      for (size_t i = 0; i < n; ++i)
        for (int h = 0; h < 2; ++h) { const float g = got[i][h], w = ref[i][h]; if (memcmp(&g, &w, 4)) { ++e; ++by_half[h]; ++by_quarter[(i & 63) >> 4]; } }
    return e;
  };
  // part 1: every victim form beside the two library GEMMs
  for (int vi = 0; vi < N_VICTIMS + N_VICTIMS32; ++vi) {
    const int v = vi < N_VICTIMS ? vi : 100 + vi - N_VICTIMS;
    if (hand) continue;
    if (quick && !(v == 1 || v == 2 || v == 3 || v == 4 || v == 8 || v == 12)) continue;
    (void)hipMemsetAsync(dout, 0xff, n * 8, sa);
    launch_victim(v, da, db, dc, dout, n, sa);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(ref.data(), dout, n * 8, hipMemcpyDeviceToHost);
    for (int ag = 0; ag < 3; ++ag) {
      long runs = 0, elems = 0, bh[2] = {0, 0}, bq[4] = {0, 0, 0, 0};
      for (int it = 0; it < rounds; ++it) { const long e = run(v, ag, bh, bq); runs += e != 0; elems += e; }
      printf("%-48s beside %-18s: %ld of %d runs differ, %8ld values (lo %ld hi %ld | lanes 0-15 %ld 16-31 %ld 32-47 %ld 48-63 %ld)\n", victim_name(v), aggr_names[ag],
             runs, rounds, elems, bh[0], bh[1], bq[0], bq[1], bq[2], bq[3]);
      const bool lint_rejects = v == 1 || v == 7 || v == 8 || v == 12 || v == 17 || v == 18 || v == 29 || v == 31;      // op_sel:[0,1...] on packed f32
      (lint_rejects ? bad_known : bad_other) += runs;
      fflush(stdout);
    }
  }
  // part 2: the known-bad form beside every hand-made aggressor
  if (!quick) {
    const int v = 1;
    (void)hipMemsetAsync(dout, 0xff, n * 8, sa);
    launch_victim(v, da, db, dc, dout, n, sa);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(ref.data(), dout, n * 8, hipMemcpyDeviceToHost);
    for (int ag = 3; ag < n_aggr; ++ag) {
      long runs = 0, elems = 0, bh[2] = {0, 0}, bq[4] = {0, 0, 0, 0};
      for (int it = 0; it < 4 * rounds; ++it) { const long e = run(v, ag, bh, bq); runs += e != 0; elems += e; }
      printf("%-48s beside %-38s: %ld of %d runs differ, %8ld values (lo %ld hi %ld | lanes 48-63 %ld)\n", victim_name(v), aggr_names[ag], runs, 4 * rounds, elems, bh[0], bh[1],
             bq[3]);
      fflush(stdout);
    }
  }
  printf("pk_opsel_probe: forms the ISA lint rejects differed in %ld runs, every other form in %ld\n", bad_known, bad_other);
  return bad_other ? 2 : (bad_known ? 1 : 0);
}
