// valu_ops.hip -- SIMD issue cost of single VALU instructions on gfx950 at 4 waves per SIMD, by opcode and by
// which lanes are active.  Every wave runs kIter x 32 independent instructions of one kind (inline asm, 8 registers
// round-robin); one 1024-thread workgroup per CU (a 96 KB LDS request keeps a second one off).  Cost =
// kernel time x in-kernel clock x SIMDs / wave-instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIter = 16384;
extern __shared__ float g_pad[];

#define OP8(INS)                                                                                  \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                            \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)   \
               : "v"(a), "v"(b), "s"(sm) : "vcc", "v20", "v21", "v22", "v23", "v24", "v25", "s20", "s21", "s22", "s23")

#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_ADDF(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define I_MULF(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define I_MAXF(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define I_CNDM(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_CMP(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define I_SQRT(k) "v_sqrt_f32 %" #k ", %" #k "\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define I_MULHI(k) "v_mul_hi_u32 %" #k ", %" #k ", %8\n"
#define I_CVTUB(k) "v_cvt_f32_ubyte0 %" #k ", %" #k "\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 5\n"
#define I_FMAC(k) "v_fmac_f32 %" #k ", %8, %9\n"
#define I_PKFMA(k) "v_pk_fma_f32 v[10:11], v[10:11], v[12:13], v[14:15]\n"
#define I_READLANE(k) "v_readfirstlane_b32 s20, %" #k "\n"
#define I_FMA_S(k) "v_fma_f32 %" #k ", %10, %" #k ", %9\n"
#define I_MUL_S(k) "v_mul_f32 %" #k ", %10, %" #k "\n"
#define I_MUL_K(k) "v_mul_f32 %" #k ", 2.0, %" #k "\n"
#define I_MUL_LIT(k) "v_mul_f32 %" #k ", 0x3f9d70a4, %" #k "\n"
#define I_SUBF(k) "v_sub_f32 %" #k ", %" #k ", %8\n"
#define I_MINF(k) "v_min_f32 %" #k ", %" #k ", %8\n"
#define I_MADU64(k) "v_mad_u64_u32 v[20:21], s[20:21], %" #k ", %8, v[20:21]\n"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %8\n"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define I_LSHR(k) "v_lshrrev_b32 %" #k ", 3, %" #k "\n"
#define I_CVTFU(k) "v_cvt_f32_u32 %" #k ", %" #k "\n"
#define I_FLOOR(k) "v_floor_f32 %" #k ", %" #k "\n"
#define I_CNDM_S(k) "v_cndmask_b32 %" #k ", %" #k ", %8, s[22:23]\n"
#define I_FMAMK(k) "v_fmamk_f32 %" #k ", %" #k ", 0x3f9d70a4, %8\n"
#define I_MIN3(k) "v_min3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MULF64(k) "v_mul_f64 v[20:21], v[20:21], v[22:23]\n"
#define I_CMP_S(k) "v_cmp_lt_f32 s[22:23], %" #k ", %8\n"
#define I_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define I_PKFMA2(k) "v_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[24:25]\n"
#define I_PKMUL(k) "v_pk_mul_f32 v[20:21], v[20:21], v[22:23]\n"
#define I_CNDM_E64V(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, vcc\n"
#define I_CMPCNDM(k) "v_cmp_lt_f32 vcc, %" #k ", %9\n v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_CMPCNDM_S(k) "v_cmp_lt_f32 s[22:23], %" #k ", %9\n v_cndmask_b32 %" #k ", %" #k ", %8, s[22:23]\n"
#define I_CNDM_K(k) "v_cndmask_b32 %" #k ", 0, %" #k ", vcc\n"
#define I_ADDC(k) "v_addc_co_u32 %" #k ", vcc, %" #k ", %8, vcc\n"
#define I_FMAMIX(k) "v_fma_mix_f32 %" #k ", %8, %9, %" #k " op_sel_hi:[1,0,0]\n"
#define I_FMAMIXH(k) "v_fma_mix_f32 %" #k ", %8, %9, %" #k " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define I_ALIGNBIT(k) "v_alignbit_b32 %" #k ", %" #k ", %8, 31\n"
#define I_CVTF16(k) "v_cvt_f32_f16 %" #k ", %" #k "\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define I_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_OR3(k) "v_or3_b32 %" #k ", %" #k ", %8, %9\n"
#define I_MBCNT(k) "v_mbcnt_lo_u32_b32 %" #k ", %10, %" #k "\n"

template <int OP>
__global__ void __launch_bounds__(1024) k_op(float* out, unsigned long long* stamps, float a, float b, unsigned long long lanes) {
  if (a == 12345.f) g_pad[threadIdx.x] = b;
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const unsigned sm = 0xffffffffu;
  asm volatile("s_mov_b64 s[22:23], 0x55\n s_mov_b64 vcc, 0x33\n" ::: "s22", "s23", "vcc");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if ((lanes >> (threadIdx.x & 63)) & 1ull) {
    for (int i = 0; i < kIter; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (OP == 0) OP8(I_FMA);
        if (OP == 1) OP8(I_ADDF);
        if (OP == 2) OP8(I_MULF);
        if (OP == 3) OP8(I_MAXF);
        if (OP == 4) OP8(I_ADDU);
        if (OP == 5) OP8(I_MOV);
        if (OP == 6) OP8(I_AND);
        if (OP == 7) OP8(I_LSHL);
        if (OP == 8) OP8(I_CNDM);
        if (OP == 9) OP8(I_CMP);
        if (OP == 10) OP8(I_RCP);
        if (OP == 11) OP8(I_SQRT);
        if (OP == 12) OP8(I_MULLO);
        if (OP == 13) OP8(I_MULHI);
        if (OP == 14) OP8(I_CVTUB);
        if (OP == 15) OP8(I_MAX3);
        if (OP == 16) OP8(I_BFE);
        if (OP == 17) OP8(I_FMAC);
        if (OP == 18) OP8(I_MBCNT);
        if (OP == 39) OP8(I_CNDM_E64V);
        if (OP == 40) OP8(I_CMPCNDM);
        if (OP == 41) OP8(I_CMPCNDM_S);
        if (OP == 42) OP8(I_CNDM_K);
        if (OP == 43) OP8(I_ADDC);
        if (OP == 44) OP8(I_FMAMIX);
        if (OP == 45) OP8(I_FMAMIXH);
        if (OP == 46) OP8(I_ALIGNBIT);
        if (OP == 47) OP8(I_CVTF16);
        if (OP == 48) OP8(I_PERM);
        if (OP == 49) OP8(I_ANDOR);
        if (OP == 50) OP8(I_MED3);
        if (OP == 51) OP8(I_OR3);
        if (OP == 19) OP8(I_FMA_S);
        if (OP == 20) OP8(I_MUL_S);
        if (OP == 21) OP8(I_MUL_K);
        if (OP == 22) OP8(I_MUL_LIT);
        if (OP == 23) OP8(I_SUBF);
        if (OP == 24) OP8(I_MINF);
        if (OP == 25) OP8(I_MADU64);
        if (OP == 26) OP8(I_LSHLADD);
        if (OP == 27) OP8(I_XOR);
        if (OP == 28) OP8(I_LSHR);
        if (OP == 29) OP8(I_CVTFU);
        if (OP == 30) OP8(I_FLOOR);
        if (OP == 31) OP8(I_CNDM_S);
        if (OP == 32) OP8(I_FMAMK);
        if (OP == 33) OP8(I_MIN3);
        if (OP == 34) OP8(I_MULF64);
        if (OP == 35) OP8(I_CMP_S);
        if (OP == 36) OP8(I_ADD3);
        if (OP == 37) OP8(I_PKFMA2);
        if (OP == 38) OP8(I_PKMUL);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if ((threadIdx.x & 63) == 0) {
    stamps[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)] = t1 - t0;
    stamps[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) + 1] = r1 - r0;
  }
}

typedef void (*kern_t)(float*, unsigned long long*, float, float, unsigned long long);
struct Op { const char* name; kern_t k; };

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* out;
  unsigned long long* st;
  CHECK(hipMalloc(&out, (size_t)cus * 1024 * sizeof(float)));
  CHECK(hipMalloc(&st, (size_t)cus * 16 * 2 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const Op ops[] = {{"v_fma_f32", k_op<0>}, {"v_add_f32", k_op<1>}, {"v_mul_f32", k_op<2>}, {"v_max_f32", k_op<3>},
                    {"v_add_u32", k_op<4>}, {"v_mov_b32", k_op<5>}, {"v_and_b32", k_op<6>}, {"v_lshlrev_b32", k_op<7>},
                    {"v_cndmask_b32", k_op<8>}, {"v_cmp_lt_f32", k_op<9>}, {"v_rcp_f32", k_op<10>}, {"v_sqrt_f32", k_op<11>},
                    {"v_mul_lo_u32", k_op<12>}, {"v_mul_hi_u32", k_op<13>}, {"v_cvt_f32_ubyte0", k_op<14>},
                    {"v_max3_f32", k_op<15>}, {"v_bfe_u32", k_op<16>}, {"v_fmac_f32", k_op<17>}, {"v_mbcnt_lo", k_op<18>},
                    {"v_fma_f32 sgpr src", k_op<19>}, {"v_mul_f32 sgpr src", k_op<20>}, {"v_mul_f32 inline 2.0", k_op<21>},
                    {"v_mul_f32 literal", k_op<22>}, {"v_sub_f32", k_op<23>}, {"v_min_f32", k_op<24>},
                    {"v_mad_u64_u32", k_op<25>}, {"v_lshl_add_u32", k_op<26>}, {"v_xor_b32", k_op<27>},
                    {"v_lshrrev_b32", k_op<28>}, {"v_cvt_f32_u32", k_op<29>}, {"v_floor_f32", k_op<30>},
                    {"v_cndmask sgpr mask", k_op<31>}, {"v_fmamk_f32", k_op<32>}, {"v_min3_f32", k_op<33>},
                    {"v_mul_f64", k_op<34>}, {"v_cmp_lt_f32 sgpr", k_op<35>}, {"v_add3_u32", k_op<36>},
                    {"v_pk_fma_f32", k_op<37>}, {"v_pk_mul_f32", k_op<38>},
                    {"v_cndmask e64 vcc", k_op<39>}, {"v_cmp+v_cndmask vcc (2 insts)", k_op<40>},
                    {"v_cmp+v_cndmask sgpr (2 insts)", k_op<41>}, {"v_cndmask 0,v,vcc", k_op<42>}, {"v_addc_co_u32 vcc", k_op<43>},
                    {"v_fma_mix_f32 f16lo", k_op<44>}, {"v_fma_mix_f32 f16hi", k_op<45>}, {"v_alignbit_b32", k_op<46>},
                    {"v_cvt_f32_f16", k_op<47>}, {"v_perm_b32", k_op<48>}, {"v_and_or_b32", k_op<49>}, {"v_med3_f32", k_op<50>},
                    {"v_or3_b32", k_op<51>}};
  struct Mask { const char* name; unsigned long long m; };
  const Mask masks[] = {{"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 32-63", 0xffffffff00000000ull},
                        {"even lanes", 0x5555555555555555ull}, {"lanes 0-15", 0xffffull}, {"lane 0", 1ull}};
  for (const Op& op : ops) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(op.k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (const Mask& mk : masks) {
      if (&op != &ops[0] && &op != &ops[4] && &op != &ops[10] && &mk != &masks[0]) continue;  // masks: three opcodes only
      op.k<<<cus, 1024, 96 * 1024>>>(out, st, 1.0001f, 0.5f, mk.m);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 2; ++r) op.k<<<cus, 1024, 96 * 1024>>>(out, st, 1.0001f, 0.5f, mk.m);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 2;
      std::vector<unsigned long long> h((size_t)cus * 16 * 2);
      CHECK(hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double ticks = 0, real = 0;
      for (size_t i = 0; i < h.size() / 2; ++i) { ticks += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
      const double ghz = ticks / real * 0.1;
      const double insts_per_simd = 4.0 * kIter * 32;  // 4 waves per SIMD
      printf("%-18s %-14s %.3f ms  clock %.2f GHz  %.2f SIMD cycles per wave-instruction\n", op.name, mk.name, ms, ghz,
             ms * 1e-3 * ghz * 1e9 / insts_per_simd);
    }
  }
  return 0;
}
