// Micro-benchmark: SIMD issue cost (cycles per wave64 instruction per SIMD, 8 waves per SIMD, independent chains) of the instruction kinds the
// ray-march and gradient kernels are made of: is everything "2 cycles" like v_fma_f32, or are integer / convert / select / compare
// instructions dearer?  Answers how to price SQ_INSTS_VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define KERNEL(NAME, ASM)                                                                                                     \
	__global__ void __launch_bounds__(256) NAME(float *out, int iters)                                                        \
	{                                                                                                                         \
		float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.5f; \
		for (int i = 0; i < iters; ++i)                                                                                       \
		{                                                                                                                     \
			_Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21", "s22", "s23"); \
		}                                                                                                                     \
		out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                          \
	}
#define L8(I) I " %0, %0, %8\n\t" I " %1, %1, %8\n\t" I " %2, %2, %8\n\t" I " %3, %3, %8\n\t" I " %4, %4, %8\n\t" I " %5, %5, %8\n\t" I " %6, %6, %8\n\t" I " %7, %7, %8"
#define U8(I) I " %0, %0\n\t" I " %1, %1\n\t" I " %2, %2\n\t" I " %3, %3\n\t" I " %4, %4\n\t" I " %5, %5\n\t" I " %6, %6\n\t" I " %7, %7"
#define T8(I) I " %0, %0, %8, %0\n\t" I " %1, %1, %8, %1\n\t" I " %2, %2, %8, %2\n\t" I " %3, %3, %8, %3\n\t" I " %4, %4, %8, %4\n\t" I " %5, %5, %8, %5\n\t" I " %6, %6, %8, %6\n\t" I " %7, %7, %8, %7"
KERNEL(k_fma, T8("v_fma_f32"))
KERNEL(k_add_f32, L8("v_add_f32"))
KERNEL(k_mul_f32, L8("v_mul_f32"))
KERNEL(k_sub_f32, L8("v_sub_f32"))
KERNEL(k_add_u32, L8("v_add_u32"))
KERNEL(k_and_b32, L8("v_and_b32"))
KERNEL(k_lshl, L8("v_lshlrev_b32"))
KERNEL(k_max_i32, L8("v_max_i32"))
KERNEL(k_med3_f32, T8("v_med3_f32"))
KERNEL(k_med3_i32, T8("v_med3_i32"))
KERNEL(k_cvt_f32_i32, U8("v_cvt_f32_i32"))
KERNEL(k_cvt_i32_f32, U8("v_cvt_i32_f32"))
KERNEL(k_cvt_ubyte0, U8("v_cvt_f32_ubyte0"))
KERNEL(k_floor, U8("v_floor_f32"))
KERNEL(k_mov, U8("v_mov_b32"))
KERNEL(k_cndmask, L8("v_cndmask_b32"))
KERNEL(k_mad_u32_u24, T8("v_mad_u32_u24"))
KERNEL(k_mul_lo_u32, L8("v_mul_lo_u32"))
KERNEL(k_bfe, T8("v_bfe_u32"))
KERNEL(k_lshl_or, T8("v_lshl_or_b32"))
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %8\n\tv_cmp_lt_f32 vcc, %1, %8\n\tv_cmp_lt_f32 vcc, %2, %8\n\tv_cmp_lt_f32 vcc, %3, %8\n\tv_cmp_lt_f32 vcc, %4, %8\n\tv_cmp_lt_f32 vcc, %5, %8\n\tv_cmp_lt_f32 vcc, %6, %8\n\tv_cmp_lt_f32 vcc, %7, %8")
KERNEL(k_sdwa_sub, "v_sub_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_0\n\tv_sub_u32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_0\n\tv_sub_u32_sdwa %2, %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_0\n\tv_sub_u32_sdwa %3, %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_0\n\tv_sub_u32_sdwa %4, %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_0\n\tv_sub_u32_sdwa %5, %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_0\n\tv_sub_u32_sdwa %6, %6, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_0\n\tv_sub_u32_sdwa %7, %7, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_0")
KERNEL(k_cndmask_s, "s_mov_b64 s[20:21], 0x5555\n\tv_cndmask_b32 %0, %0, %8, s[20:21]\n\tv_cndmask_b32 %1, %1, %8, s[20:21]\n\tv_cndmask_b32 %2, %2, %8, s[20:21]\n\tv_cndmask_b32 %3, %3, %8, s[20:21]\n\tv_cndmask_b32 %4, %4, %8, s[20:21]\n\tv_cndmask_b32 %5, %5, %8, s[20:21]\n\tv_cndmask_b32 %6, %6, %8, s[20:21]\n\tv_cndmask_b32 %7, %7, %8, s[20:21]")
KERNEL(k_cmp_s, "v_cmp_lt_f32 s[20:21], %0, %8\n\tv_cmp_lt_f32 s[22:23], %1, %8\n\tv_cmp_lt_f32 s[20:21], %2, %8\n\tv_cmp_lt_f32 s[22:23], %3, %8\n\tv_cmp_lt_f32 s[20:21], %4, %8\n\tv_cmp_lt_f32 s[22:23], %5, %8\n\tv_cmp_lt_f32 s[20:21], %6, %8\n\tv_cmp_lt_f32 s[22:23], %7, %8")
KERNEL(k_and_or, T8("v_and_or_b32"))
KERNEL(k_add3, T8("v_add3_u32"))
KERNEL(k_lshl_add, T8("v_lshl_add_u32"))
KERNEL(k_min_f32, L8("v_min_f32"))
KERNEL(k_max_f32, L8("v_max_f32"))
KERNEL(k_or_b32, L8("v_or_b32"))
KERNEL(k_xor_b32, L8("v_xor_b32"))
KERNEL(k_sub_u32, L8("v_sub_u32"))
KERNEL(k_mul_u32_u24, L8("v_mul_u32_u24"))
KERNEL(k_ceil, U8("v_ceil_f32"))
KERNEL(k_fmac, L8("v_fmac_f32"))
KERNEL(k_sqrt, U8("v_sqrt_f32"))
KERNEL(k_rcp, U8("v_rcp_f32"))
#define KERNEL64(NAME, ASM)                                                                                                   \
	__global__ void __launch_bounds__(256) NAME(float *out, int iters)                                                        \
	{                                                                                                                         \
		double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.5; \
		for (int i = 0; i < iters; ++i)                                                                                       \
		{                                                                                                                     \
			_Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); \
		}                                                                                                                     \
		out[blockIdx.x * 256 + threadIdx.x] = (float) (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                                \
	}
KERNEL64(k_pk_fma, T8("v_pk_fma_f32"))
KERNEL64(k_pk_add, L8("v_pk_add_f32"))
KERNEL64(k_pk_mul, L8("v_pk_mul_f32"))
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %8\n\tv_lshl_add_u64 %1, %1, 0, %8\n\tv_lshl_add_u64 %2, %2, 0, %8\n\tv_lshl_add_u64 %3, %3, 0, %8\n\tv_lshl_add_u64 %4, %4, 0, %8\n\tv_lshl_add_u64 %5, %5, 0, %8\n\tv_lshl_add_u64 %6, %6, 0, %8\n\tv_lshl_add_u64 %7, %7, 0, %8")
KERNEL(k_perm, T8("v_perm_b32"))
KERNEL(k_trunc, U8("v_trunc_f32"))
KERNEL(k_fract, U8("v_fract_f32"))
KERNEL(k_rndne, U8("v_rndne_f32"))
KERNEL(k_min_u32, L8("v_min_u32"))
KERNEL(k_cvt_u32_f32, U8("v_cvt_u32_f32"))
KERNEL(k_fmamk, "v_fmamk_f32 %0, %0, 0x41000000, %8\n\tv_fmamk_f32 %1, %1, 0x41000000, %8\n\tv_fmamk_f32 %2, %2, 0x41000000, %8\n\tv_fmamk_f32 %3, %3, 0x41000000, %8\n\tv_fmamk_f32 %4, %4, 0x41000000, %8\n\tv_fmamk_f32 %5, %5, 0x41000000, %8\n\tv_fmamk_f32 %6, %6, 0x41000000, %8\n\tv_fmamk_f32 %7, %7, 0x41000000, %8")
KERNEL(k_mul_lit, "v_mul_f32 %0, 0x3b808081, %0\n\tv_mul_f32 %1, 0x3b808081, %1\n\tv_mul_f32 %2, 0x3b808081, %2\n\tv_mul_f32 %3, 0x3b808081, %3\n\tv_mul_f32 %4, 0x3b808081, %4\n\tv_mul_f32 %5, 0x3b808081, %5\n\tv_mul_f32 %6, 0x3b808081, %6\n\tv_mul_f32 %7, 0x3b808081, %7")
int main()
{
	float *out;
	(void) hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
	hipEvent_t a, b;
	(void) hipEventCreate(&a), (void) hipEventCreate(&b);
	const int iters = 2000, wps = 8, blocks = 256 * wps;
	struct { const char *name; void (*k)(float *, int); } cases[] = {
		{"v_fma_f32", k_fma}, {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_sub_f32", k_sub_f32}, {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32},
		{"v_lshlrev_b32", k_lshl}, {"v_max_i32", k_max_i32}, {"v_med3_f32", k_med3_f32}, {"v_med3_i32", k_med3_i32}, {"v_cvt_f32_i32", k_cvt_f32_i32},
		{"v_cvt_i32_f32", k_cvt_i32_f32}, {"v_cvt_f32_ubyte0", k_cvt_ubyte0}, {"v_floor_f32", k_floor}, {"v_mov_b32", k_mov}, {"v_cndmask_b32 (vcc)", k_cndmask},
		{"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_bfe_u32", k_bfe}, {"v_lshl_or_b32", k_lshl_or}, {"v_cmp_lt_f32 (vcc)", k_cmp},
		{"v_sub_u32_sdwa", k_sdwa_sub}, {"v_cndmask_b32 (sgpr mask)", k_cndmask_s}, {"v_cmp_lt_f32 (sgpr dst)", k_cmp_s}, {"v_and_or_b32", k_and_or}, {"v_add3_u32", k_add3},
		{"v_lshl_add_u32", k_lshl_add}, {"v_min_f32", k_min_f32}, {"v_max_f32", k_max_f32}, {"v_or_b32", k_or_b32}, {"v_xor_b32", k_xor_b32}, {"v_sub_u32", k_sub_u32},
		{"v_mul_u32_u24", k_mul_u32_u24}, {"v_ceil_f32", k_ceil}, {"v_fmac_f32", k_fmac}, {"v_sqrt_f32", k_sqrt}, {"v_rcp_f32", k_rcp},
		{"v_pk_fma_f32", k_pk_fma}, {"v_pk_add_f32", k_pk_add}, {"v_pk_mul_f32", k_pk_mul}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_perm_b32", k_perm},
		{"v_trunc_f32", k_trunc}, {"v_fract_f32", k_fract}, {"v_rndne_f32", k_rndne}, {"v_min_u32", k_min_u32}, {"v_cvt_u32_f32", k_cvt_u32_f32},
		{"v_fmamk_f32 (literal)", k_fmamk}, {"v_mul_f32 (literal)", k_mul_lit}};
	for (auto &c : cases)
	{
		c.k<<<blocks, 256>>>(out, 10);
		(void) hipDeviceSynchronize();
		(void) hipEventRecord(a);
		c.k<<<blocks, 256>>>(out, iters);
		(void) hipEventRecord(b);
		(void) hipEventSynchronize(b);
		float ms;
		(void) hipEventElapsedTime(&ms, a, b);
		printf("%-22s %.3f ms -> %.2f cycles per wave64 instruction per SIMD (8 waves per SIMD, 2.4 GHz)\n", c.name, ms, ms * 1e-3 * 2.4e9 / ((double) iters * 64 * wps));
	}
	return 0;
}
