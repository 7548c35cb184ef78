// Micro-benchmark: issue rate of wave64 VALU instructions on gfx950 (cycles per instruction per SIMD) for plain and packed FP32 FMAs,
// at 1, 2, 4 and 8 waves per SIMD.  Settles how to turn SQ_INSTS_VALU into "VALU busy" fractions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int PACKED>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
	for (int i = 0; i < iters; ++i)
	{
		if (PACKED)
		{
#pragma unroll
			for (int j = 0; j < 16; ++j)
				asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2\n\tv_pk_fma_f32 %3, %3, %3, %3"
				             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
		}
		else
		{
#pragma unroll
			for (int j = 0; j < 8; ++j)
				asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
				             "v_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7"
				             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
int main()
{
	float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	const int iters = 4000;
	for (int packed = 0; packed < 2; ++packed)
		for (int wps : {1, 2, 4, 8})
		{
			const int blocks = 256 * wps;        // 256 CUs x (wps workgroups of 4 waves) -> wps waves per SIMD
			auto launch = [&](int it) { if (packed) k<1><<<blocks, 256>>>(out, it); else k<0><<<blocks, 256>>>(out, it); };
			launch(10); hipDeviceSynchronize();
			hipEventRecord(a); launch(iters); hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b);
			const double instr_per_simd = (double) iters * 64 * wps;        // 64 VALU instructions per iteration per wave
			printf("%s  %d waves/SIMD: %.3f ms -> %.2f cycles per wave instruction per SIMD at 2.4 GHz\n", packed ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms,
			       ms * 1e-3 * 2.4e9 / instr_per_simd);
		}
	return 0;
}
