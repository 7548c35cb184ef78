// Micro-benchmark: does a wave64 VALU instruction cost less on gfx950's SIMD32 when only 32 lanes are enabled (one 32-lane pass
// instead of two)?  Independent and dependent v_fma_f32 chains, one wave per SIMD, EXEC = all / lower half / upper half /
// even lanes / one lane.  If the half-empty wave issues twice as fast, a latency-bound kernel gains from putting its rays in
// the lower 32 lanes of twice as many waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int DEP>
__global__ void __launch_bounds__(256) k(float *out, int iters, uint64_t mask)
{
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	if ((mask >> (threadIdx.x & 63)) & 1ull)
		for (int i = 0; i < iters; ++i)
		{
			if (DEP)
			{
#pragma unroll
				for (int j = 0; j < 8; ++j)
					asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\t"
					             "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0"
					             : "+v"(a0));
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
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main()
{
	float *out;
	(void) hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
	hipEvent_t a, b;
	(void) hipEventCreate(&a), (void) hipEventCreate(&b);
	const int iters = 4000;
	struct { const char *name; uint64_t mask; } cases[] = {{"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 32-63", 0xffffffff00000000ull},
	                                                      {"even lanes", 0x5555555555555555ull}, {"lane 0", 1ull}};
	for (int dep = 0; dep < 2; ++dep)
		for (int wps : {1, 2})
			for (auto &c : cases)
			{
				const int blocks = 256 * wps;
				auto launch = [&](int it) { if (dep) k<1><<<blocks, 256>>>(out, it, c.mask); else k<0><<<blocks, 256>>>(out, it, c.mask); };
				launch(10);
				(void) hipDeviceSynchronize();
				(void) hipEventRecord(a);
				launch(iters);
				(void) hipEventRecord(b);
				(void) hipEventSynchronize(b);
				float ms;
				(void) hipEventElapsedTime(&ms, a, b);
				printf("%s chain, %d wave(s)/SIMD, %-14s %.3f ms -> %.2f cycles per instruction per wave at 2.4 GHz\n", dep ? "dependent  " : "independent", wps, c.name, ms,
				       ms * 1e-3 * 2.4e9 / ((double) iters * 64));
			}
	return 0;
}
