// Micro-benchmark: what a per-lane dword gather costs the texture addresser of a CU as a function of the EXEC mask — does a load
// with 16 active lanes cost a quarter of one with 64?  (The ray-march kernels issue loads for subsets of a wave: probing lanes
// vs sampling lanes.)  L1-resident working set, 8 waves per SIMD, so the loop is bound by the address path, not by latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) k(const uint32_t *buf, uint64_t mask, uint32_t iters, uint32_t *out)
{
	const uint32_t h = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
	uint32_t acc = 0, idx = (lane * 37u + (threadIdx.x >> 6) * 11u) & 1023u;
	if ((mask >> lane) & 1ull)
		for (uint32_t it = 0; it < iters; ++it)
		{
			const uint32_t v = buf[idx];        // 4 KiB table: L1 hits
			acc += v;
			idx = (idx + 17u + (v & 1u)) & 1023u;        // dependent, so loads are not hoisted or merged
		}
	out[h] = acc;
}
int main()
{
	const int blocks = 256 * 8;
	uint32_t *out, *buf;
	hipMalloc(&out, blocks * 256 * 4); hipMalloc(&buf, 4096); hipMemset(buf, 0, 4096);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	struct { const char *name; uint64_t mask; } cases[] = {
		{"64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"even lanes (32)", 0x5555555555555555ull}, {"lanes 0-15", 0xffffull},
		{"every 4th lane (16)", 0x1111111111111111ull}, {"lanes 0-7", 0xffull}, {"every 8th lane (8)", 0x0101010101010101ull}, {"lanes 0-3", 0xfull}, {"1 lane", 1ull}};
	for (auto &c : cases)
	{
		k<<<blocks, 256>>>(buf, c.mask, 64, out); hipDeviceSynchronize();
		hipEventRecord(a); for (int r = 0; r < 5; ++r) k<<<blocks, 256>>>(buf, c.mask, 2048, out);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
		const double wave_loads_per_cu = (double) blocks * 4 * 2048 / 256;
		printf("%-22s %.3f ms  %.1f cycles per wave-level load per CU (2.4 GHz)\n", c.name, ms, ms * 1e-3 * 2.4e9 / wave_loads_per_cu);
	}
	return 0;
}
