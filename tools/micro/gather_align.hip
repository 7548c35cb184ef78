// Micro-benchmark: throughput of 4-byte gathers at 4-byte aligned vs 2-byte aligned addresses (the packed sampling image reads
// x-pair dwords at 2-byte alignment).  Each lane walks a pseudo-random but wave-coherent sequence of bricks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32a2 __attribute__((aligned(2)));
template <int MISALIGN, int NLOADS>
__global__ void __launch_bounds__(256) k(const uint8_t *buf, uint64_t nbricks, uint32_t iters, uint32_t *out)
{
	uint32_t h = blockIdx.x * 256 + threadIdx.x;
	uint32_t acc = 0;
	const uint32_t wave = h >> 6, lane = h & 63;
	for (uint32_t it = 0; it < iters; ++it)
	{
		// bricks of a wave are neighbours (8x8 rays): brick = base(wave,it) + small lane offset
		uint64_t base = ((uint64_t) (wave * 2654435761u + it * 40503u) % (nbricks - 64));
		const uint8_t *b = buf + (base + (lane >> 2)) * 256 + ((lane & 3) * 12 + (it & 3) * 50) % 180 * 1;
		b = (const uint8_t *) (((uintptr_t) b & ~(uintptr_t) 3) + MISALIGN);
#pragma unroll
		for (int j = 0; j < NLOADS; ++j)
			acc += *reinterpret_cast<const u32a2 *>(b + (j & 1) * 12 + (j >> 1) * 48);
	}
	out[h] = acc;
}
int main()
{
	const uint64_t nbricks = 14000000; uint8_t *buf; uint32_t *out;
	hipMalloc(&buf, nbricks * 256); hipMemset(buf, 1, nbricks * 256);
	const int blocks = 256 * 8 * 4; hipMalloc(&out, blocks * 256 * 4);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	auto run = [&](auto kern, const char *name) {
		kern<<<blocks, 256>>>(buf, nbricks, 64, out); hipDeviceSynchronize();
		hipEventRecord(a); for (int r = 0; r < 5; ++r) kern<<<blocks, 256>>>(buf, nbricks, 256, out); hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
		double loads = (double) blocks * 256 * 256 * 4; printf("%-28s %.3f ms  %.1f G lane-loads/s\n", name, ms, loads / ms / 1e6);
	};
	run(k<0, 4>, "4 dwords, 4-byte aligned");
	run(k<2, 4>, "4 dwords, 2-byte aligned");
	run(k<0, 4>, "4 dwords, 4-byte aligned");
	run(k<2, 4>, "4 dwords, 2-byte aligned");
	return 0;
}
