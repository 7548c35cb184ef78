// Micro-benchmark: cost of one per-lane gather instruction as a function of its width (byte, dword, dwordx2, dwordx4) when every
// lane reads its own 16-byte record.  Answers whether a "one record per sample" layout (1 x dwordx4) is cheaper for the texture
// addresser than the packed-brick layout's 4 x dword.  WORKING SET selects L1- / L2- / HBM-resident regimes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(256) k(const uint8_t *buf, uint32_t nrec_mask, uint32_t iters, uint32_t *out)
{
	const uint32_t h = blockIdx.x * 256 + threadIdx.x;
	const uint32_t wave = h >> 6, lane = h & 63;
	uint32_t acc = 0;
	for (uint32_t it = 0; it < iters; ++it)
	{
		// an 8x8 ray tile: lanes read neighbouring records (4 per 64-byte line along x, other rows elsewhere)
		const uint32_t base = wave * 2654435761u + it * 40503u;
		const uint32_t rec = (base + (lane & 7) + (lane >> 3) * 1024u) & nrec_mask;
		const uint8_t *p = buf + (uint64_t) rec * 16;
		if (MODE == 0) acc += *p;
		if (MODE == 1) acc += *reinterpret_cast<const uint32_t *>(p);
		if (MODE == 2) { uint2 v = *reinterpret_cast<const uint2 *>(p); acc += v.x ^ v.y; }
		if (MODE == 3) { uint4 v = *reinterpret_cast<const uint4 *>(p); acc += v.x ^ v.y ^ v.z ^ v.w; }
		if (MODE == 4) { const uint32_t *q = reinterpret_cast<const uint32_t *>(p); acc += q[0] ^ q[1] ^ q[2] ^ q[3]; }   // may merge
		if (MODE >= 6 && MODE <= 9)
		{   // round 4: a 16-byte gather that is NOT 16-byte aligned - 8, 4, 2 bytes off a record boundary (an empty asm keeps it one instruction)
			typedef uint32_t u4 __attribute__((ext_vector_type(4)));
			typedef const __attribute__((address_space(1))) u4 __attribute__((aligned(2))) *gp;
			const uint32_t off = MODE == 6 ? 8u : MODE == 7 ? 4u : MODE == 8 ? 2u : 10u;
			u4 v = *(gp) (uintptr_t) (buf + (uint64_t) (rec & (nrec_mask >> 1)) * 16 + off);
			acc += v.x ^ v.y ^ v.z ^ v.w;
		}
		if (MODE == 10)
		{   // dwordx2 at 4-byte alignment
			typedef uint32_t u2 __attribute__((ext_vector_type(2)));
			typedef const __attribute__((address_space(1))) u2 __attribute__((aligned(4))) *gp;
			u2 v = *(gp) (uintptr_t) (buf + (uint64_t) (rec & (nrec_mask >> 1)) * 16 + 4);
			acc += v.x ^ v.y;
		}
		if (MODE == 5)
		{   // 4 separate dwords the compiler cannot merge (different records), like the packed brick rows
			const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
			const uint32_t o = (lane * 4u) & 12u;
			acc += q[0] ^ *(const uint32_t *) (buf + (uint64_t) ((rec + 64u) & nrec_mask) * 16 + o) ^
			       *(const uint32_t *) (buf + (uint64_t) ((rec + 8192u) & nrec_mask) * 16 + o) ^
			       *(const uint32_t *) (buf + (uint64_t) ((rec + 8256u) & nrec_mask) * 16 + o);
		}
	}
	out[h] = acc;
}
int main()
{
	const int blocks = 256 * 8 * 4;
	uint32_t *out; hipMalloc(&out, blocks * 256 * 4);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	const char *names[] = {"1 byte", "1 dword", "1 dwordx2", "1 dwordx4", "4 dword (same record)", "4 dword (4 records)", "1 dwordx4, 8 bytes off", "1 dwordx4, 4 bytes off",
	                       "1 dwordx4, 2 bytes off", "1 dwordx4, 10 bytes off", "1 dwordx2, 4 bytes off"};
	for (uint64_t log2rec : {10, 18, 26})
	{
		const uint64_t nrec = 1ull << log2rec; uint8_t *buf;
		hipMalloc(&buf, nrec * 16); hipMemset(buf, 1, nrec * 16);
		printf("working set %.2f MiB\n", nrec * 16 / 1048576.0);
		auto run = [&](auto kern, const char *name) {
			kern<<<blocks, 256>>>(buf, (uint32_t) (nrec - 1), 64, out); hipDeviceSynchronize();
			hipEventRecord(a); for (int r = 0; r < 5; ++r) kern<<<blocks, 256>>>(buf, (uint32_t) (nrec - 1), 256, out);
			hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
			const double wave_iters = (double) blocks * 4 * 256;
			printf("  %-24s %.3f ms  %.1f cycles/CU per wave-iteration (2.4 GHz, 256 CUs)\n", name, ms, ms * 1e-3 * 2.4e9 * 256 / wave_iters);
		};
		run(k<0>, names[0]); run(k<1>, names[1]); run(k<2>, names[2]); run(k<3>, names[3]); run(k<4>, names[4]); run(k<5>, names[5]);
		run(k<6>, names[6]); run(k<7>, names[7]); run(k<8>, names[8]); run(k<9>, names[9]); run(k<10>, names[10]);
		hipFree(buf);
	}
	return 0;
}
