// Micro-benchmark for the FETCH_SIZE calibration of the integrator's access pattern: every lane issues the four
// 2-byte-aligned dword loads of one trilinear footprint (rows at +0, +10, +50, +60 inside a 256-byte brick), every brick of a 1 GiB
// buffer is touched by exactly ONE lane of the whole grid, so the bytes that must come from memory are known:
//   k_gather<2>    footprint bytes [2, 66) of the brick   -> one 128-byte line per brick  (both of its 64-byte halves)
//   k_gather<100>  footprint bytes [100, 164)             -> both 128-byte lines of the brick
//   k_gather<64>   footprint bytes [64, 128)              -> one 128-byte line, ONE 64-byte half
//   k_stream       16 bytes per lane, coalesced           -> the guide's calibration case (FETCH_SIZE reports half)
// Run under `rocprofv3 --pmc FETCH_SIZE -- tools/micro/gather_fetch` and compare the counter with the printed byte counts.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32_align2 __attribute__((aligned(2)));

template <int IN>
__global__ void __launch_bounds__(256) k_gather(const uint8_t *buf, uint32_t *out)
{
	const uint32_t h = blockIdx.x * 256u + threadIdx.x;
	// a wave's 64 lanes read 64 bricks spread like the 8x8 rays of a tile (neighbours in x adjacent, rows 4096 bricks apart)
	const uint32_t wave = h >> 6, lane = h & 63u;
	const uint32_t brick = ((wave & 511u) * 8u + (lane & 7u)) + (((wave >> 9) * 8u + (lane >> 3)) << 12);
	const uint8_t *p = buf + ((uint64_t) brick << 8) + IN;
	const uint32_t a = *reinterpret_cast<const u32_align2 *>(p), b = *reinterpret_cast<const u32_align2 *>(p + 10);
	const uint32_t c = *reinterpret_cast<const u32_align2 *>(p + 50), d = *reinterpret_cast<const u32_align2 *>(p + 60);
	out[h] = a ^ b ^ c ^ d;
}

__global__ void __launch_bounds__(256) k_stream(const uint4 *buf, uint32_t *out)
{
	const uint32_t h = blockIdx.x * 256u + threadIdx.x;
	uint32_t       acc = 0;
	for (int i = 0; i < 16; ++i)
	{
		const uint4 v = buf[(uint64_t) i * gridDim.x * 256u + h];
		acc ^= v.x ^ v.y ^ v.z ^ v.w;
	}
	out[h] = acc;
}

int main()
{
	const uint64_t bytes = 1ull << 30;        // 4 Mi bricks of 256 bytes: four times the 256 MiB Infinity Cache
	const uint32_t lanes = (uint32_t) (bytes >> 8);
	uint8_t *      buf;
	uint32_t *     out;
	if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, (size_t) lanes * 4) != hipSuccess)
		return 1;
	hipMemset(buf, 1, bytes);
	hipDeviceSynchronize();
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	auto timed = [&](const char *name, auto launch, double must_fetch, const char *what) {
		hipEventRecord(e0);
		launch();
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		printf("%-14s %.3f ms  bytes that must be fetched: %.0f KiB (%s)\n", name, ms, must_fetch / 1024.0, what);
	};
	// every kernel runs ONCE (a second launch would find part of the buffer in the Infinity Cache)
	timed("k_stream", [&] { hipLaunchKernelGGL(k_stream, dim3(lanes / 256 / 4), dim3(256), 0, 0, (const uint4 *) buf, out); }, (double) bytes / 4,
	      "16 B per lane, coalesced, 256 MiB");
	timed("k_gather<2>", [&] { hipLaunchKernelGGL(k_gather<2>, dim3(lanes / 256), dim3(256), 0, 0, buf, out); }, (double) lanes * 128, "one 128-B line per brick");
	timed("k_gather<100>", [&] { hipLaunchKernelGGL(k_gather<100>, dim3(lanes / 256), dim3(256), 0, 0, buf, out); }, (double) lanes * 256, "both 128-B lines per brick");
	timed("k_gather<64>", [&] { hipLaunchKernelGGL(k_gather<64>, dim3(lanes / 256), dim3(256), 0, 0, buf, out); }, (double) lanes * 128,
	      "one 128-B line per brick, one 64-B half of it");
	hipDeviceSynchronize();
	return 0;
}
