// Micro-benchmark: would the integrator's x-pair gathers be cheaper as FORMAT loads?  buffer_load_format_d16_xyzw with an 8_8_8_8 USCALED
// descriptor hands the four bytes (v0, g0, v1, g1) of an x pair back as four f16 values in two VGPRs - no v_cvt_f32_ubyteN (16 of them
// per sample in the shipped loop) - and v_fma_mix_f32 consumes f16 operands next to an fp32 weight.  Questions answered here:
//   1. are format loads exact at the 2-byte-aligned addresses the packed volume has (the element is 4 bytes)?
//   2. what does a format gather cost the texture path next to a plain dword gather (L1-resident, address-path bound)?
//   3. what do v_fma_mix_f32 and v_pk_add_f16 cost next to v_fma_f32 (8 waves per SIMD, dependent chains of 8)?
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float    float4v __attribute__((ext_vector_type(4)));
typedef int      int4v __attribute__((ext_vector_type(4)));

__device__ half4v  buf_load_fmt_h4(int4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f16");
__device__ float4v buf_load_fmt_f4(int4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");
__device__ uint32_t buf_load_u32(int4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");

__device__ __forceinline__ int4v make_rsrc(const void *base, uint32_t bytes)
{
	const uint64_t a = (uint64_t) base;
	int4v          r;
	r.x = (int) (uint32_t) a;
	r.y = (int) (uint32_t) ((a >> 32) & 0xffffu);        // stride 0
	r.z = (int) bytes;
	r.w = (int) (0xFACu | (2u << 12) | (10u << 15));        // dst_sel xyzw = RGBA, USCALED, 8_8_8_8
	return r;
}

// 1. exactness at every byte alignment
__global__ void k_check(const uint8_t *buf, uint32_t bytes, uint32_t n, uint32_t step, uint32_t *bad_h, uint32_t *bad_f)
{
	const int4v    rs = make_rsrc(buf, bytes);
	const uint32_t i  = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const uint32_t off = i * step;
	const half4v   h   = buf_load_fmt_h4(rs, (int) off, 0, 0);
	const float4v  f   = buf_load_fmt_f4(rs, (int) off, 0, 0);
	for (int c = 0; c < 4; ++c)
	{
		const float want = (float) buf[off + c];
		if ((float) h[c] != want)
			atomicAdd(bad_h, 1u);
		if (f[c] != want)
			atomicAdd(bad_f, 1u);
	}
}

// 2. address-path cost, as tools/micro/gather_mask.hip measures it (4 KiB table, dependent indices)
template <int KIND>
__global__ void __launch_bounds__(256) k_gather(const uint8_t *buf, uint32_t iters, uint32_t align_mask, uint32_t *out)
{
	const int4v    rs   = make_rsrc(buf, 4096 + 16);
	const uint32_t lane = threadIdx.x & 63;
	uint32_t       acc = 0, idx = ((lane * 37u + (threadIdx.x >> 6) * 11u) * 2u) & 4095u & align_mask;
	for (uint32_t it = 0; it < iters; ++it)
	{
		uint32_t v;
		if (KIND == 0)
			v = *reinterpret_cast<const uint32_t *>(buf + idx);
		else if (KIND == 1)
			v = buf_load_u32(rs, (int) idx, 0, 0);
		else if (KIND == 2)
		{
			const half4v h = buf_load_fmt_h4(rs, (int) idx, 0, 0);
			v              = (uint32_t) (float) (h.x + h.w);
		}
		else
		{
			const float4v f = buf_load_fmt_f4(rs, (int) idx, 0, 0);
			v               = (uint32_t) (f.x + f.w);
		}
		acc += v;
		idx = (idx + 34u + ((v & 1u) << 1)) & 4095u & align_mask;
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// 3. issue cost of the arithmetic
template <int KIND>
__global__ void __launch_bounds__(256) k_valu(uint32_t iters, float w, uint32_t seed, float *out)
{
	float    a[8];
	half2v   h[8];
	for (int i = 0; i < 8; ++i)
	{
		a[i] = (float) (threadIdx.x + i + seed);
		h[i] = half2v{(_Float16) (float) ((threadIdx.x + i) & 255), (_Float16) (float) ((threadIdx.x * 3 + i) & 255)};
	}
	for (uint32_t it = 0; it < iters; ++it)
	{
#pragma unroll
		for (int i = 0; i < 8; ++i)
		{
			if (KIND == 0)
				a[i] = __builtin_fmaf(w, a[i], a[(i + 1) & 7]);
			else if (KIND == 1)        // v_fma_mix_f32: fp32 weight, f16 difference, f16 base
				asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(a[i]) : "v"(a[i]), "v"(h[i]), "v"(h[(i + 1) & 7]));
			else if (KIND == 2)
				asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(h[i]) : "v"(h[i]), "v"(h[(i + 1) & 7]));
			else
				asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
		}
	}
	float s = 0;
	for (int i = 0; i < 8; ++i)
		s += a[i] + (float) h[i].x + (float) h[i].y;
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
	const uint32_t       bytes = 1u << 20;
	std::vector<uint8_t> host(bytes + 16);
	for (uint32_t i = 0; i < bytes + 16; ++i)
		host[i] = (uint8_t) ((i * 2654435761u) >> 13);
	uint8_t  *buf;
	uint32_t *bad, *out;
	hipMalloc(&buf, bytes + 16); hipMemcpy(buf, host.data(), bytes + 16, hipMemcpyHostToDevice);
	hipMalloc(&bad, 8); hipMalloc(&out, 256 * 8 * 256 * 4);
	for (uint32_t step : {4u, 2u, 1u})
	{
		hipMemset(bad, 0, 8);
		const uint32_t n = (bytes - 8) / step;
		k_check<<<(n + 255) / 256, 256>>>(buf, bytes + 16, n, step, bad, bad + 1);
		uint32_t r[2]; hipMemcpy(r, bad, 8, hipMemcpyDeviceToHost);
		printf("format loads at every %u-byte offset: %u loads, d16 mismatches %u, f32 mismatches %u\n", step, n, r[0], r[1]);
	}
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	const int  blocks = 256 * 8;
	auto       time = [&](auto launch, const char *name, double per_cu) {
        launch(64); hipDeviceSynchronize();
        hipEventRecord(a); for (int r = 0; r < 5; ++r) launch(2048);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-52s %.3f ms  %.1f cycles per wave-level instruction per %s (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / per_cu, per_cu == (double) blocks * 4 * 2048 / 256 ? "CU" : "SIMD");
	};
	const double per_cu = (double) blocks * 4 * 2048 / 256, per_simd = (double) blocks * 4 * 2048 * 8 / 1024;
	for (uint32_t am : {~3u, ~1u})
	{
		printf("offsets %s:\n", am == ~3u ? "4-byte aligned" : "2-byte aligned");
		time([&](uint32_t it) { k_gather<0><<<blocks, 256>>>(buf, it, am, out); }, "  global_load_dword", per_cu);
		time([&](uint32_t it) { k_gather<1><<<blocks, 256>>>(buf, it, am, out); }, "  buffer_load_dword offen", per_cu);
		time([&](uint32_t it) { k_gather<2><<<blocks, 256>>>(buf, it, am, out); }, "  buffer_load_format_d16_xyzw (8_8_8_8 USCALED)", per_cu);
		time([&](uint32_t it) { k_gather<3><<<blocks, 256>>>(buf, it, am, out); }, "  buffer_load_format_xyzw (8_8_8_8 USCALED)", per_cu);
	}
	float *fo = reinterpret_cast<float *>(out);
	time([&](uint32_t it) { k_valu<0><<<blocks, 256>>>(it, 0.5f, 1, fo); }, "v_fma_f32", per_simd);
	time([&](uint32_t it) { k_valu<1><<<blocks, 256>>>(it, 0.5f, 1, fo); }, "v_fma_mix_f32 (f32, f16, f16)", per_simd);
	time([&](uint32_t it) { k_valu<2><<<blocks, 256>>>(it, 0.5f, 1, fo); }, "v_pk_add_f16", per_simd);
	time([&](uint32_t it) { k_valu<3><<<blocks, 256>>>(it, 0.5f, 1, fo); }, "v_cvt_f32_ubyte2", per_simd);
	return 0;
}
