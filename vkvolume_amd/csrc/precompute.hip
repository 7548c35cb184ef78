// precompute.hip — gradient map, occupancy map and Chebyshev distance transforms for gfx950.
//
// Replaces ComputeGradientMap / ComputeDistanceMap (src/compute_gradient_map.cpp, src/compute_distance_map.cpp)
// and their shaders (gradient_map.comp, occupancy_map.comp, distance_map.comp, distance_map_anisotropic.comp).
// All kernels are integer/byte streaming work bound by HBM / LDS bandwidth; no MFMA.
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "vkv_device.hpp"

using namespace vkv;

// ---------------------------------------------------------------------------------------------
// Gradient map (shaders/gradient_map.comp:35-41)
// ---------------------------------------------------------------------------------------------
// One thread per voxel, x fastest so every load/store of a wave is one contiguous 64-byte row segment.
// The four taps sit on the (±1,±1,±1) tetrahedron, i.e. in four different (y,z) rows; L1/L2 absorb the
// 4x re-read (each byte is requested by the 4 voxels diagonal to it).
__global__ void __launch_bounds__(256) k_gradient_map(const uint8_t *__restrict__ vol, uint8_t *__restrict__ grad, int W, int H, int D,
                                                      int use_gradient, float modifier, uint32_t blocks_x, uint32_t nblocks)
{
	// grid.x = column blocks x 4-row groups of one z slice, grid.y = z (keeps every grid dimension x block size below 2^32
	// for 2048^3 and larger volumes)
	(void) nblocks;
	const uint32_t bx = blockIdx.x % blocks_x;
	const int      x  = (int) (bx * 64 + (threadIdx.x & 63));
	const int      y  = (int) ((blockIdx.x / blocks_x) * 4 + (threadIdx.x >> 6)), z = (int) blockIdx.y;
	if (x >= W || y >= H)
		return;
	float g = 1.0f;        // get_gradient_compute.glsl:6-7
	if (use_gradient)
		g = gradient_on_the_fly(vol, W, H, D, x, y, z, modifier);
	grad[vidx(x, y, z, W, H)] = store_unorm8(g);
}

// LDS-tiled version (the one the launcher uses for every width >= 4; until round 5 only for dword-aligned rows, W % 4 == 0): a workgroup computes 64 x 8 x 8 blocks of
// voxels from (64+8) x 10 x 10 halo tiles staged in LDS with coalesced dword loads, so every volume byte is fetched ~1.8x (mostly from
// L2) instead of 4x with byte gathers.
//
// Correctly rounded sqrt for x == 0 or x normal with a normal root, from the reciprocal root (Markstein's final step): g = x * rsq(x) is
// within a few ulp, the residual x - g * g is exact in an fma, and g + residual * (0.5 * rsq) rounds to the nearest float of the true
// root.  Six instructions (v_max keeps x = 0 away from 0 * inf) where __builtin_sqrtf under -fhip-fp32-correctly-rounded-divide-sqrt
// takes sixteen (v_sqrt_f32, a one-ulp fix-up from two residual signs, rescaling of tiny inputs, inf / nan pass-through).  The
// gradient's sum of squares is 0 or >= ~1e-17 (squares of rounding residues of byte / 255 values).  vkv_debug_check (what = 0)
// compares it with __builtin_sqrtf for every float of a range; the GPU tests run it over 0 and all of [2^-90, 16).
__device__ __forceinline__ float sqrt_rn_normal(float x)
{
	const float y = __builtin_amdgcn_rsqf(__builtin_fmaxf(x, 0x1p-100f));
	const float g = x * y, h = 0.5f * y;
	return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}

// R8_UNORM store of clamp(g, 0, 1) in one instruction: v_cvt_pk_u8_f32 rounds to nearest even and saturates to [0, 255], so fed
// g * 255 it equals store_unorm8(g_clamp(g, 0, 1)) for every non-NaN float (vkv_debug_check what = 1 runs over all of them).
__device__ __forceinline__ uint32_t store_unorm8_clamped(float g) { return __builtin_amdgcn_cvt_pk_u8_f32(g * 255.0f, 0u, 0u); }

// what = 0: sqrt_rn_normal vs __builtin_sqrtf; what = 1: store_unorm8_clamped vs store_unorm8(g_clamp(., 0, 1));
// what = 2: recip_exact(x) vs 1.0f / x for every ordinary x (vkv_device.hpp; the others take the IEEE path in ray_setup by construction);
// what = 3: div_by(a, x, recip_refined(x)) vs a / x with eight hashed ordinary numerators per denominator x
// what = 4: numerators +0 and -0 must NOT take the reciprocal path (div_ordinary_num), and the quotient the dispatch delivers is the IEEE one
__global__ void __launch_bounds__(256) k_check_numerics(int what, uint32_t first_bits, uint64_t count, unsigned long long *mismatches)
{
	const uint64_t i = (uint64_t) blockIdx.x * 256u + threadIdx.x;
	if (i >= count)
		return;
	const uint32_t bits = first_bits + (uint32_t) i;
	const float    x    = __uint_as_float(bits);
	bool           bad  = false;
	if (what == 0)
		bad = __float_as_uint(sqrt_rn_normal(x)) != __float_as_uint(__builtin_sqrtf(x));
	else if (what == 1)
		bad = (uint8_t) store_unorm8_clamped(x) != store_unorm8(g_clamp(x, 0.0f, 1.0f));
	else if (what == 2)
		bad = div_ordinary(x) && __float_as_uint(recip_exact(x)) != __float_as_uint(1.0f / x);
	else if (what == 4)
	{        // the dispatch of ray_setup for a zero numerator: not "ordinary", so the quotient comes from the IEEE division - and it has to,
		 // the refined reciprocal path returns +0 for -0 / x (checked here too, so that nobody relaxes div_ordinary_num on the comment's word)
		const float zeros[2] = {0.0f, -0.0f};
		for (int k = 0; k < 2; ++k)
		{
			const float a = zeros[k];
			const bool  fast = div_ordinary(x) && div_ordinary_num(a);
			const float q    = fast ? div_by(a, x, recip_refined(x)) : a / x;
			bad = bad || fast || __float_as_uint(q) != __float_as_uint(a / x);
		}
	}
	else if (div_ordinary(x))
	{
		const float r = recip_refined(x);
		uint32_t    h = bits * 0x9e3779b9u + 0x7f4a7c15u;
		for (int k = 0; k < 8 && !bad; ++k)
		{
			h ^= h >> 15, h *= 0x2c1b3c6du, h ^= h >> 12, h *= 0x297a2d39u, h ^= h >> 15;
			// sign and mantissa from the hash, exponent 2^-40 .. 2^40 (biased 87 .. 167; the largest only with a zero mantissa)
			uint32_t a_bits = (h & 0x807fffffu) | ((87u + (h >> 23) % 81u) << 23);
			if (!div_ordinary(__uint_as_float(a_bits)))
				a_bits &= 0xff800000u;
			const float    a      = __uint_as_float(a_bits);
			bad                   = __float_as_uint(div_by(a, x, r)) != __float_as_uint(a / x);
		}
	}
	if (bad)
		atomicAdd(mismatches, 1ull);
}

// Four bytes of a voxel row as ONE dword, whatever the row's alignment (round 6: the tiled kernels no longer need W % 4 == 0).  Rows of a volume
// whose width is no multiple of 4 start at every byte alignment; global loads need none on gfx950 (the integrator's own footprint gathers are
// 2-byte aligned), the type only tells the compiler not to assume one.
typedef uint32_t u32_any_align __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t load_u32_any(const uint8_t *p) { return *reinterpret_cast<const u32_any_align *>(p); }
// Dword column dc (voxels 4 dc .. 4 dc + 3) of a row of W >= 4 voxels, dc < ceil(W / 4): the last, partial column of an odd width is read as the
// row's LAST four bytes and shifted down - nothing past the row is touched, the bytes of x >= W come back zero.
__device__ __forceinline__ uint32_t row_dword(const uint8_t *row, int dc, int W)
{
	const int x = 4 * dc;
	if (x + 4 <= W)
		return load_u32_any(row + x);
	return load_u32_any(row + (W - 4)) >> (8 * (x + 4 - W));
}

constexpr int kGradTileX = 64, kGradTileY = 8, kGradTileZ = 8, kGradPitch = 72;        // pitch = 64 + 4 texels of halo on each side
constexpr int kGradSegment = 10;                                                       // tiles one workgroup marches over

// byte store through a wave-uniform base (SGPR pair) and a 32-bit lane offset: no 64-bit vector address arithmetic per voxel
__device__ __forceinline__ void store_u8_uniform_base(uint8_t *base, uint32_t off, uint32_t value)
{
	asm volatile("global_store_byte %0, %1, %2" : : "v"(off), "v"(value), "s"(base) : "memory");
}

// (Measured in round 3 and dropped: TWO tiles of loads in flight - the tile after next requested as soon as the current one is staged:
// 0.670 against 0.654 ms on C3, 6.88 against 6.84 on C4: the loads are not what the arithmetic waits for.)
// A workgroup MARCHES along z over `seg` consecutive tiles with the next tile's dwords already in flight (held in registers) while the
// current one is computed: a workgroup that loads, waits, computes and leaves keeps too few bytes in flight per CU to cover the HBM
// latency (measured on the one-tile-per-workgroup kernel: neither the VALU nor the LDS busy more than half the time; 0.96 -> 0.85 ms
// on 1024 x 1024 x 795).  The tile holds 16-bit OFFSETS into the 256-entry table of b / 255 (4 b: the shift between the tap read and the
// table read is paid once per staged texel, not four times per voxel), two per half of a staged dword, which puts the texels of every
// group of four in the order 0, 2, 1, 3.
// ALIGNED: W % 4 == 0 and a dword-aligned volume - the staging loads are plain aligned dwords (scalar base + 32-bit lane offset addressing);
// otherwise (round 6) rows start at any alignment and the last column of a row may be partial: row_dword / load_u32_any, measured 11 - 18 %
// slower on aligned volumes (64-bit vector addresses per load), 2.4 x faster than the byte-wise kernel on odd ones.
template <bool ALIGNED>
__global__ void __launch_bounds__(256) k_gradient_map_tiled(const uint8_t *__restrict__ vol, uint8_t *__restrict__ grad, int W, int H, int D,
                                                            float modifier, uint32_t tiles_x, uint32_t tiles_y, uint32_t tiles_z, uint32_t seg,
                                                            uint32_t n_wgs)
{
	constexpr int kRows = (kGradTileZ + 2) * (kGradTileY + 2), kCols = kGradPitch / 4, kIter = (kRows * kCols + 255) / 256;
	__shared__ __align__(16) uint16_t s_tile[kRows * kGradPitch];
	__shared__ float                  s_unorm[256];        // b / 255 (IEEE division, once per workgroup): a tap costs one LDS read, not four VALU
	s_unorm[threadIdx.x] = unorm8(threadIdx.x);
	const uint32_t t  = xcd_remap(blockIdx.x, n_wgs);
	const int      x0 = (int) (t % tiles_x) * kGradTileX;
	const int      y0 = (int) ((t / tiles_x) % tiles_y) * kGradTileY;
	const uint32_t k0 = (t / (tiles_x * tiles_y)) * seg, k1 = min(k0 + seg, tiles_z);
	const int      wd = (W + 3) >> 2;        // dword columns of a row (round 6: the last one partial when W % 4 != 0; rows then start at any alignment)
	const float    quarter_modifier = 0.25f * modifier;
	// per-thread staging slots (100 rows x 18 dwords, 8 per thread): row / column of the tile are the same for every tile of the march,
	// only z moves; rows clamp in y and z, dword columns clamp in x
	int  row_xy[kIter], colc[kIter], rz[kIter];
	bool left[kIter], right[kIter];
#pragma unroll
	for (int j = 0; j < kIter; ++j)
	{
		const int d   = min((int) threadIdx.x + 256 * j, kRows * kCols - 1);
		const int row = d / kCols, col = d - row * kCols;
		const int gy = min(max(y0 - 1 + row % (kGradTileY + 2), 0), H - 1);
		const int gc = (x0 >> 2) - 1 + col;
		left[j] = gc < 0, right[j] = gc >= wd;        // clamp-to-edge in x: texel x = -1 is voxel 0, texel x = W is voxel W - 1
		row_xy[j] = gy * W, colc[j] = min(max(gc, 0), wd - 1);
		rz[j]     = row / (kGradTileY + 2) - 1;
	}
	const size_t plane = (size_t) H * (size_t) W;        // bytes of a z slice
	uint32_t     v[kIter];
	auto         fetch_edge = [&](uint32_t k) {
#pragma unroll
		for (int j = 0; j < kIter; ++j)
		{
			const int gz = min(max((int) k * kGradTileZ + rz[j], 0), D - 1);
			const uint8_t *row = vol + (size_t) gz * plane + (size_t) row_xy[j];
			uint32_t       w   = ALIGNED ? reinterpret_cast<const uint32_t *>(row)[colc[j]] : row_dword(row, colc[j], W);
			const int      nv  = W - 4 * colc[j];        // voxels of this column inside the row: 1 .. 3 in the last column of an odd width
			if (!ALIGNED && nv < 4)
				w |= (((w >> (8 * (nv - 1))) & 255u) * 0x01010101u) << (8 * nv);        // clamp-to-edge: the texels x >= W of the column are voxel W - 1
			w    = left[j] ? (w << 24) : w;
			v[j] = right[j] ? (w >> 24) : w;
		}
	};
	// tiles whose halo needs no clamp in x and z (nearly all of them): one wave-uniform base per tile and a constant 32-bit byte offset per
	// slot, i.e. no vector address arithmetic at all (10 slices of the volume stay below 2^32 bytes: checked by the launcher)
	uint32_t voff[kIter];
#pragma unroll
	for (int j = 0; j < kIter; ++j)
		voff[j] = (uint32_t) ((size_t) (rz[j] + 1) * plane + (size_t) row_xy[j] + (size_t) (4 * colc[j]));
	const bool inner_x = x0 > 0 && x0 + kGradTileX + 4 <= W;
	auto       fetch   = [&](uint32_t k) {
        if (inner_x && k > 0 && (int) (k + 1) * kGradTileZ < D)
        {
            const uint8_t *base = vol + ((size_t) k * kGradTileZ - 1) * plane;
#pragma unroll
            for (int j = 0; j < kIter; ++j)
            {
                uint32_t o = voff[j];
                asm volatile("" : "+v"(o));        // keeps the zero-extension next to the load: scalar base + 32-bit lane offset addressing
                v[j] = ALIGNED ? *reinterpret_cast<const uint32_t *>(base + o) : load_u32_any(base + o);
            }
        }
        else
            fetch_edge(k);
	};
	fetch(k0);
	const int      lx = threadIdx.x & 63, x = x0 + lx, ly0 = (int) (threadIdx.x >> 6) * 2;        // lane = x, each wave two y rows, all z
	const uint32_t off = (uint32_t) (y0 + ly0) * (uint32_t) W + (uint32_t) x, off1 = off + (uint32_t) W;        // inside one z slice (< 2^32 voxels)
	constexpr int  sy = kGradPitch, sz = (kGradTileY + 2) * kGradPitch;
	// position of texel i of a row in the tile (0, 2, 1, 3 within every four), for the lane's x - 1 and x + 1; texel x sits at column x - x0 + 4
	auto      column = [](int i) { return (i & ~3) | ((i & 1) << 1) | ((i >> 1) & 1); };
	const int cm = column(4 + lx - 1), cp = column(4 + lx + 1);
	for (uint32_t k = k0; k < k1; ++k)
	{
		__syncthreads();        // the previous tile has been read by everyone
#pragma unroll
		for (int j = 0; j < kIter; ++j)
			if ((int) threadIdx.x + 256 * j < kRows * kCols)
				reinterpret_cast<uint2 *>(s_tile)[threadIdx.x + 256 * j] = make_uint2((v[j] & 0x00ff00ffu) << 2, (v[j] & 0xff00ff00u) >> 6);
		__syncthreads();
		if (k + 1 < k1)
			fetch(k + 1);        // in flight during the arithmetic below
		const int z0 = (int) k * kGradTileZ;
		// the four taps of TWO voxels (rows ly0, ly0 + 1 of slice lz; k.xyy, k.yyx, k.yxy, k.xxx of get_gradient_compute.glsl:8-11): all reads
		// of a stage are issued before the first is used
		auto taps = [&](int lz, float(&a)[4], float(&b)[4]) {
			const int      r  = ((lz + 1) * (kGradTileY + 2) + (ly0 + 1)) * kGradPitch;
			const uint32_t a0 = s_tile[r + cp - sy - sz], a1 = s_tile[r + cm - sy + sz], a2 = s_tile[r + cm + sy - sz], a3 = s_tile[r + cp + sy + sz];
			const uint32_t b0 = s_tile[r + cp - sz], b1 = s_tile[r + cm + sz], b2 = s_tile[r + cm + 2 * sy - sz], b3 = s_tile[r + cp + 2 * sy + sz];
			__builtin_amdgcn_wave_barrier();
			auto at = [&](uint32_t o) { return *reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(s_unorm) + o); };
			a[0] = at(a0), a[1] = at(a1), a[2] = at(a2), a[3] = at(a3);
			b[0] = at(b0), b[1] = at(b1), b[2] = at(b2), b[3] = at(b3);
			__builtin_amdgcn_wave_barrier();
		};
		// get_gradient_compute.glsl:12-20, the operations of gradient_from_taps with the short exact sqrt.  The three factors 0.25 are
		// taken out: scaling by a power of two commutes with every rounding on the way (squares x 2^-4, their sums, the root x 2^-2;
		// nothing comes near the denormal range: the sum is 0 or >= ~1e-17), so sqrt(sum of (0.25 s)^2) * m == sqrt(sum of s^2) * (0.25 m)
		// bit for bit (quarter_modifier = 0.25 * modifier is exact as well).
		// (The empty asm statements keep the three chains scalar: the packed v_pk_add_f32 the compiler forms otherwise issue at 4.4 cycles
		// for two results, no faster than two full-rate scalar operations, and cost two v_mov to assemble their operands.)
		auto finish = [&](const float(&q)[4]) -> uint32_t {
			float tx = q[0] - q[1], ty = -q[0] - q[1];
			asm volatile("" : "+v"(tx));
			asm volatile("" : "+v"(ty));
			float       sx = (tx - q[2]) + q[3], sy_ = (ty + q[2]) + q[3];
			const float sz_ = ((-q[0] + q[1]) - q[2]) + q[3];
			asm volatile("" : "+v"(sx));
			asm volatile("" : "+v"(sy_));
			const float len = sqrt_rn_normal((sx * sx + sy_ * sy_) + sz_ * sz_);
			return store_unorm8_clamped(len * quarter_modifier);
		};
		uint8_t *gz = grad + (size_t) z0 * (size_t) W * (size_t) H;        // wave-uniform base, advanced per slice on the scalar unit
		if (x0 + kGradTileX <= W && y0 + kGradTileY <= H && z0 + kGradTileZ <= D)
		{        // interior tile: no masks, the eight slices unrolled
#pragma unroll
			for (int lz = 0; lz < kGradTileZ; ++lz)
			{
				float a[4], b[4];
				taps(lz, a, b);
				store_u8_uniform_base(gz, off, finish(a));
				store_u8_uniform_base(gz, off1, finish(b));
				gz += (size_t) W * (size_t) H;
			}
		}
		else if (x < W)
			for (int lz = 0; lz < kGradTileZ && z0 + lz < D; ++lz)
			{
				float a[4], b[4];
				taps(lz, a, b);
				if (y0 + ly0 < H)
					gz[off] = (uint8_t) finish(a);
				if (y0 + ly0 + 1 < H)
					gz[off1] = (uint8_t) finish(b);
				gz += (size_t) W * (size_t) H;
			}
	}
}

// ---------------------------------------------------------------------------------------------
// Occupancy map (shaders/occupancy_map.comp:45-73)
// ---------------------------------------------------------------------------------------------
// alpha > 0 is a pure function of (gradient texel, intensity texel) of the NEAREST-sampled TF texture, so the
// 256x256 alpha channel is first reduced to a 1-bit table (8 KiB) that every block stages in LDS.
__global__ void __launch_bounds__(256) k_tf_bits(const uint8_t *__restrict__ tf_rgba8, uint32_t *__restrict__ bits)
{
	const uint32_t w = blockIdx.x * 256 + threadIdx.x;        // 2048 words
	if (w >= 2048)
		return;
	uint32_t v = 0;
	for (int i = 0; i < 32; ++i)
		v |= (tf_rgba8[((size_t) w * 32 + i) * 4 + 3] > 0 ? 1u : 0u) << i;
	bits[w] = v;
}

// Which INTENSITY bytes can be occupied at all (alpha > 0 for some gradient byte): 256 bits behind the bit table (words 2048..2055).  A voxel
// whose intensity is not among them is empty whatever its gradient: k_occupancy_map_waves tests the lowest such byte against whole batches of
// volume dwords before it touches the gradient map or the table.
__global__ void __launch_bounds__(256) k_tf_columns(uint32_t *__restrict__ bits)
{
	const uint32_t v = threadIdx.x;        // intensity byte = texture column
	uint32_t       any = 0;
	for (uint32_t g = 0; g < 256; ++g)
		any |= bits[g * 8u + (v >> 5)];
	const unsigned long long m = __builtin_amdgcn_ballot_w64(((any >> (v & 31u)) & 1u) != 0u);
	if ((v & 63u) == 0u)
		bits[2048u + (v >> 5)] = (uint32_t) m, bits[2048u + (v >> 5) + 1u] = (uint32_t) (m >> 32);
}

// Block = 256 threads = 256 consecutive voxels in x of one cell row (cy, cz); it walks the by*bz voxel rows of
// that cell row, every load being a coalesced 64-byte segment per wave, ORs "occupied" into one LDS flag per cell.
// GRAD: 0 = use_gradient false (gradient = 1.0), 1 = precomputed map, 2 = on-the-fly tetrahedron.
// Same dword streaming for any block width: a workgroup owns `cpb` whole cells of a cell row (cpb a multiple of 4, so its first
// voxel is dword aligned), a thread ORs the occupied bits of its 4 voxels over the by x bz rows and then flags their cells in LDS.
template <int GRAD>
__global__ void __launch_bounds__(256) k_occupancy_map_dword_any(const uint8_t *__restrict__ vol, const uint8_t *__restrict__ grad,
                                                                 const uint32_t *__restrict__ tf_bits, uint8_t *__restrict__ map, int W, int H, int D,
                                                                 int mw, int mh, int bx, int by, int bz, int cpb, uint32_t blocks_x)
{
	__shared__ uint32_t s_bits[2048];
	__shared__ uint32_t s_cell[1024];
	for (int i = threadIdx.x; i < 2048; i += 256)
		s_bits[i] = tf_bits[i];
	for (int i = threadIdx.x; i < cpb; i += 256)
		s_cell[i] = 0;
	__syncthreads();
	const int c0 = (int) (blockIdx.x % blocks_x) * cpb, x0 = c0 * bx;        // first cell / voxel of the workgroup
	const int cy = (int) (blockIdx.x / blocks_x), cz = (int) blockIdx.y;
	const int y_end = min((cy + 1) * by, H), z_end = min((cz + 1) * bz, D);
	const int nd    = (min(cpb * bx, W - x0) + 3) >> 2;        // dwords of the span (W % 4 == 0: never past the row)
	for (int d = threadIdx.x; d < nd; d += 256)
	{
		const int     xd     = (x0 >> 2) + d;
		uint32_t      occ    = 0;
		constexpr int kBatch = 8;        // rows fetched together (see k_occupancy_map_dword)
		const int     ny = y_end - cy * by, nz = z_end - cz * bz, n_rows = (ny > 0 && nz > 0) ? ny * nz : 0;        // (a cell row / slice past the volume: no rows)
		for (int r0 = 0; r0 < n_rows; r0 += kBatch)
		{
			uint32_t v4[kBatch], g4[kBatch];
#pragma unroll
			for (int j = 0; j < kBatch; ++j)
			{
				const int    r   = min(r0 + j, n_rows - 1);
				const size_t row = ((size_t) (cz * bz + r / ny) * H + (size_t) (cy * by + r % ny)) * (size_t) W;
				v4[j]            = reinterpret_cast<const uint32_t *>(vol + row)[xd];
				g4[j]            = GRAD == 1 ? reinterpret_cast<const uint32_t *>(grad + row)[xd] : 0xffffffffu;
			}
#pragma unroll
			for (int j = 0; j < kBatch; ++j)
#pragma unroll
				for (int i = 0; i < 4; ++i)
				{
					const uint32_t bit = ((g4[j] >> (8 * i)) & 255u) * 256u + ((v4[j] >> (8 * i)) & 255u);
					occ |= ((s_bits[bit >> 5] >> (bit & 31u)) & 1u) << i;
				}
		}
		if (occ)
		{
			int cell = (4 * d) / bx, in = (4 * d) - cell * bx;
#pragma unroll
			for (int i = 0; i < 4; ++i)
			{
				if ((occ >> i) & 1u)
					s_cell[cell] = 1;        // benign race: every writer stores 1
				if (++in == bx)
					in = 0, ++cell;
			}
		}
	}
	__syncthreads();
	for (int c = threadIdx.x; c < cpb && c0 + c < mw; c += 256)
		map[((size_t) cz * mh + cy) * (size_t) mw + (size_t) (c0 + c)] = s_cell[c] ? 0 : 255;        // OCCUPIED = 0, EMPTY = 255
}

// Any block width at streaming rate (round 5; the sweep's block sizes 2, 3, 5, 6 - and 3 is where the reference publishes its best frame
// rates).  What the two kernels around this one lose on small or odd blocks: a workgroup per cell row streams by * bz voxel rows (9 for a 3^3
// block: 18 KB) for 8 KB of staged bit table, meets at barriers for every cell row, and fetches rows in eights, so a 9-row cell pays for 16.
//   * The map is first filled with EMPTY (255) and the kernel only ever stores OCCUPIED (0): any number of lanes may then hit the same cell
//     in any order, so nothing has to be aligned to cells and nothing is exchanged.
//   * Every WAVE is on its own: it owns 64 consecutive dwords (256 voxels) in x and walks `cy_per_wave` consecutive cell rows of them; the
//     bit table is staged once per workgroup, i.e. once per >= 256 KB of voxels, and that is the kernel's only barrier.
//   * A wave fetches exactly ROWS voxel rows per batch - ROWS / CRB rows of each of CRB cell rows (the launcher picks ROWS from {8, 9, 10, 12}
//     so that by * bz rows leave the fewest idle slots: 9 -> 9, 16 -> 8, 25 -> 27, 36 -> 36, 49 -> 50; cells of fewer than eight rows are
//     taken CRB = 2, 4 or 8 cell rows at a time: a 2^3 block's four rows two cell rows at a time).
//   * Short circuit: the VOLUME dwords of a batch come first.  If no byte of them reaches the lowest intensity the transfer function gives any
//     alpha to (k_tf_columns; four bytes per SWAR test), every voxel of the batch is empty whatever its gradient: the wave goes on to the next
//     batch without reading the gradient map or the table - on the bench volume that is most batches, i.e. close to half the bytes
//     occupancy_map.comp reads.  Otherwise the gradient rows follow and every voxel is looked up (same result either way: the test is exact).
//   * A lane ORs the occupied bits of its four voxels over a cell row's rows and stores a 0 into the cells of the voxels that have one (a
//     lane's four voxels lie in the same cells in every cell row: the x -> cell division is done once).
// a wave-uniform 64-bit value moved into scalar registers (the compiler does not always see the uniformity through the loops' counters)
__device__ __forceinline__ size_t uniform_u64(size_t x)
{
	const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) x), hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (x >> 32));
	return ((size_t) hi << 32) | lo;
}

template <int GRAD, int ROWS, int CRB>
__global__ void __launch_bounds__(256) k_occupancy_map_waves(const uint8_t *__restrict__ vol, const uint8_t *__restrict__ grad, const uint32_t *__restrict__ tf_bits,
                                                             uint8_t *__restrict__ map, int W, int H, int D, int mw, int mh, int bx, int by, int bz, uint32_t spans_x,
                                                             int cy_per_wave)
{
	__shared__ uint32_t s_bits[2048];
	for (int i = threadIdx.x; i < 2048; i += 256)
		s_bits[i] = tf_bits[i];
	// lowest intensity byte with any alpha (256: none - the map stays empty; 0: no short circuit possible)
	uint32_t vmin = 256u;
	for (int w = 7; w >= 0; --w)
	{
		const uint32_t m = tf_bits[2048 + w];        // wave-uniform scalar loads
		if (m)
			vmin = (uint32_t) w * 32u + (uint32_t) __builtin_ctz(m);
	}
	__syncthreads();
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const uint32_t task = blockIdx.x * 4u + wave;        // (span of 64 dwords in x, group of cell rows); blockIdx.y = cell slice
	const int      xd = (int) (task % spans_x) * 64 + (int) lane, cy0 = (int) (task / spans_x) * cy_per_wave, cz = (int) blockIdx.y;
	if (cy0 >= mh || xd * 4 >= W || vmin == 256u)
		return;
	// the lane's byte offset inside a row: scalar row base + 32-bit lane offset per load.  The last lane of an odd width (x + 4 > W) loads the
	// row's last four bytes instead and shifts them down (row_dword); its voxels past the row are masked out of the result
	const uint32_t xo_ = (uint32_t) xd * 4u, xo = min(xo_, (uint32_t) (W - 4)), shr = (xo_ - xo) * 8u;
	const uint32_t valid   = xo_ + 4u <= (uint32_t) W ? 0xfu : (1u << ((uint32_t) W - xo_)) - 1u;
	const bool     shifted = __builtin_amdgcn_ballot_w64(shr != 0u) != 0ull;        // wave-uniform: only the wave that holds the row's end
	// "some byte of x is >= vmin" for four bytes at once: byte + (256 - vmin) carries out of its eight bits.  The low seven bits are added
	// per byte (no carry between bytes: 127 + 127 < 256), the carry out of bit 7 is then majority(byte's bit 7, the constant's bit 7, the
	// sum's bit 7), i.e. an OR or an AND with the constant's bit 7 known
	const uint32_t cadd = 256u - vmin, c_hi = cadd >> 7, c_lo4 = (cadd & 127u) * 0x01010101u;
	// the cells of this lane's four voxels: cell0 = the first voxel's, step bit i = voxel i lies in the next cell (one store per cell)
	uint32_t cell0 = (uint32_t) (4 * xd) / (uint32_t) bx, step = 0u;
	{
		uint32_t in = (uint32_t) (4 * xd) - cell0 * (uint32_t) bx;
#pragma unroll
		for (int i = 1; i < 4; ++i)
			if (++in == (uint32_t) bx)
				in = 0, step |= 1u << i;
	}
	const int    z_end = min((cz + 1) * bz, D), nz = z_end - cz * bz;
	// a map may have more cell rows / slices than the volume fills (map extent > ceil(extent / block), e.g. 16^3 voxels under a 7^3 map:
	// block 3, cells 6 of y and z lie outside): those cells hold no voxel and stay EMPTY, like in the workgroup-per-cell-row kernels
	const int    cy_end = min(min(cy0 + cy_per_wave, mh), (H + by - 1) / by);
	if (nz <= 0 || cy0 >= cy_end)
		return;
	const size_t zs = (size_t) H * (size_t) W;
	for (int cy = cy0; cy < cy_end; cy += CRB)
	{
		uint32_t occ[CRB];
#pragma unroll
		for (int k = 0; k < CRB; ++k)
			occ[k] = 0;
		const int ny     = min((cy + 1) * by, H) - cy * by;        // CRB == 1: rows of this cell row in y (whole cells otherwise)
		const int n_rows = CRB == 1 ? ny * nz : ROWS;
		int       ry = 0, rz = 0;        // CRB == 1: the (y, z) of the next row inside the cell row (wave-uniform counters instead of a division per load)
		for (int r0 = 0; r0 < n_rows; r0 += ROWS)
		{
			size_t   row[ROWS];
			uint32_t v4[ROWS];
#pragma unroll
			for (int j = 0; j < ROWS; ++j)
			{
				if (CRB == 1)
				{
					row[j] = (size_t) (cz * bz + rz) * zs + (size_t) (cy * by + ry) * (size_t) W;
					if (!(rz == nz - 1 && ry == ny - 1))        // the tail repeats the last row (idempotent OR; the line is in L1)
					{
						if (++ry == ny)
							ry = 0, ++rz;
					}
				}
				else
				{        // by * bz = ROWS / CRB rows per cell row; the rows a ragged last cell row / slice does not have repeat its last one
					constexpr int kPer = ROWS / CRB;
					const int     c = min(cy + j / kPer, cy_end - 1), r = j % kPer;        // (a group's last batch repeats its last cell row)
					row[j]          = (size_t) (cz * bz + min(r / by, nz - 1)) * zs + (size_t) min(c * by + r % by, H - 1) * (size_t) W;
				}
				row[j] = uniform_u64(row[j]);        // wave-uniform by construction: a scalar base for the two loads of this row
				v4[j]  = load_u32_any(vol + row[j] + xo);
				if (shifted)
					v4[j] >>= shr;
			}
			if (vmin != 0u)
			{
				uint32_t acc = 0;
#pragma unroll
				for (int j = 0; j < ROWS; ++j)
				{
					const uint32_t sum = (v4[j] & 0x7f7f7f7fu) + c_lo4;
					acc |= c_hi ? (v4[j] | sum) : (v4[j] & sum);
				}
				if (__builtin_amdgcn_ballot_w64((acc & 0x80808080u) != 0u) == 0ull)
					continue;        // the whole batch of the whole wave is below the transfer function's threshold
			}
			uint32_t g4[ROWS];
#pragma unroll
			for (int j = 0; j < ROWS; ++j)
			{
				g4[j] = GRAD == 1 ? load_u32_any(grad + row[j] + xo) : 0xffffffffu;
				if (GRAD == 1 && shifted)
					g4[j] >>= shr;
			}
#pragma unroll
			for (int j = 0; j < ROWS; ++j)
#pragma unroll
				for (int i = 0; i < 4; ++i)
				{
					const uint32_t bit = ((g4[j] >> (8 * i)) & 255u) * 256u + ((v4[j] >> (8 * i)) & 255u);
					occ[CRB == 1 ? 0 : j / (ROWS / CRB)] |= ((s_bits[bit >> 5] >> (bit & 31u)) & 1u) << i;
				}
		}
#pragma unroll
		for (int k = 0; k < CRB; ++k)
		{
			occ[k] &= valid;        // (the voxels a row's last lane holds past the row's end)
			if (cy + k >= cy_end || occ[k] == 0u)
				continue;
			uint8_t *out = map + ((size_t) cz * mh + (size_t) (cy + k)) * (size_t) mw;
			// OR the bits of the voxels that share a cell, then one store per occupied cell
			uint32_t c = cell0, any = 0u;
#pragma unroll
			for (int i = 0; i < 4; ++i)
			{
				if (i > 0 && ((step >> i) & 1u))
				{
					if (any)
						out[c] = 0;        // OCCUPIED
					++c, any = 0u;
				}
				any |= (occ[k] >> i) & 1u;
			}
			if (any)
				out[c] = 0;
		}
	}
}

template <int GRAD>
__global__ void __launch_bounds__(256) k_occupancy_map(const uint8_t *__restrict__ vol, const uint8_t *__restrict__ grad,
                                                       const uint32_t *__restrict__ tf_bits, uint8_t *__restrict__ map, int W, int H, int D,
                                                       int mw, int mh, int md, int bx, int by, int bz, float modifier, uint32_t blocks_x)
{
	__shared__ uint32_t s_bits[2048];
	__shared__ uint32_t s_cell[256];
	for (int i = threadIdx.x; i < 2048; i += 256)
		s_bits[i] = tf_bits[i];
	s_cell[threadIdx.x] = 0;
	__syncthreads();

	const uint32_t bxi = blockIdx.x % blocks_x;
	const int      cy = (int) (blockIdx.x / blocks_x), cz = (int) blockIdx.y;        // grid = (x blocks * cell rows, cell slices)
	// the block covers cells [c0, c0 + cells_per_block) of this cell row
	const int cells_per_block = 256 / bx > 0 ? 256 / bx : 1;
	const int c0              = (int) bxi * cells_per_block;
	const int x0              = c0 * bx;
	const int span            = cells_per_block * bx;        // voxels handled per pass (<= 256 unless bx > 256)
	const int y_end = min((cy + 1) * by, H), z_end = min((cz + 1) * bz, D);

	uint32_t occ = 0;
	for (int lx = threadIdx.x; lx < span; lx += 256)
	{
		const int x = x0 + lx;
		if (x >= W)
			break;
		occ = 0;
		for (int z = cz * bz; z < z_end; ++z)
			for (int y = cy * by; y < y_end; ++y)
			{
				const size_t   o = vidx(x, y, z, W, H);
				const uint32_t v = vol[o];
				uint32_t       g;
				if (GRAD == 1)
					g = grad[o];        // unorm8 byte: NEAREST lookup of b/255 lands on texel b
				else if (GRAD == 2)
					g = (uint32_t) tf_texel(gradient_on_the_fly(vol, W, H, D, x, y, z, modifier));
				else
					g = 255;        // gradient = 1.0 -> texel 255
				const uint32_t bit = g * 256 + v;
				occ |= (s_bits[bit >> 5] >> (bit & 31)) & 1u;
			}
		if (occ)
			s_cell[lx / bx] = 1;        // benign race: every writer stores 1
	}
	__syncthreads();
	const int c = c0 + (int) threadIdx.x;
	if ((int) threadIdx.x < cells_per_block && c < mw)
		map[vidx(c, cy, cz, mw, mh)] = s_cell[threadIdx.x] ? 0 : 255;        // OCCUPIED = 0, EMPTY = 255
}

// Dword path of the occupancy pass for the common block sizes whose x extent divides a dword (bx = 1, 2, 4) on dword-aligned
// rows (W % 4 == 0): one thread owns 4 consecutive voxels in x, i.e. 4 / bx whole cells, and walks the by * bz voxel rows of its
// cell row with one volume dword (+ one gradient dword) per row — 256 contiguous bytes per wave-load instead of 64 — and no
// LDS flags or second barrier.  GRAD 0 / 1 only (the on-the-fly gradient variant stays on the byte path).
template <int GRAD, int BX>
__global__ void __launch_bounds__(256) k_occupancy_map_dword(const uint8_t *__restrict__ vol, const uint8_t *__restrict__ grad,
                                                             const uint32_t *__restrict__ tf_bits, uint8_t *__restrict__ map, int W, int H, int D, int mw,
                                                             int mh, int by, int bz, uint32_t blocks_x)
{
	__shared__ uint32_t s_bits[2048];
	for (int i = threadIdx.x; i < 2048; i += 256)
		s_bits[i] = tf_bits[i];
	__syncthreads();
	const int xd = (int) (blockIdx.x % blocks_x) * 256 + (int) threadIdx.x;        // dword column
	const int cy = (int) (blockIdx.x / blocks_x), cz = (int) blockIdx.y;
	if (xd * 4 >= W)
		return;
	const int y_end = min((cy + 1) * by, H), z_end = min((cz + 1) * bz, D);
	uint32_t  occ   = 0;        // bit i: voxel 4*xd + i of some row is occupied
	// the by x bz rows of the cell are fetched eight at a time (16 loads in flight per lane: the kernel is a pure stream and
	// bandwidth = bytes in flight / latency)
	constexpr int kBatch = 8;
	const int     ny = y_end - cy * by, nz = z_end - cz * bz, n_rows = (ny > 0 && nz > 0) ? ny * nz : 0;        // (a cell row / slice past the volume: no rows)
	for (int r0 = 0; r0 < n_rows; r0 += kBatch)
	{
		uint32_t v4[kBatch], g4[kBatch];
#pragma unroll
		for (int j = 0; j < kBatch; ++j)
		{
			const int    r   = min(r0 + j, n_rows - 1);        // the tail repeats the last row (idempotent OR)
			const int    z   = cz * bz + r / ny, y = cy * by + r % ny;
			const size_t row = ((size_t) z * H + y) * (size_t) W;
			v4[j]            = reinterpret_cast<const uint32_t *>(vol + row)[xd];
			g4[j]            = GRAD == 1 ? reinterpret_cast<const uint32_t *>(grad + row)[xd] : 0xffffffffu;        // gradient 1.0 -> texel 255
		}
#pragma unroll
		for (int j = 0; j < kBatch; ++j)
#pragma unroll
			for (int i = 0; i < 4; ++i)
			{
				const uint32_t bit = ((g4[j] >> (8 * i)) & 255u) * 256u + ((v4[j] >> (8 * i)) & 255u);
				occ |= ((s_bits[bit >> 5] >> (bit & 31u)) & 1u) << i;
			}
	}
	constexpr int      kCells = 4 / BX;
	constexpr uint32_t kMask  = (1u << BX) - 1u;
	uint8_t *          out    = map + ((size_t) cz * mh + cy) * (size_t) mw + (size_t) xd * kCells;
#pragma unroll
	for (int c = 0; c < kCells; ++c)
		if (xd * kCells + c < mw)
			out[c] = ((occ >> (c * BX)) & kMask) ? 0 : 255;        // OCCUPIED = 0, EMPTY = 255
}

// ---------------------------------------------------------------------------------------------
// Occupied-voxel count (shaders/occupied_voxel_count.comp + occupied_voxel_count_reduce.comp)
// ---------------------------------------------------------------------------------------------
// analytic get_color (transfer_function.glsl:41-43) > 0
__device__ __forceinline__ bool analytic_occupied(float intensity, float gradient, float imin, float iinv, float gmin, float ginv)
{
	const float ai = g_clamp((intensity - imin) * iinv, 0.0f, 1.0f);
	const float ag = g_clamp((gradient - gmin) * ginv, 0.0f, 1.0f);
	return ai * ag > 0.0f;
}

// The statistic's alpha is again a pure function of the (gradient byte, intensity byte) pair when the gradient comes
// from the R8_UNORM map: reduce the analytic TF to the same 8 KiB bit table the occupancy pass uses.
__global__ void __launch_bounds__(256) k_tf_bits_analytic(uint32_t *__restrict__ bits, float imin, float iinv, float gmin, float ginv)
{
	const uint32_t w = blockIdx.x * 256 + threadIdx.x;        // 2048 words, 32 intensity texels each
	if (w >= 2048)
		return;
	const uint32_t g = w >> 3;
	uint32_t       v = 0;
	for (uint32_t i = 0; i < 32; ++i)
		v |= (analytic_occupied(unorm8((w & 7u) * 32 + i), unorm8(g), imin, iinv, gmin, ginv) ? 1u : 0u) << i;
	bits[w] = v;
}

// GRAD as in k_occupancy_map.  Each thread walks voxels with a grid stride of whole x rows; a wave counts with
// ballot + popcount, a workgroup adds once to the 64-bit total.
template <int GRAD, int DWORDS>        // DWORDS: 0 = a voxel per lane, 1 = four per lane from aligned dwords (W % 4 == 0), 2 = four per lane at any alignment / width
__global__ void __launch_bounds__(256) k_occupied_voxel_count(const uint8_t *__restrict__ vol, const uint8_t *__restrict__ grad,
                                                              const uint32_t *__restrict__ tf_bits, unsigned long long *__restrict__ total, int W, int H,
                                                              int D, float modifier, float imin, float iinv, float gmin, float ginv, uint32_t blocks_x,
                                                              uint32_t n_row_groups)
{
	__shared__ uint32_t s_bits[2048];
	__shared__ uint32_t s_wave[4];
	for (int i = threadIdx.x; i < 2048; i += 256)
		s_bits[i] = tf_bits[i];
	__syncthreads();
	const uint32_t bx   = blockIdx.x % blocks_x;
	const uint32_t rows = (uint32_t) H * (uint32_t) D;
	uint32_t       n    = 0;
	if (GRAD != 2 && DWORDS)
	{        // dword-aligned rows: one lane tests 4 voxels per load (256 contiguous bytes per wave-load)
		const int      xd     = (int) (bx * 64 + (threadIdx.x & 63));        // dword column
		const uint32_t stride = gridDim.x / blocks_x;
		constexpr int  kBatch = 8;        // row groups fetched together: 16 loads in flight per lane
		// (any width >= 4: the row's last lane reads the row's last four bytes and shifts, row_dword; its voxels past the row do not count)
		const int n_valid = min(4, W - 4 * xd);
		if (xd * 4 < W)
			for (uint32_t rg0 = blockIdx.x / blocks_x; rg0 < n_row_groups; rg0 += stride * kBatch)
			{
				uint32_t v4[kBatch], g4[kBatch];
#pragma unroll
				for (int j = 0; j < kBatch; ++j)
				{
					const uint32_t row = (rg0 + (uint32_t) j * stride) * 4 + (threadIdx.x >> 6);
					const bool     ok  = rg0 + (uint32_t) j * stride < n_row_groups && row < rows;
					const size_t   o   = (size_t) (ok ? row : 0u) * (size_t) W;
					if (DWORDS == 1)
					{
						v4[j] = ok ? reinterpret_cast<const uint32_t *>(vol + o)[xd] : 0u;
						g4[j] = (ok && GRAD == 1) ? reinterpret_cast<const uint32_t *>(grad + o)[xd] : (ok ? 0xffffffffu : 0u);
					}
					else
					{
						v4[j] = ok ? row_dword(vol + o, xd, W) : 0u;
						g4[j] = (ok && GRAD == 1) ? row_dword(grad + o, xd, W) : (ok ? 0xffffffffu : 0u);
					}
				}
#pragma unroll
				for (int j = 0; j < kBatch; ++j)
				{
					const bool ok = rg0 + (uint32_t) j * stride < n_row_groups && (rg0 + (uint32_t) j * stride) * 4 + (threadIdx.x >> 6) < rows;
#pragma unroll
					for (int i = 0; i < 4; ++i)
					{
						const uint32_t bit = ((g4[j] >> (8 * i)) & 255u) * 256u + ((v4[j] >> (8 * i)) & 255u);
						n += (ok && (DWORDS == 1 || i < n_valid)) ? (s_bits[bit >> 5] >> (bit & 31u)) & 1u : 0u;        // per-lane partial sums, reduced below
					}
				}
			}
		for (int o2 = 32; o2 > 0; o2 >>= 1)
			n += (uint32_t) __shfl_xor((int) n, o2);
	}
	else
	{
	const int x = (int) (bx * 64 + (threadIdx.x & 63));
	for (uint32_t rg = blockIdx.x / blocks_x; rg < n_row_groups; rg += gridDim.x / blocks_x)
	{
		const uint32_t row = rg * 4 + (threadIdx.x >> 6);
		bool           occ = false;
		if (x < W && row < rows)
		{
			const int      y = (int) (row % (uint32_t) H), z = (int) (row / (uint32_t) H);
			const size_t   o = vidx(x, y, z, W, H);
			const uint32_t v = vol[o];
			if (GRAD == 2)
				occ = analytic_occupied(unorm8(v), gradient_on_the_fly(vol, W, H, D, x, y, z, modifier), imin, iinv, gmin, ginv);
			else
			{
				const uint32_t bit = (GRAD == 1 ? (uint32_t) grad[o] : 255u) * 256 + v;
				occ                = (s_bits[bit >> 5] >> (bit & 31)) & 1u;
			}
		}
		n += (uint32_t) __popcll(__ballot(occ));        // wave-uniform
	}
	}
	if ((threadIdx.x & 63) == 0)
		s_wave[threadIdx.x >> 6] = n;
	__syncthreads();
	if (threadIdx.x == 0)
		atomicAdd(total, (unsigned long long) s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3]);
}

// ---------------------------------------------------------------------------------------------
// Chebyshev distance transform (shaders/distance_map.comp, distance_map_anisotropic.comp)
// ---------------------------------------------------------------------------------------------
// x pass.  One wave stages 64 rows in LDS (coalesced), each lane scans one row serially out of LDS (row stride
// chosen so the 64 lanes hit distinct banks), then the rows are written back coalesced.
// MODE 0: isotropic two-sided in-place scan (distance_map.comp:57-71);
// MODE +1 / -1: one-sided scan of the anisotropic shader (distance_map_anisotropic.comp:44-53).
template <int MODE>
__global__ void __launch_bounds__(64) k_dm_x(const uint8_t *src, uint8_t *dst, int mw, uint32_t n_rows, int stride)        // src may alias dst (in-place)
{
	extern __shared__ __align__(16) uint8_t s_rows[];
	const uint32_t row0 = blockIdx.x * 64;
	const int      lane = threadIdx.x;
	const uint32_t rows = min(64u, n_rows - row0);
	for (uint32_t r = 0; r < rows; ++r)
	{
		const uint8_t *g = src + (size_t) (row0 + r) * (size_t) mw;
		for (int x = lane; x < mw; x += 64)
			s_rows[r * stride + x] = g[x];
	}
	__syncthreads();
	if ((uint32_t) lane < rows)
	{
		uint8_t *row = s_rows + lane * stride;
		if (MODE == 0)
		{
			uint32_t g1 = row[0];
			for (int x = 1; x < mw; ++x)
			{
				const uint32_t g = min(g1 + 1, (uint32_t) row[x]);
				row[x]           = (uint8_t) g;
				g1               = g;
			}
			for (int x = mw - 2; x >= 0; --x)
			{
				const uint32_t g = min(g1 + 1, (uint32_t) row[x]);
				row[x]           = (uint8_t) g;
				g1               = g;
			}
		}
		else
		{
			const int start = MODE > 0 ? mw - 1 : 0, end = MODE > 0 ? -1 : mw;
			uint32_t  g1    = row[start];
			for (int x = start; x != end; x -= MODE)
			{
				const uint32_t g = min(g1 + 1, (uint32_t) row[x]);
				row[x]           = (uint8_t) g;
				g1               = g;
			}
		}
	}
	__syncthreads();
	for (uint32_t r = 0; r < rows; ++r)
	{
		uint8_t *g = dst + (size_t) (row0 + r) * (size_t) mw;
		for (int x = lane; x < mw; x += 64)
			g[x] = s_rows[r * stride + x];
	}
}

// x pass for rows of up to 1024 cells: one WAVE per row, the row in registers (C cells per lane).
// Stage 0 of both shaders is the min-plus recurrence g = min(g_prev + 1, occ) (distance_map.comp:57-71 forward and backward in place,
// distance_map_anisotropic.comp:44-53 one-sided), whose closed form is
//     out(x) = min over q of (g(q) + |x - q|)            (one-sided: q >= x for dir > 0, q <= x for dir < 0),
// i.e. x + the running minimum of g(q) - q from the left, and the running minimum of g(q) + q from the right minus x: two scans - inside
// the lane over its C cells, across the lanes with six shuffle steps.  Exact for ANY byte input (not only 0 / 255 occupancy), like the
// recurrence.  Reads the whole row before it writes: dst may alias src.  MODE 0: two-sided -> dst; +1 / -1: one-sided -> dst;
// 2: dst = +1 result, dst2 = -1 result.  VEC: rows are dword-aligned (mw % 4 == 0, aligned pointers).
template <int MODE, int C, bool VEC>
__global__ void __launch_bounds__(256) k_dm_x_wave(const uint8_t *src, uint8_t *dst, uint8_t *dst2, int mw, uint32_t n_rows)
{
	// a wave takes kRows consecutive rows and has the loads of all of them in flight before it scans the first (one row is 256 bytes to 1 KB:
	// with one row per wave a CU keeps 8 KB in flight and the pass runs at 1.4 TB/s)
	constexpr int  kRows = C <= 8 ? 4 : (C <= 16 ? 2 : 1);        // (C = 32, rows of 1025 .. 2048 cells - round 6: a row is 1 - 2 KB, one per wave)
	const uint32_t row0  = (blockIdx.x * 4u + (threadIdx.x >> 6)) * kRows;
	if (row0 >= n_rows)
		return;        // wave-uniform
	const int     lane = (int) (threadIdx.x & 63u), x0 = lane * C;
	constexpr int kFar = 1 << 20;        // padding cells past the end of the row: never the minimum
	uint32_t      raw[kRows][C / 4];
	uint8_t       rawb[VEC ? 1 : kRows][VEC ? 1 : C];
#pragma unroll
	for (int r = 0; r < kRows; ++r)
	{
		const size_t ro = (size_t) min(row0 + (uint32_t) r, n_rows - 1u) * (size_t) mw;        // (rows past the end: the last row again, not stored)
		if (VEC)
		{
#pragma unroll
			for (int j = 0; j < C / 4; ++j)
				raw[r][j] = x0 + 4 * j < mw ? *reinterpret_cast<const uint32_t *>(src + ro + x0 + 4 * j) : 0u;
		}
		else
		{
#pragma unroll
			for (int i = 0; i < C; ++i)
				rawb[VEC ? 0 : r][VEC ? 0 : i] = x0 + i < mw ? src[ro + x0 + i] : (uint8_t) 0;
		}
	}
#pragma unroll
	for (int r = 0; r < kRows; ++r)
	{
		if (row0 + (uint32_t) r >= n_rows)
			break;        // wave-uniform
		const size_t ro = (size_t) (row0 + (uint32_t) r) * (size_t) mw;
		int          g[C];
#pragma unroll
		for (int i = 0; i < C; ++i)
		{
			const int v = VEC ? (int) ((raw[r][i / 4] >> (8 * (i & 3))) & 255u) : (int) rawb[VEC ? 0 : r][VEC ? 0 : i];
			g[i]        = x0 + i < mw ? v : kFar;
		}
		// running minimum of g(q) - q over q <= x (left candidates) and of g(q) + q over q >= x (right candidates), inside the lane
		int pre[C], suf[C];
		{
			int m = 2 * kFar;
#pragma unroll
			for (int i = 0; i < C; ++i)
				m = min(m, g[i] - (x0 + i)), pre[i] = m;
			m = 2 * kFar;
#pragma unroll
			for (int i = C - 1; i >= 0; --i)
				m = min(m, g[i] + (x0 + i)), suf[i] = m;
		}
		// exclusive scans over the lanes: `left` = minimum over all lower lanes, `right` = over all higher lanes
		int left = pre[C - 1], right = suf[0];
#pragma unroll
		for (int o = 1; o < 64; o <<= 1)
		{
			const int l = __shfl_up(left, o), rr = __shfl_down(right, o);
			left = lane >= o ? min(left, l) : left, right = lane + o < 64 ? min(right, rr) : right;
		}
		left = __shfl_up(left, 1), right = __shfl_down(right, 1);
		if (lane == 0)
			left = 2 * kFar;
		if (lane == 63)
			right = 2 * kFar;
		uint32_t wa[C / 4], wb[C / 4];
#pragma unroll
		for (int j = 0; j < C / 4; ++j)
		{
			wa[j] = 0, wb[j] = 0;
#pragma unroll
			for (int b = 0; b < 4; ++b)
			{
				const int i = 4 * j + b, x = x0 + i;
				const int L = min(left, pre[i]) + x, R = min(right, suf[i]) - x;        // both <= g[i] <= 255 for cells of the row
				const int a = MODE == 0 ? min(L, R) : (MODE == -1 ? L : R);
				wa[j] |= (uint32_t) (a & 255) << (8 * b);
				wb[j] |= (uint32_t) (L & 255) << (8 * b);
			}
		}
		if (VEC)
		{
#pragma unroll
			for (int j = 0; j < C / 4; ++j)
				if (x0 + 4 * j < mw)
				{
					*reinterpret_cast<uint32_t *>(dst + ro + x0 + 4 * j) = wa[j];
					if (MODE == 2)
						*reinterpret_cast<uint32_t *>(dst2 + ro + x0 + 4 * j) = wb[j];
				}
		}
		else
		{
#pragma unroll
			for (int i = 0; i < C; ++i)
				if (x0 + i < mw)
				{
					dst[ro + x0 + i] = (uint8_t) (wa[i / 4] >> (8 * (i & 3)));
					if (MODE == 2)
						dst2[ro + x0 + i] = (uint8_t) (wb[i / 4] >> (8 * (i & 3)));
				}
		}
	}
}

// ---------------------------------------------------------------------------------------------
// Packed sampling layout (see vkv_device.hpp): one 128-thread half-block per brick, thread = one of the 5^3 texels
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack_volume(const uint8_t *__restrict__ vol, const uint8_t *__restrict__ grad, uint8_t *__restrict__ packed,
                                                     int W, int H, int D, PackedDims pd, uint32_t n_bricks)
{
	// a workgroup packs 2 bricks that are neighbours in x; consecutive workgroups walk a 4 x 4 x 4 group of brick pairs
	// (8 x 4 x 4 bricks = 32 x 16 x 16 voxels) before moving on, so the overlapping 5^3 neighbourhoods are re-read from L1 / L2
	// and not from HBM (the first version walked whole brick rows and fetched every byte ~5x)
	(void) n_bricks;
	const uint32_t gx = (uint32_t) (pd.bx + 7) / 8, gy = (uint32_t) (pd.by + 3) / 4;        // groups per axis
	// workgroup ids are dealt round-robin to the 8 XCDs: XCD (id & 7) takes groups id&7, id&7 + 8, ... so a group stays in one L2;
	// grid.y = group layer in z (keeps grid.x * 256 below 2^32 for 2048^3)
	const uint32_t grp = ((blockIdx.x >> 3) >> 6) * 8u + (blockIdx.x & 7u), in = (blockIdx.x >> 3) & 63u;
	const int      bx = (int) ((grp % gx) * 8 + (in & 3u) * 2 + (threadIdx.x >> 7));
	const int      by = (int) ((grp / gx) * 4 + ((in >> 2) & 3u));
	const int      bz = (int) (blockIdx.y * 4 + (in >> 4));
	const int      t  = threadIdx.x & 127;
	if (bx >= pd.bx || by >= pd.by || bz >= pd.bz || grp >= gx * gy)
		return;
	uint16_t *dst = reinterpret_cast<uint16_t *>(packed + packed_brick_offset(bx, by, bz, pd.mx, pd.my));
	if (t >= 125)
	{
		dst[t] = 0;        // the 6 padding bytes
		return;
	}
	const int lx = t % 5, ly = (t / 5) % 5, lz = t / 25;
	const int x = min(max(bx * 4 + lx - 1, 0), W - 1), y = min(max(by * 4 + ly - 1, 0), H - 1), z = min(max(bz * 4 + lz - 1, 0), D - 1);
	const size_t   o = vidx(x, y, z, W, H);
	const uint32_t v = vol[o], g = grad ? grad[o] : 0u;
	dst[t] = (uint16_t) (v | (g << 8));
}

// ---------------------------------------------------------------------------------------------
// Axis pass of the distance transform (x, y or z), every cell an independent lane of work.
// With m_r(p) = min of g over the candidates within r cells of p (a window that only grows with r),
//     out(p) = min over q of max(|q - p|, g(q)) = the smallest r with m_r(p) <= r
// (below that r every max(r, m_r) equals m_r > r; at it the value is r itself).  This is what the reference's zig-zag search
// (distance_map.comp:72-107, up to 2 x 255 reads per cell) and the anisotropic one-sided search (distance_map_anisotropic.comp:
// 55-91) compute.  The predicate is monotone in r and true at r = g(p), so 8 bisection steps settle a cell, each one a
// range-minimum query answered with two byte reads from a sparse table (level k = minima of 2^k consecutive cells) that the
// workgroup builds in LDS four cells per operation.  Results of neighbouring cells differ by at most one (two-sided) or follow
// from the neighbour with one window test (one-sided), so a thread bisects only the first cell of its run and walks the rest.
// History: a data-dependent search out of an LDS strip (1.4 ms for the three C3 passes), then an O(N) monotone-deque sweep per
// column (0.66 ms: 51 K serial sweeps whose LDS round trips sit on one dependency chain), now 0.13 ms.
// A workgroup takes XT lines and a run of the axis: the whole line when it fits SEG cells, else `ch` outputs plus 255 cells of
// halo on both sides (a candidate further away can never win: the result never exceeds g(p) <= 255).
// MODE 0: two-sided, +1 / -1: candidates at higher / lower index, 2: both one-sided results from one table (dst = +1,
// dst2 = -1; the anisotropic schedule always needs the pair).
// ---------------------------------------------------------------------------------------------
// byte-wise unsigned minimum of two packed dwords (no carries between the bytes)
__device__ __forceinline__ uint32_t min_u8x4(uint32_t a, uint32_t b)
{
	const uint32_t d  = (a | 0x80808080u) - (b & 0x7f7f7f7fu);                        // bit 7 of a byte: low 7 bits of a >= those of b
	const uint32_t ge = ((a & ~b) | (~(a ^ b) & d)) & 0x80808080u;                    // bit 7: a >= b
	const uint32_t m  = (ge - (ge >> 7)) | ge;                                        // 0xff where a >= b
	return (b & m) | (a & ~m);
}

// up to four independent passes of one launch (blockIdx.y picks one): the anisotropic schedule runs both y passes, then all four z passes,
// as one grid each
struct DmPasses
{
	const uint8_t *src[4];
	uint8_t *      dst[4], *dst2[4];
};

template <int MODE, int XT, int SEG, int THREADS = 256>
__global__ void __launch_bounds__(THREADS) k_dm_rmq(const DmPasses passes, uint32_t n_lines, int len, size_t axis_stride, size_t other_stride, uint32_t chunks_x,
                                                uint32_t chunks_p, int ch, int vec)
{
	// A line is one run of the axis; the workgroup owns lines cx * XT .. + XT (consecutive x) of group `other`; LDS index p * XT + line
	// (a dword = 4 lines of one cell row).  The levels of the sparse table, then one (MODE 2: two) level-sized result area for a coalesced
	// write-out.  src may be dst (in place) when a workgroup stages whole lines (chunks_p == 1): it reads only the cells it writes.
	// Levels 0 .. kTop: a query never spans more than the staged cells, and a window of up to 2^(kTop + 1) cells is covered by two blocks of
	// level kTop (its first and its last 2^kTop cells overlap or touch) - so 256 staged cells need levels 0..7, not 0..8: one level less to
	// build, and 20 KB instead of 22 for the anisotropic passes (8 workgroups per CU instead of 7)
	constexpr int kTop   = SEG <= 128 ? 6 : (SEG <= 256 ? 7 : 8);
	constexpr int kLevel = SEG * XT, kOut = (kTop + 1) * kLevel;
	__shared__ __align__(16) uint8_t s_t[(kTop + 1 + (MODE == 2 ? 2 : 1)) * kLevel];
	const uint8_t *src = passes.src[blockIdx.y];
	uint8_t *      dst = passes.dst[blockIdx.y], *dst2 = passes.dst2[blockIdx.y];
	// neighbouring line groups read and write parts of the same 128-byte lines: give each XCD (own L2) a contiguous range of them
	const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
	const uint32_t cx = bid % chunks_x, cp = (bid / chunks_x) % chunks_p, other = bid / (chunks_x * chunks_p);
	const int      out0 = (int) cp * ch, out1 = min(len, out0 + ch);
	const int      seg0 = max(0, out0 - 255), n = min(len, out1 + 255) - seg0;        // staged cells [seg0, seg0 + n), n <= SEG
	const int      t      = (int) threadIdx.x;
	const size_t   base   = (size_t) other * other_stride + (size_t) cx * XT + (size_t) seg0 * axis_stride;
	auto           at     = [](int p, int line) { return p * XT + line; };
	// staging / write-out: iteration q of a thread is cell (line, p): 256 / XT cell rows of XT lines at a time
	const int n_iter = (n + THREADS / XT - 1) / (THREADS / XT);
	auto      cell   = [&](int q, int &line, int &p) { line = t % XT, p = t / XT + q * (THREADS / XT); };
	using vec_t = typename std::conditional<XT == 16, uint4, typename std::conditional<XT == 8, uint2, uint32_t>::type>::type;        // the XT lines of one cell row
	// `vec`: every group of XT lines (one XT-byte row segment) is an aligned vector in memory and in the LDS layout - one load / store
	// instruction moves what XT byte-wide ones would
	if (vec)
	{
		for (int p = t; p < n; p += THREADS)
			*reinterpret_cast<vec_t *>(s_t + p * XT) = *reinterpret_cast<const vec_t *>(src + base + (size_t) p * axis_stride);
	}
	else
	{
		constexpr int kBatch = 8;        // loads in flight per thread
		for (int q0 = 0; q0 < n_iter; q0 += kBatch)
		{
			uint8_t v[kBatch];
#pragma unroll
			for (int j = 0; j < kBatch; ++j)
			{
				int line, p;
				cell(q0 + j, line, p);
				const bool ok = q0 + j < n_iter && p < n && cx * XT + (uint32_t) line < n_lines;
				v[j]          = ok ? src[base + (size_t) line + (size_t) p * axis_stride] : (uint8_t) 255;
			}
#pragma unroll
			for (int j = 0; j < kBatch; ++j)
			{
				int line, p;
				cell(q0 + j, line, p);
				if (q0 + j < n_iter && p < n)
					s_t[at(p, line)] = v[j];
			}
		}
	}
	__syncthreads();
	// ---- sparse table, four cells per operation ------------------------------------------------------------------
	for (int k = 1; k <= kTop && (1 << (k - 1)) < n; ++k)
	{
		const int       h = 1 << (k - 1);
		const uint32_t *a = reinterpret_cast<const uint32_t *>(s_t + (k - 1) * kLevel);
		uint32_t *      b = reinterpret_cast<uint32_t *>(s_t + k * kLevel);
		// dword = 4 lines of cell row p; the partner is the same dword h rows on
		constexpr int kDw = XT / 4;
		for (int e = t; e < n * kDw; e += THREADS)
		{
			const int      p = e / kDw;
			const uint32_t o = (p + h < n) ? a[e + h * kDw] : 0xffffffffu;
			b[e]             = min_u8x4(a[e], o);
		}
		__syncthreads();
	}
	// ---- results: each thread owns a run of consecutive cells of one line; the first by bisection, the rest by walking
	// (the result of a neighbouring cell differs by at most one, so one or two window tests settle each further cell)
	{
		constexpr int kRuns = THREADS / XT;
		const int     line  = t % XT, run = t / XT;
		const int     nout  = out1 - out0, rl = (nout + kRuns - 1) / kRuns;
		const int     pa = out0 - seg0 + run * rl, pb = min(pa + rl, out1 - seg0);        // [pa, pb)
		auto rmq = [&](int l, int r) -> uint32_t {
			const int      k  = min(31 - __builtin_clz((uint32_t) (r - l + 1)), kTop);
			const uint8_t *tk = s_t + k * kLevel;
			return min((uint32_t) tk[at(l, line)], (uint32_t) tk[at(r - (1 << k) + 1, line)]);
		};
		auto bisect = [&](int p, int dir) -> uint32_t {
			uint32_t lo = 0, hi = s_t[at(p, line)];
#pragma unroll
			for (int it = 0; it < 8; ++it)
			{
				const uint32_t mid = (lo + hi) >> 1;
				const int      l = dir == 1 ? p : max(p - (int) mid, 0);
				const int      r = dir == -1 ? p : min(p + (int) mid, n - 1);
				const bool     ok = rmq(l, r) <= mid;
				hi = ok ? mid : hi;
				lo = ok ? lo : mid + 1;
			}
			return hi;
		};
		if (pa < pb)
		{
			if (MODE == 0)
			{        // (measured and dropped: the run as two half runs with a bisection each, two chains side by side - C3 iso 0.099-0.109 ms against
				 // 0.087-0.092: the second bisection's 16 reads cost more than the shorter chain returns)
				uint32_t c               = bisect(pa, 0);
				s_t[kOut + at(pa, line)] = (uint8_t) c;
				for (int p = pa + 1; p < pb; ++p)
				{        // out(p) is c - 1, c or c + 1.  Radius c - 1 failed at p - 1, and the window of radius c - 1 at p is that window
					 // minus its first cell plus cell p + c - 1: it can only succeed through the new cell - one byte, no query.
					const int      e    = p + (int) c - 1;
					const bool     down = c >= 1u && e < n && (uint32_t) s_t[at(min(e, n - 1), line)] <= c - 1u;
					const uint32_t w0   = rmq(max(p - (int) c, 0), min(p + (int) c, n - 1));
					c                   = down ? c - 1u : (w0 <= c ? c : c + 1u);
					s_t[kOut + at(p, line)] = (uint8_t) c;
				}
			}
			if (MODE == 2)
			{        // both one-sided results of the run as ONE loop: the walk down (candidates at higher index, from pb - 1) and the walk up
				 // (candidates at lower index, from pa) are independent chains of dependent LDS round trips - side by side each hides the
				 // other's latency (as two loops the LDS stores between them keep the compiler from overlapping them)
				constexpr int o2 = kOut + kLevel;
				uint32_t lo1 = 0, hi1 = s_t[at(pb - 1, line)], lo2 = 0, hi2 = s_t[at(pa, line)];
#pragma unroll
				for (int it = 0; it < 8; ++it)
				{
					const uint32_t m1 = (lo1 + hi1) >> 1, m2 = (lo2 + hi2) >> 1;
					const uint32_t w1 = rmq(pb - 1, min(pb - 1 + (int) m1, n - 1)), w2 = rmq(max(pa - (int) m2, 0), pa);
					const bool     ok1 = w1 <= m1, ok2 = w2 <= m2;
					hi1 = ok1 ? m1 : hi1, lo1 = ok1 ? lo1 : m1 + 1;
					hi2 = ok2 ? m2 : hi2, lo2 = ok2 ? lo2 : m2 + 1;
				}
				uint32_t c1 = hi1, c2 = hi2;
				s_t[kOut + at(pb - 1, line)] = (uint8_t) c1;
				s_t[o2 + at(pa, line)]       = (uint8_t) c2;
				for (int j = 1; j < pb - pa; ++j)
				{
					const int      pd = pb - 1 - j, pu = pa + j;
					const uint32_t wd = c1 >= 1u ? rmq(pd + 1, min(pd + (int) c1, n - 1)) : 255u;
					const uint32_t wu = c2 >= 1u ? rmq(max(pu - (int) c2, 0), pu - 1) : 255u;
					const uint32_t gd = s_t[at(pd, line)], gu = s_t[at(pu, line)];
					const uint32_t Td = (c1 >= 1u && wd <= c1) ? c1 : c1 + 1u, Tu = (c2 >= 1u && wu <= c2) ? c2 : c2 + 1u;
					c1 = min(gd, Td), c2 = min(gu, Tu);
					s_t[kOut + at(pd, line)] = (uint8_t) c1;
					s_t[o2 + at(pu, line)]   = (uint8_t) c2;
				}
			}
			if (MODE == 1)
			{        // candidates at higher index: walk down; the candidates above p give c or c + 1, the cell itself g(p)
				uint32_t c                   = bisect(pb - 1, 1);
				s_t[kOut + at(pb - 1, line)] = (uint8_t) c;
				for (int p = pb - 2; p >= pa; --p)
				{
					const uint32_t w = c >= 1u ? rmq(p + 1, min(p + (int) c, n - 1)) : 255u;
					const uint32_t T = (c >= 1u && w <= c) ? c : c + 1u;
					c                = min((uint32_t) s_t[at(p, line)], T);
					s_t[kOut + at(p, line)] = (uint8_t) c;
				}
			}
			if (MODE == -1)
			{
				constexpr int o2             = kOut;
				uint32_t      c              = bisect(pa, -1);
				s_t[o2 + at(pa, line)]       = (uint8_t) c;
				for (int p = pa + 1; p < pb; ++p)
				{
					const uint32_t w = c >= 1u ? rmq(max(p - (int) c, 0), p - 1) : 255u;
					const uint32_t T = (c >= 1u && w <= c) ? c : c + 1u;
					c                = min((uint32_t) s_t[at(p, line)], T);
					s_t[o2 + at(p, line)] = (uint8_t) c;
				}
			}
		}
	}
	__syncthreads();
	if (vec)
	{
		for (int p = out0 - seg0 + t; p < out1 - seg0; p += THREADS)
		{
			const size_t o = base + (size_t) p * axis_stride;
			*reinterpret_cast<vec_t *>(dst + o) = *reinterpret_cast<const vec_t *>(s_t + kOut + p * XT);
			if (MODE == 2)
				*reinterpret_cast<vec_t *>(dst2 + o) = *reinterpret_cast<const vec_t *>(s_t + kOut + kLevel + p * XT);
		}
		return;
	}
	for (int q = 0; q < n_iter; ++q)
	{
		int line, p;
		cell(q, line, p);
		if (p < out0 - seg0 || p >= out1 - seg0 || cx * XT + (uint32_t) line >= n_lines)
			continue;
		const size_t o = base + (size_t) line + (size_t) p * axis_stride;
		dst[o]         = s_t[kOut + at(p, line)];
		if (MODE == 2)
			dst2[o] = s_t[kOut + kLevel + at(p, line)];
	}
}

// LDS-staged version for dword-aligned rows (W % 4 == 0): a workgroup packs 8 x BY x BZ bricks from a 33 x (4 BY + 1) x (4 BZ + 1) texel
// tile that it stages with coalesced dword loads (volume and gradient once each), then writes the bricks as whole 256-byte lines.
// 8 x 4 x 4 (the launcher's choice for volumes of at least 16 bricks in y and z): every lane has 2 x 11 dwords in flight before the
// barrier (8 x 2 x 2: 2 x 3 - not enough outstanding bytes per CU to cover the HBM latency) and the apron re-read drops from 1.42 to 1.27.
template <int BY, int BZ, int PITCH, bool ALIGNED>        // ALIGNED: W % 4 == 0 and dword-aligned buffers (plain dword loads); else row_dword
__global__ void __launch_bounds__(256) k_pack_volume_tiled(const uint8_t *__restrict__ vol, const uint8_t *__restrict__ grad, uint8_t *__restrict__ packed,
                                                           int W, int H, int D, PackedDims pd, uint32_t groups_x)
{
	// tile: rows (jz, jy) of (v | g << 8) texels; staged dword column c (voxels 4 * (bx0 - 1 + c) ..) sits at texels 4c .. 4c + 3, so the
	// padded tile column jx (voxel x = 4 * bx0 - 1 + jx) is texel jx + 3; 36 texels staged, 33 used
	constexpr int kTX = PITCH, kRY = 4 * BY + 1, kRZ = 4 * BZ + 1, kRows = kRY * kRZ;
	__shared__ __align__(8) uint16_t s_tile[kRows * kTX];
	// x-neighbouring workgroups stage parts of the same 128-byte lines: consecutive groups go to one XCD (own L2)
	const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
	const int      bx0 = (int) (bid % groups_x) * 8, by0 = (int) (bid / groups_x) * BY, bz0 = (int) blockIdx.y * BZ;
	const int wd  = (W + 3) >> 2;        // dword columns of a row; the last one partial when W % 4 != 0 (its texels x >= W are fixed up below)
	// ---- stage: row = (jz, jy) of the padded tile, 9 dwords per row starting one dword left of the tile; a lane's (row, column) advance by
	// constants from one of its loads to the next (256 = 28 * 9 + 4), so the divisions are done once
	{
		constexpr int kIter = (kRows * 9 + 255) / 256;        // all loads of a lane are in flight before its first LDS store
		uint32_t      v4[kIter], g4[kIter];
		const int     row0 = (int) threadIdx.x / 9, c0 = (int) threadIdx.x - row0 * 9;
		int           row = row0, c = c0;
#pragma unroll
		for (int j = 0; j < kIter; ++j)
		{
			const int r  = min(row, kRows - 1);
			const int ry = r % kRY, rz = r / kRY;
			const int y = min(max(by0 * 4 + ry - 1, 0), H - 1), z = min(max(bz0 * 4 + rz - 1, 0), D - 1);
			const int dc = min(max(bx0 - 1 + c, 0), wd - 1);
			const size_t o = ((size_t) z * H + y) * (size_t) W;
			v4[j]          = ALIGNED ? reinterpret_cast<const uint32_t *>(vol + o)[dc] : row_dword(vol + o, dc, W);
			g4[j]          = grad ? (ALIGNED ? reinterpret_cast<const uint32_t *>(grad + o)[dc] : row_dword(grad + o, dc, W)) : 0u;
			row += 28, c += 4;
			if (c >= 9)
				c -= 9, ++row;
		}
		row = row0, c = c0;
#pragma unroll
		for (int j = 0; j < kIter; ++j)
		{
			if (row < kRows)
			{
				// bytes (v0 v1 v2 v3), (g0 g1 g2 g3) -> texel pairs (v0 g0 v1 g1), (v2 g2 v3 g3): one byte permute each, one 8-byte LDS store
				const uint32_t lo = __builtin_amdgcn_perm(g4[j], v4[j], 0x05010400u), hi = __builtin_amdgcn_perm(g4[j], v4[j], 0x07030602u);
				if (kTX % 4 == 0)
					*reinterpret_cast<uint2 *>(&s_tile[row * kTX + 4 * c]) = make_uint2(lo, hi);
				else
				{
					*reinterpret_cast<uint32_t *>(&s_tile[row * kTX + 4 * c])     = lo;
					*reinterpret_cast<uint32_t *>(&s_tile[row * kTX + 4 * c + 2]) = hi;
				}
			}
			row += 28, c += 4;
			if (c >= 9)
				c -= 9, ++row;
		}
	}
	__syncthreads();
	// ---- clamp-to-edge in x (dword columns were clamped as a whole): x = -1 -> voxel 0, x >= W -> voxel W - 1
	if (bx0 == 0 || bx0 * 4 + 31 >= W)
	{
		for (int it = threadIdx.x; it < kRows * 33; it += 256)
		{
			const int row = it / 33, jx = it - row * 33;
			const int x = bx0 * 4 + jx - 1;
			if (x < 0)
				s_tile[row * kTX + jx + 3] = s_tile[row * kTX + 3 + 1 - bx0 * 4];
			else if (x >= W)
				s_tile[row * kTX + jx + 3] = s_tile[row * kTX + 3 + W - bx0 * 4];
		}
		__syncthreads();
	}
	// ---- write: 8 BY BZ bricks x 16 pieces of 16 bytes (a store instruction costs the same per lane whatever its width).  A lane keeps its
	// piece q of every brick it writes (bricks b, b + 16, ...), so the positions of its eight texels inside a brick's 5^3 block are computed once
	const int q = (int) threadIdx.x & 15;
	int       off[8];
#pragma unroll
	for (int k = 0; k < 8; ++k)
	{
		const int t = 8 * q + k;        // texel of the 5^3 brick, x fastest; 125..127 are padding
		const int lx = t % 5, ly = (t / 5) % 5, lz = t / 25;
		off[k]       = t < 125 ? (lz * kRY + ly) * kTX + lx + 3 : -1;
	}
#pragma unroll
	for (int j = 0; j < (8 * BY * BZ) / 16; ++j)
	{
		const int b  = ((int) threadIdx.x >> 4) + 16 * j;
		const int lbx = b & 7, lby = (b >> 3) % BY, lbz = (b >> 3) / BY;
		const int bx = bx0 + lbx, by = by0 + lby, bz = bz0 + lbz;
		if (bx >= pd.bx || by >= pd.by || bz >= pd.bz)
			continue;
		const int base = ((lbz * 4) * kRY + lby * 4) * kTX + lbx * 4;
		uint32_t  w[4] = {0, 0, 0, 0};
#pragma unroll
		for (int k = 0; k < 8; ++k)
			if (off[k] >= 0)
				w[k >> 1] |= (uint32_t) s_tile[base + off[k]] << (16 * (k & 1));
		// non-temporal: 3.6 GB that nobody reads before the kernel is over (1.31 -> 1.23 ms; non-temporal LOADS of the volume cost the apron's
		// L2 hits: 1.6 ms)
		typedef uint32_t v4u __attribute__((ext_vector_type(4)));
		v4u              val = {w[0], w[1], w[2], w[3]};
		__builtin_nontemporal_store(val, reinterpret_cast<v4u *>(packed + packed_brick_offset(bx, by, bz, pd.mx, pd.my)) + q);
	}
}

// ---------------------------------------------------------------------------------------------
// Synthetic volumes (DESIGN.md "Synthetic inputs")
// ---------------------------------------------------------------------------------------------
struct SynthShell
{
	float cx, cy, cz, irx, iry, irz, slope, amp, lo2, hi2;
};
constexpr int kSynthShells = 40;
struct SynthArgs
{
	SynthShell sh[kSynthShells];
};

__device__ __forceinline__ uint32_t synth_hash(uint32_t seed, uint32_t x, uint32_t y, uint32_t z)
{
	uint32_t h = seed ^ (x * 0x8da6b343u) ^ (y * 0xd8163841u) ^ (z * 0xcb1ab31fu);
	h ^= h >> 16;
	h *= 0x7feb352du;
	h ^= h >> 15;
	h *= 0x846ca68bu;
	h ^= h >> 16;
	return h;
}

__global__ void __launch_bounds__(256) k_synth_sphere(uint8_t *__restrict__ vol, int W, int H, int D, uint32_t blocks_x)
{
	const uint32_t bx = blockIdx.x % blocks_x;
	const int      x  = (int) (bx * 64 + (threadIdx.x & 63));
	const int      y  = (int) ((blockIdx.x / blocks_x) * 4 + (threadIdx.x >> 6)), z = (int) blockIdx.y;        // grid.y = z
	if (x >= W || y >= H)
		return;
	const float dm = (float) max(max(W, H), D);
	const float R0 = 0.375f * dm, R1 = 0.25f * dm;
	const float cx = ((float) W - 1.0f) * 0.5f, cy = ((float) H - 1.0f) * 0.5f, cz = ((float) D - 1.0f) * 0.5f;
	const float dx = (float) x - cx, dy = (float) y - cy, dz = (float) z - cz;
	const float r  = __builtin_sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
	const float t  = g_clamp((R0 - r) / (R0 - R1), 0.0f, 1.0f);
	vol[vidx(x, y, z, W, H)] = (uint8_t) __builtin_rintf(255.0f * t);
}

__global__ void __launch_bounds__(256) k_synth_shells(uint8_t *__restrict__ vol, int W, int H, int D, uint32_t seed, uint32_t blocks_x, int n_shells, uint32_t noise_mod,
                                                      SynthArgs args)
{
	__shared__ SynthShell s_sh[kSynthShells];
	for (int i = threadIdx.x; i < kSynthShells * 10; i += 256)
		reinterpret_cast<float *>(s_sh)[i] = reinterpret_cast<const float *>(args.sh)[i];
	__syncthreads();
	const uint32_t bx = blockIdx.x % blocks_x;
	const int      x  = (int) (bx * 64 + (threadIdx.x & 63));
	const int      y  = (int) ((blockIdx.x / blocks_x) * 4 + (threadIdx.x >> 6)), z = (int) blockIdx.y;        // grid.y = z
	if (x >= W || y >= H)
		return;
	float best = 0.0f;
	for (int k = 0; k < n_shells; ++k)
	{
		const float dx = ((float) x - s_sh[k].cx) * s_sh[k].irx;
		const float dy = ((float) y - s_sh[k].cy) * s_sh[k].iry;
		const float dz = ((float) z - s_sh[k].cz) * s_sh[k].irz;
		const float q2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
		if (q2 > s_sh[k].lo2 && q2 < s_sh[k].hi2)
		{
			const float q   = __builtin_sqrtf(q2);
			const float val = s_sh[k].amp * (1.0f - __builtin_fabsf(q - 1.0f) * s_sh[k].slope);
			if (val > best)
				best = val;
		}
	}
	const uint32_t noise = synth_hash(seed, (uint32_t) x, (uint32_t) y, (uint32_t) z) % noise_mod;
	const uint32_t v     = (uint32_t) best + noise;
	vol[vidx(x, y, z, W, H)] = (uint8_t) min(v, 255u);
}

// ---------------------------------------------------------------------------------------------
// Loader conversion on the device (src/load_volume.cpp:151-169): 16 input bytes per thread
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ uint8_t normalise_voxel(T raw, bool swap, float lo, float hi)
{
	if (sizeof(T) == 2 && swap)
		raw = (T) (uint16_t) ((((uint16_t) raw) >> 8) | (((uint16_t) raw) << 8));
	const float x = ((float) raw - lo) / (hi - lo);
	const float a = (x < 1.0f) ? x : 1.0f;          // std::min(1.0f, x)
	const float t = (0.0f < a) ? a : 0.0f;          // std::max(0.0f, .)
	return (uint8_t) (255 * t);                      // truncation (load_volume.cpp:169)
}

template <typename T>
__global__ void __launch_bounds__(256) k_convert_volume(const T *__restrict__ raw, uint8_t *__restrict__ out, unsigned long long n, bool swap, float lo, float hi)
{
	constexpr int kPer = 16 / (int) sizeof(T);        // elements per 16-byte load
	const unsigned long long first = ((unsigned long long) blockIdx.x * 256 + threadIdx.x) * kPer;
	if (first >= n)
		return;
	if (first + kPer <= n && (((uintptr_t) raw) & 15u) == 0 && (((uintptr_t) out) & (kPer - 1)) == 0)
	{
		const uint4 q = *reinterpret_cast<const uint4 *>(raw + first);
		T           e[kPer];
		__builtin_memcpy(e, &q, 16);
		uint8_t o[kPer];
#pragma unroll
		for (int i = 0; i < kPer; ++i)
			o[i] = normalise_voxel<T>(e[i], swap, lo, hi);
		if (kPer == 16)
			*reinterpret_cast<uint4 *>(out + first) = *reinterpret_cast<const uint4 *>(o);
		else
			*reinterpret_cast<uint2 *>(out + first) = *reinterpret_cast<const uint2 *>(o);
	}
	else
		for (unsigned long long i = first; i < n && i < first + kPer; ++i)
			out[i] = normalise_voxel<T>(raw[i], swap, lo, hi);
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU: de-interleave gathered compact tile buffers into the final image(s) (one thread per pixel, or per four RGBA8 pixels).
// blockIdx.z = frame of the launch; each frame has its own image, source ([rank][tiles], `stride` tiles between two ranks' buffers) and
// tile rectangle (tiles numbered row-major inside it, tile t on rank t % n_ranks as its (t / n_ranks)-th); pixels outside the rectangle
// are cleared, so the image is complete after the kernel.
// ---------------------------------------------------------------------------------------------
struct ScatterFrame
{
	void *      image;
	const void *src;
	uint32_t    rx0, ry0, rw, rh;        // the tile rectangle
	uint32_t    stride;                  // tiles between the buffers of two ranks
	uint32_t    pad;
};
struct ScatterFrames
{
	ScatterFrame f[VKV_MAX_BATCH];
};

template <typename T>
__global__ void __launch_bounds__(256) k_scatter_tiles_frames(const ScatterFrames frames, uint32_t iw, uint32_t ih, uint32_t tw, uint32_t th, uint32_t n_ranks)
{
	const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
	const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (x >= iw || y >= ih)
		return;
	const ScatterFrame &F  = frames.f[blockIdx.z];
	const uint32_t      tx = x / tw - F.rx0, ty = y / th - F.ry0;        // (wraps to a huge value left of / above the rectangle)
	T                   v  = {};
	if (tx < F.rw && ty < F.rh)
	{
		const uint32_t t    = ty * F.rw + tx;
		const uint32_t rank = t % n_ranks, k = t / n_ranks;
		v = static_cast<const T *>(F.src)[(((size_t) rank * F.stride + k) * th + (y % th)) * tw + (x % tw)];
	}
	static_cast<T *>(F.image)[(size_t) y * iw + x] = v;
}

// ---------------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------------
namespace vkv
{

int launch_gradient_map(vkv_ctx *ctx, const uint8_t *d_vol, uint8_t *d_grad, VkvExtent3D e, const VkvTransferFunctionUniform *tf, hipStream_t s)
{
	const uint32_t blocks_x = (e.width + 63) / 64;
	if (e.depth > 65535u || (uint64_t) blocks_x * ((e.height + 3) / 4) > 0xffffffull)
		return set_error(ctx, VKV_E_UNSUPPORTED, "gradient_map: volume too large for one launch");
	// the tiled kernel folds the three factors 0.25 into the modifier: exact unless 0.25 * modifier is denormal (or the modifier no number)
	const float m_abs    = std::fabs(tf->grad_magnitude_modifier);
	const bool  m_normal = m_abs == 0.0f || (m_abs >= 1e-30f && m_abs <= 1e30f);
	// round 6: any width >= 4 and any alignment of the volume (the staging loads need none: row_dword / load_u32_any)
	if (tf->use_gradient && m_normal && e.width >= 4 && (uint64_t) e.width * e.height * (kGradTileZ + 2) <= 0xffffffffull)
	{
		const uint64_t tx = (e.width + kGradTileX - 1) / kGradTileX, ty = (e.height + kGradTileY - 1) / kGradTileY,
		               tz = (e.depth + kGradTileZ - 1) / kGradTileZ;
		// kGradSegment tiles per workgroup, fewer when the volume would not give every CU its eight workgroups otherwise
		const uint64_t want = (uint64_t) 8 * (uint64_t) std::max(1, ctx->cu_count);
		uint32_t       seg  = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>(kGradSegment, tx * ty * tz / want));
		if (const uint32_t forced = tuning_of(ctx).gradient_segment)        // VkvTuning: lets a test march a small volume
			seg = std::min(forced, 255u);
		const uint64_t n_wgs = tx * ty * ((tz + seg - 1) / seg);
		if (n_wgs <= 0x7fffffffull)
		{
			if ((e.width & 3u) == 0 && (((uintptr_t) d_vol) & 3u) == 0)
				hipLaunchKernelGGL(k_gradient_map_tiled<true>, dim3((uint32_t) n_wgs), dim3(256), 0, s, d_vol, d_grad, (int) e.width, (int) e.height, (int) e.depth,
				                   tf->grad_magnitude_modifier, (uint32_t) tx, (uint32_t) ty, (uint32_t) tz, seg, (uint32_t) n_wgs);
			else
				hipLaunchKernelGGL(k_gradient_map_tiled<false>, dim3((uint32_t) n_wgs), dim3(256), 0, s, d_vol, d_grad, (int) e.width, (int) e.height, (int) e.depth,
				                   tf->grad_magnitude_modifier, (uint32_t) tx, (uint32_t) ty, (uint32_t) tz, seg, (uint32_t) n_wgs);
			return check_launch(ctx, "gradient_map");
		}
	}
	hipLaunchKernelGGL(k_gradient_map, dim3(blocks_x * ((e.height + 3) / 4), e.depth), dim3(256), 0, s, d_vol, d_grad, (int) e.width, (int) e.height,
	                   (int) e.depth, (int) (tf->use_gradient != 0), tf->grad_magnitude_modifier, blocks_x, 0u);
	return check_launch(ctx, "gradient_map");
}

int launch_occupancy_map(vkv_ctx *ctx, const uint8_t *d_vol, const uint8_t *d_grad, const uint8_t *d_tf, const VkvTransferFunctionUniform *tf,
                         VkvExtent3D e, uint8_t *d_map, VkvExtent3D me, hipStream_t s)
{
	uint8_t *scratch = stream_scratch(ctx, s);        // per stream: map updates on different streams do not share the bit table
	if (!scratch)
		return VKV_E_UNSUPPORTED;
	uint32_t *d_bits = reinterpret_cast<uint32_t *>(scratch + kTfBitsOffset);
	hipLaunchKernelGGL(k_tf_bits, dim3(8), dim3(256), 0, s, d_tf, d_bits);
	hipLaunchKernelGGL(k_tf_columns, dim3(1), dim3(256), 0, s, d_bits);        // words 2048..2055 of the scratch's table area
	// src/compute_distance_map.cpp:110-113
	const int bx = (int) ((e.width + me.width - 1) / me.width), by = (int) ((e.height + me.height - 1) / me.height),
	          bz = (int) ((e.depth + me.depth - 1) / me.depth);
	const int      cells_per_block = 256 / bx > 0 ? 256 / bx : 1;
	const uint32_t blocks_x        = (me.width + cells_per_block - 1) / cells_per_block;
	if ((uint64_t) blocks_x * me.height > 0xffffffull || me.depth > 65535u)
		return set_error(ctx, VKV_E_UNSUPPORTED, "occupancy_map: map too large for one launch");
	const dim3 grid(blocks_x * me.height, me.depth);
	const int W = (int) e.width, H = (int) e.height, D = (int) e.depth, mw = (int) me.width, mh = (int) me.height, md = (int) me.depth;
	const bool precomputed = tf->use_gradient && d_grad;
	const VkvTuning T_occ = tuning_of(ctx);
	// a map extent above ceil(extent / block) leaves cells without a voxel: EMPTY.  The wave kernel fills the whole map first anyway;
	// the workgroup-per-cell-row kernels only visit cells the volume reaches in x, so such a map is filled for them too
	if ((uint64_t) (mw - 1) * bx >= (uint64_t) W || (uint64_t) (mh - 1) * by >= (uint64_t) H || (uint64_t) (md - 1) * bz >= (uint64_t) D)
	{
		const hipError_t em = hipMemsetAsync(d_map, 255, (size_t) me.width * me.height * me.depth, s);
		if (em != hipSuccess)
			return set_error(ctx, (int) em, "occupancy_map: fill: %s", hipGetErrorString(em));
	}
	if ((bx == 1 || bx == 2 || bx == 4) && T_occ.occupancy_kernel == 1 && (e.width & 3u) == 0 && (!tf->use_gradient || precomputed) && (((uintptr_t) d_vol | (uintptr_t) d_grad) & 3u) == 0)
	{
		const uint32_t dblocks = (e.width / 4 + 255) / 256;
		const dim3     dgrid(dblocks * me.height, me.depth);
#define VKV_OCC_DWORD(G, B)                                                                                                                            \
	hipLaunchKernelGGL((k_occupancy_map_dword<G, B>), dgrid, dim3(256), 0, s, d_vol, d_grad, d_bits, d_map, W, H, D, mw, mh, by, bz, dblocks)
		if (precomputed)
		{
			if (bx == 1) VKV_OCC_DWORD(1, 1); else if (bx == 2) VKV_OCC_DWORD(1, 2); else VKV_OCC_DWORD(1, 4);
		}
		else
		{
			if (bx == 1) VKV_OCC_DWORD(0, 1); else if (bx == 2) VKV_OCC_DWORD(0, 2); else VKV_OCC_DWORD(0, 4);
		}
#undef VKV_OCC_DWORD
		return check_launch(ctx, "occupancy_map");
	}
	// round 6: any width >= 4 and any alignment (the kernel's loads need none; the last lane of an odd row reads the row's last four bytes)
	const bool dword_ok = e.width >= 4 && (!tf->use_gradient || precomputed);
	if (dword_ok && T_occ.occupancy_kernel != 1)
	{        // any block width: a wave per 64 dwords (k_occupancy_map_waves)
		const uint32_t spans_x = (uint32_t) (((e.width + 3) / 4 + 63) / 64);
		const int      n_rows  = by * bz;
		int            rows = 8, crb = 1;
		if (n_rows == 1 || n_rows == 2 || n_rows == 4)
			crb = 8 / n_rows;        // eight rows per batch out of 8, 4 or 2 cell rows
		else
		{
			int waste = 1 << 30;
			for (int cand : {8, 9, 10, 12})
			{
				const int w = (n_rows + cand - 1) / cand * cand - n_rows;
				if (w < waste)
					waste = w, rows = cand;
			}
		}
		// cell rows per wave: >= 64 KB of voxels behind a wave's share of the bit table's staging, but at least 32 waves per CU in the launch
		const size_t row_bytes = (size_t) 256 * n_rows * (precomputed ? 2 : 1);
		int          cyw       = (int) std::min<size_t>(std::max<size_t>(1, (64 * 1024 + row_bytes - 1) / row_bytes), me.height);
		const uint64_t want = (uint64_t) 32 * (uint64_t) std::max(1, ctx->cu_count);
		while (cyw > crb && (uint64_t) spans_x * ((me.height + cyw - 1) / cyw) * me.depth < want)
			cyw >>= 1;
		cyw = (cyw + crb - 1) / crb * crb;
		const uint64_t tasks = (uint64_t) spans_x * ((me.height + cyw - 1) / cyw);
		if ((tasks + 3) / 4 <= 0x7fffffffull && me.depth <= 65535u)
		{
			const hipError_t em = hipMemsetAsync(d_map, 255, (size_t) me.width * me.height * me.depth, s);        // EMPTY everywhere; the kernel stores the OCCUPIED cells
			if (em != hipSuccess)
				return set_error(ctx, (int) em, "occupancy_map: fill: %s", hipGetErrorString(em));
			const dim3 wgrid((uint32_t) ((tasks + 3) / 4), me.depth);
#define VKV_OCC_WAVES(G, R, C) hipLaunchKernelGGL((k_occupancy_map_waves<G, R, C>), wgrid, dim3(256), 0, s, d_vol, d_grad, d_bits, d_map, W, H, D, mw, mh, bx, by, bz, spans_x, cyw)
#define VKV_OCC_WAVES_G(G)                                                                                                                             \
	do                                                                                                                                                 \
	{                                                                                                                                                  \
		if (crb == 8) VKV_OCC_WAVES(G, 8, 8); else if (crb == 4) VKV_OCC_WAVES(G, 8, 4); else if (crb == 2) VKV_OCC_WAVES(G, 8, 2);                     \
		else if (rows == 8) VKV_OCC_WAVES(G, 8, 1); else if (rows == 9) VKV_OCC_WAVES(G, 9, 1); else if (rows == 10) VKV_OCC_WAVES(G, 10, 1);           \
		else VKV_OCC_WAVES(G, 12, 1);                                                                                                                  \
	} while (0)
			if (precomputed)
				VKV_OCC_WAVES_G(1);
			else
				VKV_OCC_WAVES_G(0);
#undef VKV_OCC_WAVES_G
#undef VKV_OCC_WAVES
			return check_launch(ctx, "occupancy_map");
		}
	}
	if (bx <= 256 && (e.width & 3u) == 0 && (!tf->use_gradient || precomputed) && (((uintptr_t) d_vol | (uintptr_t) d_grad) & 3u) == 0)
	{
		const int      cpb     = ((1024 / bx) & ~3) > 0 ? ((1024 / bx) & ~3) : 4;        // cells per workgroup (<= 1024, 4 | cpb)
		const uint32_t ablocks = (me.width + cpb - 1) / cpb;
		const dim3     agrid(ablocks * me.height, me.depth);
		if (precomputed)
			hipLaunchKernelGGL(k_occupancy_map_dword_any<1>, agrid, dim3(256), 0, s, d_vol, d_grad, d_bits, d_map, W, H, D, mw, mh, bx, by, bz, cpb, ablocks);
		else
			hipLaunchKernelGGL(k_occupancy_map_dword_any<0>, agrid, dim3(256), 0, s, d_vol, d_grad, d_bits, d_map, W, H, D, mw, mh, bx, by, bz, cpb, ablocks);
		return check_launch(ctx, "occupancy_map");
	}
	if (!tf->use_gradient)
		hipLaunchKernelGGL(k_occupancy_map<0>, grid, dim3(256), 0, s, d_vol, d_grad, d_bits, d_map, W, H, D, mw, mh, md, bx, by,
		                   bz, tf->grad_magnitude_modifier, blocks_x);
	else if (d_grad)
		hipLaunchKernelGGL(k_occupancy_map<1>, grid, dim3(256), 0, s, d_vol, d_grad, d_bits, d_map, W, H, D, mw, mh, md, bx, by,
		                   bz, tf->grad_magnitude_modifier, blocks_x);
	else
		hipLaunchKernelGGL(k_occupancy_map<2>, grid, dim3(256), 0, s, d_vol, d_grad, d_bits, d_map, W, H, D, mw, mh, md, bx, by,
		                   bz, tf->grad_magnitude_modifier, blocks_x);
	return check_launch(ctx, "occupancy_map");
}

int launch_occupied_voxel_count(vkv_ctx *ctx, const uint8_t *d_vol, const uint8_t *d_grad, const VkvTransferFunctionUniform *tf, VkvExtent3D e,
                                uint64_t *d_count, hipStream_t s)
{
	uint8_t *scratch = stream_scratch(ctx, s);
	if (!scratch)
		return VKV_E_UNSUPPORTED;
	uint32_t *d_bits = reinterpret_cast<uint32_t *>(scratch + kTfBitsOffset);
	hipLaunchKernelGGL(k_tf_bits_analytic, dim3(8), dim3(256), 0, s, d_bits, tf->intensity_min, tf->intensity_range_inv, tf->gradient_min,
	                   tf->gradient_range_inv);
	const hipError_t me = hipMemsetAsync(d_count, 0, sizeof(uint64_t), s);
	if (me != hipSuccess)
		return set_error(ctx, (int) me, "occupied_voxel_count: %s", hipGetErrorString(me));
	// four voxels per lane for every width >= 4 and every alignment (row_dword); the byte path only for the on-the-fly gradient and tiny rows
	const bool     dwords       = e.width >= 4 && (!tf->use_gradient || d_grad);
	const bool     aligned      = (e.width & 3u) == 0 && ((((uintptr_t) d_vol) | ((uintptr_t) d_grad)) & 3u) == 0;
	const uint32_t blocks_x     = dwords ? ((e.width + 3) / 4 + 63) / 64 : (e.width + 63) / 64;
	const uint64_t rows         = (uint64_t) e.height * e.depth;
	const uint64_t n_row_groups = (rows + 3) / 4;
	if (n_row_groups > 0xffffffffull)
		return set_error(ctx, VKV_E_UNSUPPORTED, "occupied_voxel_count: volume too large");
	// ~8 workgroups per CU, grid-stride over the row groups: one atomic per workgroup, few thousand in total
	const uint32_t groups = (uint32_t) (n_row_groups < 2048 / blocks_x + 1 ? n_row_groups : 2048 / blocks_x + 1);
	const dim3     grid(blocks_x * groups);
	unsigned long long *total = reinterpret_cast<unsigned long long *>(d_count);
	const int W = (int) e.width, H = (int) e.height, D = (int) e.depth;
#define VKV_COUNT(G, DW)                                                                                                                                  \
	hipLaunchKernelGGL((k_occupied_voxel_count<G, DW>), grid, dim3(256), 0, s, d_vol, d_grad, d_bits, total, W, H, D, tf->grad_magnitude_modifier,        \
	                   tf->intensity_min, tf->intensity_range_inv, tf->gradient_min, tf->gradient_range_inv, blocks_x, (uint32_t) n_row_groups)
	if (!tf->use_gradient)
	{
		if (dwords && aligned) VKV_COUNT(0, 1); else if (dwords) VKV_COUNT(0, 2); else VKV_COUNT(0, 0);
	}
	else if (d_grad)
	{
		if (dwords && aligned) VKV_COUNT(1, 1); else if (dwords) VKV_COUNT(1, 2); else VKV_COUNT(1, 0);
	}
	else
		VKV_COUNT(2, 0);
#undef VKV_COUNT
	return check_launch(ctx, "occupied_voxel_count");
}

static int row_stride_for(int mw)
{
	int s4 = (mw + 3) / 4 + 1;
	if ((s4 & 1) == 0)
		++s4;        // odd dword stride: the 64 lanes of a scan hit distinct banks
	return s4 * 4;
}

template <int MODE>
static int launch_dm_rmq(vkv_ctx *ctx, int axis, const uint8_t *src, uint8_t *dst, uint8_t *dst2, VkvExtent3D me, hipStream_t s);
template <int MODE>
static int launch_dm_rmq_passes(vkv_ctx *ctx, int axis, const DmPasses &passes, uint32_t n_passes, VkvExtent3D me, hipStream_t s);
static bool dm_whole_lines(int axis, VkvExtent3D me) { return (axis == 1 ? me.height : me.depth) <= 512u; }        // a workgroup stages the whole line: in place is safe

template <int MODE>
static int launch_dm_x(vkv_ctx *ctx, const uint8_t *src, uint8_t *dst, VkvExtent3D me, hipStream_t s)
{
	if (me.width <= 2048)        // round 6: also rows of 1025 .. 2048 cells in registers (32 cells per lane; the serial scan below was 14 x slower per cell)
		return launch_dm_rmq<MODE>(ctx, 0, src, dst, nullptr, me, s);
	// longer rows: serial row scan out of LDS (in place, like the table kernel for short rows)
	const uint32_t n_rows = me.height * me.depth;
	const int      stride = row_stride_for((int) me.width);
	const size_t   lds    = (size_t) stride * 64;
	if (lds > 64 * 1024)
		(void) hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dm_x<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	hipLaunchKernelGGL(k_dm_x<MODE>, dim3((n_rows + 63) / 64), dim3(64), lds, s, src, dst, (int) me.width, n_rows, stride);
	return check_launch(ctx, "distance_map x pass");
}

// axis: 0 = x, 1 = y, 2 = z.  MODE 2 writes the +1 result to dst and the -1 result to dst2.  The x pass may run in place
// (a wave holds its whole row before it writes); rows longer than 1024 cells go to k_dm_x.
template <int MODE>
static int launch_dm_rmq(vkv_ctx *ctx, int axis, const uint8_t *src, uint8_t *dst, uint8_t *dst2, VkvExtent3D me, hipStream_t s)
{
	if (axis == 0)
	{        // rows of up to 1024 cells: one wave per row, in registers (k_dm_x_wave); longer rows never get here (launch_dm_x)
		const int      len    = (int) me.width;
		const uint32_t n_rows = me.height * me.depth;
		const bool     vec    = (me.width & 3u) == 0 && ((((uintptr_t) src) | ((uintptr_t) dst) | ((uintptr_t) dst2)) & 3u) == 0;
#define VKV_DM_XW(C)                                                                                                                            \
	do                                                                                                                                          \
	{                                                                                                                                           \
		const uint32_t rows_per_wg = 4u * ((C) <= 8 ? 4u : ((C) <= 16 ? 2u : 1u)); /* kRows of the kernel */                                    \
		const dim3     grid((n_rows + rows_per_wg - 1u) / rows_per_wg);                                                                         \
		if (vec)                                                                                                                                \
			hipLaunchKernelGGL((k_dm_x_wave<MODE, C, true>), grid, dim3(256), 0, s, src, dst, dst2, len, n_rows);                                \
		else                                                                                                                                    \
			hipLaunchKernelGGL((k_dm_x_wave<MODE, C, false>), grid, dim3(256), 0, s, src, dst, dst2, len, n_rows);                               \
	} while (0)
		if (len <= 256)
			VKV_DM_XW(4);
		else if (len <= 512)
			VKV_DM_XW(8);
		else if (len <= 1024)
			VKV_DM_XW(16);
		else
			VKV_DM_XW(32);        // up to 2048 cells (the widest map dm_check_extent lets through)
#undef VKV_DM_XW
		return check_launch(ctx, "distance_map x pass");
	}
	DmPasses one = {};
	one.src[0] = src, one.dst[0] = dst, one.dst2[0] = dst2;
	return launch_dm_rmq_passes<MODE>(ctx, axis, one, 1, me, s);
}

// y (axis 1) or z (axis 2) pass over n_passes independent (src, dst, dst2) triples in one grid
template <int MODE>
static int launch_dm_rmq_passes(vkv_ctx *ctx, int axis, const DmPasses &passes, uint32_t n_passes, VkvExtent3D me, hipStream_t s)
{
	const size_t   sy = me.width, sz = (size_t) me.width * me.height;
	const int      len     = axis == 1 ? (int) me.height : (int) me.depth;
	const uint32_t other   = axis == 1 ? me.depth : me.height;
	const uint32_t n_lines = me.width;
	const size_t   as = axis == 1 ? sy : sz, os = axis == 1 ? sz : sy;
	uintptr_t      al = 0;
	for (uint32_t i = 0; i < n_passes; ++i)
		al |= (uintptr_t) passes.src[i] | (uintptr_t) passes.dst[i] | (uintptr_t) passes.dst2[i];
#define VKV_DM_RMQ(XT, SEG)                                                                                                                            \
	do                                                                                                                                                  \
	{                                                                                                                                                   \
		const int      ch       = len <= (SEG) ? len : (SEG) -510;                                                                                     \
		const uint32_t chunks_p = (uint32_t) ((len + ch - 1) / ch), chunks_x = (n_lines + (XT) -1) / (XT);                                              \
		if ((uint64_t) chunks_x * chunks_p * other > 0x7fffffffull)                                                                                     \
			return set_error(ctx, VKV_E_UNSUPPORTED, "distance_map: map too large for one launch");                                                     \
		const int vec = (me.width & ((XT) -1)) == 0 && (al & ((XT) -1)) == 0;                                                                           \
		hipLaunchKernelGGL((k_dm_rmq<MODE, XT, SEG>), dim3(chunks_x * chunks_p * other, n_passes), dim3(256), 0, s, passes, n_lines, len, as, os, chunks_x, \
		                   chunks_p, ch, vec);                                                                                                          \
	} while (0)
	// (measured and dropped: resident workgroups marching over several tiles with the next tile's cells prefetched into registers - 38 -> 44 us
	// per pass on C3; the CU already overlaps one workgroup's loads with the others' table building.  Round 4: WAVE-OWNED dword columns - a
	// wave builds the nine levels of 4 lines and answers their queries alone, two workgroup barriers instead of eleven - with workgroups of
	// 2 waves x 8 lines or 4 waves x 16 lines: bit-identical, C3 iso 0.102-0.117 ms against 0.087-0.092, aniso 0.41 against 0.30; the same
	// LDS then holds half as many waves per CU (14 against 32), and the walk's dependent LDS round trips need the waves more than the
	// barriers cost: profiles/r4_dm_variants.txt)
	if (len <= 128 && me.width > 16)
		VKV_DM_RMQ(16, 128);
	else if (len <= 256)        // (4 lines per workgroup here: C3 iso 0.112-0.115 ms against 0.087 - the wave slots, not the LDS, cap the CU at this size;
		                        // 16 lines x 512 threads, i.e. 16-byte row segments at the same waves per CU: the same 0.086-0.096 / 0.26 ms) 8 lines per workgroup: 20 KB of LDS instead of 40 (8 workgroups per CU, not 4) and runs of 8 cells per thread:
		VKV_DM_RMQ(8, 256);        // C3 42.6 -> 36.8 us per isotropic pass, 63 -> 47 us per anisotropic pass
	else if (len <= 512)        // whole line, no halo; 4 lines per workgroup: 22 KB of LDS instead of 45 (7 workgroups per CU, not 3): C4 iso 0.70 -> 0.64 ms,
		VKV_DM_RMQ(4, 512);        // aniso 3.11 -> 2.60 ms
	else
		VKV_DM_RMQ(8, 768);
#undef VKV_DM_RMQ
	return check_launch(ctx, "distance_map axis pass");
}

// axis: 1 = y, 2 = z
template <int MODE>
static int launch_dm_axis(vkv_ctx *ctx, int axis, const uint8_t *src, uint8_t *dst, VkvExtent3D me, hipStream_t s)
{
	return launch_dm_rmq<MODE>(ctx, axis, src, dst, nullptr, me, s);
}

static int dm_check_extent(vkv_ctx *ctx, VkvExtent3D me)
{
	if (me.width == 0 || me.height == 0 || me.depth == 0)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "distance_map: zero map extent");
	if (me.width > 2048)
		return set_error(ctx, VKV_E_UNSUPPORTED, "distance_map: map rows longer than 2048 cells (LDS strip limit of the x pass)");
	return VKV_OK;
}

// src/compute_distance_map.cpp:142-175
int launch_distance_map(vkv_ctx *ctx, uint8_t *d_map, uint8_t *d_swap, VkvExtent3D me, hipStream_t s)
{
	int rc = dm_check_extent(ctx, me);
	if (rc) return rc;
	if ((rc = launch_dm_x<0>(ctx, d_map, d_map, me, s))) return rc;
	if ((rc = launch_dm_axis<0>(ctx, 1, d_map, d_swap, me, s))) return rc;
	return launch_dm_axis<0>(ctx, 2, d_swap, d_map, me, s);
}

// src/compute_distance_map.cpp:201-252 — same buffers, same order (stream order replaces the image barriers)
int launch_distance_map_anisotropic(vkv_ctx *ctx, uint8_t *const m[8], uint8_t *swap, VkvExtent3D me, hipStream_t s)
{
	int rc = dm_check_extent(ctx, me);
	if (rc) return rc;
	const uint8_t *occ = m[7];
	if (me.width <= 1024)
	{
		// Same results with 7 launches instead of 14: every pass answers its +1 and -1 queries from one table.  Buffers: the x
		// pass leaves x+ in m[3] and x- in m[7] (in place on the occupancy); y of x+ puts y+ into swap and y- into m[1], whose z
		// passes fill (m[2], m[3]) and then (m[0], m[1]); the x- half repeats this with m[4..7].
		if ((rc = launch_dm_rmq<2>(ctx, 0, occ, m[3], m[7], me, s))) return rc;
		if (dm_whole_lines(1, me) && dm_whole_lines(2, me))
		{        // Three launches: x; both y passes as one grid; all four z passes as one grid (every launch boundary costs a drain and a ramp-up
			 // of a grid that is only three rounds of workgroups deep).  No swap buffer: y of x+ (m[3]) goes to m[0] / m[2], y of x- (m[7]) to
			 // m[4] / m[6] (all four still free), and every z pass writes its + result IN PLACE over its source (a workgroup stages the whole
			 // lines it owns before it writes them) and its - result into the odd neighbour: m[1], m[3], m[5], m[7] (whose x results have
			 // been consumed by the y launch).
			DmPasses y = {}, z = {};
			for (int h = 0; h < 2; ++h)
			{
				uint8_t *const *q = m + 4 * h;
				y.src[h] = q[3], y.dst[h] = q[0], y.dst2[h] = q[2];
				for (int k = 0; k < 2; ++k)
					z.src[2 * h + k] = q[2 * k], z.dst[2 * h + k] = q[2 * k], z.dst2[2 * h + k] = q[2 * k + 1];
			}
			if ((rc = launch_dm_rmq_passes<2>(ctx, 1, y, 2, me, s))) return rc;
			return launch_dm_rmq_passes<2>(ctx, 2, z, 4, me, s);
		}
		for (int h = 0; h < 2; ++h)
		{
			uint8_t *const *q = m + 4 * h;
			if ((rc = launch_dm_rmq<2>(ctx, 1, q[3], swap, q[1], me, s))) return rc;
			if ((rc = launch_dm_rmq<2>(ctx, 2, q[1], q[2], q[3], me, s))) return rc;
			if ((rc = launch_dm_rmq<2>(ctx, 2, swap, q[0], q[1], me, s))) return rc;
		}
		return VKV_OK;
	}
	if ((rc = launch_dm_x<1>(ctx, occ, m[3], me, s))) return rc;                  // stage1(3, +1)
	if ((rc = launch_dm_axis<1>(ctx, 1, m[3], swap, me, s))) return rc;           // stage2(3, +1)
	if ((rc = launch_dm_axis<1>(ctx, 2, swap, m[0], me, s))) return rc;           // stage3(0, +1)
	if ((rc = launch_dm_axis<-1>(ctx, 2, swap, m[1], me, s))) return rc;          // stage3(1, -1)
	if ((rc = launch_dm_axis<-1>(ctx, 1, m[3], swap, me, s))) return rc;          // stage2(3, -1)
	if ((rc = launch_dm_axis<1>(ctx, 2, swap, m[2], me, s))) return rc;           // stage3(2, +1)
	if ((rc = launch_dm_axis<-1>(ctx, 2, swap, m[3], me, s))) return rc;          // stage3(3, -1)
	if ((rc = launch_dm_x<-1>(ctx, occ, m[7], me, s))) return rc;                 // stage1(7, -1) in place
	if ((rc = launch_dm_axis<1>(ctx, 1, m[7], swap, me, s))) return rc;           // stage2(7, +1)
	if ((rc = launch_dm_axis<1>(ctx, 2, swap, m[4], me, s))) return rc;           // stage3(4, +1)
	if ((rc = launch_dm_axis<-1>(ctx, 2, swap, m[5], me, s))) return rc;          // stage3(5, -1)
	if ((rc = launch_dm_axis<-1>(ctx, 1, m[7], swap, me, s))) return rc;          // stage2(7, -1)
	if ((rc = launch_dm_axis<1>(ctx, 2, swap, m[6], me, s))) return rc;           // stage3(6, +1)
	return launch_dm_axis<-1>(ctx, 2, swap, m[7], me, s);                         // stage3(7, -1)
}

int launch_check_numerics(vkv_ctx *ctx, int what, uint32_t first_bits, uint64_t count, unsigned long long *d_mismatches, hipStream_t s)
{
	if (count == 0)
		return VKV_OK;
	if ((count + 255) / 256 > 0x7fffffffull || what < 0 || what > 4)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "check_numerics: bad arguments");
	hipLaunchKernelGGL(k_check_numerics, dim3((uint32_t) ((count + 255) / 256)), dim3(256), 0, s, what, first_bits, count, d_mismatches);
	return check_launch(ctx, "check_numerics");
}

int launch_pack_volume(vkv_ctx *ctx, const uint8_t *d_vol, const uint8_t *d_grad, VkvExtent3D e, void *d_packed, hipStream_t s)
{
	const PackedDims pd = packed_dims((int) e.width, (int) e.height, (int) e.depth);
	if (e.width >= 4)        // round 6: any width and alignment (row_dword: the staging loads need no alignment, an odd row's last column is shifted in)
	{
		const int tile_env = tuning_of(ctx).pack_tile;        // A/B switch: 2 or 4
		const bool     big = tile_env ? tile_env == 4 : (pd.by >= 16 && pd.bz >= 16);
		const uint32_t t   = big ? 4u : 2u;
		const uint32_t gx = (uint32_t) (pd.bx + 7) / 8, gy = ((uint32_t) pd.by + t - 1) / t, gz = ((uint32_t) pd.bz + t - 1) / t;
		if ((uint64_t) gx * gy <= 0xffffffull && gz <= 65535u && (uint64_t) pd.mx * pd.my * pd.mz * 512 <= 0xffffffffull)
		{
			// row pitch 40 texels; 38 (19 banks, odd: no bank conflicts, 70 % of the LDS cycles otherwise) measured the same 1.34 ms: the
			// kernel follows its 5.3 GB of traffic, not the LDS or the VALU (50 % busy)
			const bool aligned = (e.width & 3u) == 0 && ((((uintptr_t) d_vol) | ((uintptr_t) d_grad)) & 3u) == 0;
#define VKV_PACK(T, A)                                                                                                                              \
	hipLaunchKernelGGL((k_pack_volume_tiled<T, T, 40, A>), dim3(gx * gy, gz), dim3(256), 0, s, d_vol, d_grad, (uint8_t *) d_packed, (int) e.width,     \
	                   (int) e.height, (int) e.depth, pd, gx)
			if (big)
			{
				if (aligned) VKV_PACK(4, true); else VKV_PACK(4, false);
			}
			else
			{
				if (aligned) VKV_PACK(2, true); else VKV_PACK(2, false);
			}
#undef VKV_PACK
			return check_launch(ctx, "pack_volume");
		}
	}
	const uint64_t   nb = (uint64_t) pd.bx * pd.by;        // bricks per z layer
	if (nb > 0xffffffull || pd.bz > 65535 || (uint64_t) pd.mx * pd.my * pd.mz * 512 > 0xffffffffull)
		return set_error(ctx, VKV_E_UNSUPPORTED, "pack_volume: volume too large");
	// macro-brick padding (bricks that exist only because of the 8x8x8 grouping) is never addressed by the sampler
	const uint64_t groups = (uint64_t) ((pd.bx + 7) / 8) * ((pd.by + 3) / 4);        // per layer of 4 bricks in z
	if ((groups + 8) * 64 * 256 > 0xffffffffull || (pd.bz + 3) / 4 > 65535)
		return set_error(ctx, VKV_E_UNSUPPORTED, "pack_volume: volume too large");
	hipLaunchKernelGGL(k_pack_volume, dim3((uint32_t) (((groups + 7) / 8) * 8 * 64), (uint32_t) ((pd.bz + 3) / 4)), dim3(256), 0, s, d_vol, d_grad, (uint8_t *) d_packed, (int) e.width, (int) e.height,
	                   (int) e.depth, pd, (uint32_t) nb);
	return check_launch(ctx, "pack_volume");
}

// Separable alpha tables of the reference's transfer function (src/volume_component.cpp:246-261) from the uniform's fields, and
// the flag word.  has_tf == 0: no claim, flag clear.
__global__ void __launch_bounds__(256) k_tf_tables_init(uint32_t *__restrict__ tables, int has_tf, float imin, float iinv, float gmin, float ginv, int use_gradient)
{
	const int   i = threadIdx.x;
	const float x = (float) i / 255.0f;
	float       ai = (x - imin) * iinv, ag = 1.0f;
	ai = (ai < 0.0f) ? 0.0f : ai, ai = (1.0f < ai) ? 1.0f : ai;        // std::max / std::min as the host writes them
	if (use_gradient)
	{
		ag = (x - gmin) * ginv;
		ag = (ag < 0.0f) ? 0.0f : ag, ag = (1.0f < ag) ? 1.0f : ag;
	}
	tables[kTfAiWord + i] = __float_as_uint(ai);
	tables[kTfAgWord + i] = __float_as_uint(ag);
	if (i < 4)
		tables[kTfFlagWord + i] = (i == 0 && has_tf) ? kTfFlagSeparable : 0u;
}

// alpha > 0 bit table + the check of the separable claim against every texel (clears the flag on the first mismatch)
__global__ void __launch_bounds__(256) k_tf_tables(const uint8_t *__restrict__ tf_rgba8, uint32_t *__restrict__ tables, int has_tf)
{
	const uint32_t w = blockIdx.x * 256 + threadIdx.x;        // 2048 words of 32 texels
	if (w >= 2048)
		return;
	uint32_t v = 0;
	bool     ok = true;
	for (int i = 0; i < 32; ++i)
	{
		const uint32_t t     = w * 32 + i;
		const uint32_t texel = reinterpret_cast<const uint32_t *>(tf_rgba8)[t];
		v |= ((texel >> 24) > 0 ? 1u : 0u) << i;
		if (has_tf)
		{
			const uint32_t b = tf_separable_alpha(__uint_as_float(tables[kTfAiWord + (t & 255u)]), __uint_as_float(tables[kTfAgWord + (t >> 8)]));
			ok               = ok && texel == b * 0x01010101u;
		}
	}
	tables[w] = v;
	if (has_tf && !ok)
		atomicAnd(&tables[kTfFlagWord], ~kTfFlagSeparable);
}

int launch_tf_tables(vkv_ctx *ctx, const uint8_t *d_tf, const VkvTransferFunctionUniform *tf, uint32_t *d_tables, hipStream_t s)
{
	const int has_tf = tf != nullptr;
	hipLaunchKernelGGL(k_tf_tables_init, dim3(1), dim3(256), 0, s, d_tables, has_tf, has_tf ? tf->intensity_min : 0.0f, has_tf ? tf->intensity_range_inv : 0.0f,
	                   has_tf ? tf->gradient_min : 0.0f, has_tf ? tf->gradient_range_inv : 0.0f, has_tf ? (int) (tf->use_gradient != 0) : 0);
	hipLaunchKernelGGL(k_tf_tables, dim3(8), dim3(256), 0, s, d_tf, d_tables, has_tf);
	return check_launch(ctx, "transfer_function_tables");
}

// --- synthetic volume: host builds the shell table (same definition as DESIGN.md "Synthetic inputs") ---
static uint64_t splitmix64(uint64_t *s)
{
	uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
	z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static float u01(uint64_t *s) { return (float) (splitmix64(s) >> 40) * (1.0f / 16777216.0f); }

int launch_synth_volume(vkv_ctx *ctx, uint8_t *d_vol, VkvExtent3D e, uint32_t kind, uint32_t seed, hipStream_t s)
{
	const uint32_t blocks_x = (e.width + 63) / 64;
	if (e.depth > 65535u || (uint64_t) blocks_x * ((e.height + 3) / 4) > 0xffffffull)
		return set_error(ctx, VKV_E_UNSUPPORTED, "synth_volume: volume too large for one launch");
	const dim3 grid(blocks_x * ((e.height + 3) / 4), e.depth);
	const int  W = (int) e.width, H = (int) e.height, D = (int) e.depth;
	if ((kind & 255u) == 0)
	{
		hipLaunchKernelGGL(k_synth_sphere, grid, dim3(256), 0, s, d_vol, W, H, D, blocks_x);
		return check_launch(ctx, "synth_volume");
	}
	if ((kind & 255u) != 1)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "synth_volume: unknown kind %u", kind & 255u);
	// kind = 1 | shells << 8 | thickness << 16 | noise << 28: the first `shells` (0 = all 40) of the seed's shells, their thickness scaled by
	// thickness / 256 (12 bits, 0 = 1), hash noise 0 .. noise (4 bits, 0 = the default 0 .. 20)
	const uint32_t nq = (kind >> 8) & 255u, tq = (kind >> 16) & 0xfffu, noise_mod = (kind >> 28) ? (kind >> 28) + 1u : 21u;
	const int      n_shells = nq && nq < (uint32_t) kSynthShells ? (int) nq : kSynthShells;
	SynthArgs   args;
	uint64_t    st = (0x5EEDull << 32) | (uint64_t) seed;
	const float Wf = (float) e.width, Hf = (float) e.height, Df = (float) e.depth;
	const float dm = fmaxf(fmaxf(Wf, Hf), Df);
	const float th = (0.001f * dm + 1.0f) * (tq ? (float) tq * (1.0f / 256.0f) : 1.0f);
	for (int k = 0; k < kSynthShells; ++k)
	{
		SynthShell &sh = args.sh[k];
		sh.cx          = (0.15f + 0.70f * u01(&st)) * Wf;
		sh.cy          = (0.15f + 0.70f * u01(&st)) * Hf;
		sh.cz          = (0.15f + 0.70f * u01(&st)) * Df;
		const float r  = (0.05f + 0.13f * u01(&st)) * dm;
		const float rx = r * (0.7f + 0.6f * u01(&st));
		const float ry = r * (0.7f + 0.6f * u01(&st));
		const float rz = r * (0.7f + 0.6f * u01(&st));
		sh.irx = 1.0f / rx, sh.iry = 1.0f / ry, sh.irz = 1.0f / rz;
		sh.slope       = fminf(fminf(rx, ry), rz) / th;
		sh.amp         = 110.0f + 145.0f * u01(&st);
		const float w  = 1.0f / sh.slope + 0.001f;
		const float lo = 1.0f - w, hi = 1.0f + w;
		sh.lo2 = lo > 0.0f ? lo * lo : 0.0f;
		sh.hi2 = hi * hi;
	}
	hipLaunchKernelGGL(k_synth_shells, grid, dim3(256), 0, s, d_vol, W, H, D, seed, blocks_x, n_shells, noise_mod, args);
	return check_launch(ctx, "synth_volume");
}

int launch_convert_volume(vkv_ctx *ctx, const void *d_raw, int type, bool big_endian, float lo, float hi, uint64_t n, uint8_t *d_out, hipStream_t s)
{
	const uint16_t probe          = 1;
	const bool     host_is_little = *reinterpret_cast<const uint8_t *>(&probe) == 1;        // the device shares the host's byte order
	const bool     swap           = big_endian == host_is_little;
	const int      per            = (type == VKV_VOXEL_UINT16 || type == VKV_VOXEL_INT16) ? 8 : 16;
	const uint64_t blocks         = (n + (uint64_t) per * 256 - 1) / ((uint64_t) per * 256);
	if (blocks == 0)
		return VKV_OK;
	if (blocks > 0xffffffull)
		return set_error(ctx, VKV_E_UNSUPPORTED, "convert_volume: volume too large for one launch");
	const unsigned long long nn = n;
	switch (type)
	{
		case VKV_VOXEL_UINT8: hipLaunchKernelGGL(k_convert_volume<uint8_t>, dim3((uint32_t) blocks), dim3(256), 0, s, (const uint8_t *) d_raw, d_out, nn, swap, lo, hi); break;
		case VKV_VOXEL_INT8: hipLaunchKernelGGL(k_convert_volume<int8_t>, dim3((uint32_t) blocks), dim3(256), 0, s, (const int8_t *) d_raw, d_out, nn, swap, lo, hi); break;
		case VKV_VOXEL_UINT16: hipLaunchKernelGGL(k_convert_volume<uint16_t>, dim3((uint32_t) blocks), dim3(256), 0, s, (const uint16_t *) d_raw, d_out, nn, swap, lo, hi); break;
		case VKV_VOXEL_INT16: hipLaunchKernelGGL(k_convert_volume<int16_t>, dim3((uint32_t) blocks), dim3(256), 0, s, (const int16_t *) d_raw, d_out, nn, swap, lo, hi); break;
		default: return set_error(ctx, VKV_E_INVALID_ARGUMENT, "convert_volume: unsupported image data type %d", type);
	}
	return check_launch(ctx, "convert_volume");
}

// `n` frames (1 .. VKV_MAX_BATCH) described by images[f], srcs[f], rects[f] (a rectangle of whole tiles inside the image), strides[f]
int launch_scatter_tiles_frames(vkv_ctx *ctx, uint32_t n, void *const *images, const void *const *srcs, const VkvTileRect *rects, const uint32_t *strides, uint32_t iw,
                                uint32_t ih, uint32_t tw, uint32_t th, uint32_t n_ranks, uint32_t bpp, hipStream_t s)
{
	ScatterFrames fr{};
	bool          aligned = true;
	for (uint32_t f = 0; f < n; ++f)
	{
		fr.f[f] = ScatterFrame{images[f], srcs[f], rects[f].x0, rects[f].y0, rects[f].w, rects[f].h, strides[f], 0u};
		aligned = aligned && (((uintptr_t) images[f] | (uintptr_t) srcs[f]) & 15u) == 0;
	}
	if (bpp == 4 && (iw % 4) == 0 && (tw % 4) == 0 && aligned)
		// RGBA8: four pixels per thread (a 16-pixel tile row = 64 B = 4 threads)
		hipLaunchKernelGGL(k_scatter_tiles_frames<uint4>, dim3((iw / 4 + 63) / 64, (ih + 3) / 4, n), dim3(256), 0, s, fr, iw / 4, ih, tw / 4, th, n_ranks);
	else if (bpp == 4)
		hipLaunchKernelGGL(k_scatter_tiles_frames<uint32_t>, dim3((iw + 63) / 64, (ih + 3) / 4, n), dim3(256), 0, s, fr, iw, ih, tw, th, n_ranks);
	else if (bpp == 16)
		hipLaunchKernelGGL(k_scatter_tiles_frames<uint4>, dim3((iw + 63) / 64, (ih + 3) / 4, n), dim3(256), 0, s, fr, iw, ih, tw, th, n_ranks);
	else
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "scatter_tiles: bytes_per_pixel must be 4 or 16");
	return check_launch(ctx, "scatter_tiles");
}

}        // namespace vkv
