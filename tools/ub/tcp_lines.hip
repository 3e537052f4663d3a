// Microbenchmark: what does a divergent 16-byte load instruction cost as a function of the number of DISTINCT 128-byte
// lines its 64 lanes touch?  (Decides whether a node should be fetched by its own lane with 3 instructions, or by 3
// neighbouring lanes with one.)   hipcc -O3 --offload-arch=gfx950 tools/ub/tcp_lines.hip -o tools/ub/tcp_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// lanes_per_node = 1: every lane loads 16 B of its own random 48-byte node (64 lines per instruction);
// 3: lanes 3g..3g+2 load the three 16-byte pieces of node g (21 lines per instruction); 4 (64-byte nodes): 16 lines.
template <int LPN>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ tab, uint32_t n_nodes, int iters, uint32_t *out) {
	const uint32_t lane = threadIdx.x & 63u, gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t group = lane / LPN, piece = lane % LPN;
	uint32_t h = mix((gid / 64u) * 64u + group + 12345u); // same chain for the lanes of a group
	uint32_t acc = 0;
	constexpr int stride = LPN == 4 ? 4 : 3; // node size in 16-byte units
	for (int i = 0; i < iters; ++i) {
		const uint32_t node = h % n_nodes;
		const uint4 v = tab[(size_t)node * stride + (LPN == 1 ? (i % 3) : piece)];
		acc += v.x ^ v.w;
		h = mix(h + v.y); // dependent chain, like a traversal
	}
	out[gid] = acc;
}

int main(int argc, char **argv) {
	const uint32_t mb = argc > 1 ? atoi(argv[1]) : 24;
	const int iters = 2000;
	const uint32_t n_nodes = mb * 1048576u / 64u;
	uint4 *tab; uint32_t *out;
	CK(hipMalloc(&tab, (size_t)n_nodes * 64)); CK(hipMemset(tab, 1, (size_t)n_nodes * 64));
	const int grid = 256 * 8, block = 256; // 32 waves per CU
	CK(hipMalloc(&out, grid * block * 4));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto run = [&](auto kern, const char *name, int lpn) {
		kern<<<grid, block>>>(tab, n_nodes, 10, out);
		CK(hipDeviceSynchronize());
		CK(hipEventRecord(e0));
		kern<<<grid, block>>>(tab, n_nodes, iters, out);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		const double wave_loads = (double)grid * block / 64 * iters;
		const double cyc = ms * 1e6 / (wave_loads / 256) * 2.4;
		const int lines = lpn == 1 ? 64 : (63 / lpn + 1);
		printf("%-28s table %3u MB: %.3f ms, %.0f cycles per wave-load per CU at 2.4 GHz = %.1f cycles per distinct line\n", name, mb, ms, cyc, cyc / lines);
	};
	run(k<1>, "1 lane per node (64 lines)", 1);
	run(k<3>, "3 lanes per node (22 lines)", 3);
	run(k<4>, "4 lanes per node (16 lines)", 4);
	return 0;
}
