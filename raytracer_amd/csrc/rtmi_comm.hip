// librtmi — the radiosity solve across GPUs from inside the library (include/rtmi.h: rtmi_comm_*,
// rtmi_radiosity_solve_sharded_dev).  One process per GPU; F is row-sharded (rank r owns the r-th block of ceil(n / W)
// rows); per sweep the ranks all-gather their rows of B over RCCL (xGMI), and at every convergence check all-reduce the
// three norms.  Everything between two checks is ENQUEUED on the caller's stream — sweep kernel, all-gather, sweep
// kernel, ... — the host only waits where it reads the norms (the reference's solver is one thread,
// Render2/radiosity.cpp:483-563).
// RCCL is loaded at run time (dlopen), not linked: a single-GPU host needs no RCCL at all, and a process that already
// holds one (PyTorch) shares it through the soname.
#include "rtmi_rad_device.hpp"

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstring>

#include <rccl/rccl.h>

struct rtmi_comm {
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1, device = -1;
};

namespace rtmi {

struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static std::mutex g_rccl_mutex;
static Rccl g_rccl;

static const Rccl *rccl(const char *who) {
	std::lock_guard<std::mutex> lock(g_rccl_mutex);
	if (g_rccl.lib) return &g_rccl;
	const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	void *h = nullptr;
	for (const char *nm : names)
		if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
	if (!h) { fail(std::string(who) + ": RCCL is not available (dlopen librccl.so.1 failed): " + dlerror()); return nullptr; }
	Rccl r;
	r.lib = h;
	r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
	r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
	r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
	r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
	r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
	r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
	if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce || !r.GetErrorString) {
		fail(std::string(who) + ": librccl lacks an expected symbol");
		return nullptr;
	}
	g_rccl = r;
	return &g_rccl;
}

// the 64 norm slots of a sweep (k_jacobi_sweep spreads its atomics over them) -> {Σd², Σ(B−E)²} and {max|d|}
__global__ void k_fold_norms(const double *__restrict__ slots, uint32_t n_slots, double *__restrict__ sums, double *__restrict__ mx) {
	const uint32_t l = threadIdx.x;
	double a = 0.0, b = 0.0, m = 0.0;
	if (l < n_slots) { a = slots[3 * l]; b = slots[3 * l + 1]; m = slots[3 * l + 2]; }
	for (int off = 32; off > 0; off >>= 1) {
		a += __shfl_xor(a, off, 64);
		b += __shfl_xor(b, off, 64);
		m = fmax(m, __shfl_xor(m, off, 64));
	}
	if (l == 0) { sums[0] = a; sums[1] = b; mx[0] = m; }
}

} // namespace rtmi

using namespace rtmi;

extern "C" {

int rtmi_comm_unique_id(uint8_t id[RTMI_COMM_ID_BYTES]) {
	const char *who = "rtmi_comm_unique_id";
	if (!id) return fail(std::string(who) + ": NULL argument");
	const Rccl *R = rccl(who);
	if (!R) return -1;
	ncclUniqueId u;
	static_assert(sizeof(ncclUniqueId) == RTMI_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
	const ncclResult_t rc = R->GetUniqueId(&u);
	if (rc != ncclSuccess) return fail(std::string(who) + ": ncclGetUniqueId: " + R->GetErrorString(rc));
	std::memcpy(id, &u, sizeof u);
	return 0;
}

rtmi_comm *rtmi_comm_create(const uint8_t id[RTMI_COMM_ID_BYTES], int rank, int world) {
	const char *who = "rtmi_comm_create";
	if (!id || world < 1 || rank < 0 || rank >= world) { fail(std::string(who) + ": bad argument"); return nullptr; }
	const Rccl *R = rccl(who);
	if (!R) return nullptr;
	rtmi_comm *c = new (std::nothrow) rtmi_comm();
	if (!c) { fail(std::string(who) + ": out of memory"); return nullptr; }
	if (hipGetDevice(&c->device) != hipSuccess) { delete c; fail(std::string(who) + ": hipGetDevice failed"); return nullptr; }
	ncclUniqueId u;
	std::memcpy(&u, id, sizeof u);
	const ncclResult_t rc = R->CommInitRank(&c->comm, world, u, rank); // collective: every rank calls it with the same id
	if (rc != ncclSuccess) { delete c; fail(std::string(who) + ": ncclCommInitRank: " + R->GetErrorString(rc)); return nullptr; }
	c->rank = rank;
	c->world = world;
	return c;
}

void rtmi_comm_destroy(rtmi_comm *c) {
	if (!c) return;
	if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
	delete c;
}

int rtmi_radiosity_solve_sharded_dev(rtmi_comm *c, const float *d_F, uint64_t ld, uint32_t n, uint32_t row0, uint32_t n_rows,
                                     const double rho[3], const float *d_E, float *d_B, float tol_rel, uint32_t max_sweeps,
                                     uint32_t check_every, rtmi_solve_stats *stats, void *stream) {
	const char *who = "rtmi_radiosity_solve_sharded_dev";
	if (!c || (!d_F && n_rows) || !rho || !d_E || !d_B) return fail(std::string(who) + ": NULL argument"); // (a rank may own no rows)
	if (n == 0) return fail(std::string(who) + ": n == 0");
	int dev = -1;
	if (hipGetDevice(&dev) != hipSuccess || dev != c->device) return fail(std::string(who) + ": the communicator was created on another device");
	const Rccl *R = rccl(who);
	if (!R) return -1;
	const uint32_t W = (uint32_t)c->world, blk = (n + W - 1) / W, want0 = std::min(n, (uint32_t)c->rank * blk);
	// rank r owns rows [min(n, r·blk), min(n, (r+1)·blk)): the blocks the all-gather puts side by side.  (With ONE rank any
	// row range is accepted: tools measure a shard's sweep + exchange cost on a single GPU that way.)
	if (W > 1 && (row0 != want0 || n_rows != std::min(n, want0 + blk) - want0))
		return fail(std::string(who) + ": rows [row0, row0 + n_rows) are not this rank's block of ceil(n / world) rows");
	if ((uint64_t)row0 + n_rows > n) return fail(std::string(who) + ": row range exceeds n");
	if (check_every == 0) check_every = 1;
	hipStream_t st = (hipStream_t)stream;
	const size_t n_pad = (size_t)blk * W, vb = sizeof(float) * 3 * n_pad;
	float *buf[2] = {nullptr, nullptr};
	double *d_norms = nullptr, *d_red = nullptr; // check_every × 64 slots × 3; {Σd², Σ(B−E)², max|d|}
	const size_t nb = 3 * sizeof(double) * kNormSlots;
	int rc = 0;
	if (hipMalloc(&buf[0], vb) != hipSuccess || hipMalloc(&buf[1], vb) != hipSuccess || hipMalloc(&d_norms, nb * check_every) != hipSuccess ||
	    hipMalloc(&d_red, 3 * sizeof(double)) != hipSuccess)
		rc = fail(std::string(who) + ": hipMalloc failed");
	auto nccl_ok = [&](ncclResult_t r, const char *what) { return r == ncclSuccess ? 0 : fail(std::string(who) + ": " + what + ": " + R->GetErrorString(r)); };
	uint32_t sweeps = 0;
	double rel = INFINITY, mx = INFINITY;
	int at = 0;
	const auto t0 = std::chrono::steady_clock::now();
	if (!rc) {
		// B_0 = E (Render2/radiosity.cpp:506-509); the padding rows behind n stay zero in both vectors
		if (hipMemsetAsync(buf[0], 0, vb, st) != hipSuccess || hipMemsetAsync(buf[1], 0, vb, st) != hipSuccess ||
		    hipMemcpyAsync(buf[0], d_E, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess)
			rc = fail(std::string(who) + ": initialising B failed");
	}
	while (!rc && sweeps < max_sweeps) {
		const uint32_t k = std::min(check_every, max_sweeps - sweeps);
		if (hipMemsetAsync(d_norms, 0, nb * k, st) != hipSuccess) { rc = fail(std::string(who) + ": memset failed"); break; }
		for (uint32_t i = 0; i < k && !rc; ++i) { // enqueue only: sweep, gather, sweep, gather, ...
			float *Bo = buf[(at + i) & 1], *Bn = buf[(at + i + 1) & 1];
			rc = sweep_launch_slots(d_F, ld, n, row0, n_rows, rho, d_E, Bo, Bn, d_norms + 3 * kNormSlots * i, kNormSlots, st);
			// in place: rank r's block already sits at Bn + r·blk·3
			if (!rc) rc = nccl_ok(R->AllGather(Bn + (size_t)c->rank * blk * 3, Bn, (size_t)blk * 3, ncclFloat, c->comm, st), "ncclAllGather");
		}
		if (rc) break;
		at = (at + (int)k) & 1;
		sweeps += k;
		hipLaunchKernelGGL(k_fold_norms, dim3(1), dim3(64), 0, st, d_norms + 3 * kNormSlots * (k - 1), kNormSlots, d_red, d_red + 2);
		if (launch_check(who)) { rc = -1; break; }
		rc = nccl_ok(R->AllReduce(d_red, d_red, 2, ncclDouble, ncclSum, c->comm, st), "ncclAllReduce");
		if (!rc) rc = nccl_ok(R->AllReduce(d_red + 2, d_red + 2, 1, ncclDouble, ncclMax, c->comm, st), "ncclAllReduce");
		double h[3];
		if (!rc && (hipMemcpyAsync(h, d_red, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
			rc = fail(std::string(who) + ": reading the norms failed: " + hipGetErrorString(hipGetLastError()));
		if (rc) break;
		rel = h[1] > 0.0 ? std::sqrt(h[0] / h[1]) : 0.0;
		mx = h[2];
		if (rel <= (double)tol_rel) break; // the same numbers on every rank: every rank stops at the same sweep
	}
	if (!rc && (hipMemcpyAsync(d_B, buf[at], sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
		rc = fail(std::string(who) + ": copying B out failed");
	if (!rc && stats) {
		stats->sweeps = sweeps;
		stats->rel_change = rel;
		stats->max_change = mx;
		stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); // wall: sweeps + exchange + checks
		stats->bytes_per_sweep = 4.0 * (double)n_rows * (double)n;
	}
	for (float *b : buf) if (b) (void)hipFree(b);
	if (d_norms) (void)hipFree(d_norms);
	if (d_red) (void)hipFree(d_red);
	return rc;
}

} // extern "C"
