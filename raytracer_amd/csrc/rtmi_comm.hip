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
	bool dead = false; // an RCCL call failed and the communicator was aborted: every later call fails at once
};

namespace rtmi {

struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommAbort)(ncclComm_t) = nullptr; // optional
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
	r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(h, "ncclCommAbort"));
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

// ---- the solve loop's exchange over RCCL (SolveExchange, rtmi_rad_device.hpp) --------------------------------------------
struct RcclExchange {
	rtmi_comm *c;
	const Rccl *R;
	uint32_t blk; // rows per rank
};
// An RCCL call that returns an error leaves the communicator in an unknown state: abort it, so that nothing of this rank
// stays queued on it and later calls fail at once instead of hanging.
static int nccl_failed(RcclExchange *x, ncclResult_t r, const char *what) {
	const std::string msg = std::string(what) + ": " + x->R->GetErrorString(r);
	if (!x->c->dead) {
		x->c->dead = true;
		if (x->R->CommAbort && x->c->comm) { (void)x->R->CommAbort(x->c->comm); x->c->comm = nullptr; }
	}
	return fail(msg + " (the communicator was aborted: destroy it and create a new one)");
}
static int rccl_gather(void *self, float *d_B_new, double *, hipStream_t st) {
	RcclExchange *x = static_cast<RcclExchange *>(self);
	// in place: rank r's block already sits at d_B_new + r·blk·3
	const ncclResult_t r = x->R->AllGather(d_B_new + (size_t)x->c->rank * x->blk * 3, d_B_new, (size_t)x->blk * 3, ncclFloat, x->c->comm, st);
	return r == ncclSuccess ? 0 : nccl_failed(x, r, "ncclAllGather");
}
static int rccl_reduce(void *self, double *d_red, hipStream_t st) {
	RcclExchange *x = static_cast<RcclExchange *>(self);
	ncclResult_t r = x->R->AllReduce(d_red, d_red, 3, ncclDouble, ncclSum, x->c->comm, st);
	if (r == ncclSuccess) r = x->R->AllReduce(d_red + 3, d_red + 3, 1, ncclDouble, ncclMax, x->c->comm, st);
	return r == ncclSuccess ? 0 : nccl_failed(x, r, "ncclAllReduce");
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
	const Rccl *R = rccl("rtmi_comm_destroy"); // (takes the table's lock; the table exists if the communicator does)
	if (c->comm && R && R->CommDestroy) (void)R->CommDestroy(c->comm);
	delete c;
}

int rtmi_radiosity_solve_sharded_dev(rtmi_comm *c, const float *d_F, uint64_t ld, uint32_t n, uint32_t row0, uint32_t n_rows,
                                     const double rho[3], const float *d_E, float *d_B, float tol_rel, uint32_t max_sweeps,
                                     uint32_t check_every, rtmi_solve_stats *stats, void *stream) {
	const char *who = "rtmi_radiosity_solve_sharded_dev";
	if (!c || (!d_F && n_rows) || !rho || !d_E || !d_B) return fail(std::string(who) + ": NULL argument"); // (a rank may own no rows)
	if (n == 0) return fail(std::string(who) + ": n == 0");
	if (c->dead || !c->comm) return fail(std::string(who) + ": the communicator was aborted after an RCCL error; destroy it and create a new one");
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
	// From here on every rank reaches solve_loop's status agreement, whatever happens to it: a failed allocation is handed
	// in as setup_rc and told to the others there.  (The small block first: it carries the word the ranks agree on.)
	const size_t n_pad = (size_t)blk * W, vb = sizeof(float) * 3 * n_pad;
	if (n_pad > 0xffffffffull) return fail(std::string(who) + ": n too large");
	float *buf[2] = {nullptr, nullptr};
	double *d_norms = nullptr; // check_every windows of 64 slots × 3, then the 4 words of a check
	const size_t norm_doubles = 3 * (size_t)kNormSlots * check_every;
	int rc = 0;
	if (hipMalloc(&d_norms, sizeof(double) * (norm_doubles + 4)) != hipSuccess) d_norms = nullptr;
	if (!d_norms || env_uint("RTMI_DEBUG_FAIL_SOLVE_SETUP", 0) || hipMalloc(&buf[0], vb) != hipSuccess || hipMalloc(&buf[1], vb) != hipSuccess)
		rc = fail(std::string(who) + ": hipMalloc failed");
	RcclExchange xs = {c, R, blk};
	const SolveExchange X = {&xs, rccl_gather, rccl_reduce, false};
	rc = solve_loop(who, d_F, ld, n, (uint32_t)n_pad, row0, n_rows, rho, d_E, d_B, tol_rel, max_sweeps, check_every, &X, stats, (hipStream_t)stream,
	                buf[0], buf[1], d_norms, kNormSlots, d_norms ? d_norms + norm_doubles : nullptr, nullptr, nullptr, rc);
	for (float *b : buf) if (b) (void)hipFree(b);
	if (d_norms) (void)hipFree(d_norms);
	return rc;
}

} // extern "C"
