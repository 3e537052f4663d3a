// Host-side declarations shared by the translation units of librtmi.
#pragma once
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "bvh_build.hpp"
#include "rtmi_device.hpp"

struct rtmi_scene {
	rtmi::SceneDev dev;
	uint64_t n_tris;
	int device;
	uint32_t stack_bytes;        // dynamic LDS of the traversal kernels: room for the deepest path of either hierarchy, ints per lane
	uint32_t stack_bytes_binary; // ... of the kernels that walk the binary nodes only: (depth + 1) ints per lane
	rtmi_scene_info info;
	int *d_queue;                     // work-queue counters: ONE PER STREAM that has launched on this scene (kQueueSlots of them).
	                                  // Launches on one stream are ordered, so they can share a counter (zeroed on the stream
	                                  // before each kernel); launches on different streams never share one — no in-flight limit.
	std::mutex queue_mutex;
	std::unordered_map<void *, uint32_t> stream_slot;
	std::vector<int *> queue_more;    // further blocks of kQueueSlots counters, added when a host has launched on more distinct streams than
	                                  // the blocks so far hold (a server that makes a stream per request): slot k lives in block k / kQueueSlots
	std::unordered_map<const void *, int> blocks_per_cu; // hipOccupancyMaxActiveBlocksPerMultiprocessor per (kernel) at stack_bytes of LDS
	void *h_small = nullptr, *d_small = nullptr; // host-mapped staging of the host-pointer entries' small batches (lazily made)
	std::mutex small_mutex;
	int *h_error;                     // host-mapped word a kernel sets when a wave gives up (SceneDev::error points at it)
	int n_cus;
	uint32_t n_objects; // 1 + the largest tri_object seen at creation (materials are looked up by it)
	double *d_v1_orig; // n_tris × 3: triVert1 of every triangle in ORIGINAL order (hit points, object.cpp:23)
	uint64_t xyz_hash; // of the vertex array the scene was built from (rtmi_patches_create must be handed the same one)
	std::vector<uint32_t> wide_of_node; // per binary device node: rtmi::Wide4::cover of its subtree (the form-factor path tables start walks there)
};

namespace rtmi {
// 64-bit digest of a vertex array (bit patterns; not cryptographic): tells "the same array" from another one
inline uint64_t hash_doubles(const double *p, uint64_t n) {
	uint64_t h = 0x9E3779B97F4A7C15ull ^ n;
	for (uint64_t k = 0; k < n; ++k) {
		uint64_t w;
		__builtin_memcpy(&w, p + k, 8);
		h = (h ^ w) * 0xBF58476D1CE4E5B9ull;
		h ^= h >> 29;
	}
	return h;
}
constexpr uint32_t kQueueSlots = 256;
extern thread_local std::string g_err;
int fail(const std::string &msg);        // records msg for rtmi_last_error(), returns -1
int launch_check(const char *who);
int check_scene(const rtmi_scene *s, const char *who); // NULL / wrong current device -> -1
unsigned env_uint(const char *name, unsigned dflt); // tuning overrides from the environment
Tune launch_tune(Tune t, uint32_t grid, uint64_t n_items); // claim sizes for a launch of n_items on `grid` persistent workgroups
Tune refill_threshold();                          // RTMI_REFILL / RTMI_VOTE_NODE / RTMI_VOTE_LEAF, read once per process
// ray queries whose ray count lives on the device: min(cap, *d_n · mul) rays of the buffer are traced
int closest_hit_counted(rtmi_scene *s, const rtmi_ray *d_rays, uint64_t cap, const uint32_t *d_n, uint32_t mul, rtmi_hit *d_out,
                        void *stream, const char *who);
int any_hit_counted(rtmi_scene *s, const rtmi_ray *d_rays, uint64_t cap, const uint32_t *d_n, uint32_t mul, uint8_t *d_occ,
                    void *stream, const char *who);
// for queue kernels (trace_queue) defined in other translation units: a zeroed counter of the scene's ring, and the
// size of a persistent grid for `kernel` (workgroups of kBlock threads, rtmi_scene::stack_bytes of dynamic LDS)
int *queue_slot(rtmi_scene *s, hipStream_t st, const char *who);
uint32_t persistent_grid_of(rtmi_scene *s, const void *kernel, uint64_t n_items);
} // namespace rtmi
