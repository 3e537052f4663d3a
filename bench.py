#!/usr/bin/env python3
"""bench.py — the hot path on BASELINE.json's metric: "closest-hit Mrays/s on 1M-tri scene; form-factor+solve
time @64K patches".

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the closest-hit path over one batch of synthetic primary rays: the 3840x2160 frame of
the 1 Mi-triangle soup, rays generated on the device, hit records written to HBM.  With N
ranks the job is N samples per pixel of that frame (weak scaling: 8 294 400 rays per GPU per step); every
sample's 64x64 image tiles are dealt round-robin over the ranks, no collective on the data path.
`value` = rays of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline            the Jacobi sweep (the HBM-bound kernel of the metric's second half), 4·n_rows·n bytes/launch
  roofline_traversal  the closest-hit kernel the headline `value` times: SURVEY.md §8d's max(flop / 157.3 TFLOP/s,
                      cache bytes / 34.5 TB/s L2) from the algorithmic work per ray, both terms printed
  roofline_formfactor the form-factor kernel (98 % of "form-factor + solve time"): fp64 vector-ALU bound
  tile_launch         N=1 only: the frame as the 8 launches one rank of 8 issues, against the one full-frame launch
  strong_scaling_one_frame  N>1 only: ONE 1-spp frame dealt over the ranks, ms and efficiency against the full frame on one rank
  solver / formfactor the 64 K-patch solve (synthetic F, row-sharded over the ranks) and the form-factor build
  cpu_baseline        the oracle (CPU restatement, OpenMP) on a bounded sample of the same rays, N=1 only
  cpu_baseline_radiosity  the metric's second half on the CPU: the oracle (OpenMP) on the C2 box, and the COMPILED REFERENCE's
                      own findFormFactor / progressiveRefinement (1 core, as shipped) on its own element list
  reference_so        false (loudly) when oracle/_ref/*.so did not travel with the tree: the "reference" legs are then absent
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector
FP64_PEAK_TFLOPS = 78.6        # AMD's published MI355X FP64 vector peak (the guide gives no fp64 vector figure) = 256 CUs x 4 SIMDs x
                               # 16 lanes x 2 flop x 2.4 GHz, i.e. one v_fma_f64 per 4 cycles and SIMD; measured issue cost 5.1 cycles
                               # (tools/ub/valu_rate, profiles/ab_r03.md) = 61.7 TFLOP/s attainable
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md: L2, 8 XCDs aggregate
# Algorithmic fp64 flop of one form-factor sample: two sample points from their hashes (2 x 24), Unit(xj − xi)
# with its length (3 + 5 + sqrt + 3 div), the two cosines and the ray origin (5 + 6 + 6) = 77; and per ray that is shot
# (both cosines positive): the reference's Möller–Trumbore on triangle j (46), hit point (22), |p − xj| (9), delF (5) = 82.
FLOP_FF_SAMPLE = 77.0
FLOP_FF_RAY = 82.0
# algorithmic work per traversal event: one node visit = the four fp32 slab tests of a 4-wide node,
# one triangle test = the reference's Möller–Trumbore in fp64 (priced at 2 fp32-equivalents per flop)
FLOP_NODE_VISIT = 87.0          # 4 boxes x (6 fma = 12 flop + 6 min/max) + per node 6 fma + 3 ldexp
FLOP_TRI_TEST = 46.0 * 2.0
BYTES_NODE, BYTES_TRI, BYTES_HIT = 48.0, 80.0, 24.0


def halton(i, b):
    f, r = 1.0, 0.0
    while i > 0:
        f /= b
        r += f * (i % b)
        i //= b
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tris", type=int, default=1 << 20)
    ap.add_argument("--res", default="3840x2160")
    ap.add_argument("--patches", type=int, default=65536)
    ap.add_argument("--ff-rows", type=int, default=-1, help="-1: build the whole 64K-patch form-factor matrix and solve on it; K>0: a sample of K rows; 0: skip")
    ap.add_argument("--ff-launch-rows", type=int, default=0, help="rows per rtmi_form_factors_dev call of the build; 0 (default): this rank's whole block in ONE call")
    ap.add_argument("--no-solver", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-tiles", action="store_true", help="skip the tile_launch section (profiling runs: its launches would mix into the per-kernel averages)")
    ap.add_argument("--cpu-row-stride", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd import multi, workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))
    # RTMI_DIST_BACKEND=gloo rehearses the N>1 code path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives staged through the host): a correctness rehearsal, never a measurement.
    backend = os.environ.get("RTMI_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    stream = torch.cuda.current_stream().cuda_stream

    # ------------------------------------------------------------------ closest hit, C3
    x_res, y_res = (int(v) for v in args.res.lower().split("x"))
    tris = workloads.soup(args.tris, seed=1)
    scene = rt.Scene(tris)                      # host BVH build + upload, replicated per rank; not timed
    info = scene.info()
    cam = rt.Camera.from_cam14(workloads.SOUP_CAM14, (x_res, y_res))
    n_pix = x_res * y_res
    spp = world
    offs = [(0.5, 0.5)] + [(halton(s, 2), halton(s, 3)) for s in range(1, spp)]
    TILE = (64, 64)
    frames = [torch.empty(n_pix * rt.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(spp)]
    rays_rank = sum(multi.pixels_of_rank(x_res, y_res, TILE[0], TILE[1], (rank + s) % world, world) for s in range(spp))
    rays_total = spp * n_pix

    # At N > 1 a rank traces 1/N of the tiles of each of N samples per step.  As N launches they would each end in their own
    # tail (tile_launch below measures it at N = 1: 8 launches on one stream take 1.9x the one full-frame launch); the
    # library takes the samples of a step in ONE launch with one work queue (rtmi_primary_closest_hit_samples_dev).
    MAXS = rt.api.MAX_SAMPLES_PER_LAUNCH

    def step():
        if spp == 1:
            scene.primary_dev(cam, frames[0].data_ptr(), sx=offs[0][0], sy=offs[0][1], tile=TILE, stream=stream)
            return
        for s0 in range(0, spp, MAXS):
            scene.primary_samples_dev(cam, [(offs[s][0], offs[s][1], (rank + s) % world, frames[s].data_ptr()) for s in range(s0, min(spp, s0 + MAXS))],
                                      tile=TILE, tile_stride=world, stream=stream)

    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step()
        ev[k][1].record()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    launches_per_step = (spp + MAXS - 1) // MAXS
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / (args.steps * launches_per_step)   # average duration of one launch (HIP events)
    mrays = rays_total * args.steps / dt / 1e6

    # ------------------------------------------------------------------ strong scaling: ONE 1-spp frame dealt over the N ranks
    # (north_star's "near-linear tile scaling"; the reference's loop is one frame, rasterizer.cpp:219-233).  Every rank
    # traces every N-th 64x64 tile of the same frame into its own buffer; no collective on the data path (the final gather of
    # the disjoint tiles is a D2H copy per rank, not timed, as at N = 1).  efficiency = the full frame on ONE rank (measured
    # on every rank, the slowest taken) / (N x the dealt frame).
    strong = None
    if world > 1:
        def one_frame_dealt():
            scene.primary_dev(cam, frames[0].data_ptr(), tile=TILE, tile_first=rank, tile_stride=world, stream=stream)

        def one_frame_alone():
            scene.primary_dev(cam, frames[0].data_ptr(), tile=TILE, stream=stream)
        for fn in (one_frame_alone, one_frame_dealt):
            for _ in range(max(1, args.warmup)):
                fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_frame_alone()
        torch.cuda.synchronize()
        alone_ms = max_over_ranks((time.perf_counter() - t0) * 1e3 / args.steps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_frame_dealt()
        barrier()
        dealt_ms = max_over_ranks((time.perf_counter() - t0) * 1e3 / args.steps)
        strong = {"workload": "ONE C3 frame at 1 spp, its 64x64 tiles dealt round-robin over the %d ranks" % world,
                  "ms_per_frame": round(dealt_ms, 4), "mrays_per_s": round(n_pix / dealt_ms / 1e3, 1),
                  "full_frame_on_one_rank_ms": round(alone_ms, 4), "speedup": round(alone_ms / dealt_ms, 3),
                  "efficiency": round(alone_ms / dealt_ms / world, 4), "scaling": "strong"}

    # traversal work of the frame (diagnostic kernel, untimed) -> roofline of the closest-hit kernel
    counts = torch.zeros(6, dtype=torch.int64, device="cuda")
    scene.primary_work_dev(cam, counts.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    visits, tests, vslots, tslots, wleaf, widle = (int(v) for v in counts.cpu())
    flop_per_ray = (visits * FLOP_NODE_VISIT + tests * FLOP_TRI_TEST) / n_pix
    l2_bytes_per_ray = (visits * BYTES_NODE + tests * BYTES_TRI) / n_pix + BYTES_HIT
    rays_per_launch = rays_rank / launches_per_step
    trav_tflops = flop_per_ray * rays_per_launch / (kernel_ms * 1e-3) / 1e12
    cache_gbs = l2_bytes_per_ray * rays_per_launch / (kernel_ms * 1e-3) / 1e9
    valu_frac, cache_frac = trav_tflops / FP32_PEAK_TFLOPS, cache_gbs / L2_PEAK_GBS
    # fraction of roofline = max(flop / vector-ALU peak, cache bytes / L2 bandwidth), algorithmic work per ray
    roofline_traversal = {
        "kernel": "k_primary_closest" if spp == 1 else "k_primary_closest_samples", "bound": "l2" if cache_frac >= valu_frac else "valu",
        "achieved": round(cache_gbs if cache_frac >= valu_frac else trav_tflops, 3),
        "peak": L2_PEAK_GBS if cache_frac >= valu_frac else FP32_PEAK_TFLOPS,
        "unit": "GB/s through L1/L2 (algorithmic node + triangle + hit bytes)" if cache_frac >= valu_frac else "TFLOP/s (fp32-equivalent; fp64 flop = 2)",
        "frac": round(max(valu_frac, cache_frac), 4),
        "frac_valu": round(valu_frac, 4), "valu_tflops": round(trav_tflops, 3), "valu_peak_tflops": FP32_PEAK_TFLOPS,
        "frac_cache": round(cache_frac, 4), "cache_gbs": round(cache_gbs, 1), "cache_peak_gbs": L2_PEAK_GBS,
        "kernel_ms": round(kernel_ms, 4), "rays_per_launch": int(rays_per_launch),
        "node_visits_per_ray": round(visits / n_pix, 2), "tri_tests_per_ray": round(tests / n_pix, 2),
        "cache_bytes_per_ray": round(l2_bytes_per_ray, 1), "flop_per_ray": round(flop_per_ray, 1),
        "lane_occupancy_node_loop": round(visits / max(1, vslots), 3), "lane_occupancy_tri_loop": round(tests / max(1, tslots), 3),
        "node_step_lanes_holding_a_leaf": round(wleaf / max(1, vslots), 3), "node_step_lanes_without_a_ray": round(widle / max(1, vslots), 3),
        "traffic": None,
    }

    # ------------------------------------------------------------------ what a rank's launch shape at N = 8 costs, measured at N = 1
    # The same frame as 8 launches of every 8th tile (what one rank of 8 issues per step), on one stream and dealt over
    # 4 streams as step() does, against the one full-frame launch: efficiency = full-frame time / time of the 8 launches.
    tile_launch = None
    if world == 1 and not args.no_tiles:
        side = [torch.cuda.Stream() for _ in range(4)]

        def eighths(handles):
            for s in range(8):
                scene.primary_dev(cam, frames[0].data_ptr(), tile=TILE, tile_first=s, tile_stride=8, stream=handles[s % len(handles)])

        def wall_ms(fn, reps):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / reps

        reps = max(3, min(10, args.steps))
        full_ms = wall_ms(lambda: scene.primary_dev(cam, frames[0].data_ptr(), tile=TILE, stream=stream), reps)
        one_ms = wall_ms(lambda: eighths([stream]), reps)
        four_ms = wall_ms(lambda: eighths([st.cuda_stream for st in side]), reps)
        # the same rays as ONE launch of 8 samples x every 8th tile: what bench.py --gpus 8 issues per rank and step
        set_ms = wall_ms(lambda: scene.primary_samples_dev(cam, [(0.5, 0.5, s, frames[0].data_ptr()) for s in range(8)], tile=TILE, tile_stride=8,
                                                           stream=stream), reps)
        scene.primary_dev(cam, frames[0].data_ptr(), tile=TILE, stream=stream)      # frames[0] again holds the full frame
        torch.cuda.synchronize()
        tile_launch = {"workload": "the C3 frame as 8 launches of every 8th 64x64 tile (one rank's launches at N = 8)",
                       "full_frame_ms": round(full_ms, 3), "eight_launches_one_stream_ms": round(one_ms, 3),
                       "eight_launches_four_streams_ms": round(four_ms, 3), "one_launch_of_eight_samples_ms": round(set_ms, 3),
                       "tile_launch_efficiency_one_stream": round(full_ms / one_ms, 3),
                       "tile_launch_efficiency_four_streams": round(full_ms / four_ms, 3),
                       "tile_launch_efficiency": round(full_ms / set_ms, 3),
                       "note": "bench.py --gpus N issues a rank's N tile sets as one launch (rtmi_primary_closest_hit_samples_dev), so "
                               "`tile_launch_efficiency` = full frame / one launch of 8 samples is the one that applies; the other two are "
                               "what a caller pays who launches per tile set (4 streams: depends on how the streams fall on the 4 hardware queues)"}

    # ------------------------------------------------------------------ radiosity solve, C5 (synthetic F)
    solver = None
    roofline = None
    formfactor = None
    roofline_formfactor = None
    if not args.no_solver:
        n = args.patches
        row0, n_rows, _ = multi.rows_of_rank(n, rank, world)
        ld = (n + 31) // 32 * 32
        F = workloads.synthetic_F(n, n_rows=n_rows, row0=row0, seed=1, ld=ld)
        rho = np.array([0.7, 0.7, 0.7])
        E_h = np.zeros((n, 3), np.float32)
        E_h[::97] = 68500.0 / 97.0
        E_h[n - 1] = 68500.0
        E = torch.from_numpy(E_h).cuda()

        def sweep(B_old, B_new, norms):
            rt.jacobi_sweep_dev(F.data_ptr(), ld, n, row0, n_rows, rho, E.data_ptr(), B_old.data_ptr(), B_new.data_ptr(),
                                norms.data_ptr(), stream)

        # (a) the kernel alone: K sweeps back to back, HIP events on the launch stream
        Ba = E.clone(); Bb = torch.zeros_like(Ba)
        norms = torch.zeros(3, dtype=torch.float64, device="cuda")
        for _ in range(3):
            sweep(Ba, Bb, norms)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_sw = max(10, args.steps)
        barrier()
        e0.record()
        for _ in range(n_sw // 2):
            sweep(Ba, Bb, norms)
            sweep(Bb, Ba, norms)
        e1.record()
        torch.cuda.synchronize()
        sweep_ms = e0.elapsed_time(e1) / (2 * (n_sw // 2))
        bytes_launch = 4.0 * n_rows * n
        gbs = bytes_launch / (sweep_ms * 1e-3) / 1e9
        roofline = {"kernel": "k_jacobi_sweep", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                    "bytes_per_launch": bytes_launch, "kernel_ms": round(sweep_ms, 4)}
        tr = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch, measured separately
        if os.path.exists(tr):
            try:
                tj = json.load(open(tr))
                # `traffic` is a STORED measurement (PMC passes of an earlier run of this same launch shape), not of this run
                if tj.get("k_jacobi_sweep", {}).get("bytes_per_launch") == bytes_launch:
                    roofline["traffic"] = tj["k_jacobi_sweep"]["hbm_bytes"]
                    roofline["traffic_source"] = "stored: profiles/%s_rocprof.md (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same launch shape)" % tj["k_jacobi_sweep"].get("source", "?")
                if tj.get("k_primary_closest", {}).get("rays_per_launch") == int(rays_per_launch):
                    roofline_traversal["traffic"] = tj["k_primary_closest"]["hbm_bytes"]
                    # PMC shares of the same kernel (rocprofv3 passes of tools/profile_gpu.sh; profiles/<source>_rocprof.md)
                    for key in ("l2_hit_rate", "valu_active_share_of_wave_cycles", "wait_any_share_of_wave_cycles", "lanes_active_per_valu_instruction",
                                "valu_issue_utilisation", "valu_instructions_per_ray"):
                        if key in tj["k_primary_closest"]:
                            roofline_traversal[key] = tj["k_primary_closest"][key]
                    roofline_traversal["pmc_source"] = "stored: profiles/%s_rocprof.md" % tj["k_primary_closest"].get("source", "?")
                    roofline_traversal["traffic_source"] = roofline_traversal["pmc_source"]
            except Exception:
                pass
        # (b) the whole solve to 1e-4 relative change (north star), exchange included when N > 1
        barrier()
        t0 = time.perf_counter()
        if world == 1:
            B = torch.zeros_like(E)
            stats = rt.api.rtmi_solve_stats()
            rc = rt.lib.rtmi_radiosity_solve_dev(F.data_ptr(), ld, n, 0, n, rho.ctypes.data, E.data_ptr(), B.data_ptr(), 1e-4,
                                                 1000, 4, None, None, rt.api.C.byref(stats), stream)
            if rc != 0:
                raise SystemExit(rt.lib.rtmi_last_error().decode())
            sweeps, rel = stats.sweeps, stats.rel_change
            solve_path = "rtmi_radiosity_solve_dev"
        elif backend == "nccl":
            # the sharded loop inside the library: sweep kernel + RCCL all-gather enqueued back to back, host waits at the checks
            try:
                B, stats = multi.solve_sharded_library(F.data_ptr(), ld, n, row0, n_rows, rho, E, tol_rel=1e-4, max_sweeps=1000, check_every=4, stream=stream)
                sweeps, rel = stats.sweeps, stats.rel_change
                solve_path = "rtmi_radiosity_solve_sharded_dev (RCCL inside the library)"
            except rt.RtmiError as e:     # e.g. no librccl for dlopen: the Python loop over torch.distributed does the same iteration
                print("bench.py: library exchange unavailable (%s): falling back to multi.solve_sharded" % e, file=sys.stderr, flush=True)
                r = multi.solve_sharded(sweep, E, n, tol_rel=1e-4, max_sweeps=1000, check_every=4)
                B, sweeps, rel = r.B, r.sweeps, r.rel_change
                solve_path = "multi.solve_sharded (torch.distributed)"
        else:
            r = multi.solve_sharded(sweep, E, n, tol_rel=1e-4, max_sweeps=1000, check_every=4)
            B, sweeps, rel = r.B, r.sweeps, r.rel_change
            solve_path = "multi.solve_sharded (torch.distributed, %s)" % backend
        barrier()
        solve_s = max_over_ranks(time.perf_counter() - t0)
        # residual of the solution: one more sweep must not move it
        norms.zero_()
        Bc = torch.zeros((B.shape[0], 3), dtype=torch.float32, device="cuda")
        sweep(B.contiguous(), Bc, norms)
        torch.cuda.synchronize()
        solver = {"patches": n, "F": "synthetic uniform(0,1), zero diagonal, row sums 0.9, fp32, %.2f GiB/rank" % (n_rows * ld * 4 / 2**30),
                  "scaling": "strong (rows sharded over ranks, RCCL all-gather of B per sweep)" if world > 1 else "single GPU",
                  "path": solve_path, "sweeps": int(sweeps), "rel_change": rel, "solve_seconds": round(solve_s, 5),
                  "ms_per_sweep_kernel": round(sweep_ms, 4), "ms_per_sweep_in_solve": round(solve_s * 1e3 / max(1, sweeps), 4)}
        del F
        torch.cuda.empty_cache()

        # (c) form factors at 64 K patches: box with 12·74² − (ceiling opening) patches, light above the opening.
        #     --ff-rows -1 (default): this rank's whole row block in ONE call (--ff-launch-rows K: in calls of K rows; one launch of
        #     the whole block runs 15 % faster than launches of 4096 rows, profiles/ab_r03.md), then the solve on that
        #     real F; --ff-rows K > 0: a sample of K rows only (extrapolated, marked as such).
        if args.ff_rows != 0:
            k = max(1, int(round((n / 12.0) ** 0.5)))
            verts, normals, light = workloads.closed_box(k, hole=0.3)
            # one-time set-up of the radiosity path, timed: scene build (host BVH + upload), patch set
            # (rtmi_patches_create), and below the first rtmi_form_factors_dev call (path tables on the host + the library's
            # timing of its two kernels on up to 1024 rows)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            box = rt.Scene(verts)
            torch.cuda.synchronize()
            setup_scene_s = time.perf_counter() - ts
            ts = time.perf_counter()
            rad = rt.Radiosity(box, normals, light)
            torch.cuda.synchronize()
            setup_patches_s = time.perf_counter() - ts
            nb = rad.n
            ldb = (nb + 31) // 32 * 32   # rows on 128-byte lines (include/rtmi.h: the sweeps read such a matrix 8 % faster)
            r0, nr, _ = multi.rows_of_rank(nb, rank, world)
            full = args.ff_rows < 0
            nr_s = nr if full else min(args.ff_rows, nr)
            Fb = torch.empty((nr_s, ldb), dtype=torch.float32, device="cuda")
            lc = np.ascontiguousarray(light, np.float64)

            def build(first, count):
                rc = rt.lib.rtmi_form_factors_dev(rad.handle, lc.ctypes.data, 10, 30, 1, 0, r0 + first, count,
                                                  Fb[first:].data_ptr(), ldb, stream)
                if rc != 0:
                    raise SystemExit(rt.lib.rtmi_last_error().decode())
            torch.cuda.synchronize()
            ts = time.perf_counter()
            build(0, min(1024, nr_s))                                                # first call: builds the path tables and lets the library time its two kernels
            torch.cuda.synchronize()
            setup_first_call_s = time.perf_counter() - ts
            barrier()
            fe0, fe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_launch = 0
            t0 = time.perf_counter()
            fe0.record()
            lrows = nr_s if args.ff_launch_rows <= 0 else min(args.ff_launch_rows, nr_s)
            for first in range(0, nr_s, lrows):
                build(first, min(lrows, nr_s - first))
                n_launch += 1
            fe1.record()
            barrier()
            ff_s = max_over_ranks(time.perf_counter() - t0)
            ff_kernel_ms = fe0.elapsed_time(fe1) / n_launch      # k_form_factors (+ the light column's rays) per launch, HIP events
            box.check()
            ne = rad.numOfElements
            vis_rays = 10.0 * (ne - 1) * (nb - 1 if full else nr_s * world)
            # samples that shoot a ray: both cosines positive <=> the two patches do not lie in one plane facing the same way
            # (a box: pairs on different walls); counted on the host from the normals
            _, inv, cnt = np.unique(np.round(normals, 9), axis=0, return_inverse=True, return_counts=True)
            shoot = 1.0 - float((cnt.astype(np.float64) ** 2).sum()) / float(ne) ** 2
            pairs_launch = 10.0 * (ne - 1) * nr_s / n_launch
            ff_tflops = (FLOP_FF_SAMPLE + FLOP_FF_RAY * shoot) * pairs_launch / (ff_kernel_ms * 1e-3) / 1e12
            roofline_formfactor = {
                "kernel": "k_form_factors", "bound": "valu (fp64 vector)", "achieved": round(ff_tflops, 3), "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s fp64 (algorithmic: %.0f flop per sample + %.0f per ray shot; divisions and roots count 1)" % (FLOP_FF_SAMPLE, FLOP_FF_RAY),
                "frac": round(ff_tflops / FP64_PEAK_TFLOPS, 4), "kernel_ms": round(ff_kernel_ms, 3), "rows_per_launch": lrows,
                "pair_samples_per_launch": pairs_launch, "rays_shot_fraction": round(shoot, 4),
                "G_pair_samples_per_s": round(pairs_launch / (ff_kernel_ms * 1e-3) / 1e9, 2), "traffic": None}
            tr = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tr):
                try:
                    tj = json.load(open(tr)).get("k_form_factors", {})
                    for key in ("hbm_bytes_per_entry_fetch", "hbm_bytes_per_entry_write", "l2_hit_rate", "valu_issue_utilisation",
                                "valu_instructions_per_lane_sample", "scratch_bytes", "vgprs"):
                        if key in tj:
                            roofline_formfactor[key] = tj[key]
                    if "hbm_bytes_per_launch" in tj and tj.get("rows_per_launch"):   # per launch of THIS size: the stored bytes scale with the rows
                        roofline_formfactor["traffic"] = tj["hbm_bytes_per_launch"] * lrows / tj["rows_per_launch"]
                        roofline_formfactor["traffic_note"] = "scaled from a %d-row launch (bytes per matrix entry do not depend on the launch size)" % tj["rows_per_launch"]
                    if tj:
                        roofline_formfactor["pmc_source"] = "stored: profiles/%s" % tj.get("source", "?")
                except Exception:
                    pass
            formfactor = {"patches": ne, "rows_built_per_rank": nr_s, "seconds": round(ff_s, 4),
                          "kernel_chosen_by_the_library": {0: "not measured", 1: "k_form_factors (path walk)", 2: "k_form_factors_flat (root traversal)"}.get(
                              int(rt.lib.rtmi_debug_form_factor_kernel(rad.handle)), "?"),
                          "pair_samples_M_per_s": round(vis_rays / ff_s / 1e6, 1), "samples_per_pair": 10,
                          "note": "a visibility ray is shot only for samples whose two cosines are positive (others add 0 whatever the ray says)",
                          "full_matrix": full, "F_GiB_per_rank": round(nr_s * ldb * 4 / 2**30, 2)}
            # the rows tests/test_gpu_full_size.py compares with the oracle: CRC of the matrix that was just timed
            import zlib
            pinned = [r for r in (0, 21000, 43210, 64743) if r0 <= r < r0 + nr_s and r < ne]
            formfactor["crc32_of_pinned_rows"] = {str(r): "%08x" % zlib.crc32(Fb[r - r0, : nb].cpu().numpy().tobytes()) for r in pinned}
            try:   # the oracle's CRCs of the same rows (oracle/gen_golden_c5_rows.py; tests/test_gpu_full_size.py asserts them too)
                pin = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_box_rows.json")))
                if pin.get("patches") == ne:
                    formfactor["pinned_rows_equal_the_oracle"] = all(formfactor["crc32_of_pinned_rows"][str(r)] == pin["crc32"][str(r)] for r in pinned)
            except Exception:
                pass
            setup_s = max_over_ranks(setup_scene_s + setup_patches_s + setup_first_call_s)
            formfactor["setup_seconds"] = round(setup_s, 4)
            formfactor["setup"] = {"scene_build_and_upload": round(setup_scene_s, 4), "patch_set": round(setup_patches_s, 4),
                                   "first_call_tables_and_kernel_choice": round(setup_first_call_s, 4),
                                   "note": "once per scene / patch set, outside `seconds`; the first call also writes its (up to 1024) rows"}
            if not full:
                formfactor["extrapolated_full_build_seconds"] = round(ff_s * nr / nr_s, 1)
            else:
                # the solve on the F just built (the metric's "form-factor + solve time")
                rad.FF, rad.ld, rad.row0, rad.n_rows = Fb, ldb, r0, nr
                barrier()
                t0 = time.perf_counter()
                rad.progressiveRefinement(tol_rel=1e-4, max_sweeps=1000, check_every=4)
                barrier()
                sv = max_over_ranks(time.perf_counter() - t0)
                Bh = rad.B[: rad.numOfElements, 0].float().cpu().numpy()
                formfactor.update({"solve_seconds": round(sv, 4), "solve_sweeps": int(rad.stats.sweeps),
                                   "solve_rel_change": rad.stats.rel_change, "formfactor_plus_solve_seconds": round(ff_s + sv, 3),
                                   "formfactor_plus_solve_seconds_cold": round(ff_s + sv + setup_s, 3),
                                   "B_mean": float(Bh.mean()), "B_max": float(Bh.max()),
                                   "row_sum_mean_of_local_rows": float(Fb[:, : rad.numOfElements].sum(1).mean())})
            del Fb
            rad.FF = None
            rad.close(); box.close()

    # ------------------------------------------------------------------ CPU baseline (oracle), rank 0, N = 1
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle_py import Oracle
        ora = Oracle()
        rows = np.arange(0, y_res, max(1, args.cpu_row_stride), dtype=np.uint32)
        ii, jj = np.meshgrid(rows, np.arange(x_res, dtype=np.uint32), indexing="ij")
        pix = np.stack([ii.ravel(), jj.ravel()], 1)
        d = ora.primary_dirs(ora.raster_details(workloads.SOUP_CAM14, x_res, y_res), pix)
        o = np.tile(workloads.SOUP_CAM14[:3], (len(d), 1))
        h = ora.bvh_build(tris)
        secs, reps = 0.0, 0
        while secs < args.cpu_seconds and reps < 64:          # bounded: about cpu_seconds of CPU tracing
            (tri, t, u, v), s1 = ora.bvh_closest_hit(h, o, d)
            secs += s1
            reps += 1
        ora.bvh_free(h)
        got = frames[0].cpu().numpy().view(rt.HIT_DTYPE)[pix[:, 0].astype(np.int64) * x_res + pix[:, 1]]
        mism = int((got["tri"] != np.where(tri < 0, rt.MISS, tri).astype(np.uint32)).sum() + (got["t"] != t).sum())
        cpu = {"value": round(reps * len(d) / secs / 1e6, 3), "unit": "Mrays/s", "cores": ora.num_threads(), "kind": "port",
               "sample": "every %s row of the %dx%d frame (%d of the step's rays) traced %d times = %.1f s: oracle BVH + the reference's double Moller-Trumbore" % (
                   "" if args.cpu_row_stride == 1 else "%d-th" % args.cpu_row_stride, x_res, y_res, len(d), reps, secs),
               "gpu_vs_oracle_mismatches_on_sample": mism}

    # ------------------------------------------------------------------ CPU baseline of the metric's second half
    # (i) the oracle (OpenMP, its own hierarchy) on the C2 box: form factors + direct solve, beside the GPU on the same box;
    # (ii) the COMPILED REFERENCE's own Radiosity::findFormFactor / progressiveRefinement (1 thread, as shipped) on the
    #      element list its BSP build makes of test1, beside the GPU on the same elements.
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libref_render2.so")
    have_ref_so = os.path.exists(ref_so)
    if rank == 0 and not have_ref_so:
        print("bench.py: oracle/_ref/libref_render2.so is ABSENT (it is built where /root/reference exists and travels with the "
              "tree): every 'kind: reference' leg below is missing", file=sys.stderr, flush=True)
    cpu_rad = None
    if world == 1 and rank == 0 and not args.no_cpu and not args.no_solver:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle_py import Oracle
        ora = Oracle()
        v2, n2, l2 = workloads.closed_box(18, hole=0.3)             # C2: 12·18² patches less the ceiling opening the light shines through
        nb2 = len(v2)
        # GPU: whole matrix + solve
        sc2 = rt.Scene(v2)
        rad2 = rt.Radiosity(sc2, n2, l2)
        rad2.findFormFactor(samples=10, light_samples=30, seed=1, row0=0, n_rows=1024)     # warm-up; the library times its two kernels on it
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        F2 = rad2.findFormFactor(samples=10, light_samples=30, seed=1)
        torch.cuda.synchronize()
        g_ff = time.perf_counter() - t0
        t0 = time.perf_counter()
        rad2.progressiveRefinement(tol_rel=1e-7, check_every=4)
        torch.cuda.synchronize()
        g_sv = time.perf_counter() - t0
        # oracle: a bounded sample of rows (about 10 s), extrapolated to the matrix; the direct solve of the GPU's F
        hb = ora.bvh_build(v2)
        rows_done, secs = 0, 0.0
        same = True
        while secs < 10.0 and rows_done < nb2:
            k = min(64, nb2 - rows_done)
            t0 = time.perf_counter()
            Fo = ora.form_factors_bvh(v2, n2, 10, 1, row0=rows_done, n_rows=k, bvh=hb)
            secs += time.perf_counter() - t0
            same = same and bool(np.array_equal(Fo.astype(np.float32), F2[rows_done:rows_done + k, :nb2].cpu().numpy()))
            rows_done += k
        ora.bvh_free(hb)
        t0 = time.perf_counter()
        Fh = F2[:, : nb2 + 1].cpu().numpy().astype(np.float64)
        K2 = np.eye(nb2 + 1) - 0.70 * Fh                           # radiosity.cpp:413-428; the light's row of F is zero
        E2 = np.zeros(nb2 + 1); E2[nb2] = 68500.0
        Bd = np.linalg.solve(K2, E2)
        o_sv = time.perf_counter() - t0
        Bg = rad2.B[:nb2, 0].double().cpu().numpy()
        cpu_rad = {"c2_oracle": {"workload": "C2: closed box, %d patches, 10 samples per pair: form factors + solve" % nb2, "kind": "port",
                                 "cores": ora.num_threads(), "formfactor_seconds": round(secs * nb2 / rows_done, 2),
                                 "sample": "%d of %d rows through the oracle's own hierarchy = %.1f s, extrapolated" % (rows_done, nb2, secs),
                                 "solve_seconds": round(o_sv, 3), "solve": "numpy/LAPACK direct solve of (I − rho F) B = E in double",
                                 "gpu_formfactor_seconds": round(g_ff, 4), "gpu_solve_seconds": round(g_sv, 4), "gpu_solve_sweeps": int(rad2.stats.sweeps),
                                 "gpu_rows_equal_oracle_rows": same,
                                 "gpu_vs_direct_solve_rel_l2": float(np.linalg.norm(Bg - Bd[:nb2]) / np.linalg.norm(Bd[:nb2]))}}
        rad2.close(); sc2.close()
        if have_ref_so:
            import tempfile
            from oracle_py import Reference
            g = np.load(os.path.join(ROOT, "tests", "golden", "g2_test1.npz"))
            ref = Reference(ref_so)
            work = tempfile.mkdtemp()
            hr, ok = None, False
            for _ in range(10):                       # the reference's BSP build uses an unseeded RNG and may throw
                hr = ref.scene_from_triangles(g["tris"], g["tri_obj"], g["light_center"])
                if ref.build_bsp(hr, work) == 0:
                    ok = True
                    break
            if ok:
                ne_r = ref.rad_setup(hr)              # elements = the leaves of the reference's BSP (radiosity.cpp:137-163)
                ev, en, _ = ref.rad_elements(hr, ne_r)
                r_ff = ref.rad_find_form_factor(hr)   # the reference's own Monte-Carlo fill (unseeded), seconds
                r_sv = ref.rad_solve(hr)              # its progressiveRefinement on that F
                sc3 = rt.Scene(ev)
                rad3 = rt.Radiosity(sc3, en, g["light_center"])
                rad3.findFormFactor(samples=10, light_samples=30, seed=1, row0=0, n_rows=1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rad3.findFormFactor(samples=10, light_samples=30, seed=1)
                torch.cuda.synchronize()
                g3_ff = time.perf_counter() - t0
                t0 = time.perf_counter()
                rad3.progressiveRefinement(tol_rel=1e-7, check_every=4)
                torch.cuda.synchronize()
                g3_sv = time.perf_counter() - t0
                cpu_rad["test1_reference"] = {
                    "workload": "Render2/scenes/test1: the %d elements the reference's BSP build made of its 24 triangles (+ the light)" % ne_r,
                    "kind": "reference", "cores": 1, "findFormFactor_seconds": round(r_ff, 3), "progressiveRefinement_seconds": round(r_sv, 7),
                    "reference_residual": float(ref.rad_residual(hr)),
                    "gpu_formfactor_seconds": round(g3_ff, 5), "gpu_solve_seconds": round(g3_sv, 5),
                    "note": "the reference's sample points are unseeded: its F differs from run to run; the GPU's F on the same elements is the oracle's bit for bit (tests/test_gpu_adapter.py)"}
                rad3.close(); sc3.close()

    # ------------------------------------------------------------------ C1: the compiled reference itself (configs[0])
    # The reference cannot run C3 (its BSP build is quadratic and throws on real meshes), so its own CPU path is
    # timed on its own scene: test1, 500x400 primary rays through Scene::Cast, beside the GPU on the same frame.
    # Needs oracle/_ref/libref_render2.so (built where /root/reference exists; the .so travels to the GPU box).
    c1 = None
    if world == 1 and rank == 0 and not args.no_cpu and have_ref_so:
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle_py import Reference
        g = np.load(os.path.join(ROOT, "tests", "golden", "g2_test1.npz"))
        ref = Reference(ref_so)
        work = tempfile.mkdtemp()
        h, ok = None, False
        for _ in range(10):                       # the reference's BSP build uses an unseeded RNG and may throw
            h = ref.scene_from_triangles(g["tris"], g["tri_obj"], g["light_center"])
            if ref.build_bsp(h, work) == 0:
                ok = True
                break
        if ok:
            s1 = rt.Scene(g["tris"], g["tri_obj"])
            cam1 = rt.Camera.from_cam14(g["cam"], g["res"])
            n1 = cam1.x_res * cam1.y_res
            r1 = torch.zeros(n1 * 64, dtype=torch.uint8, device="cuda")
            o1 = torch.zeros(n1 * 24, dtype=torch.uint8, device="cuda")
            rt.primary_rays_dev(cam1, r1.data_ptr(), stream=stream)
            s1.primary_dev(cam1, o1.data_ptr(), tile=TILE, stream=stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                s1.primary_dev(cam1, o1.data_ptr(), tile=TILE, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            gpu_ms = e0.elapsed_time(e1) / 20
            hr = r1.cpu().numpy().view(rt.RAY_DTYPE)
            hh = o1.cpu().numpy().view(rt.HIT_DTYPE)
            secs, reps, c = 0.0, 0, None
            while secs < 10.0 and reps < 40:       # about 10 s of the reference's Scene::Cast, 1 thread as shipped
                c = ref.cast(h, hr["o"], hr["d"])
                secs += c["seconds"]
                reps += 1
            hit = hh["tri"] != rt.MISS
            same = bool(np.array_equal(c["hit"].astype(bool), hit))
            dmax = float(np.max(np.abs(c["dist"][hit] - hh["t"][hit]) / hh["t"][hit])) if hit.any() else 0.0
            c1 = {"workload": "C1: Render2/scenes/test1 (24 triangles), 500x400 primary rays, closest hit",
                  "reference_Mrays_per_s": round(reps * n1 / secs / 1e6, 4), "reference_cores": 1, "kind": "reference",
                  "sample": "%d frames of Scene::Cast through the reference's BSP = %.1f s" % (reps, secs),
                  "gpu_Mrays_per_s": round(n1 / gpu_ms / 1e3, 1), "gpu_ms": round(gpu_ms, 4),
                  "same_hit_miss": same, "max_rel_distance_diff": dmax}

    # ------------------------------------------------------------------ frames: Scene::Trace + Shader::Shade on the device
    # (the caller of the ray queries.)  The shipped test1 frame (500x400, ambient 0.1, 10 AO rays,
    # depth 1) beside the compiled reference's own Scene::Trace loop on the same scene, and a 1920x1080 frame of a
    # 147 K-triangle studio.  Not part of `value`.
    frame = None
    if world == 1 and rank == 0 and not args.no_cpu:
        from raytracer_amd import workloads as W

        def frame_ms(sh, cam, reps):
            n = cam.x_res * cam.y_res
            rgb = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
            pix = torch.zeros(n, 3, dtype=torch.uint8, device="cuda")
            sh.render_frame_dev(cam, rgb.data_ptr(), pix.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                sh.render_frame_dev(cam, rgb.data_ptr(), pix.data_ptr(), stream=stream)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps, rgb

        g6 = np.load(os.path.join(ROOT, "tests", "golden", "g6_shade_test1.npz"))
        sc = rt.Scene(g6["tris"], g6["tri_obj"])
        shd = rt.Shader(sc, g6["normals"], g6["mats"], g6["lights"], max_tree_depth=1, ao_rays=10)
        cam6 = rt.Camera.from_cam14(g6["cam14"], g6["res"])
        ms1, rgb1 = frame_ms(shd, cam6, 20)
        frame = {"c1_frame": {"workload": "Render2/scenes/test1 as main.cpp renders it: 500x400, 10 AO rays + 1 shadow ray per hit, depth 1",
                              "gpu_ms": round(ms1, 4), "gpu_Mpixels_per_s": round(0.2 / ms1 * 1e3, 1)}}
        if c1 is not None and ok:
            ref.set_materials(h, g6["mats"])
            o6 = np.tile(g6["cam14"][:3], (int(g6["res"][0]) * int(g6["res"][1]), 1))
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            from oracle_py import Oracle as _O
            _o = _O()
            pix6 = np.array([(i, j) for i in range(int(g6["res"][1])) for j in range(int(g6["res"][0]))], np.uint32)
            d6 = _o.primary_dirs(_o.raster_details(g6["cam14"], int(g6["res"][0]), int(g6["res"][1])), pix6)
            secs, reps, ref_rgb = 0.0, 0, None
            while secs < 5.0 and reps < 10:
                ref_rgb, s_ = ref.trace(h, o6, d6)
                secs += s_
                reps += 1
            dmax = float(np.abs(rgb1.cpu().numpy() - ref_rgb).max())
            frame["c1_frame"].update({"reference_ms": round(1e3 * secs / reps, 1), "reference_cores": 1, "kind": "reference",
                                      "sample": "%d frames of the compiled reference's Scene::Trace loop = %.1f s" % (reps, secs),
                                      "max_colour_diff_to_reference": round(dmax, 4),
                                      "note": "the reference draws its AO rays from an unseeded generator (colours may differ by multiples of 0.0085, at most 0.085); on this scene no AO ray is occluded, so 0.0 means the two frames are identical"})
        vs, ob, ms_, ls = W.studio(64, 0.1)
        sc2 = rt.Scene(vs, ob)
        shd2 = rt.Shader(sc2, W.face_normals(vs), ms_, ls, max_tree_depth=1, ao_rays=10)
        cam7 = rt.Camera.from_cam14(W.STUDIO_CAM14, (1920, 1080))
        ms2, _ = frame_ms(shd2, cam7, 10)
        shd3 = rt.Shader(sc2, W.face_normals(vs), ms_, ls, max_tree_depth=3, ao_rays=10)
        ms3, _ = frame_ms(shd3, cam7, 5)
        frame["studio_1080p"] = {"workload": "workloads.studio(64): %d triangles, 6 materials (Phong, microfacet, mirror, 2 refracting), 2 lights, 1920x1080, 10 AO rays" % len(vs),
                                 "depth1_ms": round(ms2, 3), "depth1_Mpixels_per_s": round(1920 * 1080 / ms2 / 1e3, 1),
                                 "depth3_ms": round(ms3, 3)}
        # C4 (BASELINE config 4): ~100 K-triangle room, 1920x1080, primary closest hit, then per pixel 16 ambient-occlusion rays
        # (radius 2, any hit), 1 hard-shadow ray (any hit) and — for a quarter of the frame — 100 soft-shadow rays (closest hit),
        # every secondary ray built inside the query kernel (rtmi_ao_occlusion_dev, rtmi_shadow_occlusion_dev, rtmi_soft_shadow_dev)
        room, rn, light4 = W.closed_box(91)
        inner, inn, _ = W.closed_box(6, side=2.0)
        inner = inner + np.tile([4.0, 4.0, 0.0], 3)[None, :]
        tr4 = np.concatenate([room, inner])
        nr4 = torch.from_numpy(np.ascontiguousarray(np.concatenate([rn, -inn]))).cuda()
        sc4 = rt.Scene(tr4)
        cam4 = rt.Camera.from_cam14(np.array([1.0, 1.0, 5.0, 6.0, 6.0, 1.0, 0, 0, 1, 1.0, -1.0, 1.0, -0.5625, 0.5625]), (1920, 1080))
        n4, spp4, np4 = cam4.x_res * cam4.y_res, 16, 100
        m4 = n4 // 4
        rays4 = torch.zeros(n4 * 64, dtype=torch.uint8, device="cuda")
        hits4 = torch.zeros(n4 * 24, dtype=torch.uint8, device="cuda")
        occ4 = torch.zeros(n4 * spp4, dtype=torch.uint8, device="cuda")
        socc4 = torch.zeros(n4, dtype=torch.uint8, device="cuda")
        contrib4 = torch.zeros(m4 * np4, dtype=torch.uint8, device="cuda")
        fac4 = torch.zeros(m4, dtype=torch.float64, device="cuda")
        rt.primary_rays_dev(cam4, rays4.data_ptr(), stream=stream)
        stages = (("primary_ms", n4, lambda: sc4.primary_dev(cam4, hits4.data_ptr(), stream=stream)),
                  ("ao_16spp_ms", n4 * spp4, lambda: sc4.ao_occlusion_dev(nr4.data_ptr(), rays4.data_ptr(), hits4.data_ptr(), n4, spp4, 2.0, 1, occ4.data_ptr(), stream)),
                  ("hard_shadow_ms", n4, lambda: sc4.shadow_occlusion_dev(nr4.data_ptr(), rays4.data_ptr(), hits4.data_ptr(), n4, light4, socc4.data_ptr(), stream)),
                  ("soft_shadow_100_rays_quarter_frame_ms", m4 * np4, lambda: sc4.soft_shadow_dev(nr4.data_ptr(), rays4.data_ptr(), hits4.data_ptr(), m4, light4, np4, contrib4.data_ptr(), fac4.data_ptr(), stream)))
        c4 = {"workload": "C4: closed room + inner box, %d triangles, 1920x1080: primary closest hit, 16 AO rays (radius 2) + 1 hard-shadow ray per pixel (any hit), 100 soft-shadow rays for a quarter of the pixels (closest hit); secondary rays built inside the query kernels" % len(tr4)}
        a4, b4 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for name, n_rays, call in stages:
            call()
            torch.cuda.synchronize()
            a4.record()
            for _ in range(5):
                call()
            b4.record()
            torch.cuda.synchronize()
            ms = a4.elapsed_time(b4) / 5
            c4[name] = round(ms, 3)
            c4[name.replace("_ms", "_Mrays_per_s")] = round(n_rays / ms / 1e3, 1)
        c4["frame_primary_ao_hard_shadow_ms"] = round(c4["primary_ms"] + c4["ao_16spp_ms"] + c4["hard_shadow_ms"], 3)
        c4["frame_Mrays_per_s"] = round(n4 * (spp4 + 2) / c4["frame_primary_ao_hard_shadow_ms"] / 1e3, 1)
        frame["c4_room"] = c4

    if rank == 0:
        line = {
            "metric": "closest-hit Mrays/s on 1M-tri scene; form-factor+solve time @64K patches",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (triangle test) + f32 (box tests)",
            "data": "synthetic" if backend == "nccl" else "synthetic (REHEARSAL over %s, ranks share GPUs: not a measurement)" % backend,
            "config": {"workload": "C3: %d-triangle random soup, %dx%d primary rays x %d spp (1 spp per GPU), closest hit" % (
                           args.tris, x_res, y_res, spp),
                       "rays_per_step": rays_total, "tiles": "%dx%d round-robin over ranks" % TILE,
                       "bvh": {"nodes": int(info.n_nodes), "leaves": int(info.n_leaves), "depth": int(info.max_depth),
                               "max_leaf": int(info.max_leaf_size), "build_seconds": round(info.build_seconds, 2),
                               "device_MB": round(info.device_bytes / 1e6, 1)}},
            "roofline": roofline, "roofline_traversal": roofline_traversal, "roofline_formfactor": roofline_formfactor,
            "roofline_note": "`roofline` is k_jacobi_sweep, the HBM-bound kernel of the metric's second half; the kernel `value` times is in roofline_traversal (bound: L2), the one that dominates form-factor + solve time in roofline_formfactor (bound: fp64 vector ALU)",
            "tile_launch": tile_launch, "strong_scaling_one_frame": strong,
            "solver": solver, "formfactor": formfactor,
            "cpu_baseline": cpu, "cpu_baseline_radiosity": cpu_rad, "reference_so": have_ref_so, "c1_reference": c1, "frame": frame,
        }
        print(json.dumps(line), flush=True)
    multi.close_library_comms()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
