"""Worker of tests/test_gpu_solver.py::test_library_loop_with_two_ranks_sharing_the_gpu (started by torch.distributed.run, 2 ranks,
gloo).  Each rank owns a block of rows of F on the SAME GPU and runs the library's solve loop with an exchange callback that
carries B and the norms through gloo (host staging): a rehearsal of the multi-GPU loop on a one-GPU box, never a measurement."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as rt          # noqa: E402
from raytracer_amd import multi     # noqa: E402
from test_gpu_solver import system, direct   # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
n = 1001                                            # not a multiple of the rank count: the last block is short
F, Fp, E, ld = system(n, 12)
rho = np.array([0.84, 0.37, 0.7])
row0, n_rows, blk = multi.rows_of_rank(n, rank, world)
dF_all, dE = torch.from_numpy(Fp).cuda(), torch.from_numpy(E).cuda()
dF_mine = dF_all[row0:row0 + n_rows].contiguous()
st = torch.cuda.current_stream().cuda_stream


_hip = rt.api.C.CDLL("libamdhip64.so")
_hip.hipMemcpy.argtypes = [rt.api.C.c_void_p, rt.api.C.c_void_p, rt.api.C.c_size_t, rt.api.C.c_int]


def memcpy(dst, src, nbytes, kind):          # kind: 1 = host to device, 2 = device to host (blocking)
    if _hip.hipMemcpy(dst, src, nbytes, kind) != 0:
        raise RuntimeError("hipMemcpy failed")


def exchange(user, d_B, d_norms, stream):
    """all-gather the ranks' row blocks of B (n x 3 floats at d_B) and reduce the norm triple, through the host"""
    try:
        torch.cuda.synchronize()
        hB = torch.empty(3 * n, dtype=torch.float32)
        hN = torch.empty(3, dtype=torch.float64)
        memcpy(hB.data_ptr(), d_B, 12 * n, 2)
        memcpy(hN.data_ptr(), d_norms, 24, 2)
        parts = [torch.zeros(3 * blk, dtype=torch.float32) for _ in range(world)]
        mine = torch.zeros(3 * blk, dtype=torch.float32)
        mine[: 3 * n_rows] = hB[3 * row0: 3 * (row0 + n_rows)]
        dist.all_gather(parts, mine)
        hB.copy_(torch.cat(parts)[: 3 * n])
        s, m = hN[:2].clone(), hN[2:].clone()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        hN[:2], hN[2:] = s, m
        memcpy(d_B, hB.data_ptr(), 12 * n, 1)
        memcpy(d_norms, hN.data_ptr(), 24, 1)
        return 0
    except Exception as e:      # never let an exception cross the C boundary
        print("exchange failed:", e, file=sys.stderr, flush=True)
        return 1


cb = rt.api.EXCHANGE_FN(exchange)


def solve_sharded():
    dB = torch.zeros_like(dE)
    stats = rt.api.rtmi_solve_stats()
    rc = rt.lib.rtmi_radiosity_solve_dev(dF_mine.data_ptr(), ld, n, row0, n_rows, rho.ctypes.data, dE.data_ptr(), dB.data_ptr(), 1e-7, 500, 3, cb, None,
                                         rt.api.C.byref(stats), st)
    return rc, dB, stats


ref = torch.zeros_like(dE)
s0 = rt.api.radiosity_solve_dev(dF_all.data_ptr(), ld, n, rho, dE.data_ptr(), ref.data_ptr(), 1e-7, 500, 3, st)
rc, dB, s1 = solve_sharded()
assert rc == 0, rt.lib.rtmi_last_error().decode()
want = direct(F, rho, E)
res = {"rank": rank, "sweeps": int(s1.sweeps), "sweeps_single": int(s0.sweeps), "accelerated": int(s1.accelerated_sweeps),
       "equal_bits": bool(np.array_equal(dB.cpu().numpy().view(np.uint32), ref.cpu().numpy().view(np.uint32))),
       "rel_err_vs_direct": float(np.linalg.norm(dB.cpu().numpy() - want) / np.linalg.norm(want))}
# one rank fails its set-up: every rank must come back with -1
dist.barrier()
if rank == 1:
    os.environ["RTMI_DEBUG_FAIL_SOLVE_SETUP"] = "1"
t0 = time.perf_counter()
rc, _, _ = solve_sharded()
res["fail_rc"], res["fail_seconds"], res["fail_msg"] = int(rc), time.perf_counter() - t0, rt.lib.rtmi_last_error().decode()
os.environ.pop("RTMI_DEBUG_FAIL_SOLVE_SETUP", None)
dist.barrier()
rc, dB, _ = solve_sharded()
res["after_failure_equal_bits"] = bool(rc == 0 and np.array_equal(dB.cpu().numpy().view(np.uint32), ref.cpu().numpy().view(np.uint32)))
json.dump(res, open(os.path.join(os.environ["RTMI_MP_OUT"], "rank%d.json" % rank), "w"))
dist.barrier()
dist.destroy_process_group()
