"""One process per GPU: how the path shards across ranks.

  * ray queries: BVH and triangles replicated; image tiles dealt round-robin (rank r traces tiles
    r, r+W, r+2W, ...) — no collective, each rank writes its own pixels of the frame.
  * form factors: rank r owns rows rows_of_rank(n, r, W) of F — no collective.
  * whole frames: each rank renders its tiles into a zeroed full-size buffer; render_frame_sharded() assembles the image
    on every rank with ONE all-reduce (sum: every pixel is written by exactly one rank).
  * solve: F stays row-sharded; after each Jacobi sweep the ranks all-gather their rows of B and, at every convergence
    check, all-reduce the three norms.  In production that loop runs INSIDE the library (rtmi_radiosity_solve_sharded_dev:
    sweep kernel and RCCL all-gather enqueued back to back on one stream, the host waits only at the checks) —
    solve_sharded_library() below only hands it the communicator id; solve_sharded() is the same iteration driven from
    Python over torch.distributed (gloo in the CPU tests, a rehearsal path on the GPU box).

Nothing here computes: the sweep itself is the HIP kernel behind the C ABI.
"""
import ctypes as C
import math

_COMMS = {}


def library_comm(group=None):
    """The library's own RCCL communicator (rtmi_comm) over the ranks of `group`: rank 0 draws the id, torch.distributed
    carries it to the others, every rank joins on its current GPU.  One per (process, group); world size 1 needs no
    torch.distributed at all."""
    import torch.distributed as dist
    from . import api
    live = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if live else 1
    rank = dist.get_rank(group) if live else 0
    key = (id(group), world, rank)
    if key in _COMMS:
        return _COMMS[key]
    ident = (C.c_uint8 * 128)()
    box = [None]
    if rank == 0:      # a failure here (no RCCL to load) must reach the other ranks too, or they would wait for the id for ever
        box = [bytes(ident)] if api.lib.rtmi_comm_unique_id(ident) == 0 else [api.lib.rtmi_last_error().decode()]
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if not isinstance(box[0], bytes):
        raise api.RtmiError(box[0] or "rtmi_comm_unique_id failed")
    ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
    h = api.lib.rtmi_comm_create(ident, rank, world)
    if not h:
        raise api.RtmiError(api.lib.rtmi_last_error().decode())
    _COMMS[key] = (C.c_void_p(h), rank, world)
    return _COMMS[key]


def close_library_comms():
    """Destroys the communicators library_comm() made (call before torch.distributed is shut down)."""
    from . import api
    for comm, _, _ in _COMMS.values():
        api.lib.rtmi_comm_destroy(comm)
    _COMMS.clear()


def solve_sharded_library(d_F, ld, n, row0, n_rows, rho, E, tol_rel=1e-7, max_sweeps=1000, check_every=4, group=None, stream=None):
    """rtmi_radiosity_solve_sharded_dev: the whole sharded Jacobi loop inside the library, exchange over RCCL.  d_F: device
    pointer of this rank's first row; rows [row0, row0 + n_rows) must be rows_of_rank(n, rank, world).  Returns
    (B [n, 3] on the device, rtmi_solve_stats)."""
    import numpy as np
    import torch
    from . import api
    comm, rank, world = library_comm(group)
    rho = np.ascontiguousarray(rho, np.float64)
    B = torch.zeros((n, 3), dtype=torch.float32, device=E.device)
    stats = api.rtmi_solve_stats()
    st = torch.cuda.current_stream().cuda_stream if stream is None else stream
    api._check(api.lib.rtmi_radiosity_solve_sharded_dev(comm, d_F, ld, n, row0, n_rows, rho.ctypes.data, E.data_ptr(), B.data_ptr(), tol_rel,
                                                        max_sweeps, check_every, C.byref(stats), st))
    return B, stats


def rows_of_rank(n, rank, world):
    """Contiguous, equal-size row blocks (the last ranks may own fewer, or none): (row0, n_rows, block)."""
    blk = (n + world - 1) // world
    row0 = min(n, rank * blk)
    return row0, min(n, row0 + blk) - row0, blk


def tiles_of_rank(x_res, y_res, tile_w, tile_h, rank, world):
    """Tile numbers (row-major) rank traces: rank, rank+world, ... as rtmi_primary_closest_hit_dev deals them."""
    tiles = ((x_res + tile_w - 1) // tile_w) * ((y_res + tile_h - 1) // tile_h)
    return range(rank, tiles, world)


def pixels_of_rank(x_res, y_res, tile_w, tile_h, rank, world):
    tx = (x_res + tile_w - 1) // tile_w
    n = 0
    for t in tiles_of_rank(x_res, y_res, tile_w, tile_h, rank, world):
        x0, y0 = (t % tx) * tile_w, (t // tx) * tile_h
        n += (min(x_res, x0 + tile_w) - x0) * (min(y_res, y0 + tile_h) - y0)
    return n


class SolveResult:
    def __init__(self, B, sweeps, rel_change, max_change):
        self.B, self.sweeps, self.rel_change, self.max_change = B, sweeps, rel_change, max_change


def solve_sharded(sweep, E, n, tol_rel=1e-7, max_sweeps=1000, check_every=4, group=None, after_sweep=None):
    """Jacobi iteration B <- E + rho∘(F·B) with F row-sharded over the ranks of `group`.

    sweep(B_old, B_new, norms): computes THIS rank's rows of B_new (a full [n_pad, 3] tensor; other rows are
    ignored) from the full B_old and adds its rows' (Σd², Σ(B−E)², max|d|) to `norms` (float64[3], zeroed here).
    E: [n, 3] float32 tensor on the compute device.  Returns every row on every rank.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    row0, n_rows, blk = rows_of_rank(n, rank, world)
    n_pad = blk * world
    dev = E.device
    staged = world > 1 and E.is_cuda and dist.get_backend(group) == "gloo"
    B = torch.zeros((n_pad, 3), dtype=torch.float32, device=dev)
    B[:n] = E                                                  # B_0 = E, Render2/radiosity.cpp:506-509
    Bn = torch.zeros_like(B)
    norms = torch.zeros(3, dtype=torch.float64, device=dev)
    sweeps, rel, mx = 0, math.inf, math.inf
    while sweeps < max_sweeps:
        norms.zero_()
        sweep(B, Bn, norms)
        if world > 1:
            # every rank contributes its block of rows (blk·3 floats: 96 KiB at 64 K patches on 8 GPUs)
            mine = Bn[rank * blk:(rank + 1) * blk].reshape(-1).clone()
            if staged:   # rehearsal only: gloo cannot gather device tensors, go through the host
                parts = [torch.empty(blk * 3, dtype=torch.float32) for _ in range(world)]
                dist.all_gather(parts, mine.cpu(), group=group)
                Bn.view(-1).copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(Bn.view(-1), mine, group=group)
        B, Bn = Bn, B
        sweeps += 1
        if after_sweep is not None:
            after_sweep(sweeps, B)
        if sweeps % check_every == 0 or sweeps == max_sweeps:
            if world > 1:
                s = norms[:2].clone(); m = norms[2:].clone()
                if staged:
                    s, m = s.cpu(), m.cpu()
                dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
                dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
                h = torch.cat([s, m]).cpu()
            else:
                h = norms.cpu()
            rel = math.sqrt(h[0].item() / h[1].item()) if h[1].item() > 0 else 0.0
            mx = h[2].item()
            if rel <= tol_rel:
                break
    return SolveResult(B[:n], sweeps, rel, mx)


def assemble_frame(frame, group=None):
    """frame: a full-size tensor (bytes or doubles) holding THIS rank's tiles and zeros elsewhere.  Returns the whole
    image on every rank: every pixel belongs to exactly one rank, so a sum over ranks is the image."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return frame
    if frame.is_cuda and dist.get_backend(group) == "gloo":   # rehearsal only: stage through the host
        h = frame.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        frame.copy_(h)
    else:
        dist.all_reduce(frame, op=dist.ReduceOp.SUM, group=group)
    return frame


def render_frame_sharded(render, x_res, y_res, tile=(64, 64), dtype=None, device="cuda", group=None):
    """render(out, tile, tile_first, tile_stride) fills this rank's tiles of `out` (y_res x x_res x 3, zeroed here) —
    Shader.render_frame_dev / rtmi_radiosity_raster_dev with tile_first = rank, tile_stride = world in production.
    Returns the assembled frame on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    out = torch.zeros((y_res, x_res, 3), dtype=dtype or torch.uint8, device=device)
    render(out, tile, rank, world)
    return assemble_frame(out, group)
