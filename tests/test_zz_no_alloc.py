"""rtmi_radiosity_solve_noalloc_dev — the solve for callers that cannot allocate (include/rtmi.h): its scratch is a
static device array inside librtmi.so.  Checked against rtmi_radiosity_solve_dev (same kernels): B must be equal bit
for bit, on two calls in a row, the second with a smaller input (the array keeps the first call's contents); device
memory in use does not move across the call; its fixed-sweep form is captured into a HIP graph and replayed.
Collected last (file name) so that a failure here hides no other test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    import raytracer_amd
    return raytracer_amd


def system(n, seed):
    rng = np.random.default_rng(seed)
    F = rng.random((n, n)).astype(np.float32)
    np.fill_diagonal(F, 0)
    F *= (0.9 / F.sum(1))[:, None].astype(np.float32)
    E = (rng.random((n, 3)) * 100).astype(np.float32)
    ld = (n + 3) // 4 * 4
    Fp = np.zeros((n, ld), np.float32)
    Fp[:, :n] = F
    return Fp, E, ld


def solve_both(rt, n, seed, tol, max_sweeps, check_every):
    import torch
    Fp, E, ld = system(n, seed)
    rho = np.array([0.84, 0.37, 0.7])
    dF = torch.from_numpy(Fp).cuda()
    dE = torch.from_numpy(E).cuda()
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for noalloc in (False, True):
        dB = torch.full_like(dE, float("nan"))
        stats = rt.api.radiosity_solve_dev(dF.data_ptr(), ld, n, rho, dE.data_ptr(), dB.data_ptr(), tol, max_sweeps, check_every, st,
                                           noalloc=noalloc)
        torch.cuda.synchronize()
        out.append((dB.cpu().numpy(), stats))
    return out


def test_noalloc_solve_equals_the_allocating_solve_twice_in_a_row(rt):
    # first call: 1000 patches; second call: 257 patches (smaller: the scratch array still holds the first call's vector)
    for n, seed in ((1000, 5), (257, 6)):
        # a fixed number of sweeps (tolerance 0): both entry points run the very same launches -> identical bits
        (B0, s0), (B1, s1) = solve_both(rt, n, seed, 0.0, 23, 4)
        assert s0.sweeps == s1.sweeps == 23
        assert np.isfinite(B0).all() and np.array_equal(B0.view(np.uint32), B1.view(np.uint32))
        assert np.isnan(s0.rel_change) and np.isnan(s1.rel_change)       # nothing is read back in this form
        # to tolerance: same stopping sweep, same B
        (B0, s0), (B1, s1) = solve_both(rt, n, seed, 1e-6, 500, 3)
        assert s0.sweeps == s1.sweeps and 5 < s1.sweeps < 500 and s1.rel_change <= 1e-6
        assert np.array_equal(B0.view(np.uint32), B1.view(np.uint32))


def test_noalloc_solve_allocates_nothing_and_can_be_captured_into_a_graph(rt):
    """(a) device memory in use does not move across calls of the entry (hipMemGetInfo before and after, tolerance form and
    fixed form); (b) the fixed form (tol_rel = 0: exactly max_sweeps sweeps, nothing read back) recorded into a HIP graph
    (torch.cuda.graph = hipStreamBeginCapture) does not run at capture, and replayed — twice — writes the allocating entry's
    bits; (c) the tolerance form is refused on a capturing stream (it would have to wait for the norms)."""
    import torch
    n = 1200
    Fp, E, ld = system(n, 11)
    rho = np.array([0.84, 0.37, 0.7])
    dF, dE = torch.from_numpy(Fp).cuda(), torch.from_numpy(E).cuda()
    ref = torch.zeros_like(dE); dB = torch.zeros_like(dE)
    st = torch.cuda.current_stream().cuda_stream
    rt.api.radiosity_solve_dev(dF.data_ptr(), ld, n, rho, dE.data_ptr(), ref.data_ptr(), 0.0, 17, 4, st)
    rt.api.radiosity_solve_dev(dF.data_ptr(), ld, n, rho, dE.data_ptr(), dB.data_ptr(), 0.0, 17, 4, st, noalloc=True)   # (first use: looks up the array's address)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for tol, mx in ((0.0, 17), (1e-6, 400)):
        rt.api.radiosity_solve_dev(dF.data_ptr(), ld, n, rho, dE.data_ptr(), dB.data_ptr(), tol, mx, 4, st, noalloc=True)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    out = torch.zeros_like(dE)
    stats = rt.api.rtmi_solve_stats()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        cs = torch.cuda.current_stream().cuda_stream
        rc = rt.lib.rtmi_radiosity_solve_noalloc_dev(dF.data_ptr(), ld, n, rho.ctypes.data, dE.data_ptr(), out.data_ptr(), 0.0, 17, 4, rt.api.C.byref(stats), cs)
        rc_tol = rt.lib.rtmi_radiosity_solve_noalloc_dev(dF.data_ptr(), ld, n, rho.ctypes.data, dE.data_ptr(), out.data_ptr(), 1e-6, 400, 4, None, cs)
        msg = rt.lib.rtmi_last_error()
    assert rc == 0 and stats.sweeps == 17
    assert rc_tol == -1 and b"captured" in msg
    assert float(out.abs().sum()) == 0.0                           # captured, not run
    for _ in range(2):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32))


def test_noalloc_solve_rejects_sizes_beyond_its_static_array(rt):
    import torch
    x = torch.zeros(64, device="cuda")
    rho = np.array([0.7, 0.7, 0.7])
    p = x.data_ptr()
    with pytest.raises(rt.RtmiError, match="RTMI_NOALLOC_MAX_N"):
        rt.api.radiosity_solve_dev(p, rt.api.NOALLOC_MAX_N + 4, rt.api.NOALLOC_MAX_N + 1, rho, p, p, 1e-4, 10, 1, 0, noalloc=True)
    with pytest.raises(rt.RtmiError, match="RTMI_NOALLOC_MAX_CHECK_EVERY"):
        rt.api.radiosity_solve_dev(p, 4, 4, rho, p, p, 1e-4, 10, rt.api.NOALLOC_MAX_CHECK_EVERY + 1, 0, noalloc=True)
    # the allocating entry refuses a partial row range without an exchange callback (it would iterate on rows nobody fills)
    stats = rt.api.rtmi_solve_stats()
    rc = rt.lib.rtmi_radiosity_solve_dev(p, 4, 4, 0, 2, rho.ctypes.data, p, p, 1e-4, 10, 1, None, None, rt.api.C.byref(stats), 0)
    assert rc != 0 and b"exchange" in rt.lib.rtmi_last_error()
