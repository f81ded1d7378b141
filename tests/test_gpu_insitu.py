"""GPU tests of two pieces of round 6's measurement / safety plumbing:
 (1) the kernels' cross-row sums through v_permlane16_swap / v_permlane32_swap (csrc/vibo_device.hpp: xor16_add / xor32_add, inline
     asm with a hand-placed s_nop) against the __shfl_xor form they replace, lane by lane, bit for bit;
 (2) the in-situ launch timer (vibo_set_insitu_timer): the matrix row-split kernel stamps its own entry / exit; the counters count
     every launch -- eager and replayed from a hipGraph, where HIP events cannot be recorded --, agree with HIP events around
     eager launches, and switching the hook on changes no result bit.
"""
import ctypes

import pytest
import torch

from test_gpu_parity import dev
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec

pytestmark = pytest.mark.gpu


def test_lane_swap_sums_equal_the_shuffle_form():
    d = dev()
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(5)
    for trial in range(8):
        v = torch.randn(64, generator=g) * (10.0 ** (trial - 4))
        if trial == 7:
            v = torch.arange(64, dtype=torch.float32)          # (every lane distinguishable: a wrong partner shows)
        vin = v.to(d)
        out = torch.full((6, 64), float('nan'), device=d)
        stream = ctypes.c_void_p(torch.cuda.current_stream(d).cuda_stream)
        _lib.check(lib.vibo_selftest_lane_swaps(ops._ptr(vin), ops._ptr(out), stream), 'vibo_selftest_lane_swaps')
        torch.cuda.synchronize()
        o = out.cpu()
        lanes = torch.arange(64)
        assert torch.equal(o[2], v + v[lanes ^ 16]) and torch.equal(o[3], v + v[lanes ^ 32])      # the shuffle form is what it says
        assert torch.equal(o[0].view(torch.int32), o[2].view(torch.int32)), 'xor16_add != v + shfl_xor(v, 16)'
        assert torch.equal(o[1].view(torch.int32), o[3].view(torch.int32)), 'xor32_add != v + shfl_xor(v, 32)'
        assert torch.equal(o[4].view(torch.int32), o[5].view(torch.int32)), 'chained swap sums differ from the chained shuffles'


def _problem(P, I, A, d, seed=3):
    g = torch.Generator(device=d).manual_seed(seed)
    r = (torch.rand(P, I, device=d, generator=g) < 0.5).float()
    mk = (torch.rand(P, I, device=d, generator=g) >= 0.1)
    spec = ElboSpec(irt_model=2, ability_dim=A)
    table = torch.randn(2, 2 * A, device=d, generator=g) * 0.5
    item = torch.randn(I, A + 1, device=d, generator=g)
    eps = torch.randn(P, A, device=d, generator=g)
    r2, m8, code = ops.prepare_rows(r, mk)
    return spec, r2, m8, code, table, item, eps


def test_insitu_timer_counts_launches_and_matches_events():
    d = dev()
    P, I, A = 200_000, 1000, 8
    spec, r, m8, code, table, item, eps = _problem(P, I, A, d)
    assert ops.plan_kernel(spec, P, I, code, True).startswith('matrix')
    call = lambda: ops._hip_launch_elbo(spec, r, m8, code, None, table, item, eps, None, _lib.REG_KL, True, P)
    ref = call()                       # hook off
    torch.cuda.synchronize()
    tm = ops.InsituTimer(d)
    assert tm.read() == {'launches': 0}
    with tm:
        for _ in range(3):
            timed = call()
        torch.cuda.synchronize()
        # the hook changes no result bit
        assert torch.equal(ref.flat.view(torch.int32), timed.flat.view(torch.int32))
        assert torch.equal(ref.ability_mu.view(torch.int32), timed.ability_mu.view(torch.int32))
        tm.reset()
        n = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            call()
        e1.record()
        torch.cuda.synchronize()
        ev_ms = e0.elapsed_time(e1) / n         # kernel + finalize launch + gaps: an upper bound of the kernel
        got = tm.read()
        assert got['launches'] == n and not got['in_flight']
        assert 0.02 < got['min_ms'] <= got['mean_ms'] <= got['max_ms'] < 5.0
        assert got['mean_ms'] <= ev_ms * 1.02, (got, ev_ms)
        assert got['mean_ms'] >= ev_ms * 0.6, (got, ev_ms)          # (the finalize launch and the gaps are a small part of it)
        # replayed from a hipGraph: the captured launches carry the block's address
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            call()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            call()
        tm.reset()
        for _ in range(7):
            gr.replay()
        torch.cuda.synchronize()
        rep = tm.read()
        assert rep['launches'] == 7
        assert abs(rep['mean_ms'] - got['mean_ms']) < 0.25 * got['mean_ms']
    # hook off again: launches no longer count
    tm.reset()
    call()
    torch.cuda.synchronize()
    assert tm.read() == {'launches': 0}


def test_insitu_timer_counts_narrow_kernel_launches():
    """The narrow-row kernel (BASELINE configs[0] / [3] widths) carries the same stamps."""
    d = dev()
    P, I, A = 100_000, 96, 1
    spec, r, m8, code, table, item, eps = _problem(P, I, A, d, seed=5)
    assert ops.plan_kernel(spec, P, I, code, True).startswith('narrow')
    call = lambda: ops._hip_launch_elbo(spec, r, m8, code, None, table, item, eps, None, _lib.REG_KL, True, P)
    ref = call()
    torch.cuda.synchronize()
    tm = ops.InsituTimer(d)
    with tm:
        for _ in range(6):
            got = call()
        torch.cuda.synchronize()
        k = tm.read()
    assert k['launches'] == 6 and 0.002 < k['min_ms'] <= k['max_ms'] < 1.0
    assert torch.equal(ref.flat.view(torch.int32), got.flat.view(torch.int32))


def test_insitu_timer_inside_the_folded_train_step_graph():
    """The benchmark's use: the folded step captured as one hipGraph, the timer armed at capture time."""
    from vibo_amd.torch_core.models import VIBO_2PL
    from vibo_amd.trainer import FusedTrainer
    d = dev()
    P, I, A = 65_536, 1000, 8
    spec, r, m8, code, table, item, eps = _problem(P, I, A, d, seed=9)
    torch.manual_seed(1)
    model = VIBO_2PL(A, I, ability_merge='product').to(d)
    tr = FusedTrainer(model, lr=5e-3, rng='native', seed=7)
    tm = ops.InsituTimer(d)
    with tm:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                tr.step(r, m8)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            loss = tr.step(r, m8)
    tm.reset()
    for _ in range(12):
        gr.replay()
    torch.cuda.synchronize()
    got = tm.read()
    assert got['launches'] == 12 and torch.isfinite(loss)
    assert 0.01 < got['mean_ms'] < 1.0
