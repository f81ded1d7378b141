#!/usr/bin/env python3
"""Check the planner's kernel choice (want_msplit in csrc/vibo_capi.hip: matrix row-split kernel vs VALU row-split kernel)
against this box: time both kernels (vibo_desc.flags pins them) over a grid of persons x items x ability_dim, print the
measured crossover and every shape where the planner's default is the slower one by more than 5 %.
   python tools/calibrate_planner.py [--grad 1] [--codes] > profiles/rNN_planner_calibration.txt"""
import argparse, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec

ap = argparse.ArgumentParser()
ap.add_argument('--codes', action='store_true', help='rows as 1-byte cell codes')
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--cond', action='store_true', help='conditional posterior: matrix-pipe passes (VIBO_FLAG_COND_MATRIX) vs the VALU passes (VIBO_FLAG_COND_VALU), and what the planner picks')
a = ap.parse_args()
d = torch.device('cuda:0')
g = torch.Generator(device=d).manual_seed(0)


def time_call(fn):
    """us per call, replayed from a hipGraph (how the CLI and bench.py launch the step: the per-launch cost of an eager call
    would hide the kernels -- an eager call of either kernel takes 23-28 us up to 4 096 persons)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(a.iters):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * a.iters) * 1e3          # us


if a.cond:
    print(f'# conditional posterior, 2PL, 10 % missing, forward + backward, hipGraph replays; rows: {"cell codes" if a.codes else "fp32 + mask"}')
    print(f'{"persons":>8s} {"items":>6s} {"A":>2s} {"valu us":>9s} {"matrix us":>10s}  planner')
    lib = _lib.load()
    for A in (1, 2, 3, 4, 8):
        for I in (100, 1000):
            for P in (16, 256, 1024, 4096, 8192, 16384, 65536, 262144):
                spec = ElboSpec(irt_model=2, ability_dim=A, conditional=True)
                resp = (torch.rand(P, I, device=d, generator=g) < 0.5).float()
                mask = torch.rand(P, I, device=d, generator=g) >= 0.1
                table = torch.randn(2, I, 2 * A, device=d, generator=g) * 0.5
                item = torch.randn(I, A + 1, device=d, generator=g)
                eps = torch.randn(P, A, device=d, generator=g)
                if a.codes:
                    r, m, code = ops.prepare_rows(ops.pack_cell_codes(resp, mask), None)
                else:
                    r_, m_ = ops.pad_rows(resp, mask)
                    r, m, code = ops.prepare_rows(r_, m_)
                t = {}
                for name, fl in (('valu', _lib.FLAG_COND_VALU), ('default', _lib.FLAG_COND_MATRIX)):
                    ops.DESC_FLAGS = fl
                    t[name] = time_call(lambda: ops._hip_launch_elbo(spec, r, m, code, None, table, item, eps, None, _lib.REG_KL, True, P))
                ops.DESC_FLAGS = 0
                dd = ops._make_desc(spec, P, I, code, _lib.REG_KL, True, r.stride(0), m.stride(0))
                import ctypes
                pk = lib.vibo_plan_cond_passes(ctypes.byref(dd))
                best = 'matrix' if t['default'] < t['valu'] else 'valu'
                print(f'{P:8d} {I:6d} {A:2d} {t["valu"]:9.1f} {t["default"]:10.1f}  faster: {best:6s}  planner puts on the matrix pipe: {"pre " if pk & 1 else ""}{"post" if pk & 2 else ""}{" (first pass folded into the matrix kernel)" if pk & 4 else ""}{"nothing" if pk == 0 else ""}')
    sys.exit(0)
print(f'# device: {torch.cuda.get_device_name(0)}; rows: {"cell codes" if a.codes else "fp32 + mask"}; 2PL, 10 % missing, forward + backward')
print(f'{"persons":>8s} {"items":>6s} {"A":>2s} {"valu us":>9s} {"matrix us":>10s} {"faster":>7s} {"planner":>8s}  note')
bad = 0
for A in (1, 8):
    for I in (128, 256, 384, 512, 768, 1000):
        for P in (256, 1024, 2048, 4096, 16384, 65536):
            spec = ElboSpec(irt_model=2, ability_dim=A)
            resp = (torch.rand(P, I, device=d, generator=g) < 0.5).float()
            mask = torch.rand(P, I, device=d, generator=g) >= 0.1
            table = torch.randn(2, 2 * A, device=d, generator=g) * 0.5
            item = torch.randn(I, A + 1, device=d, generator=g)
            eps = torch.randn(P, A, device=d, generator=g)
            if a.codes:
                r, m, code = ops.prepare_rows(ops.pack_cell_codes(resp, mask), None)
            else:
                r = resp
                m, code = ops.prepare_mask(mask)
            t = {}
            for name, fl in (('valu', _lib.FLAG_KERNEL_VALU), ('matrix', _lib.FLAG_KERNEL_MATRIX)):
                ops.DESC_FLAGS = fl
                t[name] = time_call(lambda: ops._hip_launch_elbo(spec, r, m, code, None, table, item, eps, None, _lib.REG_KL, True, P))
            ops.DESC_FLAGS = 0
            pick = ops.plan_kernel(spec, P, I, code)
            pick = 'matrix' if pick.startswith('matrix') else 'valu'
            best = min(t, key=t.get)
            loss = t[pick] / t[best] - 1.0
            note = f'planner loses {100 * loss:.0f} %' if loss > 0.05 else ''
            bad += loss > 0.05
            print(f'{P:8d} {I:6d} {A:2d} {t["valu"]:9.1f} {t["matrix"]:10.1f} {best:>7s} {pick:>8s}  {note}')
print(f'# shapes where the planner is more than 5 % off the faster kernel: {bad}')
