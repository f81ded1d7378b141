"""The planner's kernel choice (csrc/vibo_capi.hip: want_msplit / want_narrow, hand-written thresholds) against the committed
calibration tables (profiles/*planner_calibration*.txt, written by tools/calibrate_planner.py on two MI355X boxes: hipGraph
replays of both row-split kernels over persons x items x ability_dim).  No GPU needed: vibo_plan_kernel reads the descriptor.

    python tools/check_planner_table.py            # prints every row where the choice was measured slower, exits 1 if any
                                                   # row loses more than --tol (default 8 %) on EVERY box that measured it

The narrow-row kernel (round 5) is newer than the tables: rows it takes (<= 128 items, ability_dim <= 4) are listed as such
and not judged (its own A/B against the VALU kernel: profiles/r05_other_paths_kernel_trace.txt).
"""
import argparse
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'variational-item-response-theory-public_amd'))

from vibo_amd import _lib  # noqa: E402


def plan(B, I, A):
    lib = _lib.load()
    d = _lib.ViboDesc()
    d.abi_version = _lib.ABI_VERSION
    d.num_person, d.num_item, d.ability_dim, d.irt_model = B, I, A, 2
    d.mask_dtype, d.want_grad, d.flags = _lib.MASK_U8, 1, 0
    d.response_row_stride = d.mask_row_stride = (I + 3) & ~3
    return {1: 'matrix', 2: 'valu', 6: 'narrow'}.get(lib.vibo_plan_kernel(ctypes.byref(d)), 'other')


def read_tables():
    rows = {}
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*planner_calibration*.txt'))):
        for ln in open(path):
            t = ln.split()
            if len(t) >= 6 and t[0].isdigit():
                key = (int(t[0]), int(t[1]), int(t[2]))
                rows.setdefault(key, []).append((float(t[3]), float(t[4]), os.path.basename(path)))
    return rows


def check(tol=0.08, verbose=True):
    bad = []
    rows = read_tables()
    for (B, I, A), meas in sorted(rows.items()):
        choice = plan(B, I, A)
        if choice not in ('matrix', 'valu'):
            if verbose:
                print(f'{B:8d} {I:6d} {A:2d}  planner: {choice} (not in the tables)')
            continue
        losses = []
        for valu_us, matrix_us, name in meas:
            mine, other = (matrix_us, valu_us) if choice == 'matrix' else (valu_us, matrix_us)
            losses.append(mine / other - 1.0)
        if min(losses) > tol:
            bad.append((B, I, A, choice, losses))
        if verbose and max(losses) > 0.0:
            print(f'{B:8d} {I:6d} {A:2d}  planner: {choice:6s} slower by ' + ', '.join(f'{100 * x:+.0f} %' for x in losses)
                  + ('   <-- on every box' if min(losses) > tol else ''))
    return bad, len(rows)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--tol', type=float, default=0.08)
    a = ap.parse_args()
    bad, n = check(a.tol)
    print(f'{n} calibrated shapes, {len(bad)} where the planner picks the kernel that was slower by more than {100 * a.tol:.0f} % on every box')
    sys.exit(1 if bad else 0)
