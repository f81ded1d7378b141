#!/usr/bin/env python3
"""Phase timing of the matrix row-split kernel (development build: make -C .../csrc TIMING=1).
   python tools/ms_timing.py [profile_kernel.py args]  -> mean shader-clock cycles per batch and phase, per wave"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.argv = ['profile_kernel.py'] + sys.argv[1:]
exec(open(os.path.join(ROOT, 'tools', 'profile_kernel.py')).read())
from vibo_amd import _lib
lib = _lib.load() if hasattr(_lib, 'load') else ctypes.CDLL(os.path.join(ROOT, 'variational-item-response-theory-public_amd', 'vibo_amd', 'libvibo_hip.so'))
n = 1024 * 8 * 16
buf = (ctypes.c_longlong * n)()
fn = lib.vibo_debug_ms_timing_c if '--codes' in sys.argv else lib.vibo_debug_ms_timing      # (one buffer per translation unit)
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = fn(buf, n)
raw = np.frombuffer(buf, dtype=np.int64).reshape(1024, 8, 16)[:256]
t = raw[:, :, :12].astype(np.float64)
nb = (P + 31) // 32 / 256
names = ['tiles u0', 'pack0', 'tiles u1', 'pack1', 'counts+gth', 'barrier A', 'forward', 'barrier B', 'read ops', 'backward', 'wait loads x2', 'issue loads x2']
print('rc', rc, 'batches per workgroup %.1f' % nb)
print('phase           ' + ''.join(f' wave{w:1d}  ' for w in range(8)) + '   (cycles per batch, mean over workgroups)')
for k, nm in enumerate(names):
    print(f'{nm:14s}' + ''.join(f'{t[:, w, k].mean() / nb:8.0f}' for w in range(8)))
print(f'{"total":14s}' + ''.join(f'{t[:, w, :].sum(axis=1).mean() / nb:8.0f}' for w in range(8)))

# fixed cost of a launch: kernel prologue / epilogue in shader cycles, and the launch's wall-clock span (100 MHz real-time counter)
print('prologue cycles (mean over workgroups, per wave):', ' '.join(f'{raw[:, w, 12].mean():8.0f}' for w in range(8)))
print('epilogue cycles (mean over workgroups, per wave):', ' '.join(f'{raw[:, w, 13].mean():8.0f}' for w in range(8)))
ent, ext = raw[:, :, 14].astype(np.float64), raw[:, :, 15].astype(np.float64)
live = ent > 0
t0 = ent[live].min()
print(f'wall clock (us from the first wave\'s entry): entries {((ent[live] - t0) / 100).min():.2f} .. {((ent[live] - t0) / 100).max():.2f}, '
      f'exits {((ext[live] - t0) / 100).min():.2f} .. {((ext[live] - t0) / 100).max():.2f}; loop cycles per workgroup (wave 0): '
      f'mean {t[:, 0, :].sum(axis=1).mean():.0f}, min {t[:, 0, :].sum(axis=1).min():.0f}, max {t[:, 0, :].sum(axis=1).max():.0f}')
