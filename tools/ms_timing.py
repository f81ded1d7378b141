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
SLOTS = 32
n = 1024 * 8 * SLOTS
buf = (ctypes.c_longlong * n)()
fused = '--cond' in sys.argv and '--cond-three-pass' not in sys.argv and '--codes' not in sys.argv and '--flows' not in sys.argv and A == 1
fn = (lib.vibo_debug_ms_timing_xa if fused else lib.vibo_debug_ms_timing_fc if ('--codes' in sys.argv and '--flows' in sys.argv) else
      lib.vibo_debug_ms_timing_c if '--codes' in sys.argv else lib.vibo_debug_ms_timing)      # (one buffer per translation unit)
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = fn(buf, n)
raw = np.frombuffer(buf, dtype=np.int64).reshape(1024, 8, SLOTS)[:256]
t = raw[:, :, :12].astype(np.float64)
nb = (P + 31) // 32 / (256 // max(1, (I + 1023) // 1024) if I > 1024 else 256)
names = ['tiles u0', 'pack0', 'tiles u1', 'pack1', 'counts+gth', 'barrier A', 'forward', 'barrier B', 'read ops', 'backward', 'put_counts', 'put_stats']
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

# one-shot marks: cycles since the wave's entry (prologue) / since the end of the batch loop (end code), mean over workgroups
pn = ['rows+eps requested', 'table in LDS', 'item sample in', 'barrier 1', 'image written', 'barrier 2', 'M-tile 0 packed', 'M-tile 1 packed']
en = ['last backward', 'll -> barrier', 'scalars', 'table grads', 'item grads staged']
nbt = (P + 31) // 32
rem = nbt % 256
classes = [('all workgroups', slice(0, 256))] if rem == 0 else [(f'long workgroups (0..{rem - 1}: one batch more)', slice(0, rem)), (f'late workgroups ({rem}..255)', slice(rem, 256))]
for cname, sl in classes:
    print(f'prologue marks, {cname} (cumulative cycles)   ' + ''.join(f' wave{w:1d}  ' for w in range(8)))
    for k, nm in enumerate(pn):
        print(f'  {nm:22s}' + ''.join(f'{raw[sl, w, 16 + k].mean():8.0f}' for w in range(8)) + f'   max (wave 0): {raw[sl, 0, 16 + k].max():.0f}')
    ex = (ext[sl][live[sl]] - t0) / 100
    print(f'  exits {ex.min():.2f} .. {ex.max():.2f} us; loop cycles (wave 0) mean {t[sl, 0, :].sum(axis=1).mean():.0f}')
print('end-code marks (cumulative cycles)')
for k, nm in enumerate(en):
    print(f'  {nm:22s}' + ''.join(f'{raw[:, w, 24 + k].mean():8.0f}' for w in range(8)))
