#!/usr/bin/env python3
"""What the hot loop of a kernel is made of, from the object file's disassembly (no GPU needed):
   python tools/loop_isa.py <object stem under csrc/build> <substring of the mangled kernel name> [--invariants]
prints, for the kernel's largest loop (the backward branch with the longest span): the instruction mix, the share of pure moves
(v_mov / v_accvgpr) in the VALU stream, scratch instructions (spill reloads: each is followed by an s_waitcnt vmcnt(0) that also
drains whatever prefetch is in flight), every s_waitcnt vmcnt(0), LDS shuffles (ds_bpermute / ds_swizzle) and DPP moves that were
not fused into their add, and -- with --invariants -- the vector registers the loop only reads (candidates for forming in place
instead of keeping live: DESIGN 8.2), each with the instruction that defined it.
Example: python tools/loop_isa.py vibo_msplit_a ILi2ELb1ELi0ELb0ELb1ELi0E --invariants"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
LLVM = '/opt/rocm/lib/llvm/bin'
NO_DST = ('ds_write', 'global_store', 'scratch_store', 'buffer_store', 'ds_add', 'v_cmp', 's_', 'v_readlane', 'v_readfirstlane', 'global_atomic')


def regs(tok):
    out = []
    for m in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        out += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
    return out


def main():
    stem, pat = sys.argv[1], sys.argv[2]
    obj = os.path.join(ROOT, 'variational-item-response-theory-public_amd', 'csrc', 'build', stem + '.o')
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'dev.co')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, obj], check=True)
        subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        '--input=' + fat, '--output=' + co, '--unbundle'], check=True)
        txt = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', co], check=True, capture_output=True, text=True).stdout
    m = re.search(r'<(\S*' + re.escape(pat) + r'\S*)>:\n(.*?)(?=\n[0-9a-f]+ <|\Z)', txt, re.S)
    if not m:
        sys.exit('no kernel matching ' + pat)
    lines = m.group(2).split('\n')
    addr = {}
    for i, l in enumerate(lines):
        mm = re.search(r'//\s*([0-9A-F]{12}):', l)
        if mm:
            addr[int(mm.group(1), 16)] = i
    best = None
    for i, l in enumerate(lines):
        if 's_cbranch' in l or 's_branch' in l:
            a = int(re.search(r'//\s*([0-9A-F]{12}):', l).group(1), 16)
            off = int(l.split()[1])
            off = off - 65536 if off > 32767 else off
            t = a + 4 + 4 * off
            if t < a and t in addr and (best is None or a - t > best[2]):
                best = (addr[t], i, a - t)
    if best is None:
        sys.exit('no loop found')
    lo, hi, _ = best
    body = [l.split('//')[0].strip() for l in lines[lo:hi + 1]]
    body = [b for b in body if b]
    ops = collections.Counter(b.split()[0] for b in body)
    valu = sum(v for k, v in ops.items() if k.startswith('v_'))
    moves = sum(v for k, v in ops.items() if k in ('v_mov_b32_e32', 'v_mov_b64_e32', 'v_accvgpr_read_b32', 'v_accvgpr_write_b32', 'v_mov_b32_dpp'))
    print(m.group(1))
    print(f'largest loop: {len(body)} instructions (of {len(lines)}), {valu} VALU, {moves} of them moves ({100.0 * moves / max(valu, 1):.0f} %), '
          f'{sum(v for k, v in ops.items() if k.startswith("v_mfma"))} MFMA, {sum(v for k, v in ops.items() if k.startswith("ds_"))} LDS, '
          f'{sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_")))} global / buffer, {ops["s_waitcnt"]} s_waitcnt, {ops["s_nop"]} s_nop')
    print('scratch instructions:', sum(v for k, v in ops.items() if k.startswith('scratch_')),
          '| s_waitcnt vmcnt(0):', sum(1 for b in body if b.startswith('s_waitcnt') and 'vmcnt(0)' in b),
          '| ds_bpermute / ds_swizzle:', ops['ds_bpermute_b32'] + ops['ds_swizzle_b32'],
          '| v_mov_b32_dpp (unfused):', ops['v_mov_b32_dpp'], '| fused *_dpp:', sum(v for k, v in ops.items() if k.endswith('_dpp') and k != 'v_mov_b32_dpp'),
          '| v_permlane*_swap:', ops['v_permlane16_swap_b32_e32'] + ops['v_permlane32_swap_b32_e32'])
    print('mix:', ', '.join(f'{k} {v}' for k, v in ops.most_common(24)))
    if '--invariants' in sys.argv:
        written, read = set(), {}
        for idx, b in enumerate(body):
            parts = b.split(None, 1)
            if len(parts) < 2:
                continue
            op, toks = parts[0], [t.strip() for t in parts[1].split(',')]
            srcs = toks if op.startswith(NO_DST) else toks[1:]
            if not op.startswith(NO_DST):
                written.update(regs(toks[0]))
            for t in srcs:
                for r in regs(t):
                    read.setdefault(r, []).append(op)
        inv = sorted(set(read) - written)
        print(f'{len(inv)} vector registers the loop only reads:')
        pre = [l.split('//')[0].strip() for l in lines[:lo]]
        for r in inv:
            d = next((b for b in reversed(pre) if b and not b.startswith(NO_DST) and len(b.split(None, 1)) > 1
                      and r in regs(b.split(None, 1)[1].split(',')[0])), '?')
            print(f'  v{r}: {len(read[r])} reads ({", ".join(sorted(set(read[r]))[:3])}) <- {d[:70]}')


if __name__ == '__main__':
    main()
