#!/usr/bin/env python3
"""Compile one csrc/*.hip for gfx950 with the library's flags and look at the code the compiler made (no GPU needed).

    python tools/isa.py d2d_rollout.hip                      # every kernel: VGPR / SGPR / spills / LDS / static instruction mix
    python tools/isa.py d2d_rollout.hip 'rollout_kernelILi0ELi2' --dump      # one kernel's instructions (comments stripped)

The static mix counts every instruction of the kernel, cold arms included; the per-wave dynamic counts come from the PMC
passes (tools/pmc_step.sh)."""
import argparse
import re
import subprocess
import sys
import tempfile
from collections import Counter
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def classify(op):
    if op.startswith('v_'):
        return 'valu'
    if op.startswith(('s_load', 's_buffer_load')):
        return 'smem'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    return op


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('source')
    ap.add_argument('kernel', nargs='?', help='substring of the mangled kernel name')
    ap.add_argument('--dump', action='store_true')
    ap.add_argument('--defines', default='')
    a = ap.parse_args()
    from gym_d2d_amd import build
    src = build.CSRC / a.source
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [build._hipcc(), *build.FLAGS, *a.defines.split(), '-I', str(build.INCLUDE), '-c', str(src), '-save-temps', '-o', 'x.o']
        r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-4000:])
        asm = next(Path(tmp).glob('*gfx950*.s')).read_text()
    meta = {}
    for blk in re.split(r'\n  - ', asm[asm.find('amdhsa.kernels'):]):
        name = re.search(r'\.name:\s+(\S+)', blk)
        if name:
            f = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
            meta[name.group(1)] = dict(vgpr=f('vgpr_count'), sgpr=f('sgpr_count'), sgpr_spills=f('sgpr_spill_count'),
                                       vgpr_spills=f('vgpr_spill_count'), scratch=f('private_segment_fixed_size'))
    for name, m in meta.items():
        if a.kernel and a.kernel not in name:
            continue
        i = asm.index(name + ':')
        body = asm[i:asm.index('.Lfunc_end', i)]
        lines = [l.strip() for l in body.splitlines()]
        ins = [l for l in lines if l and not l.startswith(('.', ';', '_')) and not l.endswith(':') and not re.match(r'^\.?L?BB\d+_\d+:', l)]
        mix = Counter(classify(l.split()[0]) for l in ins)
        print(name, m, dict(mix))
        if a.dump:
            for l in body.splitlines():
                t = l.split(';')[0].rstrip()
                if t.strip() and not t.strip().startswith(('.p2align', '.loc', '.cfi')):
                    print(t)


if __name__ == '__main__':
    main()
