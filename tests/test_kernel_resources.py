"""Register budget of the step kernels, read from the compiler's own metadata (no GPU needed: hipcc cross-compiles gfx950).

VERDICT r2 item 7: every non-rollout specialisation spilled scalars (14-61 SGPR spills with one or two links per thread,
318-355 in the strided kernels, 20-132 bytes of scratch).  The cause was not the argument block but LLVM's loop vectoriser
widening the reward rules' rare search loops into hundreds of instructions; with those loops marked cold and the strided
kernels' per-link bodies no longer unrolled, the power-law and table kernels spill nothing and use no scratch."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope='module')
def kernels(tmp_path_factory):
    from gym_d2d_amd import build
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not Path(hipcc).exists():
        pytest.skip('hipcc missing')
    tmp = tmp_path_factory.mktemp('isa')
    cmd = [hipcc, *build.FLAGS, '-I', str(build.INCLUDE), '-c', str(build.CSRC / 'd2d_step.hip'), '-save-temps', '-o', 'step.o']
    r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    asm = next(tmp.glob('*gfx950*.s')).read_text()
    out = {}
    for blk in re.split(r'\n  - ', asm[asm.find('amdhsa.kernels'):]):
        name = re.search(r'\.name:\s+(\S+)', blk)
        m = name and re.match(r'_ZN3d2d11step_kernelILi(\d)ELi(\d)ELb([01])ELi(\d)ELi(\d+)EEEvNS_8StepArgsE', name.group(1))
        if not m:
            continue
        field = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
        out[tuple(int(x) for x in m.groups())] = {'vgpr': field('vgpr_count'), 'sgpr': field('sgpr_count'),
                                                  'sgpr_spills': field('sgpr_spill_count'), 'vgpr_spills': field('vgpr_spill_count'),
                                                  'scratch': field('private_segment_fixed_size')}
    assert len(out) >= 30, sorted(out)
    return out


def test_no_kernel_uses_scratch_or_spills_vector_registers(kernels):
    for key, k in kernels.items():
        assert k['scratch'] == 0 and k['vgpr_spills'] == 0, (key, k)


def test_power_law_and_table_kernels_spill_no_scalars(kernels):
    """<MODE in {inverse-square, power law, table}, LPT in {1, 2}, *, *, *>: the kernels behind the single-env D2DEnv, the
    Shannon / CueSinrShannon rewards, the table route and the member lists."""
    checked = 0
    for (mode, lpt, full, hot, opt), k in kernels.items():
        if mode in (0, 1, 2, 4) and lpt in (1, 2):
            # (PL_POWK, two links per thread sharing a workgroup with other envs - the coldest of its generic variants - keeps up to
            # eight scalars in lanes: the uniform trip count and parity of k beside everything the general kernel holds)
            assert k['sgpr_spills'] <= (8 if (mode, lpt, full) == (4, 2, 0) else 0), ((mode, lpt, full, hot, opt), k)
            checked += 1
    assert checked >= 24


def test_strided_and_shadowing_kernels_stay_within_a_few_lane_spills(kernels):
    """The strided kernels (threads per env forced below the link count) and the shadowing kernels (a Philox-driven Gaussian
    per pair, inlined at every pair evaluation) keep a few scalars in VGPR lanes - never in memory."""
    for (mode, lpt, full, hot, opt), k in kernels.items():
        if mode == 3:
            # (round 6: the Box-Muller behind every shadowed pair grew a log1p arm and an exact-quadrant cosine; the strided
            # exact-position variant - the coldest kernel in the library - keeps 93 scalars in lanes, still none in memory)
            assert k['sgpr_spills'] < (100 if opt & 16 else 80), ((mode, lpt, full, hot, opt), k)
        elif lpt == 0:
            # (PL_POWK's strided variants hold the eight tests on k as scalar masks: a few lanes more)
            assert k['sgpr_spills'] < (44 if mode == 4 else 32), ((mode, lpt, full, hot, opt), k)


@pytest.fixture(scope='module')
def rollout_kernels(tmp_path_factory):
    from gym_d2d_amd import build
    tmp = tmp_path_factory.mktemp('isa_rollout')
    cmd = [build._hipcc(), *build.FLAGS, '-I', str(build.INCLUDE), '-c', str(build.CSRC / 'd2d_rollout.hip'), '-save-temps', '-o', 'ro.o']
    r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    asm = next(tmp.glob('*gfx950*.s')).read_text()
    out = {}
    for blk in re.split(r'\n  - ', asm[asm.find('amdhsa.kernels'):]):
        name = re.search(r'\.name:\s+(\S+)', blk)
        m = name and re.match(r'_ZN3d2d14rollout_kernelILi(\d)ELi(\d+)ELi(\d)EEEvNS_8StepArgsE', name.group(1))
        if m:
            field = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
            out[tuple(int(x) for x in m.groups())] = {'vgpr': field('vgpr_count'), 'sgpr_spills': field('sgpr_spill_count'),
                                                      'vgpr_spills': field('vgpr_spill_count'), 'scratch': field('private_segment_fixed_size'),
                                                      'static_lds': field('group_segment_fixed_size')}
    # 3 path-loss modes (1 / d^2, general power law, one-integer-k power law) x (4 option sets x 2 links per thread + 2 padded
    # variants + 6 exact-position variants: option bit 16)
    assert len(out) == 48, sorted(out)
    return out


def test_rollout_kernel_of_round5_keeps_full_occupancy_and_no_static_lds(rollout_kernels):
    """csrc/d2d_rollout.hip, <path-loss mode, options, links per thread>: nothing spills; with scalar records (option bit 2: what
    BASELINE configs 3 - 5 run) one and two links per thread stay within 64 VGPRs = 8 waves per SIMD; no static LDS in front of
    the dynamic block (the kernel addresses LDS by raw byte offsets)."""
    for key, k in rollout_kernels.items():
        assert k['scratch'] == 0 and k['vgpr_spills'] == 0 and k['sgpr_spills'] == 0 and k['static_lds'] == 0, (key, k)
        if key[1] & 16:                                  # exact positions (d2d_set_positions_f64): a fourth 16-byte row per link
            assert k['vgpr'] <= 80, (key, k)             # 6 waves per SIMD
        elif key[1] & 10:                                # scalar records (2) or a padded link count (8: one link per thread)
            # (the one-integer-k power law with a padded link count or two links per thread: 66 - 69, seven waves per SIMD)
            assert k['vgpr'] <= (72 if key[0] == 4 and (key[1] & 8 or key[2] == 2) else 64), (key, k)


def test_rollout_kernel_keeps_full_occupancy(kernels):
    """8 waves per SIMD (four 512-thread workgroups per CU) needs <= 64 VGPRs; the scalar-record variant holds the records in
    SGPRs and frees vector registers."""
    for mode in (0, 1, 4):
        for opt in (0, 2, 4, 6):
            k = kernels[(mode, 1, 1, 1, opt)]
            assert k['vgpr'] <= 64 and k['sgpr_spills'] == 0, (mode, opt, k)
        assert kernels[(mode, 1, 1, 1, 2)]['vgpr'] <= kernels[(mode, 1, 1, 1, 0)]['vgpr']
