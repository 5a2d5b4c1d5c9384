"""ISA audit of the kernels that request vector memory from inline asm (hipcc does not know that the destination
registers are written later): on every control-flow path from such a request to the first hand-placed wait no
instruction may touch the destination registers (tools/audit_async_regs.py).  Cross-compiles on the CPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_group_record_kernels_never_touch_panels_in_flight(tmp_path):
    asm = str(tmp_path / 'spmm_rec.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', asm,
                    os.path.join(ROOT, 'ndcn_amd', 'csrc', 'spmm_rec.hip')], check=True, stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'audit_async_regs.py'), asm, 'spmm_rec_kernel'],
                         capture_output=True, text=True)
    assert out.returncode == 0 and 'TOTAL problems 0' in out.stdout, out.stdout[-2000:]
    assert out.stdout.count('asm loads') == 24            # three record shapes x halo x {plain, combine, error, rk4}
    text = open(asm).read()
    assert '.vgpr_spill_count: 0' in text and 'vgpr_spill_count:' in text
    assert all(l.strip().endswith(' 0') for l in text.split('\n') if '.vgpr_spill_count:' in l)


def text_of(asm, symbol):
    """the instruction lines of one kernel in a -S listing"""
    on = False
    for line in open(asm):
        if line.startswith(symbol + ':'):
            on = True
            continue
        if on and (line.startswith('\t.end_amdhsa_kernel') or '.Lfunc_end' in line):
            return
        if on and line.startswith('\t') and not line.startswith('\t.') and not line.strip().startswith(';'):
            yield line


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_fused3_producers_never_touch_panels_in_flight(tmp_path):
    """rhs_fused3.hip: the producer waves' row-local panels are requested a step ahead from inline asm and awaited by a
    run-time vmcnt count; no spill anywhere (a reload would queue behind the producers' requests)."""
    asm = str(tmp_path / 'rhs_fused3.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', asm,
                    os.path.join(ROOT, 'ndcn_amd', 'csrc', 'rhs_fused3.hip')], check=True, stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'audit_async_regs.py'), asm, 'rhs_fused3_kernel'],
                         capture_output=True, text=True)
    assert out.returncode == 0 and 'TOTAL problems 0' in out.stdout, out.stdout[-2000:]
    # halo x {plain, combine 0-5, error 1 | 5, rk4 0-3} + the two X + c Xadd variants + the 13 no-halo variants once more with
    # plain instead of non-temporal epilogue stores (panels that live in the Infinity Cache)
    # every instantiation: 41 of the inference path + 10 masked-input + 10 S-output (odeint_adjoint's halves) + 6 without the store of K
    # (RkOpt::no_k as a template argument: COMBINE with no earlier stage and RK4's fourth stage x {halo, nt, plain stores})
    assert out.stdout.count('asm loads') == 67
    # the dopri5 launches carry no trace of it: the <COMBINE, 4> variant is back at its size before the run-time test of a.K (2026)
    n_ins = sum(1 for l in text_of(asm, '_ZN4ndcn17rhs_fused3_kernelILb0ELi1ELi4ELi0ELb1ELb0ELb0EEEvNS_6F3ArgsENS_5F3EpiE'))
    assert n_ins <= 2040, n_ins
    text = open(asm).read()
    spills = [l for l in text.split('\n') if '.vgpr_spill_count:' in l]
    assert spills and all(l.strip().endswith(' 0') for l in spills)
