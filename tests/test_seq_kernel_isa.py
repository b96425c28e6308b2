"""sbm_iex_seq_kernel<M, rotated> fills its ring of step tables with LDS-direct loads issued from inline assembly and
waits for them with a hand-written `s_waitcnt vmcnt(2 (NRING - 1))` (csrc/sbm_implicit_extrap_seq.hpp).  That count is
right only while NOTHING ELSE in the loop touches vector memory: a compiler-generated global / scratch / buffer access
between the ring's loads (a spill, say, after some unrelated change) would make the wait return before the table it is
meant for has landed.  This test reads the ISA of the built stiff50 plugin -- the code object is unbundled from the
shared object and disassembled, about a second -- and checks exactly that.  No GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def _disassemble(plugin, tmp):
    fat = os.path.join(tmp, 'fat.bin')
    co = os.path.join(tmp, 'dev.co')
    subprocess.run(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', plugin, fat], check=True)
    subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--input=' + fat,
                    '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co, '--unbundle'], check=True)
    return subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', co], check=True, stdout=subprocess.PIPE,
                          text=True).stdout


def test_ring_loop_of_the_rotated_kernel_has_no_other_vector_memory_access(tmp_path):
    plugin = os.path.join(REPO, 'sysbio_modeling_amd', '_build', 'sbm_model_stiff50.so')
    if not (os.path.exists(plugin) and shutil.which('objcopy') and os.path.exists(os.path.join(LLVM, 'llvm-objdump'))):
        pytest.skip("needs the built stiff50 plugin and the LLVM binutils of ROCm")
    text = _disassemble(plugin, str(tmp_path))
    # the kernel: from its symbol to the next symbol
    m = re.search(r'^[0-9a-f]+ <(_Z18sbm_iex_seq_kernelI8SbmModelLb1EE[^>]*)>:\n(.*?)(?=^[0-9a-f]+ <)', text, re.S | re.M)
    assert m, "sbm_iex_seq_kernel<SbmModel, true> not found in the code object"
    body = [ln.split('//')[0].strip() for ln in m.group(2).splitlines() if ln.strip()]
    lds_loads = [i for i, ln in enumerate(body) if ln.startswith('global_load_lds_dwordx4')]
    # two sites (the prologue loop and the step loop), two loads each (RC by all lanes, A by lanes 0-31)
    assert len(lds_loads) == 4, lds_loads
    # the hand-written wait: the first s_waitcnt vmcnt(8) behind the step loop's loads (the compiler may have one of its
    # own elsewhere in the kernel)
    waits = [i for i, ln in enumerate(body) if re.match(r's_waitcnt vmcnt\(8\)', ln) and i > lds_loads[3]]
    assert waits and waits[0] - lds_loads[3] <= 12, (waits, lds_loads)
    # the step loop: from the first of its two loads to the backward branch that closes it -- everything the wavefront
    # executes between two visits of the wait.  Find the loop by its branch target: the nearest label before the loads.
    start = lds_loads[2]
    # walk forward to the s_waitcnt vmcnt(0) that drains the ring after the loop (the first one after the wait)
    drain = next(i for i in range(waits[0] + 1, len(body)) if re.match(r's_waitcnt vmcnt\(0\)', body[i]))
    region = body[start:drain]
    vmem = [ln for ln in region if re.match(r'(global_|scratch_|buffer_|flat_)', ln)]
    assert all(ln.startswith('global_load_lds_dwordx4') for ln in vmem), [ln for ln in vmem if not ln.startswith('global_load_lds')]
    assert len(vmem) == 2
    # and the column step is there in full: 50 rows, one multiply and one fused multiply-add each
    n_valu64 = sum(1 for ln in region if re.match(r'v_(fma|fmac|mul|add)_f64', ln))
    assert n_valu64 >= 2 * 50, n_valu64
