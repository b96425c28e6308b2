"""Condense the rocprofv3 output of scripts/profile_gpu.sh into the files kept under profiles/.

usage: python scripts/summarize_profile.py gpurun_out/<tag> profiles/<round>

  <round>/<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  <round>/<workload>_pmc_summary.json   per kernel: launch resources + mean of every collected counter per launch
  <round>/<workload>_plain.json         the un-profiled `bench.py --only <workload>` line (steps per pass)
  profiles/kernel_counters.json         what bench.py's roofline objects read, per kernel key:
        hbm_bytes_per_launch   (2*FETCH_SIZE + WRITE_SIZE) KiB per pass of the workload -- FETCH_SIZE doubled as
                               MI355X_MICROARCH.md's HBM section prescribes for gfx950, separate PMC passes
        valu_insts_per_step    SQ_INSTS_VALU / accepted steps
        cycles_per_valu_inst   4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU   (the counter ticks in quad-cycles)
        valu_busy_fraction     SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES * resident waves per SIMD  (= share of the SIMD's
                               issue slots; waves per SIMD from the launch's register and LDS allocation)
        lds_conflict_share     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

# key -> (workload, kernel-name regex, path of "steps per pass" inside the workload's plain.json, launches per pass)
# ... and the native module the kernel lives in ('core' = libsbm_hip.so, else the model plugin's name): its build stamp
# (bench.py's "build" object in plain.json) is stored with the entry, and bench.py prints the entry only while the
# module it runs carries the same stamp.
KEYS = {
    'sens_rowgroup_cascade20_dopri45': ('headline', r'sbm_sens_rowgroup_kernel<.*, 1>', ('headline', 'steps_per_pass'), 1, 'cascade20'),
    'state_packed_cascade20_dopri45': ('configs1', r'sbm_state_packed_kernel<.*, 1, ', ('configs1', 'dopri45', 'steps'), 1, 'cascade20'),
    'state_packed_cascade20_rk4_fixed_4096': ('configs1', r'sbm_state_packed_kernel<.*, 0, ', ('configs1', 'rk4_fixed_4096', 'steps'), 1, 'cascade20'),
    'iex_stiff50': ('configs4', r'sbm_iex(_seq)?_kernel', ('configs4', 'euler_steps'), 1, 'stiff50'),
    'imid_stiff50': ('configs4_fixed', r'sbm_imid_kernel', ('configs4_fixed', 'steps'), None, 'stiff50'),
    'lm_step': ('fit', r'k_lm_step', None, None, 'core'),
    'lm_trust_step': ('fit', r'k_lm_trust', None, None, 'core'),
    'lm_update': ('fit', r'k_lm_update', None, None, 'core'),
    'assemble': ('headline', r'k_assemble', None, 1, 'core'),
    'sens_rowgroup_cascade20_dop853': ('dop853', r'sbm_sens_rowgroup_kernel<.*RG2, 5>', ('dop853', 'steps'), 1, 'cascade20'),
    'dense20_valu': ('dense', r'sbm_sens_row(lane|group)_kernel', ('dense', 'valu', 'steps'), 1, 'dense20'),
    'dense20_mfma': ('dense', r'sbm_sens_mfma_kernel', ('dense', 'mfma', 'steps'), 1, 'dense20'),
}


def waves_per_simd(vgpr, agpr, lds):
    """Resident waves per SIMD of a 64-thread workgroup kernel.  rocprofv3's VGPR_Count on gfx950 is in units of two
    registers (the row-group kernel's 256 VGPRs read "128"); LDS: 160 KiB per CU, four SIMDs."""
    try:
        regs = 2 * (int(vgpr) + int(agpr or 0))
        alloc = -(-regs // 8) * 8
        by_regs = max(1, min(8, 512 // max(alloc, 1)))
        by_lds = 8 if not int(lds or 0) else max(1, (160 * 1024 // int(lds)) // 4)
    except (TypeError, ValueError):
        return None
    return min(by_regs, by_lds)


def read_pmc(src):
    summary = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(src, 'pmc_*', '**', '*_counter_collection.csv'), recursive=True)):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        launch = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row['Kernel_Name']
                per[k][row['Counter_Name']].append(float(row['Counter_Value']))
                launch[k] = {'vgpr': row.get('VGPR_Count'), 'agpr': row.get('Accum_VGPR_Count'),
                             'sgpr': row.get('SGPR_Count'), 'lds': row.get('LDS_Block_Size'),
                             'scratch': row.get('Scratch_Size'), 'wg': row.get('Workgroup_Size'),
                             'grid': row.get('Grid_Size')}
        for k, d in per.items():
            e = summary.setdefault(k, collections.OrderedDict(launch=launch[k]))
            for c, v in sorted(d.items()):
                e[c] = {'launches': len(v), 'mean_per_launch': sum(v) / len(v), 'total': sum(v)}
    return summary


def dig(d, path):
    for p in path:
        d = d[p]
    return d


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    cpath = os.path.join(os.path.dirname(os.path.abspath(dst)), 'kernel_counters.json')
    try:
        with open(cpath) as fh:
            counters = json.load(fh)
    except (OSError, ValueError):
        counters = {}
    for wdir in sorted(glob.glob(os.path.join(src, '*', ''))):
        w = os.path.basename(os.path.dirname(wdir))
        ks = glob.glob(os.path.join(wdir, 'trace', '**', '*_kernel_stats.csv'), recursive=True)
        if ks:
            shutil.copy(ks[0], os.path.join(dst, '%s_kernel_stats.csv' % w))
        plain = None
        try:
            with open(os.path.join(wdir, 'plain.json')) as fh:
                plain = json.loads([ln for ln in fh if ln.startswith('{')][-1])
            with open(os.path.join(dst, '%s_plain.json' % w), 'w') as fh:
                json.dump(plain, fh, indent=1)
        except (OSError, ValueError, IndexError):
            pass
        summary = read_pmc(wdir)
        if summary:
            with open(os.path.join(dst, '%s_pmc_summary.json' % w), 'w') as fh:
                json.dump(summary, fh, indent=1)
        else:      # the raw counter files are gone (they stay on the GPU box): work from the kept summary
            try:
                with open(os.path.join(dst, '%s_pmc_summary.json' % w)) as fh:
                    summary = json.load(fh, object_pairs_hook=collections.OrderedDict)
                if plain is None:
                    with open(os.path.join(dst, '%s_plain.json' % w)) as fh:
                        plain = json.load(fh)
            except (OSError, ValueError):
                continue
        for key, (kw, rx, steps_path, per_pass, module) in KEYS.items():
            if kw != w:
                continue
            hits = [k for k in summary if re.search(rx, k)]
            if not hits:
                continue
            # several instantiations may match (e.g. two chunk layouts): take the one with most VALU work
            k = max(hits, key=lambda kk: summary[kk].get('SQ_INSTS_VALU', {}).get('total', 0.0))
            e = summary[k]
            out = {'kernel': k, 'launch': e['launch'], 'source': os.path.join(dst, '%s_pmc_summary.json' % w),
                   'module': module, 'build_stamp': ((plain or {}).get('build') or {}).get(module)}
            n_launch = e.get('SQ_INSTS_VALU', e.get('WRITE_SIZE', {'launches': 0}))['launches']
            steps_pass = None
            if plain is not None and steps_path is not None:
                try:
                    steps_pass = float(dig(plain, steps_path))
                except (KeyError, TypeError):
                    steps_pass = None
            lpp = per_pass
            if lpp is None and plain is not None:
                try:
                    lpp = float(plain[w].get('launches_per_pass', 1))
                except (KeyError, AttributeError):
                    lpp = 1
            lpp = lpp or 1
            if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
                f_, w_ = e['FETCH_SIZE']['mean_per_launch'], e['WRITE_SIZE']['mean_per_launch']
                out['hbm_bytes_per_launch'] = (2.0 * f_ + w_) * 1024.0 * lpp
                out['fetch_kib_per_launch'], out['write_kib_per_launch'] = f_, w_
                out['launches_per_pass'] = lpp
            if 'SQ_INSTS_VALU' in e and e['SQ_INSTS_VALU']['total'] > 0:
                iv = e['SQ_INSTS_VALU']
                av = e.get('SQ_ACTIVE_INST_VALU')
                if steps_pass:
                    out['valu_insts_per_step'] = iv['mean_per_launch'] * lpp / steps_pass
                if av:
                    out['cycles_per_valu_inst'] = 4.0 * av['total'] / iv['total']
                    wc = e.get('SQ_WAVE_CYCLES')
                    wps = waves_per_simd(e['launch']['vgpr'], e['launch']['agpr'], e['launch']['lds'])
                    # ... and no more than the launch has: a grid of fewer waves than the chip's 1024 SIMDs can hold
                    try:
                        n_waves = int(e['launch']['grid']) // 64
                        if wps:
                            wps = min(float(wps), n_waves / 1024.0)
                    except (TypeError, ValueError):
                        pass
                    if wc and wps:
                        out['waves_per_simd'] = wps
                        out['valu_active_share_of_wave_lifetime'] = av['total'] / wc['total']
                        out['valu_busy_fraction'] = min(1.0, av['total'] / wc['total'] * wps)
            if 'SQ_INSTS_SALU' in e and steps_pass:
                out['salu_insts_per_step'] = e['SQ_INSTS_SALU']['mean_per_launch'] * lpp / steps_pass
            if 'SQ_ACTIVE_INST_ANY' in e and e.get('SQ_WAVE_CYCLES', {}).get('total'):
                # share of a wavefront's lifetime in which it has an instruction of ANY kind in flight
                out['any_inst_active_share_of_wave_lifetime'] = e['SQ_ACTIVE_INST_ANY']['total'] / e['SQ_WAVE_CYCLES']['total']
            if 'SQ_LDS_BANK_CONFLICT' in e and e.get('SQ_LDS_IDX_ACTIVE', {}).get('total'):
                out['lds_conflict_share'] = e['SQ_LDS_BANK_CONFLICT']['total'] / e['SQ_LDS_IDX_ACTIVE']['total']
                if 'SQ_INSTS_LDS' in e and steps_pass:
                    out['lds_insts_per_step'] = e['SQ_INSTS_LDS']['mean_per_launch'] * lpp / steps_pass
            counters[key] = out
    with open(cpath, 'w') as fh:
        json.dump(counters, fh, indent=1)
    print("wrote", dst, "and", cpath, "keys:", sorted(counters))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
