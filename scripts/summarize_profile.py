"""Condense the rocprofv3 output of scripts/profile_gpu.sh into the files kept under profiles/.

usage: python scripts/summarize_profile.py gpurun_out/<tag> profiles/<round>

  <round>/kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  <round>/pmc_summary.json   per kernel: launch resources + mean of every collected counter per launch
  profiles/hbm_traffic.json  HBM bytes per launch of the dominant sensitivity kernel (bench.py's
                             roofline.traffic): (2*FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE doubled as
                             MI355X_MICROARCH.md's HBM section prescribes for gfx950, separate passes.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    ks = glob.glob(os.path.join(src, 'trace', '**', '*_kernel_stats.csv'), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(dst, 'kernel_stats.csv'))
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
                e[c] = {'launches': len(v), 'mean_per_launch': sum(v) / len(v)}
    with open(os.path.join(dst, 'pmc_summary.json'), 'w') as fh:
        json.dump(summary, fh, indent=1)
    # dominant sensitivity kernel = the one with the largest WRITE_SIZE total
    best = None
    for k, e in summary.items():
        if 'sbm_sens' in k and 'WRITE_SIZE' in e and 'FETCH_SIZE' in e:
            tot = e['WRITE_SIZE']['mean_per_launch'] * e['WRITE_SIZE']['launches']
            if best is None or tot > best[0]:
                best = (tot, k, e)
    if best:
        _, k, e = best
        fetch, write = e['FETCH_SIZE']['mean_per_launch'], e['WRITE_SIZE']['mean_per_launch']
        tpath = os.path.join(os.path.dirname(os.path.abspath(dst)), 'hbm_traffic.json')
        out = {'dopri45': {
            'hbm_bytes_per_launch': (2.0 * fetch + write) * 1024.0, 'fetch_kib': fetch, 'write_kib': write,
            'note': "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), %s, V=4096, 17 output rows; "
                    "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md "
                    "section HBM; the read side is scalar / 8-byte traffic, so this is an upper bound). Writes = the "
                    "sampled Y/S rows: 4096*17*820*8 B = 456.7e6 B." % k,
            'source': os.path.join(dst, 'pmc_summary.json')}}
        with open(tpath, 'w') as fh:
            json.dump(out, fh, indent=1)
    print("wrote", dst, "kernels:", len(summary))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
