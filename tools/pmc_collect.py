"""Collect the rocprofv3 passes of tools/profile_all.sh into the files bench.py and the judge read:

    python tools/pmc_collect.py <tag> [gpurun_out]      ->  profiles/<tag>_pmc.json        {workload: {kernel: {counter: mean per dispatch}}}
                                                            profiles/<tag>_<wl>_pmc.txt    the same as text
                                                            profiles/<tag>_<wl>_kernel_stats.{csv,txt}  rocprofv3 --stats summary

FETCH_SIZE / WRITE_SIZE are in KB.  `fetch_doubled` marks the kernels whose reads are 16-B-per-lane coalesced
streams, for which gfx950's FETCH_SIZE reports half the bytes (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'gpurun_out')
# kernels that stream their input with 16-byte-per-lane loads (float4 rows / planes)
WIDE_READERS = ("chamfer_nn_mfma_kernel", "chamfer_fixup_kernel")


def short(name):
    n = name.split('(')[0]
    n = n.replace('void ', '').replace('vpn::', '')
    # the two instantiations of raster_total_kernel (training step / module path) never meet in one workload: both go
    # under the name the library's launch profile and bench.py use
    n = n.replace('raster_total_kernel<true>', 'raster_total_kernel').replace('raster_total_kernel<false>', 'raster_total_kernel')
    return n.strip()


out = {}
for wl in ('c3', 'c2', 'c5'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(src, '%s_%s_pmc*' % (tag, wl)))):
        # gpurun merges every call's output into gpurun_out/: a pass directory may hold files of earlier runs of
        # the same tag; only the newest one describes the current kernels
        for f in sorted(glob.glob(d + '/*/*counter_collection.csv'), key=os.path.getmtime)[-1:]:
            for r in csv.DictReader(open(f)):
                acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    if not acc:
        continue
    ker = {}
    lines = ['# rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 10 --warmup 3 --no-extras --no-graph '
             '--workload %s  (separate passes, tools/profile_all.sh; mean per dispatch)' % wl]
    for k, c in sorted(acc.items()):
        if not k.startswith(('chamfer', 'raster', 'sample', 'loss_', 'emd_', 'vpdiv', 'trainstep', 'camera')):
            continue
        ker[k] = {cn: sum(v) / len(v) for cn, v in c.items()}
        ker[k]['dispatches'] = max(len(v) for v in c.values())
        ker[k]['fetch_doubled'] = k.startswith(WIDE_READERS)
        lines.append(k)
        for cn, v in sorted(c.items()):
            lines.append('   %-28s n=%-4d mean=%.4g' % (cn, len(v), sum(v) / len(v)))
    out[wl] = ker
    open(os.path.join(ROOT, 'profiles', '%s_%s_pmc.txt' % (tag, wl)), 'w').write('\n'.join(lines) + '\n')
    st = sorted(glob.glob(os.path.join(src, '%s_%s_stats' % (tag, wl), '*', '*kernel_stats.csv')), key=os.path.getmtime)[::-1]
    if st:
        shutil.copy(st[0], os.path.join(ROOT, 'profiles', '%s_%s_kernel_stats.csv' % (tag, wl)))
        with open(os.path.join(ROOT, 'profiles', '%s_%s_kernel_stats.txt' % (tag, wl)), 'w') as fh:
            fh.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-extras --no-graph --workload %s\n' % wl)
            for r in list(csv.DictReader(open(st[0])))[:16]:
                fh.write('%-72s calls=%-5s avg_us=%9.2f min=%9.2f max=%9.2f pct=%s\n' % (
                    r['Name'][:72], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3,
                    float(r['MaxNs']) / 1e3, r['Percentage']))
out['_source'] = ('rocprofv3 passes of tools/profile_all.sh %s; FETCH_SIZE / WRITE_SIZE in KB (separate passes); fetch_doubled = '
                  'kernel streams 16 B per lane, FETCH_SIZE doubled by bench.py per MI355X_MICROARCH.md HBM section' % tag)
json.dump(out, open(os.path.join(ROOT, 'profiles', '%s_pmc.json' % tag), 'w'), indent=1, sort_keys=True)
print('wrote profiles/%s_pmc.json:' % tag, {k: len(v) for k, v in out.items() if not k.startswith('_')})
