#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py into profiles/.

usage: tools/pmc_summary.py <tag> <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [<dir with VALUBusy pass>]
Each dir is the -d output of   rocprofv3 --pmc <COUNTER..> --kernel-trace --output-format csv -d <dir> -- python bench.py ...
(separate passes, as MI355X_MICROARCH.md prescribes).  Writes profiles/<tag>_pmc_hbm_traffic.md and refreshes
profiles/pmc_render_backward.json (the per-launch HBM bytes bench.py copies into roofline.traffic).
FETCH_SIZE / WRITE_SIZE are in KB; gfx950 correction from the guide: traffic = 2 * FETCH_SIZE + WRITE_SIZE."""
import csv, glob, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r'(\w+_kernel)(<[^>]*>)?\(', name)
    if m:
        return m.group(1) + (m.group(2) or '')
    m = re.search(r'(\w+)\(', name)
    return m.group(1) if m else name[:60]


def load(d):
    out = defaultdict(lambda: defaultdict(list))
    files = sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # the newest pass only (gpurun merges successive runs into the same directory)
        for r in csv.DictReader(open(f)):
            out[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return out


def main():
    tag, dfetch, dwrite = sys.argv[1:4]
    dvalu = sys.argv[4] if len(sys.argv) > 4 else None
    fetch, write = load(dfetch), load(dwrite)
    valu = load(dvalu) if dvalu else {}
    rows = []
    for k in sorted(set(fetch) | set(write)):
        if 'at::' in k or 'rocclr' in k or 'elementwise' in k:
            continue
        f = fetch.get(k, {}).get('FETCH_SIZE', [])
        w = write.get(k, {}).get('WRITE_SIZE', [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        vb = valu.get(k, {})
        extra = {c: sum(v) / len(v) for c, v in vb.items()}
        rows.append((k, max(len(f), len(w)), fm, wm, (2 * fm + wm) * 1024 / 1e6, extra))
    rows.sort(key=lambda r: -r[4])
    cols = sorted({c for r in rows for c in r[5]})
    md = [f'# HBM traffic per launch from PMC counters ({tag}, ' + (os.environ.get('SKGS_PROFILE_TITLE') or os.environ.get('SKGS_PROFILE_CONFIG', 'config #1: 100k Gaussians, 20 bones, 800x800')) + ')', '',
          'Separate passes (MI355X_MICROARCH.md): `rocprofv3 --pmc FETCH_SIZE --kernel-trace ...`, `--pmc WRITE_SIZE ...`'
          + (', `--pmc ' + ' '.join(cols) + ' ...`' if cols else '') + ' on `python bench.py --steps 10 --warmup 2 --no-cpu-baseline`.',
          'Counters are in KB; gfx950 correction from the guide: `traffic = 2 * FETCH_SIZE + WRITE_SIZE`.', '',
          '| kernel | launches | FETCH_SIZE [KB] | WRITE_SIZE [KB] | traffic = 2F+W [MB] |' + ''.join(f' {c} |' for c in cols),
          '|---|---|---|---|---|' + '---|' * len(cols)]
    for k, n, fm, wm, t, extra in rows:
        def cell(c):
            v = extra.get(c, float('nan'))
            if c in ('VALUBusy', 'VALUUtilization') and v > 100.0:    # a derived PERCENTAGE: see the note under the table
                return f' 100 (raw {v:.1f}) |'
            return f' {v:.1f} |'
        md.append(f'| {k} | {n} | {fm:.1f} | {wm:.1f} | {t:.1f} |' + ''.join(cell(c) for c in cols))
    if any(extra.get(c, 0.0) > 100.0 for _, _, _, _, _, extra in rows for c in ('VALUBusy', 'VALUUtilization')):
        md += ['', 'Note: `VALUBusy` is rocprofv3\'s DERIVED metric `100 * SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / GRBM_GUI_ACTIVE`; the two raw',
               'counters are sampled by different blocks (SQ per XCD, summed; GRBM once) and on launches of several hundred microseconds',
               'the ratio overshoots 100 % by up to ~20 % on this chip.  Values above 100 are shown clamped with the raw reading in',
               'brackets; read them as "the SIMDs issue VALU work in (nearly) every cycle", not as a measurement above the peak.']
    open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_hbm_traffic.md'), 'w').write('\n'.join(md) + '\n')
    rb = [r for r in rows if r[0].startswith('render_backward_kernel')]
    if rb:
        k, n, fm, wm, t, extra = rb[0]
        import subprocess
        # the commit the counters were taken at: SKGS_PROFILE_COMMIT when the summary is made later than the run, else HEAD
        try:
            commit = os.environ.get('SKGS_PROFILE_COMMIT') or subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        except Exception:
            commit = None
        rec = {'config': os.environ.get('SKGS_PROFILE_CONFIG', 'hook-like-100k-800'), 'kernel': 'render_backward',
               'file': f'profiles/{tag}_pmc_hbm_traffic.md', 'commit': commit, 'fetch_size_kb': fm, 'write_size_kb': wm,
               'hbm_bytes_per_launch': int((2 * fm + wm) * 1024),
               'method': f'2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes, see {tag}_pmc_hbm_traffic.md'}
        rec.update({c.lower(): v for c, v in extra.items()})
        if 'sq_insts_valu' in rec:
            rec['valu_insts_per_launch'] = rec['sq_insts_valu']
        # average kernel duration from the --stats pass of the same profile (profiles/<tag>_kernel_stats_*.csv)
        for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats*.csv'))):
            for r in csv.DictReader(open(f)):
                if 'render_backward_kernel' in r.get('Name', ''):
                    rec['avg_us'] = float(r['AverageNs']) / 1e3
        # the step's launches by rocprofv3 (average duration of each, and their sum): what bench.py prints under `kernels_sum`
        sys.path.insert(0, ROOT)
        from benchlib.options import STEP_KERNELS_SK
        for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats*.csv'))):
            per = {}
            for r in csv.DictReader(open(f)):
                for nm, pat in STEP_KERNELS_SK.items():
                    if re.search(pat + r'\(', r.get('Name', '')) and 'at::' not in r['Name'][:20]:
                        # (several instantiations match a family only for the preprocess pair: the step's is the one with a job)
                        if nm not in per or int(r['Calls']) < per[nm]['calls']:
                            per[nm] = dict(us=round(float(r['AverageNs']) / 1e3, 2), calls=int(r['Calls']))
            if per:
                rec['step_kernels_rocprof'] = dict(file=os.path.relpath(f, ROOT), kernels=per, sum_us=round(sum(v['us'] for v in per.values()), 1),
                                                   missing=[n for n in STEP_KERNELS_SK if n not in per])
        # (SKGS_PMC_JSON: another file name, e.g. pmc_render_backward_sp.json for the stage-sp profile)
        json.dump(rec, open(os.path.join(ROOT, 'profiles', os.environ.get('SKGS_PMC_JSON', 'pmc_render_backward.json')), 'w'), indent=1)
    print('\n'.join(md[:14]))


if __name__ == '__main__':
    main()
