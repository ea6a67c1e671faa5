"""The `roofline` object of bench.py's JSON line (stage sk): the dominant kernel against the HBM peak, the committed counter
profile it is read beside, and the whole step against the same roof."""
import json
import os

from benchlib.options import HBM_PEAK_GBPS, alg_bytes, whole_step_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VALU_PEAK = 256 * 4 * 2.4e9 / 2  # wave-instructions / s: 256 CUs x 4 SIMDs x one wave64 instruction per 2 cycles at 2.4 GHz


def committed_counter_profile(cfg_name, file='pmc_render_backward.json'):
    """counters are not collected in a bench run: what the last committed PMC profile of render_backward on this workload says,
    with the file and the commit it was taken at (tools/pmc_summary.py writes both); None when there is none"""
    pmc = os.path.join(ROOT, 'profiles', file)
    if not os.path.exists(pmc):
        return None
    try:
        rec = json.load(open(pmc))
        if rec.get('config') != cfg_name:
            return None
        out = {k: rec.get(k) for k in ('file', 'commit', 'hbm_bytes_per_launch', 'valubusy', 'valuutilization',
                                       'valu_insts_per_launch', 'avg_us', 'step_kernels_rocprof') if k in rec}
        if rec.get('valu_insts_per_launch') and rec.get('avg_us'):
            # VALU issue roofline (MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles on a SIMD-32)
            ach = rec['valu_insts_per_launch'] / (rec['avg_us'] * 1e-6)
            out['valu'] = dict(
                bound='valu', achieved=round(ach / 1e9, 1), peak=round(VALU_PEAK / 1e9, 1), unit='G wave-instructions/s',
                frac=round(ach / VALU_PEAK, 4),
                # measured on this chip (tools/micro/valu_issue_rate.hip, profiles/*_valu_issue_rate.txt): plain fp32 / integer
                # ops ~1000 G/s, DPP / compares / selects ~570, permlane swaps, exp, rcp ~300
                peak_measured_plain=1000.0, frac_of_measured=round(ach / 1000.0e9, 4),
                busy=round(rec['valubusy'] / 100.0, 4) if rec.get('valubusy') else None,
                note='busy = share of the kernel\'s time its SIMDs spend issuing VALU work: the distance from the ceiling of ITS '
                     'OWN instruction mix (40 % of the issue clocks of a visit are the cross-lane reduction: 17 DPP adds, 2 '
                     'permlane swaps)')
        return out
    except Exception:
        return None


def render_backward_roofline(prof, cfg, R_mean, n_params, ms_step):
    """`prof`: {kernel: (ms, launches)} of the timed kernels; `n_params`: optimizer elements (28 B each per step)"""
    P, M, K, W, H = cfg['P'], cfg['M'], cfg['K'], cfg['W'], cfg['H']
    rb_ms, rb_n = prof.get('render_backward', (0.0, 0))
    rb_us = rb_ms / max(rb_n, 1) * 1e3
    rb_bytes = alg_bytes('render_backward', P, M, K, W, H, R_mean)
    achieved = rb_bytes / (rb_us * 1e-6) / 1e9 if rb_us > 0 else 0.0
    from_profile = committed_counter_profile(cfg['name'])
    b_alg, b_adam = whole_step_bytes(P, M, K, W, H, R_mean), 28 * n_params
    whole_step = dict(alg_bytes_render=int(b_alg), alg_bytes_adam=int(b_adam), ms=round(ms_step, 4),
                      frac=round((b_alg + b_adam) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                      frac_render_only=round(b_alg / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                      note='B_alg of SURVEY 8(d) (+ 28 B per optimizer element) / ms_per_step / 8 TB/s')
    return {'bound': 'hbm', 'kernel': 'render_backward', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBPS, 5),
            # HBM bytes per launch from the PMC counters (2 x FETCH_SIZE + WRITE_SIZE, separate passes): not collected in this
            # run -- the figure of the last committed counter profile of this kernel on this workload, with the file and commit it
            # comes from (null when there is none for this workload)
            'traffic': (from_profile or {}).get('hbm_bytes_per_launch'),
            'traffic_source': ({k: from_profile.get(k) for k in ('file', 'commit')} if from_profile else None),
            'whole_step': whole_step, 'avg_us': round(rb_us, 2), 'launches': rb_n, 'alg_bytes_per_launch': int(rb_bytes),
            'limiter': 'valu', 'from_profile': from_profile,
            'note': 'the dominant kernel is VALU-issue bound, not HBM-bound (SURVEY 8d caveat): the HBM fraction above is the '
                    'contract figure, from_profile.valu the one that bounds it; streaming kernels are listed under "kernels" with '
                    'their GB/s; traffic comes from the committed counter profile named in traffic_source (no counter is '
                    'collected in a bench run), null when there is none for this workload'}


def kernels_sum(kernels, cfg_name, ms_step):
    """the three ways the step's kernel time is stated, side by side: the sum of the `kernels` table (every entry is a
    HIP-event bracket around ONE eager launch: ~2-4 us of bracket each, so the sum sits ABOVE the step), the step itself (graph
    replay, the contract's number), and the sum of the same launches' rocprofv3 averages from the committed `--kernel-trace --stats`
    profile of this workload (no brackets: equals the step when the graph has no gaps)"""
    prof = (committed_counter_profile(cfg_name) or {}).get('step_kernels_rocprof')
    return {'event_bracketed_us': round(sum(v['us'] * v['launches_per_step'] for v in kernels.values()), 1),
            'graph_step_us': round(ms_step * 1e3, 1),
            'rocprof_us': None if prof is None else prof.get('sum_us'),
            'rocprof_source': None if prof is None else {'file': prof.get('file'), 'missing': prof.get('missing')},
            'note': 'event_bracketed = sum of the `kernels` table, each entry timed as one eager launch between two HIP events (the '
                    'bracket adds 2-4 us per entry); rocprof = sum of the same launches\' average durations in the committed '
                    'rocprofv3 kernel-stats profile; graph_step = ms_per_step'}
