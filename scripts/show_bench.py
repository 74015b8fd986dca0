"""one-line summaries of bench.py JSON lines: python scripts/show_bench.py file.json ..."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e); continue
    c = d['config']; r = d['roofline']
    s = '%-30s ms/step %7.2f  its %s/%s/%s' % (f.split('/')[-1], d['ms_per_step'], c.get('newton_its_per_step'), c.get('bicgstab_its_per_step'), c.get('poisson_cg_its_per_step'))
    s += '  roof %.3f (%.1f us)' % (r['frac'] or 0, (r['ms_per_launch'] or 0) * 1e3)
    if r.get('cold_cache'): s += ' cold %.3f (%.1f us)' % (r['cold_cache']['frac'], r['cold_cache']['ms_per_launch'] * 1e3)
    a = d.get('assembly')
    if a and a.get('frac'): s += ' asm %.3f (%.1f us)' % (a['frac'], (a.get('ms_per_application') or 0) * 1e3)
    if 'validation' in c: s += ' val %.1e/%.1e exact %.2f ms' % (c['validation']['max_rel_diff_velocity_vs_exact'], c['validation']['max_rel_diff_pressure_vs_exact'], c['ms_per_step_with_krylov_rtol_1e-12_exact_newton'])
    if 'max_abs_velocity_error_vs_analytic' in c: s += ' err %.2e' % c['max_abs_velocity_error_vs_analytic']
    print(s)
