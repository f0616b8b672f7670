"""Condensed view of a bench.py JSON line (gpurun prints only the tail of stdout)."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])   # (RCCL may print banner lines first)
short = len(sys.argv) > 2


def leg(name, x):
    r = x.get('roofline') or {}
    return (f"{name}: {x['iters_per_s']:.0f} it/s {x['ms_per_step']:.4f} ms | kernel {r.get('avg_launch_ms')} ms "
            f"(w {r.get('avg_launch_ms_w_step')} h {r.get('avg_launch_ms_h_step')}) frac {r.get('frac')} outside {r.get('outside_fused_kernels_ms')}")


print(leg(d['dtype'], d))
for k in ('bf16_mode', 'parity_mode'):
    if d.get(k):
        print('  ' + leg(k + ' ' + d[k]['precision'], d[k]))
if short:
    sys.exit(0)
print('  blocks', d.get('blocks_ms_per_step'), 'preroll', d.get('preroll_steps'))
if d.get('parity'):
    print('  parity k=%d' % d['parity']['k'], {m: {k: v for k, v in e.items()} for m, e in d['parity']['modes'].items()})
if d.get('cpu_baseline'):
    c = d['cpu_baseline']
    print('  cpu', c['iters_per_s'], 'it/s on', c['cores'], 'cores')
for b, e in ((d.get('beta_sweep') or {}).get('betas') or {}).items():
    print(f"  beta={b}: {e['iters_per_s']:.0f} it/s, kernel {e['kernel_avg_launch_ms']} ms frac {e['kernel_frac']}, parity {e.get('parity')}")
if d.get('nmfd'):
    n = d['nmfd']
    print(f"  nmfd: {n['iters_per_s']:.0f} it/s {n['ms_per_step']} ms; per gemm {n['roofline'].get('per_gemm')}; parity {n.get('parity')}")
if d.get('nmf2d'):
    n = d['nmf2d']
    print(f"  nmf2d ({n['dtype']}): {n['iters_per_s']:.0f} it/s {n['ms_per_step']} ms; per gemm {n['roofline'].get('per_gemm')}; parity {(n.get('parity') or {}).get('modes')}; fit loop {(n.get('fit') or {}).get('iters_per_s_loop')} it/s")
r = d.get('roofline') or {}
if r.get('ceiling'):
    c = r['ceiling']
    print(f"  ceiling: with stream {c['with_stream']}, mfma only {c['mfma_only']}; frac_of_ceiling {r.get('frac_of_ceiling')}; traffic {r.get('traffic')} ({(r.get('traffic_source') or '')[:80]})")
if d.get('ref_notebook'):
    for b, e in d['ref_notebook']['betas'].items():
        print(f"  ref_notebook beta={b}: auto {e['auto']} | f16x {e['f16x_forced']} | cpu port {e.get('cpu_port_s_per_iter')} s/it | notebook {e['notebook_context_s_per_iter']}")
