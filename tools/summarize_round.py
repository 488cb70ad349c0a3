"""Print the figures DESIGN.md / README.md quote from one tools/gpu_round.sh output directory: python tools/summarize_round.py gpurun_out/<tag>"""
import json, os, sys

d = sys.argv[1]


def load(f):
    l = open(os.path.join(d, f)).read()
    return json.loads(l[l.index('{'):])


def par(p):
    return (p['sequences_checked'], p['sequences_outside_tolerance_at_last_frame'],
            [(x['sequence'], x['first_frame_outside_tolerance'], x['outside_tolerance_at_last_frame']) for x in p['departures']],
            p['departures_elsewhere'], p['max_abs_dVW_while_inside_tolerance'])


j = load('bench.json')
print('default', j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], j['roofline'].get('launch_us'))
print(j['kernel_us_per_step'], 'sum', round(sum(j['kernel_us_per_step'].values())))
print('single', j['single_sequence_ms_per_frame'], [(b['sequences_per_launch'], b['frames_per_s']) for b in j['batch_sweep']])
print({k: v['value'] for k, v in j['cpu_baseline']['modes'].items()})
ex = j['extras']
print({k: (v.get('value'), v.get('unit')) for k, v in ex.items() if isinstance(v, dict)})
print('pcie', j['pcie_inclusive']['rgb24']['value'], j['pcie_inclusive']['grey8']['value'])
print('parity default', par(j['pose_rmse']['free_running_parity']))
print('parity hetero', par(j['heterogeneous']['free_running_parity']), j['heterogeneous'].get('value'))
print('parity tum', par(ex['tum_undistort']['pose_rmse']['free_running_parity']))
print('nav', j['config'].get('nav_gather'), j['config'].get('nav_gather_info'))
for k, v in j['roofline_kernels'].items():
    print(' ', k, v)
if os.path.exists(os.path.join(d, 'bench_driver_form.json')):
    x = load('bench_driver_form.json')
    print('driver', x['value'], x['ms_per_step'], x['roofline']['kernel'], x['roofline']['frac'], x['roofline'].get('launch_us'))
    print(x['kernel_us_per_step'])
    print('parity driver', par(x['pose_rmse']['free_running_parity']), par(x['heterogeneous']['free_running_parity']))
for f in ('bench_stage_a', 'bench_tum_undistort', 'bench_imu'):
    x = load(f + '.json')
    print(f, x['value'], x['unit'], x['ms_per_step'])
