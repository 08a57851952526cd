"""Condenses the GPU test session's parity_errors.json (every gradient / tensor comparison: achieved error relative to
the tensor's scale, and the fraction of the tolerance it used) to one line per test: python tools/parity_summary.py IN OUT"""
import collections
import json
import sys

d = json.load(open(sys.argv[1]))
per = collections.OrderedDict()
for key, v in d['comparisons'].items():
    test = key.split(' :: ')[0]
    e = per.setdefault(test, {'comparisons': 0, 'elements': 0, 'worst_tol_used': 0.0, 'worst_err_over_scale': 0.0, 'worst_what': ''})
    e['comparisons'] += 1
    e['elements'] += int(v.get('n', 0))
    if v['tol_used'] >= e['worst_tol_used']:
        e['worst_tol_used'], e['worst_what'] = v['tol_used'], key.split(' :: ', 1)[1] if ' :: ' in key else ''
    e['worst_err_over_scale'] = max(e['worst_err_over_scale'], v['max_err_over_scale'])
out = {'_what': 'per test: number of tensor comparisons, elements compared, the largest |hip - ref| / max|ref| seen, and the largest '
                'fraction of the tolerance (tests/golden_util.py: 1e-6 + 1e-4 |ref| + 6e-5 max|ref| for gradients) any element used',
       'worst_tol_used': d.get('worst_tol_used'), 'tests': per,
       '_relu_decisions': 'bench-shape gradient tests feed the oracle the relu decisions of the device (tests/test_gpu_bench_shape.py:'
                          'DeviceReluDecisions); per test and relu site: how many decisions differ from the oracle\'s own x > 0, the bound '
                          '(8 + 2e-5 N), the largest |x| at which one differs and its bound (2e-5)',
       'relu_decisions': d.get('relu_decisions', {})}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print('%d tests, %d comparisons, worst tolerance use %.3f' % (len(per), len(d['comparisons']), d.get('worst_tol_used', -1)))
