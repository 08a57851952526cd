"""Selected short legs of bench.py's `configs` block on their own (a kernel change that touches one configuration only):
   python3 tools/config_legs.py 4q 4bq 4b      -> one JSON object per line"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    import torch
    import bench
    from lirec_amd import _lib, ops
    want = sys.argv[1:] or ['4q', '4bq', '4b']
    mode = _lib.default_gemm_mode()
    ops.set_gemm_mode(mode)
    B, T, R = 64, 32, 18
    kw = dict(rels_n_clips=R)
    legs = {
        '4': lambda: bench.config_leg('4', 'int_rel_ch', kw, 'int_rel_ch', dict(T=T, R=R), B, 101, 15, True, torch.bfloat16, mode),
        '4q': lambda: bench.config_leg('4q', 'int_rel_ch', kw, 'int_rel_ch', dict(T=T, R=R), B, 101, 15, True, 'q16', mode),
        '4c': lambda: bench.config_leg('4c', 'int_rel_ch', kw, 'int_rel_ch', dict(T=T, R=R), B, 101, 15, True, 'q32', mode),
        '4bq': lambda: bench.config_leg('4bq', 'int_rel_ch', kw, 'int_rel_ch', dict(T=T, R=R), B, 101, 15, True, 'q16', mode, set_mode=3),
        '4b': lambda: bench.config_leg('4b', 'int_rel_ch', kw, 'int_rel_ch', dict(T=T, R=R), B, 101, 15, True, torch.bfloat16, mode, set_mode=3),
    }
    legs['2b'] = lambda: bench.config_leg('2b', 'int_rel_ch', dict(rels_n_clips=R), 'int_rel_ch', dict(T=16, R=R), 256, 101, 15, True, torch.float32, mode, steps=10, warmup=3)
    legs['2b_noside'] = lambda: bench.config_leg('2b (weight gradients on the step\'s own stream)', 'int_rel_ch', dict(rels_n_clips=R, wgrad_side_stream=False),
                                                 'int_rel_ch', dict(T=16, R=R), 256, 101, 15, True, torch.float32, mode, steps=10, warmup=3)
    for k in want:
        r = legs[k]()
        print(json.dumps({'leg': k, 'ms_per_step': r.get('ms_per_step'), 'value': r.get('value'), 'roofline': r.get('roofline'),
                          'site_ms': r.get('site_ms'), 'layer1': r.get('layer1'), 'step_launch': r.get('step_launch')}))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
