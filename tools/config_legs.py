"""Selected short legs of bench.py's `configs` block on their own (a kernel change that touches one configuration only):
   python3 tools/config_legs.py 4q 4bq 4b      -> one JSON object per line
   a leg name with the suffix _noside runs with opt.wgrad_side_stream off (weight gradients on the step's own stream);
   h<B> = the headline recipe at B clips x 16 pairs (fp32 block), e.g. h64, h128, h256"""
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
    R = 18
    table = {                                # (T, B, feature dtype, set_mode)
        '4': (32, 64, torch.bfloat16, None), '4q': (32, 64, 'q16', None), '4c': (32, 64, 'q32', None),
        '4bq': (32, 64, 'q16', 3), '4b': (32, 64, torch.bfloat16, 3), '2b': (16, 256, torch.float32, None),
    }
    for k in want:
        base, noside = (k[:-7], True) if k.endswith('_noside') else (k, False)
        if base.startswith('h') and base[1:].isdigit():
            T, B, fd, sm = 16, int(base[1:]), 'q32', None
        else:
            T, B, fd, sm = table[base]
        kw = dict(rels_n_clips=R)
        if noside:
            kw['wgrad_side_stream'] = False
        r = bench.config_leg(k, 'int_rel_ch', kw, 'int_rel_ch', dict(T=T, R=R), B, 101, 15, True, fd, mode, set_mode=sm,
                             steps=10 if B > 128 else 20, warmup=3)
        print(json.dumps({'leg': k, 'ms_per_step': r.get('ms_per_step'), 'value': r.get('value'), 'roofline': r.get('roofline'),
                          'site_ms': r.get('site_ms'), 'layer1': r.get('layer1'), 'step_launch': r.get('step_launch')}))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
