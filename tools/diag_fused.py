"""diagnostic: eager loop vs recorded step (fused first-layer update), buffer by buffer after every step"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from test_gpu_recorded_bench_shape import _fresh, _eager_step, B, T, R, host_batch
from lirec_amd.data import to_device_batch
from lirec_amd.graph import RecordedTrainStep
from lirec_amd import ops

hb = host_batch(B, T, R, 'survey')
m1, l1, o1 = _fresh(False)
m2, l2, o2 = _fresh(False)
b1, b2 = to_device_batch(hb, 'cuda'), to_device_batch(hb, 'cuda')
for _ in range(3):
    _eager_step(m1, l1, o1, b1)
g = RecordedTrainStep(m2, l2, o2, b2, warmup=2)
print('fused', g.fused, 'overwrite', g.overwrite)
lo, hi, n = m2.first_layer_range()


def report(tag):
    torch.cuda.synchronize()
    for name, a, b in (('params', m1.flat_params(), m2.flat_params()), ('grads', m1.flat_grads(attach=False), m2.flat_grads(attach=False)),
                       ('m', o1._m, o2._m), ('v', o1._v, o2._v)):
        d0 = int((a[:lo] != b[:lo]).sum()); d1 = int((a[lo:] != b[lo:]).sum())
        print(tag, name, 'differ: before first layers', d0, ' first layers', d1, float((a - b).abs().max()))
    if g.fused:
        pd = dict(m2.named_parameters())
        base = m2._w1q_buf.data_ptr()
        for nme, addr in m2._w1q.items():
            ref = ops.to_q32b(pd[nme].data.contiguous()).data
            k = 4 * pd[nme].numel()
            got = m2._w1q_buf[addr - base:addr - base + k]
            print(tag, 'shadow', nme, int((got != ref[:k]).sum()))
        Wg = pd['gates_ints.fc_out.weight'].data
        ref = ops.to_q32b(Wg.contiguous()).data
        print(tag, 'gate valid', m2._wgq_valid, 'gate shadow', int((m2._gate_ws[:4 * Wg.numel()] != ref[:4 * Wg.numel()]).sum()))
        glo, ghi, _, _ = m2.gate_range()
        for name, a, b in (('params', m1.flat_params(), m2.flat_params()), ('grads', m1.flat_grads(attach=False), m2.flat_grads(attach=False)),
                           ('m', o1._m, o2._m), ('v', o1._v, o2._v)):
            print(tag, name, 'gate range differ', int((a[glo:ghi] != b[glo:ghi]).sum()))


report('after recording (3 steps)')
for i in range(2):
    _eager_step(m1, l1, o1, b1)
    g.step()
    report('step %d' % (4 + i))
