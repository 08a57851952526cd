"""DESIGN.md section 4.3's per-site table from a bench line and the counter summary of the same session:
    python tools/design_table.py profiles/bench_r06.json profiles/traffic.json        (prints the markdown rows)"""
import json
import sys

ROWS = [
    ('`stage`', '`stage_fused_kernel`: the row lists of both heads (the rows themselves are GATHERED from the q32b block), the dropout keep bytes of H1 (Philox), the forward partition bound', 'stage'),
    ('`embed_l1_fwd`', '**K1** `gemm_p2_ntg_kernel` (both heads, 8 problems; rows gathered by per-lane LDS-DMA addresses)', 'embed_l1_fwd'),
    ('`pool_fwd`', '`pool_rows_kernel`: masked mean + sign bits of H1', 'pool_fwd'),
    ('`embed_l2_fwd`', 'layer 2 + tanh + dropout (on-the-fly core, grouped)', 'embed_l2_fwd'),
    ('`gate_stage` ×3', '`split_q32b_dual_kernel` ×3: Wg + Wg^T (side stream), EE + EE^T, dZg + dZg^T', 'gate_stage'),
    ('`gate_fwd`', '`gemm_p3_kernel<4, 3, 0>`', 'gate_fwd'),
    ('`linear_fwd`', 'heads forward (split-K + reduce)', 'linear_fwd'),
    ('`loss`', '`margin_loss_kernel` (+ finalize inside)', 'loss'),
    ('`linear_dA`', 'heads\' data gradient', 'linear_dA'),
    ('`linear_dW`', 'heads\' weight gradient (side stream)', 'linear_dW'),
    ('`gate_dEE`', '`gemm_p3_kernel<4, 3, 1>` (through `Wg^T`; two column ranges, one launch)', 'gate_dEE'),
    ('`gate_dW`', '`gemm_p3_kernel<4, 3, 2>` (`dZg^T`, `EE^T`; side stream; three tiles per workgroup)', 'gate_dW'),
    ('`embed_dZ1`', 'hidden-layer gradient (on-the-fly core)', 'embed_dZ1'),
    ('`embed_dW2`', 'second layers\' weight gradients (third stream)', 'embed_dW2'),
    ('`pool_bwd`', '`unpool_rows_kernel`: un-pool from the sign bits → dZ1 planes', 'pool_bwd'),
    ('`embed_dW1`', '**dW1** `gemm_p2_tn_kernel<0, true, 2>` (stream-K over the gathered rows)', 'embed_dW1'),
    ('`embed_dW1_reduce`', '`gemm_p2_tn_reduce_kernel<true>`: slab reduce + Adam of the first layers + W1 → q32b', 'embed_dW1_reduce'),
    ('`adam` ×2', '`adam_kernel` ×2: heads + gate (side stream, counts its own step), second layers', 'adam'),
]


def main(bench, traffic):
    d, t = json.load(open(bench)), json.load(open(traffic))
    k, busy = d['kernels'], t.get('_mfma_busy', {})
    out = ['| site | kernel | in step µs | alone µs | bound | frac (in step) | counter bytes / launch | MFMA-pipe busy |', '|---|---|---|---|---|---|---|---|']
    for a, b, site in ROWS:
        v = k[site]
        mb = ('%d MB' % round(t[site] / 1e6)) if t.get(site) else '—'
        out.append('| %s | %s | %.1f | %.1f | %s | %.3f | %s | %s |' % (a, b, 1e3 * v['avg_ms'], 1e3 * (v.get('alone_avg_ms') or 0), v['bound'], v['frac'], mb,
                                                                       ('%.2f' % busy[site]) if site in busy else '—'))
    print('\n'.join(out))


if __name__ == '__main__':
    main(*sys.argv[1:3])
