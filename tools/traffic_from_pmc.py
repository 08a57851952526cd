"""HBM-side bytes per call-site launch from the two rocprofv3 counter passes of tools/make_profiles.sh.

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  Corrections (MI355X_MICROARCH.md, HBM section): on gfx950
FETCH_SIZE counts 128-B requests at 64 B, so wide streaming reads are doubled; WRITE_SIZE is exact.  Bytes are summed
over the kernels of a call site (split-K reduce kernels are listed on their own) and divided by the site's launches,
the same averaging as bench.py's roofline.achieved.
usage: traffic_from_pmc.py pmc_fetch.csv pmc_write.csv out.json [pmc_mfma.csv [key=value ...]]   (key=value: run configuration / commit for `_meta`)

With the optional MFMA pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES): `_mfma_busy` per site = the fraction of the
kernel's duration in which a SIMD's matrix pipe is busy, averaged over the 1024 SIMDs:
SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)  (SQ_BUSY_CYCLES is summed over the 32 shader engines, so the
kernel lasts SQ_BUSY_CYCLES / 32 cycles; checked against the MFMA count x 32 cycles per v_mfma_f32_32x32x16_bf16).
"""
import collections
import csv
import json
import sys

SITES = [  # (site, kernel-name fragments that belong to it)
    ('embed_l1_fwd', ['gemm_bf16x3_kernel<0, 3, 1, true', 'gemm_bf16x3_kernel<0, 0, 1, true', 'gemm_mfma_kernel<0, 2, 2, 1', 'gemm_mfma_kernel<0, 1, 1, 1',
                      'gemm_p2_nt_kernel', 'gemm_p2_ntg_kernel', 'gemm_p2_ntg1_kernel', 'gemm_p2_ntg1o_kernel']),
    ('embed_dW1', ['gemm_bf16x3_kernel<2, 2, 3, true', 'gemm_bf16x3_kernel<2, 2, 2, true', 'gemm_bf16x3_kernel<2, 3, 2, true', 'gemm_bf16x3_kernel<2, 3, 3, true',
                   'gemm_p2_tn_kernel']),
    ('embed_dW1_reduce', ['gemm_p2_tn_reduce_kernel']),
    ('stage', ['stage_fused_kernel', 'stage_rows_q32b_kernel', 'split_planes_kernel']),
    ('gate_stage', ['split_q32b_kernel', 'split_q32b_dual_kernel']),
    # gemm_p3_kernel<MI, NI, EPI, ONE, ABL>: EPI 0 forward, 1 data gradient, 2 weight gradient (round 4's kernel: <KIND, ABL>)
    ('gate_fwd', ['gemm_p3_kernel<4, 3, 0, ', 'gemm_p3_kernel<4, 4, 0, ', 'gemm_p3_kernel<0, ']),
    ('gate_dEE', ['gemm_p3_kernel<4, 3, 1, ', 'gemm_p3_kernel<4, 4, 1, ', 'gemm_p3_kernel<1, ']),
    ('gate_dW', ['gemm_p3_kernel<4, 3, 2, ', 'gemm_p3_kernel<2, ']),
    ('splitk_reduce', ['splitk_reduce_flat_kernel', 'splitk_reduce_kernel']),
    ('pool_fwd', ['pool_fwd_kernel', 'pool_compact_kernel', 'pool_rows_kernel']),
    ('pool_bwd', ['pool_bwd_kernel', 'unpool_relu_kernel', 'unpool_relu_compact_kernel', 'unpool_rows_kernel']),
    ('compact_rows', ['compact_count_kernel', 'compact_place_kernel', 'compact_rows_serial_kernel']),
    ('loss', ['margin_loss_kernel']),
    ('adam', ['adam_kernel']),
]


def matches(kernel_name, frags):
    """a fragment names a kernel when it starts a C++ identifier of the demangled name ('pool_rows_kernel' must not claim
    'unpool_rows_kernel')"""
    for f in frags:
        i = kernel_name.find(f)
        while i >= 0:
            if i == 0 or not (kernel_name[i - 1].isalnum() or kernel_name[i - 1] == '_'):
                return True
            i = kernel_name.find(f, i + 1)
    return False


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            tot[r['Kernel_Name']] += float(r['Counter_Value']) * 1024.0
            cnt[r['Kernel_Name']] += 1
    return tot, cnt


def main(fetch_csv, write_csv, out, mfma_csv=None, *meta_args):
    rd, rc = per_kernel(fetch_csv, 'FETCH_SIZE')
    wr, wc = per_kernel(write_csv, 'WRITE_SIZE')
    res = {}
    for site, frags in SITES:
        names_r = [k for k in rd if matches(k, frags)]
        names_w = [k for k in wr if matches(k, frags)]
        # (a stream-K reduce kernel belongs to its GEMM's launch: its bytes count, its dispatches do not)
        nr = sum(rc[k] for k in names_r if 'reduce' not in k or 'reduce' in site)
        nw = sum(wc[k] for k in names_w if 'reduce' not in k or 'reduce' in site)
        if not nr or not nw:
            continue
        read = 2.0 * sum(rd[k] for k in names_r) / nr
        write = sum(wr[k] for k in names_w) / nw
        res[site] = int(read + write)
        res['_' + site] = {'read_bytes_per_launch': int(read), 'write_bytes_per_launch': int(write), 'launches_seen': nr,
                           'kernels': sorted(set(k[:60] for k in names_r))}
    if mfma_csv:
        mb, mc = per_kernel(mfma_csv, 'SQ_VALU_MFMA_BUSY_CYCLES')
        bb, bc = per_kernel(mfma_csv, 'SQ_BUSY_CYCLES')
        busy = {}
        for site, frags in SITES:
            names = [k for k in mb if matches(k, frags)]
            num = sum(mb[k] for k in names) / 1024.0          # per_kernel scales by 1024 (KB counters); undo
            den = sum(bb[k] for k in names) / 1024.0
            if den > 0 and num > 0:
                busy[site] = round(num / (32.0 * den), 4)
        res['_mfma_busy'] = busy
    res['_note'] = ('HBM-side bytes per site launch (read + write) from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE '
                    '(separate passes, KB per dispatch), FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies '
                    '128-B requests at 64 B); embed_* sites average their two launches per step (interaction + context '
                    'head) like roofline.achieved; split-K reduce kernels are listed as their own site')
    # the configuration these passes ran (bench.py attaches the numbers to a run only when it is the same one)
    meta = {'batch': 64, 'tracks': 16, 'ctx_clips': 18, 'fill': 'survey', 'gemm_mode': 2, 'feature_dtype': 'f32', 'compact': 1, 'layer1_planes': 1, 'storage': 'q32b'}
    for kv in (meta_args or []):
        k, v = kv.split('=', 1)
        meta[k] = int(v) if v.lstrip('-').isdigit() else v
    res['_meta'] = meta
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.startswith('_')}))


if __name__ == '__main__':
    main(*sys.argv[1:])
