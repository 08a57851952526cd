"""Carry the numbers of a new profiles/bench_r04.json (+ traffic.json) into DESIGN.md / README.md: every figure the documents quote
from the PREVIOUS bench file (git HEAD's copy) is replaced by the new file's.  Usage: python tools/refresh_docs.py [old.json]"""
import json
import re
import subprocess
import sys


def cfgmap(x):
    return {c['config'].split(':')[0]: c for c in (x.get('configs') or [])}


def main():
    if len(sys.argv) > 1:
        o = json.load(open(sys.argv[1]))
    else:
        o = json.loads(subprocess.check_output(['git', 'show', 'HEAD:profiles/bench_r04.json']))
    d = json.load(open('profiles/bench_r04.json'))
    t = json.load(open('profiles/traffic.json'))
    busy = t.get('_mfma_busy', {})
    co, cn = cfgmap(o), cfgmap(d)
    s, r = open('DESIGN.md').read(), open('README.md').read()
    missing = []

    def rep(txt, a, b):
        if a == b:
            return txt
        if a not in txt:
            missing.append(a[:70])
            return txt
        return txt.replace(a, b)
    ov, om, nv, nm = o['value'] / 1e3, o['ms_per_step'], d['value'] / 1e3, d['ms_per_step']
    s = rep(s, f'**{ov:.1f} k clips/s, {om:.3f} ms per step**', f'**{nv:.1f} k clips/s, {nm:.3f} ms per step**')
    s = rep(s, f"{o['roofline']['avg_launch_ms'] * 1e3:.1f} µs per launch in the timed step", f"{d['roofline']['avg_launch_ms'] * 1e3:.1f} µs per launch in the timed step")
    s = rep(s, f"`frac` {o['roofline']['frac']:.3f} of the 2.5 PF dense bf16 peak", f"`frac` {d['roofline']['frac']:.3f} of the 2.5 PF dense bf16 peak")
    s = rep(s, f'against {om:.3f} ms for the step', f'against {nm:.3f} ms for the step')
    s = rep(s, f'the step {om:.3f} ms)', f'the step {nm:.3f} ms)')
    # the kernel table: in step / alone / frac columns, MFMA-busy and counter bytes
    K = d['kernels']
    lines = s.split('\n')
    for i, l in enumerate(lines):
        m = re.match(r'\| `([a-zA-Z0-9_]+)`( ×\d)? \| (.*?) \| [\d.]+ \| [\d.]+ \| (hbm|mfma) \| [\d.]+ \| (.*?) \| (.*?) \|$', l)
        if m and m.group(1) in K:
            v = K[m.group(1)]
            tr, b = t.get(m.group(1)), busy.get(m.group(1))
            lines[i] = '| `%s`%s | %s | %.1f | %.1f | %s | %.3f | %s | %s |' % (
                m.group(1), m.group(2) or '', m.group(3), v['avg_ms'] * 1e3, (v.get('alone_avg_ms') or 0) * 1e3, m.group(4), v['frac'],
                ('%.0f MB' % (tr / 1e6)) if isinstance(tr, (int, float)) else m.group(5), ('%.2f' % b) if b else m.group(6))
    s = '\n'.join(lines)
    ev = d['eval']
    summ = (f"{nv:.1f} k clips/s ({nm:.3f} ms); with the input pipeline {d['input_pipeline']['value'] / 1e3:.1f} k; on q32b storage {d['q32_storage']['value'] / 1e3:.1f} k; "
            f"eval {ev['value'] / 1e3:.0f} k (q32b storage: {ev['q32_storage']['value'] / 1e3:.0f} k; fp32 block staged for one use: {ev['fp32_staged']['value'] / 1e3:.0f} k); "
            f"dense fill {d['dense_fill']['value'] / 1e3:.1f} k; exact-f32 core {d['strict_f32']['value'] / 1e3:.1f} k; "
            f"configs (train legs in the recorded launch form): 1 → {cn['1']['value'] / 1e6:.2f} M clips/s (`frac` {cn['1']['roofline']['frac']:.3f}; `1q`, rows stored as q32b: "
            f"{cn['1q']['value'] / 1e6:.2f} M, {cn['1q']['roofline']['frac']:.3f}), 3 → {cn['3']['value'] / 1e3:.0f} k, "
            f"4 → {cn['4']['value'] / 1e3:.1f} k / {cn['4']['ms_per_step']:.2f} ms (row-major bf16 block, staged as q16b; `4q`, stored as q16b: {cn['4q']['value'] / 1e3:.1f} k / {cn['4q']['ms_per_step']:.2f} ms), "
            f"4c → {cn['4c']['value'] / 1e3:.1f} k, 4b → {cn['4b']['value'] / 1e3:.1f} k (`4bq`: {cn['4bq']['value'] / 1e3:.1f} k); "
            f"`cpu_baseline` {d['cpu_baseline']['value']:.1f} clips/s train / {d['cpu_baseline']['eval_value']:.1f} eval on {d['cpu_baseline']['cores']} threads.  The whole default run takes ~1 minute on the box.")
    i = s.index('**[measured]** `profiles/bench_r04.json`: ')
    j = s.index('\n', i)
    s = s[:i] + '**[measured]** `profiles/bench_r04.json`: ' + summ + s[j:]
    s = rep(s, f"(`input_pipeline` leg: {o['input_pipeline']['value'] / 1e3:.1f} k)", f"(`input_pipeline` leg: {d['input_pipeline']['value'] / 1e3:.1f} k)")
    s = rep(s, f"own leg {o['q32_storage']['value'] / 1e3:.1f} k clips/s", f"own leg {d['q32_storage']['value'] / 1e3:.1f} k clips/s")
    s = rep(s, f"explicit, opt-in storage: {o['q32_storage']['value'] / 1e3:.1f} k clips/s", f"explicit, opt-in storage: {d['q32_storage']['value'] / 1e3:.1f} k clips/s")
    eo = o['eval']
    r = rep(r, f"{ov:.1f} k clips/s on the SURVEY generator's ragged masks", f"{nv:.1f} k clips/s on the SURVEY generator's ragged masks")
    r = rep(r, f"{o['input_pipeline']['value'] / 1e3:.1f} k with the next batch's rows staged beside the step, {o['q32_storage']['value'] / 1e3:.1f} k when",
            f"{d['input_pipeline']['value'] / 1e3:.1f} k with the next batch's rows staged beside the step, {d['q32_storage']['value'] / 1e3:.1f} k when")
    r = rep(r, f"{o['dense_fill']['value'] / 1e3:.1f} k with every mask entry valid, {eo['value'] / 1e3:.0f} k clips/s for the evaluation loop body ({eo['q32_storage']['value'] / 1e3:.0f} k on q32b storage), {o['strict_f32']['value'] / 1e3:.1f} k with the exact f32-MFMA core;",
            f"{d['dense_fill']['value'] / 1e3:.1f} k with every mask entry valid, {ev['value'] / 1e3:.0f} k clips/s for the evaluation loop body ({ev['q32_storage']['value'] / 1e3:.0f} k on q32b storage), {d['strict_f32']['value'] / 1e3:.1f} k with the exact f32-MFMA core;")
    r = rep(r, f"The CPU oracle does {o['cpu_baseline']['value']:.0f} (train) / {o['cpu_baseline']['eval_value']:.0f} (eval) clips/s",
            f"The CPU oracle does {d['cpu_baseline']['value']:.0f} (train) / {d['cpu_baseline']['eval_value']:.0f} (eval) clips/s")
    open('DESIGN.md', 'w').write(s)
    open('README.md', 'w').write(r)
    print('not found (left as they are):', missing)


if __name__ == '__main__':
    main()
