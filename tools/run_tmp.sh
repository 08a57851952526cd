mkdir -p gpurun_out/r05e
Q="--no-cpu-baseline --no-configs --no-pcie --no-strict --no-eval --no-dense --no-parity-check"
for rep in 1 2; do
for v in "A" "B --ablate 32768"; do
  set -- $v; tag=$1; shift
  timeout 300 python bench.py $Q "$@" > gpurun_out/r05e/b_$tag$rep.json 2> gpurun_out/r05e/b_$tag$rep.err
done; done
python - <<'PY'
import json,glob
for t in 'AB':
    for f in sorted(glob.glob('gpurun_out/r05e/b_%s?.json'%t)):
        d=json.load(open(f)); k=d['kernels']
        print(t, d['ms_per_step'], {n: (round(1e3*k[n]['avg_ms'],1), round(1e3*k[n].get('alone_avg_ms',0),1)) for n in ('gate_dW','embed_dW1','embed_dW1_reduce','adam','pool_bwd','embed_dZ1')})
PY
