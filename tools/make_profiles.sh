#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun; results land in gpurun_out/profiles/):
#   1. the default bench line                          -> bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command -> kernel_stats.csv (+ kernel_trace.csv for the step timeline)
#   3. separate counter passes (FETCH_SIZE; WRITE_SIZE; MFMA busy) -> pmc_fetch.csv, pmc_write.csv, pmc_mfma.csv
#      (tools/traffic_from_pmc.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
timeout 600 python3 $B --no-cpu-baseline --no-dense --no-pcie --no-configs --steps 30 --warmup 10 > /dev/null 2>&1   # page the image in
timeout 900 python3 $B > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $B --no-cpu-baseline --no-profile --no-dense --no-pcie --no-configs --no-parity-check --steps 50 --warmup 10 > $O/kt.log 2>&1; echo "kernel-trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o pmc -- python3 $B --no-cpu-baseline --no-profile --no-dense --no-pcie --no-configs --no-parity-check --steps 5 --warmup 3 > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o pmc -- python3 $B --no-cpu-baseline --no-profile --no-dense --no-pcie --no-configs --no-parity-check --steps 5 --warmup 3 > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $O/pmc_mfma -o pmc -- python3 $B --no-cpu-baseline --no-profile --no-dense --no-pcie --no-configs --no-parity-check --steps 5 --warmup 3 > $O/pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
cp $O/pmc_mfma/pmc_counter_collection.csv $O/pmc_mfma.csv 2>/dev/null
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
cp $O/pmc_fetch/pmc_counter_collection.csv $O/pmc_fetch.csv 2>/dev/null
cp $O/pmc_write/pmc_counter_collection.csv $O/pmc_write.csv 2>/dev/null
python3 $R/tools/traffic_from_pmc.py $O/pmc_fetch.csv $O/pmc_write.csv $O/traffic.json $O/pmc_mfma.csv commit=${LIREC_COMMIT:-unknown} > $O/traffic.log 2>&1
rm -rf $O/kt/*kernel_trace.csv $O/pmc_fetch $O/pmc_write $O/pmc_mfma
ls -la $O
