#!/bin/bash
# gpurun_out/profiles/* (written by tools/make_profiles.sh on the GPU box) -> profiles/rNN_* (tracked):  bash tools/copy_profiles.sh r03
R=${1:-r04}; S=gpurun_out/profiles; D=profiles
cp $S/bench.json $D/bench_$R.json
cp $S/kernel_stats.csv $D/${R}_kernel_stats.csv; cp $S/kernel_stats_eval.csv $D/${R}_kernel_stats_eval.csv
cp $S/pmc_fetch.csv $D/${R}_pmc_fetch_size.csv; cp $S/pmc_write.csv $D/${R}_pmc_write_size.csv; cp $S/pmc_mfma.csv $D/${R}_pmc_mfma_busy.csv
cp $S/pmc_gemm_pipes.txt $D/${R}_pmc_gemm_pipes.txt; cp $S/traffic.json $D/traffic.json
cp $S/parity_errors.json $D/${R}_parity_errors.json; cp $S/ablate_planes.txt $D/${R}_ablate_planes.txt
cp $S/calib_fetch.txt $D/${R}_calib_fetch.txt; cp $S/launch_modes.txt $D/${R}_launch_modes.txt
cp $S/tile_order_fetch.txt $D/${R}_tile_order_fetch.txt; cp $S/step_trace.csv $D/${R}_step_trace.csv
cp $S/pieces_kernel_stats.csv $D/${R}_pieces_kernel_stats.csv; cp $S/pieces_pmc_FETCH_SIZE.txt $D/${R}_pieces_pmc_fetch.txt
cp $S/pieces_pmc_WRITE_SIZE.txt $D/${R}_pieces_pmc_write.txt; cp $S/pieces_sites.txt $D/${R}_pieces_sites.txt
cp $S/training_entry.txt $D/${R}_training_entry.txt; cp $S/mode3_sites.txt $D/${R}_mode3_sites.txt
cp $S/l2_lds_intake.txt $D/${R}_l2_lds_intake.txt; cp $S/p3_bench.txt $D/${R}_p3_bench.txt; cp $S/p2_bench.txt $D/${R}_p2_bench.txt
cp $S/p2x_bench.txt $D/${R}_p2x_bench.txt; cp $S/p2o_bench.txt $D/${R}_p2o_bench.txt
python3 - $S $D/${R}_trajectory_and_onepass.json <<'P'
import glob, json, sys
out = {}
for f in sorted(glob.glob(sys.argv[1] + '/trajectory_*.json') + glob.glob(sys.argv[1] + '/onepass_*.json')):
    out[f.split('/')[-1][:-5]] = json.load(open(f))
json.dump(out, open(sys.argv[2], 'w'), indent=1)
P
ls $D | wc -l
