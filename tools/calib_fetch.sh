R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/calib; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/calib_fetch.py > /dev/null 2>&1
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u > $O/tcc_counters.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o c -- python3 $R/tools/calib_fetch.py > $O/f.log 2>&1
python3 $R/tools/calib_fetch_report.py $O/f/c_counter_collection.csv > $O/fetch.txt 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace --output-format csv -d $O/g -o c -- python3 $R/tools/calib_fetch.py > $O/g.log 2>&1
python3 $R/tools/calib_fetch_report.py $O/g/c_counter_collection.csv > $O/rdreq.txt 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/h -o c -- python3 $R/tools/calib_fetch.py > $O/h.log 2>&1
python3 $R/tools/calib_fetch_report.py $O/h/c_counter_collection.csv > $O/hit.txt 2>&1
rm -rf $O/f $O/g $O/h
