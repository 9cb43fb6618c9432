#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp; O=$R/gpurun_out/calib; mkdir -p $O; cd /tmp
$R/scripts/calib_stream
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_REQ_sum --output-format csv -d $O/ea -- $R/scripts/calib_stream > $O/ea.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fs -- $R/scripts/calib_stream > $O/fs.log 2>&1
python3 - <<PY
import csv,glob
for d in ['ea','fs']:
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            print(d, r['Kernel_Name'][:60], r['Counter_Name'], r['Counter_Value'])
PY
