#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp; O=$R/gpurun_out/prof_reortho; mkdir -p $O; cd /tmp
python3 $R/scripts/reortho_run.py > $O/run.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/scripts/reortho_run.py > $O/trace.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob('$O/trace/*/*_kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    with open('$O/kernel_stats.csv','w') as g:
        w=csv.writer(g); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','Percentage'])
        for r in rows: w.writerow([r['Name'][:110],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['Percentage']])
PY
rm -rf $O/trace; cat $O/run.log; head -9 $O/kernel_stats.csv
