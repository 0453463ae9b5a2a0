#!/bin/bash
# PMC passes over the query kernels of the resident index (config 5's text and query set); separate runs,
# --kernel-trace only.  Output: gpurun_out/qpmc/*.csv reduced to the query kernels' rows.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD; OUT=$ROOT/gpurun_out/qpmc; mkdir -p $OUT; export TMPDIR=/tmp
export PROBE_SETS=survey_8d,text_bytes_len6 PROBE_INDEX_ONLY=1
CMD="python $ROOT/scripts/gpu_query_probe.py ${1:-1000000000}"
cd /tmp
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $CMD > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE
cd $ROOT
for d in tcp tcc sq; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' > $OUT/$d.summary.txt
import csv, sys, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "query" not in k: continue
    key = (r["Dispatch_Id"], k.split("(")[0][-40:])
    rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k), v in rows.items():
    print(d, k, " ".join(f"{a}={b:.4g}" for a, b in v.items()))
PY
  cat $OUT/$d.summary.txt; tail -2 $OUT/$d.log
done
find $OUT -name "*.csv" -size +2M -delete
