set -u
export TMPDIR=/tmp; REPO=$(pwd); OUT=$REPO/gpurun_out/r05h; mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|translation" | cut -c1-160 | sort -u | head -40 > $OUT/utcl_counters.txt
for i in 1 2 3 4 5 6; do
  timeout 300 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d $OUT/p$i -o pmc -- python $REPO/bench.py --workload gaussian_4k --steps 5 --warmup 3 --no-cpu-baseline --also none > $OUT/run$i.log 2>&1
  ms=$(grep '^{' $OUT/run$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'])")
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  python - "$f" "$ms" "$i" <<'PY' | tee -a $OUT/tlb_vs_time.txt
import csv, sys, collections
f, ms, i = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if 'sep_roll4' in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print(f"process {i}: {ms} ms/step  " + "  ".join(f"{k}={v[0]/max(v[1],1):.0f}" for k, v in sorted(acc.items())))
PY
  rm -rf $OUT/p$i
done
cd $REPO
for i in 1 2 3 4; do python bench.py --workload gaussian_4k --steps 5 --warmup 3 --no-cpu-baseline --also none 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('plain process', j['ms_per_step'])" | tee -a $OUT/tlb_vs_time.txt; done
