set -u
export TMPDIR=/tmp; REPO=$(pwd); OUT=$REPO/gpurun_out/r05k; mkdir -p $OUT; cd /tmp
for v in prod halfdots onestep noloads nostores prod; do
  if [ $v = prod ]; then unset KORNIA_HIP_LIB; else export KORNIA_HIP_LIB=$REPO/kornia-rs_amd/lib/libkornia_hip_abl_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$v -o kt -- python $REPO/bench.py --workload resize_u8_224 --no-cpu-baseline --also none > $OUT/$v.log 2>&1
  f=$(find $OUT/p_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'sep_h' in r['Name'] or 'sep_v' in r['Name']:
        print('%-44s calls %4s  avg %8.1f us  min %8.1f us' % (r['Name'].split('(anonymous namespace)::')[1][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
" "$f"
  rm -rf $OUT/p_$v
done 2>&1 | tee $OUT/sep_h_ablation.txt
