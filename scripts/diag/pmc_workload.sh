#!/bin/bash
# Dev: SQ / LDS counters for one bench workload.   bash scripts/diag/pmc_workload.sh <workload> <tag> [counter groups...]
set -u
WL=$1; TAG=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$(pwd); export TMPDIR=/tmp
GROUPS_=("$@"); [ ${#GROUPS_[@]} -eq 0 ] && GROUPS_=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM")
cd /tmp
i=0
for g in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $g -d "$REPO/$OUT/pmc_$i" -o pmc -- python "$REPO/bench.py" --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --also none > "$REPO/$OUT/pmc_$i.log" 2>&1
  db=$(find "$REPO/$OUT/pmc_$i" -name '*.db' | head -1)
  [ -n "$db" ] && python "$REPO/scripts/rocpd_summary.py" "$db" | grep -v rocclr | sed -n '/counter,mean/,$p' >> "$REPO/$OUT/pmc_$WL.txt"
done
python - "$REPO/$OUT/pmc_$WL.txt" <<'PY'
import re, sys
for line in open(sys.argv[1]):
    m = re.match(r'"(.*)",(\w+),([\d.]+),(\d+)', line.strip())
    if m:
        name = re.sub(r"\(.*", "", m.group(1).replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))
        print(f"  {name:44s} {m.group(2):24s} {float(m.group(3)):18.1f}")
PY
find "$REPO/$OUT" -name '*.db' -delete
