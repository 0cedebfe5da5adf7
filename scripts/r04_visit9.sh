#!/bin/bash
# Round-4 visit 9: LDS row pitch of the staged u8 gather (test option warp_u8_lds_pitch; scripts/diag/lds_bank_sim.py predicts the
# tap reads at 2.5x their conflict-free cycles with the box pitch and 2.0x with a pitch of 0 mod 32), interleaved rounds + LDS counters.
set -u
TAG=${1:-r04x}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
timeout 600 python -m pytest tests/test_u8_gpu.py tests/test_dev_options_gpu.py -q -x 2>&1 | tail -2 | tee "$OUT/pytest.log"
for r in 1 2; do
  for v in ${MODS:--1 1 6 4}; do
    echo "round $r warp_u8_lds_pitch=$v" | tee -a "$OUT/lds_pitch_ab.txt"
    timeout 300 python bench.py --workload warp_affine_u8_4k --no-cpu-baseline --also warp_perspective_u8_4k,remap_u8_4k --dev-option warp_u8_lds_pitch=$v 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/lds_pitch_ab.txt"
  done
done
for v in -1 1; do
  bash scripts/diag/pmc_cmd.sh $TAG/lds_$v "python $REPO/bench.py --workload warp_affine_u8_4k --steps 2 --warmup 1 --no-cpu-baseline --also warp_perspective_u8_4k,remap_u8_4k --dev-option warp_u8_lds_pitch=$v" \
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
    "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU" 2>&1 | tail -8 | cut -c1-600
  cp "$OUT/lds_$v/pmc_table.txt" "$OUT/lds_${v}_counters.csv" 2>/dev/null
done
