#!/bin/bash
# PMC passes over scripts/bench_encode_fwd.py (the encode forward alone): bash scripts/pmc_encode_fwd.sh <outdir> ; env PAG_LIB_VARIANT / PAG_NO_FAST_ENCODE select the kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/pmc_enc}
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE TCC_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 scripts/bench_encode_fwd.py > $out/p$i.log 2>&1 || echo "fail $i"
done
python3 - <<PY
import csv, glob, collections
tot=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "permuto_fwd" in k:
            tot[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in tot.items():
    print(k)
    for c,vals in sorted(v.items()):
        print("   %-40s n=%3d mean %.4g" % (c, len(vals), sum(vals)/len(vals)))
PY
