# PMC passes over the engine phase probe (x3_engine_rounds, -DNO_PH build), one counter group per run (GPU box):  bash tools/micro/pmc_x3_rounds.sh [rounds] [wgs]
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/probe_x3_rounds_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=$R/tools/micro/bin/x3_engine_rounds_noph
A="${1:-40} ${2:-256}"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/lds --output-format csv -- $P $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum -d $O/tcp --output-format csv -- $P $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA -d $O/issue --output-format csv -- $P $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQC_ICACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/icache --output-format csv -- $P $A > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
res = collections.OrderedDict()
for g in ('lds', 'tcp', 'issue', 'icache'):
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % g, recursive=True):
        for r in csv.DictReader(open(f)):
            res.setdefault(r['Kernel_Name'], collections.OrderedDict())[r['Counter_Name']] = float(r['Counter_Value'])   # (the later of the two repetitions wins)
for k, v in res.items():
    print(k)
    print('   ' + '  '.join('%s=%.4g' % (n, x) for n, x in v.items()))
PY
find $O -name "*.csv" -size +2M -delete
