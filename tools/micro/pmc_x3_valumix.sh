set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/probe_x3_rounds_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=$R/tools/micro/bin/x3_engine_rounds_noph
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 -d $O/valumix --output-format csv -- $P 40 256 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
res = collections.OrderedDict()
for f in glob.glob('$O/valumix/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        res.setdefault(r['Kernel_Name'], collections.OrderedDict())[r['Counter_Name']] = float(r['Counter_Value'])
for k, v in res.items():
    print(k); print('   ' + '  '.join('%s=%.4g' % (n, x) for n, x in v.items()))
PY
