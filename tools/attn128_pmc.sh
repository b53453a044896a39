#!/bin/bash
# Wave-cycle anatomy + LDS bank conflicts of the attention kernels on the bench block: two rocprofv3 --pmc passes (kernel-trace only) over tools/attn128_check.py --only bench
#     gpurun -- 'bash tools/attn128_pmc.sh [--bwd]'     -> gpurun_out/attn128_pmc.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
i=0
for set in "$P1" "$P2"; do
  i=$((i + 1))
  rm -rf $R/gpurun_out/attn128_pmc_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/attn128_pmc_$i -o p -- python $R/tools/attn128_check.py --only bench --no-time --out attn128_pmc_check.json "$@" > $R/gpurun_out/attn128_pmc_$i.log 2>&1
  find $R/gpurun_out/attn128_pmc_$i -name "*kernel_trace.csv" -delete
done
python - <<'PY' > $R/gpurun_out/attn128_pmc.txt
import csv, glob, collections, os
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(R + '/gpurun_out/attn128_pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'attn' not in k:
            continue
        k = k.split('(')[0].replace('void ', '') + ' grid ' + r.get('Grid_Size', '?')
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[k][r['Counter_Name']] += 1
for k in sorted(acc):
    c = {m: acc[k][m] / max(n[k][m], 1) for m in acc[k]}          # per launch
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    print(k)
    for m in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS'):
        if m in c:
            print(f'    {m:28s} {100 * c[m] / wc:6.1f} % of SQ_WAVE_CYCLES')
    w = c.get('SQ_WAVES', 0) or 1
    print(f'    SQ_WAVE_CYCLES per wave      {wc / w:12.0f} quad-cycles')
    for m in ('SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU'):
        if m in c:
            print(f'    {m:28s} {c[m] / w:10.1f} per wave')
    if 'SQ_LDS_IDX_ACTIVE' in c:
        print(f'    LDS bank-conflict share      {100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / (c["SQ_LDS_IDX_ACTIVE"] or 1):6.2f} % of LDS-active cycles')
    if 'SQ_BUSY_CYCLES' in c:
        print(f'    matrix pipe busy             {100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["SQ_BUSY_CYCLES"] or 1):6.1f} % of SQ_BUSY_CYCLES (gfx94x formula)')
PY
cat $R/gpurun_out/attn128_pmc.txt
