"""Aggregate rocprofv3 --pmc csv output (p_counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv, glob, re, sys, collections
def short(n): return re.sub(r'\(.*$', '', n).replace('void ', '')[:70]
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in agg.items():
            if 'attn' in k or 'gemm' in k or 'Cijk' in k:
                print(f.split('/')[-3][-40:], k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, 'n=', len(next(iter(cs.values()))))
