"""Summarise a rocprofv3 --pmc run of tools/lds_b128_probe.hip: SQ_LDS_BANK_CONFLICT / SQ_LDS_ACTIVE per probe kernel (profiles/r04_lds_b128_conflict_probe.txt)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
names = {'<16, 1, false>': 'b128_linear  (conflict-free by construction)', '<8, 1, false>': 'b64_linear', '<4, 1, false>': 'b32_linear', '<16, 2, false>': 'b128_stride2 (2-way by construction)', '<16, 1, true>': 'b128_same    (broadcast)',
         'Li16ELi1ELb0': 'b128_linear  (conflict-free by construction)', 'Li8ELi1ELb0': 'b64_linear', 'Li4ELi1ELb0': 'b32_linear', 'Li16ELi2ELb0': 'b128_stride2 (2-way by construction)', 'Li16ELi1ELb1': 'b128_same    (broadcast)'}
print('# rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- tools/lds_b128_probe (MI355X): what the counter the convolution PMC tables')
print('# quote as "conflicts" reads for access patterns whose conflicts are known by construction')
print('%-48s %18s %18s %10s' % ('pattern', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'ratio'))
if not any(key in k for k in acc for key in names):
    print('kernel names seen:', list(acc)[:8])
for k, m in sorted(acc.items()):
    for key, nm in names.items():
        if key in k:
            c, a = m.get('SQ_LDS_BANK_CONFLICT', 0), m.get('SQ_LDS_IDX_ACTIVE', 0)
            print('%-48s %18.0f %18.0f %9.1f%%' % (nm, c, a, 100 * c / a if a else float('nan')))
