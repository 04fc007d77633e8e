import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0].replace('void ','')
    acc[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k in sys.argv[2:]:
    for n,v in acc.items():
        if k in n: print('%-40s n=%3d mean %.1f us'%(n[:40],len(v),sum(v)/len(v)))
