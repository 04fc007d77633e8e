import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find last k_sphere_trace index -> print one step from the one before
idx=[i for i,r in enumerate(rows) if 'k_sphere_trace' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
prev_end=t0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%8.1f %8.1f  gap %6.1f  q%s  %s'%((s-t0)/1e3,(e-s)/1e3,(s-prev_end)/1e3,r.get('Queue_Id','?'),r['Kernel_Name'][:60]))
    prev_end=max(prev_end,e)
print('step span us', (int(rows[b]['Start_Timestamp'])-t0)/1e3, 'launches', b-a)
