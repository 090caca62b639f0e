import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'stream_gather_kernel' in r['Kernel_Name'] and ', 1>' in r['Kernel_Name']]
steps=[(a,c) for a,b,c in zip(idx,idx[1:],idx[2:]) if c-a>25 and b-a<12]
i0,i1=steps[len(steps)//2]
out=[]
for r in rows[i0:i0+3]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.append('%s %.1f'%(r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:28],(e-s)/1e3))
print(sys.argv[1], ' | '.join(out), 'span %.1f'%((int(rows[i1]['Start_Timestamp'])-int(rows[i0]['Start_Timestamp']))/1e3))
