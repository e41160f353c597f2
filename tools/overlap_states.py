#!/usr/bin/env python3
"""Wall time spent with n extend and m shade kernels running at once, from rocprofv3 kernel traces (CSV):
python tools/overlap_states.py <..._kernel_trace.csv>...  (tools/ab_trace.sh writes such traces)"""
import csv, sys
def load(p):
    ev=[]
    for r in csv.DictReader(open(p)):
        n=r['Kernel_Name']
        k='E' if 'k_extend' in n else 'S' if 'k_shade' in n else None
        if k: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),k,r.get('Queue_Id') or r.get('Stream_Id')))
    ev.sort()
    return ev
for p in sys.argv[1:]:
    ev=load(p)
    # sweep: time in each state (#E running, #S running)
    pts=[]
    for s,e,k,q in ev:
        pts.append((s,1,k)); pts.append((e,-1,k))
    pts.sort()
    cnt={'E':0,'S':0}; last=None; state={}
    for t,d,k in pts:
        if last is not None and (cnt['E'] or cnt['S']):
            key=(cnt['E'],cnt['S']); state[key]=state.get(key,0)+t-last
        cnt[k]+=d; last=t
    tot=sum(state.values())
    print(p)
    for k in sorted(state): print('  E=%d S=%d  %.1f ms  %.1f%%'%(k[0],k[1],state[k]/1e6,100*state[k]/tot))
    print('  queues', sorted(set(q for *_,q in ev)))
