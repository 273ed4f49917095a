import sys
lines=open('gpurun_out/k4_trace.txt').read().split('\n')
waves={};cur=None
for l in lines:
    if l.startswith('wave'): cur=int(l.split()[1]); waves[cur]=[]
    elif l.strip(): waves[cur].append(int(l))
W=8
for w in range(W):
    ev=waves.get(W+w,[])
    if w==0: 
        print('w0 block0', [ev[i+1]-ev[i] for i in range(6)]); ev=ev[7:]
    for i in range(len(ev)//9):
        e=ev[9*i:9*i+9]
        print('w%d blk%d t=%d opwait %d upd %d | solve q0 %d q1 %d q2 %d q3 %d | pub %d'%(w,i,e[0],e[1]-e[0],e[2]-e[1],e[4]-e[3],e[5]-e[4],e[6]-e[5],e[7]-e[6],e[8]-e[7]))
