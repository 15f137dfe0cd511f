import sys, subprocess, collections, os
pkl=sys.argv[1]; n=int(sys.argv[2]); total=int(sys.argv[3])
import concurrent.futures as cf
def one(i):
    p=subprocess.run([sys.executable,'/tmp/diff_case.py',pkl,str(i)]+sys.argv[4:5],capture_output=True,text=True)
    out=p.stdout
    head=out.splitlines()[0] if out else 'none'
    # find first differing contact line pair
    lines=out.splitlines()
    if '--- f32' not in lines: return (i,head,None)
    k=lines.index('--- f32')
    A=[l for l in lines[2:k]]; B=[l for l in lines[k+2:]]
    import re
    def key(l):
        d=dict(re.findall(r'(\w+)=(-?\d+)\b',l)); return (d.get('a'),d.get('b'))
    ka=[key(l) for l in A]; kb=[key(l) for l in B]
    if ka!=kb: return (i,head,('structure',tuple(sorted(set(ka)^set(kb)))))
    for x,y in zip(A,B):
        nx=[float(v) for v in re.search(r'n=(\S+ \S+ \S+)',x).group(1).split()]
        ny=[float(v) for v in re.search(r'n=(\S+ \S+ \S+)',y).group(1).split()]
        dx=float(re.search(r'dist=(\S+)',x).group(1)); dy=float(re.search(r'dist=(\S+)',y).group(1))
        if max(abs(u-v) for u,v in zip(nx,ny))>float(os.environ.get("TN","1e-4")) or abs(dx-dy)>float(os.environ.get("TD","2e-5")): return (i,head,('value',key(x)))
    return (i,head,('point',None))
idx=list(range(0,total,max(1,total//n)))
cnt=collections.Counter()
with cf.ThreadPoolExecutor(8) as ex:
    for i,head,c in ex.map(one,idx):
        cnt[str(c)]+=1
for k,v in cnt.most_common(): print(v,k)
