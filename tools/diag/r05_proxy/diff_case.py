import sys, subprocess, os, re
pkl=sys.argv[1]; idx=sys.argv[2]
def run(which):
    env=dict(os.environ, EXP_TRACE='1')
    p=subprocess.run([sys.executable,'/tmp/trace_case.py',pkl,idx,which],env=env,capture_output=True,text=True)
    tr=p.stderr.split('BEGIN\n')[-1]
    subs=[]
    for line in tr.splitlines():
        if line.startswith('SUB'): subs.append([line,[]])
        elif line.startswith(' C'): subs[-1][1].append(line)
    return subs,p.stdout
A,oa=run(sys.argv[3] if len(sys.argv)>3 else 'f64'); B,ob=run('f32')
def parse(c):
    d=dict(re.findall(r'(\w+)=(-?[\d.e+-]+(?: -?[\d.e+-]+)*)',c))
    return int(d['a']),int(d['b']),float(d['dist']),[float(x) for x in d['n'].split()],[float(x) for x in d['pa'].split()]
for s,(x,y) in enumerate(zip(A,B)):
    ca=[parse(c) for c in x[1]]; cb=[parse(c) for c in y[1]]
    bad=None
    if [(c[0],c[1]) for c in ca]!=[(c[0],c[1]) for c in cb]: bad='structure'
    else:
        for p,q in zip(ca,cb):
            if max(abs(u-v) for u,v in zip(p[3],q[3]))>float(os.environ.get("TN","1e-4")): bad="normal"; break
            if abs(p[2]-q[2])>float(os.environ.get("TD","2e-5")): bad="dist"; break
            if max(abs(u-v) for u,v in zip(p[4],q[4]))>1e-3: bad='point'; break
    if bad:
        print('case',idx,'first divergence at substep',s,bad)
        print(x[0]); print('\n'.join(x[1])); print('--- f32'); print(y[0]); print('\n'.join(y[1]))
        break
else: print('case',idx,'no contact-list divergence')
