import sqlite3,collections,sys,glob
f=glob.glob(sys.argv[1]+'/*.db')[0]
c=sqlite3.connect(f)
rows=list(c.execute("select name, start, end, grid_x from kernels order by start"))
g=collections.defaultdict(lambda:[0,0.0])
for n,s,e,gx in rows:
    g[(n[:60],gx)][0]+=1; g[(n[:60],gx)][1]+=(e-s)/1e3
tot=sum(v[1] for v in g.values())
print('total ms per stack', tot/int(sys.argv[2])/1e3)
for k,v in sorted(g.items(), key=lambda kv:-kv[1][1]):
    if any(w in k[0] for w in sys.argv[3:]): print('%-62s grid %9d n=%3d avg %.1f us'%(k[0],k[1],v[0],v[1]/v[0]))
