import numpy as np, sys
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
la,lb=a["log"],b["log"]
T,bs,_=la.shape
print("T",T,"bs",bs, "mean hits base %.3f new %.3f" % (la[:,:,0].clip(0).mean(), lb[:,:,0].clip(0).mean()))
nset=nord=0
first=None
for s in range(T):
    for e in range(bs):
        na,nb_=la[s,e,0],lb[s,e,0]
        ea=la[s,e,1:1+max(na,0)]; eb=lb[s,e,1:1+max(nb_,0)]
        if na!=nb_ or not np.array_equal(ea,eb):
            if sorted(ea.tolist())==sorted(eb.tolist()): nord+=1
            else:
                nset+=1
                if first is None: first=(s,e,[hex(x) for x in ea],[hex(x) for x in eb])
print("env-steps with a different hit SET:",nset," same set, different ORDER:",nord, " first set difference:", first)
for s in range(T):
    ta,tb=a["traj"][s],b["traj"][s]
    if not np.array_equal(ta,tb):
        d=np.abs(ta-tb); i=np.unravel_index(d.argmax(),d.shape); print("first traj difference at step",s,"max",d.max(),"at",i); break
else: print("trajectories bit-identical")
