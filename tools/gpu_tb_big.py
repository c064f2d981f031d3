"""Developer tool (GPU box): the traceback pass on 20 000 pairs of the bundled-dataset stand-in, once with the default scratch
(one or a few passes) and once with 1 GiB (many passes); 300 sampled pairs are compared with the oracle's walk."""
import sys, time; sys.path.insert(0,'.')
import numpy as np, agatha_amd
from agatha_amd import workload as W
from oracle import oracle as O
qs,ts=W.cfg_c0(n=20000)
qb,qo,ql=W.make_batch(qs); tb,to,tl=W.make_batch(ts)
eng=agatha_amd.Engine(0)
b=eng.batch(qb,tb,qo,to,ql,tl); b.upload(); b.pack(); eng.synchronize()
sc=agatha_amd.Scores.make(m=1,x=4,q=6,r=2,w=751)
for cap in (None, 1<<30):
    t0=time.time(); s,qe,te,c=b.align_traceback(sc, scratch_bytes=cap); dt=time.time()-t0
    print('scratch',cap,'time %.2f s'%dt,'no path',sum(x is None for x in c),'mean ops',np.mean([len(x) for x in c if x]))
    idx=np.random.default_rng(1).choice(20000,300,replace=False)
    es,eq,et,ec=O.traceback_pairs([qs[i] for i in idx],[ts[i] for i in idx],O.make_params(m=1,x=4,q=6,r=2,w=751),threads=16)
    assert all(s[i]==es[k] and qe[i]==eq[k] and te[i]==et[k] and c[i]==ec[k] for k,i in enumerate(idx)), 'mismatch'
    print('300 sampled pairs identical to the oracle')
