import sys; sys.path.insert(0,'/root/repo')
import numpy as np, aero_amd
from tests import oracle_lib
orc=oracle_lib.load(); ctx=aero_amd.Context(0)
for (log_n,W,kw) in [(3,2,dict(fri_log_max_remainder=3,grinding_factor=8,num_queries=8)),(6,2,dict(fri_log_max_remainder=5,grinding_factor=8)),(10,2,{})]:
    o=aero_amd.ProofOptions.with_96_bit_security()
    for k,v in kw.items(): setattr(o,k,v)
    got,pub=ctx.prove_fib(aero_amd.fib_trace(W,log_n),o)
    want,_,_=orc.prove_fib(W,log_n,o.to_list())
    diff=[i for i in range(min(len(got),len(want))) if got[i]!=want[i]]
    print(log_n,W,len(got),len(want),'ndiff',len(diff),diff[:5])
    try: orc.verify(got,pub,air_kind=1,W=W,log_n=log_n); print(' verify ok')
    except Exception as e: print(' verify:',e)
