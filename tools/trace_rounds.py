"""Per-round durations of k2_estep / k2_mstep for one 512-UTR headline batch (stderr lines `[trace] kind 4|5 ...`)
and the histogram of EM round counts.  Run on the GPU box: python tools/trace_rounds.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scape_amd.engine import Engine, HipBatch
from scape_amd.host import prepare_utr
from scape_amd.synth import synth_utr
U=512
preps=[]
for i in range(U):
    g,df,_=synth_utr(i,2000,k_cap=10,base_seed=20250225)
    preps.append(prepare_utr(df,gene_info_str=g,n_max_apa=10,n_min_apa=1))
eng=Engine(0)
plan=eng.plan(preps,[(20250225+i)%2**32 for i in range(U)])
b=HipBatch(eng.ctx,preps)
eng.process(b,preps,plan,False)
os.environ["SCAPE_HIP_ROUND_TIMING"]="1"; os.environ["SCAPE_HIP_ROUND_TRACE"]="1"
if len(sys.argv) > 1 and sys.argv[1] == "bytes":
    os.environ["SCAPE_HIP_DEBUG"]="1"       # also: the M-step's cumulative tensor bytes after every round (synchronises each round)
eng.ctx.lib.scape_hip_timing_reset(eng.ctx.h)
b.build(); out=b.em_packed(plan["main"])
eng.ctx.lib.scape_hip_timing_reset(eng.ctx.h)
nlb=out[4]
print("rounds hist", np.bincount(nlb, minlength=51).tolist())
