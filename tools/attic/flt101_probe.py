import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import binding as orc
from trgt_amd.wfaligner import flank_filter_batch
from test_filter_gpu import _flank_jobs
rng = np.random.default_rng(11)
pats, txts = _flank_jobs(rng, 120)
r = flank_filter_batch(pats, txts, 175, scoring=(1, 0, 1))
bad = 0; cells = 0
for j in range(len(pats)):
    pp = orc.wfa_params(metric="affine", x=1, o1=0, e1=1, span="endsfree", pbf=0, pef=0, tbf=len(txts[j]), tef=len(txts[j]), heuristic="none")
    o = orc.wfa_align(pp, pats[j], txts[j]); cells += o["cells"]
    if int(r["score"][j]) != o["score"] or int(r["bound"][j]) < o["n_match"]:
        bad += 1
        if bad < 12: print(j, len(pats[j]), len(txts[j]), "gpu", int(r["score"][j]), int(r["bound"][j]), int(r["keep"][j]), "oracle", o["score"], o["n_match"], o["status"], pats[j][:12], txts[j][:12])
print("bad", bad, "offsets gpu", r["offsets"], "oracle", cells)
