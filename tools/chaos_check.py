import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests import test_gpu_dp as D
def run(mode): 
    l, p, sd, opt = D._one_process_steps(mode, 5, False)
    return l, p
le, pe = run("eager"); le2, pe2 = run("eager"); lg, pg = run("graph")
d = (pe - pe2).abs(); print("eager vs eager: mean %.3e max %.3e first %s" % (d.mean().item(), d.max().item(), d[:3].tolist()))
d = (pg - pe).abs(); print("graph vs eager: mean %.3e max %.3e first %s" % (d.mean().item(), d.max().item(), d[:3].tolist()))
print("losses e", le, "g", lg)
