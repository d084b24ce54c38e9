import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads(), flush=True)
from parity_common import *
sc, sd_np, sd, osc = setup()
for th in [int(a) for a in sys.argv[1:]]:
    torch.set_num_threads(th)
    b, inp = oracle_input(sc, sd, [1], 8, 8)
    t0 = time.time()
    oo = ho.holdnet_forward(osc, sd, inp, False)
    print("threads", th, "eval fwd 64 rays: %.2f s" % (time.time() - t0), flush=True)
