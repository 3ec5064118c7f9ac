"""Plan lifetime check: free device memory after LEAK_ITERS x 6 create / exec / destroy cycles (run twice with different
counts: the delta must not grow). usage: LEAK_ITERS=600 python tools/check_leaks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, __graft_entry__ as g
g.build()
import tensor_fft_amd as tf
free0, _ = torch.cuda.mem_get_info()
x = (torch.rand(8 * 2 * 65536, device="cuda") * 2 - 1).half(); y = torch.empty_like(x)
import os
for i in range(int(os.environ.get("LEAK_ITERS", "300"))):
    for n, b in ((256, 8), (1024, 8), (4096, 8), (8192, 8), (65536, 8), (1 << 17, 4)):
        p = tf.TfftPlan(n, b, 0)
        p.exec(x, x[n:], y, y[n:])
        del p
    if i % 50 == 0:
        p2 = tf.TfftPlan2D(256, 256, 2, 0); re = x[:2*65536]; p2.exec(re, re, y[:2*65536], y[2*65536:4*65536]); del p2
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
print("free before %.1f MiB, after %.1f MiB, delta %.1f MiB" % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
