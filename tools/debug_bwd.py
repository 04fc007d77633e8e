import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import sdf_packed_net, t
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
from oracle import oracle_np as ON
def rel(a,b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return float(np.abs(a-b).max()/max(np.abs(b).max(),1e-12))
W=64
sd = synth.make_state_dict(W, 0)
net, onet = sdf_packed_net(sd), ON.sdf_net(sd)
rs = np.random.RandomState(4)
M = 150
x = rs.uniform(-1, 1, size=(M, 3)).astype(np.float32)
dy = (rs.normal(size=(M, 258)) * 0.1).astype(np.float32)
dn = rs.normal(size=(M, 3)).astype(np.float32)
y, n, ctx = ops.sdf_forward(net, t(x), M)
oy, on, cache = ON.sdf_forward(onet, x)
print('fwd', rel(y,oy), rel(n,on))
for name, dnn in (('value-only', None), ('with-normal', dn)):
    oW, ob, odx = ON.sdf_backward(onet, cache, dy, dnn)
    dWs, dbs, dx = ops.sdf_backward(net, t(x), M, M, M, t(dy), t(dnn) if dnn is not None else None, ctx, True)
    print(name, 'dx', rel(dx, odx))
    for l in range(9):
        print('  layer', l, 'dW', rel(dWs[l], oW[l]), 'db', rel(dbs[l], ob[l]))
# only-normal
oW, ob, odx = ON.sdf_backward(onet, cache, np.zeros_like(dy), dn)
dWs, dbs, dx = ops.sdf_backward(net, t(x), M, M, M, t(np.zeros_like(dy)), t(dn), ctx, True)
print('normal-only dx', rel(dx, odx))
for l in range(9):
    print('  layer', l, 'dW', rel(dWs[l], oW[l]), 'db', rel(dbs[l], ob[l]))
