import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import sdf_packed_net
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
net = sdf_packed_net(synth.make_state_dict(256, 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
x = torch.rand(n, 3, device='cuda') * 2 - 1
for _ in range(5): y = ops.sdf_col0(net, x, mt=mt)
torch.cuda.synchronize()
