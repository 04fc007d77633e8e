"""Workload for `rocprofv3 --pmc ... -- python3 tools/pmc_ro.py <dtype> <n> <mt>`: the tracing MLP alone (mvsdf_sdf_col0), 5 launches (dev tool)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import sdf_packed_net
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16x2'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
mt = int(sys.argv[3]) if len(sys.argv) > 3 else 64
net = ops.pack_trace_net(sdf_packed_net(synth.make_state_dict(256, 0)), dt)
x = torch.rand(n, 3, device='cuda') * 2 - 1
for _ in range(5): y = ops.sdf_col0(net, x, mt=mt)
torch.cuda.synchronize()
