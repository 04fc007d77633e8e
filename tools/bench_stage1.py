import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import sdf_packed_net, t, trace_params
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
net = sdf_packed_net(synth.make_state_dict(256, 0))
inp, _ = synth.make_batch(8, 256, 0, seed=0, with_features=False)
dirs, cam = ops.camera_rays(t(inp['uv']), t(inp['pose']), t(inp['intrinsics']))
om = torch.ones(2048, dtype=torch.bool, device='cuda'); iv = torch.linspace(0, 1, 100).cuda(); steps = torch.rand(100).cuda()
for ls in (3, 1, 0):
    ev = []
    for _ in range(8): out = ops.trace(net, cam, dirs, om, trace_params(256, line_step_iters=ls), True, iv, steps, mt=1, mt_samples=2, events=ev)
    torch.cuda.synchronize()
    s1 = np.mean([e[0].elapsed_time(e[1]) for e in ev[3:]]); s2 = np.mean([e[1].elapsed_time(e[2]) for e in ev[3:]])
    c = out[3].cpu().numpy()
    print('line_step_iters=%d: sphere %.3f ms (rows %d)  stage2 %.3f ms (rows %d)  hits %d' % (ls, s1, c[0], s2, c[1:4].sum(), int(out[1].sum())))
