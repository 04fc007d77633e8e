"""One-off evidence run: the f32x3 tracer on the full ray batches of BASELINE configs[1] (idr_c2: 2048 rays, 8x256) and one GPU's share of configs[4]
(idr_c5share: 4096 rays) against the instruction-model oracle, bit for bit.  (The suite does this on subsets: the oracle costs ~1 ms per MLP row and thread.)
python3 tools/micro/f32s/bitexact_c2.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import oracle
from conftest import golden
from helpers import sdf_packed_net, t, trace_params
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
oracle.lib()
for name in sys.argv[1:] or ['idr_c2']:
    g = golden(name)
    W, B, P, V, seed = int(g['W']), int(g['B']), int(g['P']), int(g['V']), int(g['seed'])
    sd = synth.make_state_dict(W, seed)
    inp, _ = synth.make_batch(B, P, V, seed=seed, size=float(g['scene_size']), center=tuple(g['scene_center']), feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    dirs, cam = ops.camera_rays(t(inp['uv']), t(inp['pose']), t(inp['intrinsics']))
    steps = np.random.RandomState(seed).uniform(size=100).astype(np.float32)
    om = np.asarray(inp['object_mask']).reshape(-1).astype(bool)
    iv = torch.linspace(0, 1, 100)
    net = ops.pack_bf16_net(sdf_packed_net(sd), terms=3, weight_terms=3)
    pts, mask, dists, cnt = ops.trace(net, cam, dirs, t(om), trace_params(W), True, iv.cuda(), t(steps), mt=1, mt_samples=4)
    t0 = time.time()
    p_o, m_o, d_o, rows = oracle.trace(oracle.Net(sd, bf16='f32x3'), cam.cpu().numpy(), dirs.cpu().numpy(), om, True, steps, iv.numpy(), **synth.model_conf(W)['ray_tracer'])
    print('%s: %d rays, %d oracle MLP rows in %.0f s; masks equal %s, dists equal %s, points equal %s, row counters equal %s; hits %d' % (
        name, B * P, int(rows.sum()), time.time() - t0, np.array_equal(mask.cpu().numpy(), m_o), np.array_equal(dists.cpu().numpy(), d_o),
        np.array_equal(pts.cpu().numpy(), p_o), np.array_equal(cnt.cpu().numpy()[:4], rows), int(m_o.sum())), flush=True)
