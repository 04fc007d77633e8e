"""The throughput ("row-owner") form of the bf16-family tracing MLP (csrc/rows_engine_bf16.h: 128 rows per workgroup, a wave owns 32 rows for the
whole network, weights staged once per workgroup through an LDS ring, v_mfma_f32_32x32x16_bf16) against the column-split engines and the oracle.
Same arithmetic definition (rounding points, softplus forms); the matrix core's internal summation order differs with the instruction shape."""
import numpy as np
import pytest
import torch

from helpers import sdf_packed_net, t
from mvsdf_amd import ops
from mvsdf_amd.utils import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', ['bf16x2', 'bf16x3'])
@pytest.mark.parametrize('W', [64, 256])
def test_row_owner_mlp_vs_column_split_engine_and_oracle(oracle, W, dtype):
    sd = synth.make_state_dict(W, 0)
    net = ops.pack_trace_net(sdf_packed_net(sd), dtype)
    rs = np.random.RandomState(3)
    for n in (4000, 128, 77):                                      # ragged last workgroup, a single workgroup, less than one
        x = rs.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
        y_ro = ops.sdf_col0(net, t(x), mt=64).cpu().numpy()
        y_cs = ops.sdf_col0(net, t(x), mt=2).cpu().numpy()
        d = np.abs(y_ro - y_cs)
        if dtype == 'bf16':
            ref = oracle.sdf_forward(oracle.Net(sd, bf16=True), x, ncols=1)[:, 0]
            tol = (6e-3, 1e-4)                                       # the bf16 engine's own distance to its twin (a flipped rounding of one activation moves the output by ~1e-3)
        else:
            ref = oracle.sdf_forward(oracle.Net(sd, bf16='weights'), x, ncols=1)[:, 0]
            tol = {'bf16x2': (4e-5, 4e-6), 'bf16x3': (5e-6, 6e-7)}[dtype]
        e = np.abs(y_ro - ref)
        print('W=%d %s n=%d: row-owner vs column-split max %.3g mean %.3g; vs oracle max %.3g mean %.3g' % (W, dtype, n, d.max(), d.mean(), e.max(), e.mean()))
        assert e.max() < tol[0] and e.mean() < tol[1]
        assert d.max() < 2 * tol[0] and d.mean() < 2 * tol[1]
