"""Every matrix instruction of one f32x3 MLP evaluation (oracle arithmetic, Python mirror of oracle_mvsdf.c::sdf_row_f32x3) as tiles for mfma_file"""
import sys, ctypes as C; sys.path.insert(0, '/tmp/mfma'); sys.path.insert(0, '/root/repo')
from hw2 import *
from oracle import oracle
from mvsdf_amd.utils import synth
L = oracle.lib()
def softplus(z):
    z = np.ascontiguousarray(z, np.float32); y = np.empty_like(z)
    L.orc_softplus100(z.ctypes.data_as(C.c_void_p), C.c_int(z.size), y.ctypes.data_as(C.c_void_p)); return y
def bfbits(v):
    u = np.float32(v).view(np.uint32); u = np.uint32(u + 0x7fff + ((u >> 16) & 1)); return np.uint16(u >> 16)
def bfval(h): return np.uint32(int(h) << 16).view(np.float32)
def split3(v):
    v = np.float32(v)
    if abs(v) < 2.0 ** -60: v = np.float32(0)
    t0 = bfbits(v); v = np.float32(v - bfval(t0)); t1 = bfbits(v); v = np.float32(v - bfval(t1)); return t0, t1, bfbits(v)
def trace(W, seed, x):
    sd = synth.make_state_dict(W, seed); net = oracle.Net(sd)
    pe = oracle.pe(np.asarray([x], np.float32), net.multires)[0]
    d0 = pe.size
    at = [list(split3(v)) for v in pe]              # per column: 3 terms
    instr = []
    OS = (0, 1, 2, 0, 1, 0); OJ = (2, 1, 0, 1, 0, 0)
    for l in range(net.n_layers):
        if l > 0 and (net.skip_mask >> l) & 1:
            at += [list(split3(np.float32(v) * np.float32(0.7071067690849304))) for v in pe]
        Wl = net.W[l]; K = Wl.shape[1]; kp = (K + 31) & ~31
        assert len(at) == K, (l, len(at), K)
        A3 = np.zeros((3, kp), np.uint16)
        for k in range(K):
            for s in range(3): A3[s, k] = at[k][s]
        last = l == net.n_layers - 1
        no = 1 if last else Wl.shape[0]
        W3 = np.zeros((3, no, kp), np.uint16)
        for j in range(no):
            for k in range(K):
                t = split3(Wl[j, k])
                for s in range(3): W3[s, j, k] = t[s]
        z = np.zeros(no, np.float32)
        for j in range(no):
            acc = np.float32(net.b[l][j])
            for kb in range(0, kp, 32):
                for o in range(6):
                    a = A3[OS[o], kb:kb + 32]; w = W3[OJ[o], j, kb:kb + 32]
                    out = mfma(acc, w, a)
                    instr.append((l, j, kb, o, acc, w.copy(), a.copy(), out))
                    acc = out
            z[j] = acc
        if last: return z[0], instr
        h = softplus(z)
        if (net.skip_mask >> (l + 1)) & 1: h = (h * np.float32(0.7071067690849304)).astype(np.float32)
        at = [list(split3(v)) for v in h]
if __name__ == '__main__':
    W, seed, idx = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    x = np.random.RandomState(3 + seed).uniform(-1.2, 1.2, size=({64: 4000, 256: 1500, 512: 400}[W], 3)).astype(np.float32)
    x[:8] *= 1e-3; x[8:16] = 0.0
    y, instr = trace(W, seed, x[idx])
    ref = oracle.sdf_forward(oracle.Net(synth.make_state_dict(W, seed), bf16='f32x3'), x[idx:idx + 1], ncols=1)[0, 0]
    print('python mirror %.9g, C oracle %.9g, instructions %d' % (y, ref, len(instr)))
    n = len(instr); T = (n + 15) // 16
    A = np.zeros((T, 16, 32), np.uint16); B = np.zeros((T, 16, 32), np.uint16); Cc = np.zeros((T, 16, 16), np.float32); E = np.zeros(n, np.float32)
    for q, (l, j, kb, o, acc, w, a, out) in enumerate(instr):
        t, i = divmod(q, 16); A[t, i] = w; B[t, i] = a; Cc[t, i, i] = acc; E[q] = out
    pre = '/root/repo/tools/micro/mfma_bf16_model/cases/row'
    A.tofile(pre + '_A.bin'); B.tofile(pre + '_B.bin'); Cc.tofile(pre + '_C.bin'); E.tofile(pre + '_E.bin')
    import pickle; pickle.dump([(l, j, kb, o) for (l, j, kb, o, *_rest) in instr], open(pre + '_idx.pkl', 'wb'))
