"""Chunked evaluation helpers of the reference's eval / plot loops (reference code/utils/general.py:23-53): split a full-image
model input into pixel chunks and merge the per-chunk outputs back.  Used by callers that render whole images through
IDRNetwork.forward in eval mode (eval.py:145-151, idr_train.py:221-230)."""
import torch


def split_input(model_input, total_pixels, n_pixels=10000):
    """-> list of model-input dicts, each with at most n_pixels pixels of uv / object_mask (general.py:23-37)."""
    out = []
    for idx in torch.split(torch.arange(total_pixels, device=model_input['uv'].device), n_pixels, dim=0):
        d = dict(model_input)
        d['uv'] = torch.index_select(model_input['uv'], 1, idx)
        d['object_mask'] = torch.index_select(model_input['object_mask'], 1, idx)
        out.append(d)
    return out


def merge_output(res, total_pixels, batch_size):
    """Concatenate per-chunk output dicts back to full-image tensors (general.py:39-53); None entries are dropped."""
    merged = {}
    for k in res[0]:
        if res[0][k] is None:
            continue
        parts = [r[k].reshape(batch_size, -1, 1) if r[k].dim() == 1 else r[k].reshape(batch_size, -1, r[k].shape[-1]) for r in res]
        merged[k] = torch.cat(parts, 1).reshape(batch_size * total_pixels, -1 if parts[0].shape[-1] > 1 else 1)
        if parts[0].shape[-1] == 1:
            merged[k] = merged[k].reshape(batch_size * total_pixels)
    return merged


class PinnedUniform:
    """Uniform draws from torch's CPU generator (the reference draws eikonal points and min-sdf steps on the CPU, idr.py:216-221,
    ray_tracing.py:287) delivered to the device WITHOUT a blocking copy: the values are drawn straight into one of two pinned
    staging buffers (same RNG stream, same values as torch.empty(shape).uniform_(lo, hi)) and copied asynchronously.  Two buffers
    alternate; a buffer is only redrawn once the copy that last read it has completed (an event per buffer, normally long done)."""

    def __init__(self):
        self._bufs, self._k = {}, 0

    def __call__(self, shape, lo, hi, device):
        shape = tuple(shape)
        key = (shape, self._k & 1)
        self._k += 1
        slot = self._bufs.get(key)
        if slot is None:
            slot = self._bufs[key] = [torch.empty(shape).pin_memory(), None]
        buf, ev = slot
        if ev is not None:
            ev.synchronize()
        buf.uniform_(lo, hi)
        out = buf.to(device, non_blocking=True)
        if out.is_cuda:
            slot[1] = torch.cuda.Event()
            slot[1].record()
        return out

    STAGE_DEPTH = 8

    def pair_staged(self, shape_a, lo_a, hi_a, shape_b, lo_b, hi_b, device):
        """Two consecutive draws (a, then b: the same values as two separate calls) left in a pinned staging buffer for the CONSUMER to read on the device
        (the native step's first kernel reads the pinned memory directly: no copy node in front of the step).  -> (a, b, staging, slot): a / b are
        UNINITIALISED device tensors (views of one buffer) the consumer fills; the consumer sets slot[2] = (step, seq) of the forward that reads the
        staging buffer, and the buffer -- one of STAGE_DEPTH that take turns -- is redrawn only after that forward's ray partition has run
        (NativeStep.partition_done: a host that enqueues steps ahead of the GPU waits here once it is STAGE_DEPTH forwards ahead)."""
        shape_a, shape_b = tuple(shape_a), tuple(shape_b)
        na, nb = 1, 1
        for v in shape_a:
            na *= v
        for v in shape_b:
            nb *= v
        key = ('staged', shape_a, shape_b, self._k % self.STAGE_DEPTH)
        self._k += 1
        slot = self._bufs.get(key)
        if slot is None:
            slot = self._bufs[key] = [torch.empty(na + nb).pin_memory(), None, None]
        buf, _, reader = slot
        if reader is not None:
            reader[0].partition_done(reader[1])
            slot[2] = None
        buf[:na].uniform_(lo_a, hi_a)
        buf[na:].uniform_(lo_b, hi_b)
        out = torch.empty(na + nb, dtype=buf.dtype, device=device)
        return out[:na].view(shape_a), out[na:].view(shape_b), buf, slot

    def pair(self, shape_a, lo_a, hi_a, shape_b, lo_b, hi_b, device, defer=False):
        """Two consecutive draws (a, then b: the same values as two separate calls) staged in ONE pinned buffer and delivered by ONE async copy.
        defer=True: no copy at all -- returns (a, b, staging) with a / b UNINITIALISED device tensors (views of one buffer) and the pinned staging
        tensor; the consumer fills a / b from it on the device (the native step's first kernel reads the pinned memory directly) and must have
        done so before the next-but-one draw (two staging buffers alternate; the native step's host wait per forward guarantees it)."""
        shape_a, shape_b = tuple(shape_a), tuple(shape_b)
        na, nb = 1, 1
        for v in shape_a:
            na *= v
        for v in shape_b:
            nb *= v
        key = ('pair', shape_a, shape_b, self._k & 1)
        self._k += 1
        slot = self._bufs.get(key)
        if slot is None:
            slot = self._bufs[key] = [torch.empty(na + nb).pin_memory(), None]
        buf, ev = slot
        if ev is not None:
            ev.synchronize()
        buf[:na].uniform_(lo_a, hi_a)
        buf[na:].uniform_(lo_b, hi_b)
        if defer and torch.device(device).type == 'cuda':
            slot[1] = None
            out = torch.empty(na + nb, dtype=buf.dtype, device=device)
            return out[:na].view(shape_a), out[na:].view(shape_b), buf
        out = buf.to(device, non_blocking=True)
        if out.is_cuda:
            slot[1] = torch.cuda.Event()
            slot[1].record()
        if defer:
            return out[:na].view(shape_a), out[na:].view(shape_b), None
        return out[:na].view(shape_a), out[na:].view(shape_b)
