"""TEST-ONLY process-group shim: the collectives the gloo backend does not run on device tensors
(all_gather_into_tensor, reduce_scatter_tensor, all_to_all_single), staged through pinned host memory.

Why: RCCL refuses two ranks on one device, and the GPU boxes have one.  tests/test_hip_multirank.py therefore runs W
processes on cuda:0 over gloo.  gloo itself moves device tensors for all_reduce / broadcast (the exchanges of
`dp_exchange='grads'`, north_star's form -- those go through the backend untouched, async_op included); whether it also
runs the three collectives above on device tensors depends on the torch build, so install() PROBES them and wraps the
ones that raise: wait for the current stream, copy the input to the host, run
the SAME collective on the host copies over the same group, copy the result back on the current stream.  Stream
semantics as the product expects of RCCL: the result is ordered behind prior work of the current stream and visible to
later work on it; an async call returns a finished work handle.  Nothing under plnlp_amd/ imports this module."""
import torch
import torch.distributed as dist


class _Finished:
    """the Work handle of a collective that completed before the call returned"""

    def wait(self, timeout=None):
        return True

    def is_completed(self):
        return True


STAGED_CALLS = {"all_gather_into_tensor": 0, "reduce_scatter_tensor": 0, "all_to_all_single": 0}


def _host(t: torch.Tensor) -> torch.Tensor:
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t.detach())          # (blocking D2H on the current stream: ordered behind everything queued on it)
    return h


NATIVE_CALLS = {name: 0 for name in STAGED_CALLS}
NEEDS_STAGING = {}


def _probe(native, name, world):
    """does this build's gloo run `name` on device tensors?  A tiny call of the collective itself -- every rank makes it,
    at the same point (install() is called by all ranks right after init_process_group)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    try:
        if name == "all_gather_into_tensor":
            native[name](torch.empty(2 * world, device=dev), torch.ones(2, device=dev))
        elif name == "reduce_scatter_tensor":
            native[name](torch.empty(2, device=dev), torch.ones(2 * world, device=dev))
        else:
            native[name](torch.empty(world, device=dev), torch.ones(world, device=dev))
        torch.cuda.synchronize()
        return True
    except (RuntimeError, NotImplementedError, ValueError):
        return False


def install():
    """wrap the three collectives on torch.distributed (looked up at call time by plnlp_amd.model / plnlp_amd.shard): every
    call is counted; a collective this build's gloo does NOT run on device tensors (probed here, collectively) is staged
    through the host, the others pass straight through.  (torch 2.10's gloo was found to run all three natively on the
    MI355X boxes -- profiles/r05_gloo_device_collectives.txt -- so there the wrapper only counts.)  Idempotent."""
    if getattr(dist, "_plnlp_gloo_shim", False):
        return
    native = {name: getattr(dist, name) for name in STAGED_CALLS}
    world = dist.get_world_size()
    for name in STAGED_CALLS:
        NEEDS_STAGING[name] = dist.get_backend() == "gloo" and not _probe(native, name, world)

    def staged(name):
        def call(output, input, *args, group=None, async_op=False, **kw):      # noqa: A002 -- torch's own argument name
            if not output.is_cuda or not NEEDS_STAGING[name]:
                NATIVE_CALLS[name] += int(output.is_cuda)
                return native[name](output, input, *args, group=group, async_op=async_op, **kw)
            STAGED_CALLS[name] += 1
            out_h = torch.empty(output.shape, dtype=output.dtype, pin_memory=True)
            native[name](out_h, _host(input), *args, group=group, **kw)
            output.copy_(out_h)
            return _Finished() if async_op else None
        call.__name__ = name
        return call
    for name in STAGED_CALLS:
        setattr(dist, name, staged(name))
    dist._plnlp_gloo_shim = True
