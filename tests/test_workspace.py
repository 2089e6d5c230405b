"""The optional workspaces of the probe-routed organisations (interpol/_hip.py: _optional_workspace): taken when they fit
comfortably, declined -- without asking the allocator, hence without its synchronise-and-flush cycle -- when they do not, and a
failed request is remembered (every default call used to allocate 1.2 - 1.7 GB at config 2 and relied on catching
torch.cuda.OutOfMemoryError)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd"))

pytestmark = pytest.mark.gpu


def test_optional_workspace_policy(monkeypatch):
    import interpol
    from interpol import _hip
    dev = torch.device("cuda", 0)
    _hip.release_workspaces()
    big = 128 << 20
    ws = _hip._optional_workspace(big, dev)
    assert ws is not None and ws.numel() == big and ws.device.type == "cuda"
    # one buffer per (device, stream, host thread), kept between calls and shared by smaller requests
    assert _hip._optional_workspace(big // 2, dev) is ws
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        assert _hip._optional_workspace(1 << 20, dev) is not ws
    # ... and per HOST THREAD (ctypes releases the GIL inside the library -- two threads on one stream must not
    # interleave their kernels on one buffer)
    import threading
    other = []
    th = threading.Thread(target=lambda: other.append(_hip._optional_workspace(big // 2, dev)))
    th.start(); th.join()
    assert other[0] is not None and other[0] is not ws
    # at most _WS_MAX buffers are kept, least recently used first out
    streams = [torch.cuda.Stream(dev) for _ in range(_hip._WS_MAX + 2)]
    for st in streams:
        with torch.cuda.stream(st):
            _hip._optional_workspace(1 << 20, dev)
    assert len(_hip._WS_CACHE) == _hip._WS_MAX
    del ws, other
    _hip.release_workspaces()
    assert _hip._optional_workspace(0, dev) is None
    # not even half of what is available: declined without an allocation attempt
    calls = []
    real_empty = torch.empty
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (big, 1 << 40))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a, **k: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda *a, **k: 0)
    monkeypatch.setattr(torch, "empty", lambda *a, **k: (calls.append(a), real_empty(*a, **k))[1])
    assert _hip._optional_workspace(big, dev) is None and not calls
    # a failure is remembered and not retried while memory has not grown
    def boom(*a, **k):
        calls.append(a)
        raise torch.cuda.OutOfMemoryError("test")
    monkeypatch.setattr(torch, "empty", boom)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (8 * big, 1 << 40))
    assert _hip._optional_workspace(big, dev) is None and len(calls) == 1
    assert _hip._optional_workspace(big, dev) is None and len(calls) == 1
    # ... and is retried once noticeably more memory is available
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (10 * big, 1 << 40))
    monkeypatch.setattr(torch, "empty", lambda *a, **k: (calls.append(a), real_empty(*a, **k))[1])
    ws = _hip._optional_workspace(big, dev)
    assert ws is not None and len(calls) == 2 and 0 not in _hip._WS_DENIED
    monkeypatch.undo()
    _hip.release_workspaces()


def test_push_and_pull_do_without_a_workspace(monkeypatch):
    """With no room for the workspace the default calls still run (sample tiles) and agree with the routed ones."""
    import interpol
    from interpol import _hip
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(1, 2, 48, 48, 48, generator=g, device=dev)
    grid = interpol.identity_grid([48, 48, 48], device=dev)[None] + torch.randn(1, 48, 48, 48, 3, generator=g, device=dev)
    want_pull = interpol.grid_pull(x, grid, interpolation=3, bound="dct2", extrapolate=True)
    want_push = interpol.grid_push(x, grid, interpolation=3, bound="dct2", extrapolate=True)
    monkeypatch.setattr(_hip, "_optional_workspace", lambda nbytes, dev: None)
    got_pull = interpol.grid_pull(x, grid, interpolation=3, bound="dct2", extrapolate=True)
    got_push = interpol.grid_push(x, grid, interpolation=3, bound="dct2", extrapolate=True)
    tol = 1e-5
    assert float((got_pull - want_pull).abs().max()) <= tol * float(want_pull.abs().max())
    assert float((got_push - want_push).abs().max()) <= tol * float(want_push.abs().max())


def test_shared_target_goes_on_when_the_merged_bricks_workspace_is_denied(monkeypatch):
    """a dense shared-target push asks for the merged bricks' workspace (~22 B per sample over all sources); when that is
    denied the call must go on to the other organisations (and still be right), not stop at a half-configured route."""
    import interpol
    from interpol import _hip, ops
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(4)
    n, m = 32, 48
    x = torch.randn(6, 2, n, n, n, generator=g, device=dev)
    grid = (interpol.identity_grid([n, n, n], device=dev)[None] * (m / n) + torch.randn(6, n, n, n, 3, generator=g, device=dev)).contiguous()
    b, o = [1] * 3, [3] * 3
    want = _hip.scatter("push", x, grid, [m] * 3, b, o, 1, flags=_hip.FLAG_NO_FASTPATH).sum(0, keepdim=True)
    assert ops.kernels().dense_when_merged(grid, [m] * 3, o)
    for deny in (False, True):
        if deny:
            monkeypatch.setattr(_hip, "_optional_workspace", lambda nbytes, dev: None)
        out = torch.zeros(1, 2, m, m, m, device=dev)
        ops.kernels().push_shared_(out, x, grid, b, o, 1)
        assert float((out - want).abs().max()) <= 1e-5 * float(want.abs().max()), deny
