"""GPU tests (`-m gpu`, MI355X) of the SAFETY of the bf16 mode's hidden-layer chain (m360_mlp_chain_bf16_safe: six layers of
model.py:134-146 in one launch whose workgroups hand activations over through an XCD's L2).

The hand-over relies on two things no API promises - all workgroups of a slot on one XCD, all 256 workgroups resident at once - so the
kernel checks both itself, reports in a status block, and the host queues a GATED layer-by-layer re-run behind every launch.  These
tests break each assumption on purpose (injected faults, a shrunk wait bound, kernels on a second stream that hold CUs) and require the
SAME BITS as six plain launches every time, plus honest counters.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from mipnerf360_amd import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test needs a HIP device")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore_chain_switches():
    from mipnerf360_amd import ops
    yield
    ops.set_chain_debug(0, 0)
    ops.set_chain_cooperative(False)
    ops.set_hidden_chain(True)
    ops.set_paired_rows(True)


@pytest.fixture
def diag():
    """The fault-injection hooks (m360_hyper_t.chain_debug_*) exist in the DIAGNOSTICS build only - the same sources with -DM360_DIAG
    (make -C mipnerf360_amd/csrc diag); the product library refuses a call that carries them (test_product_library_has_no_test_hooks).
    Tests that break the chain's assumptions on purpose run every call of theirs through that build."""
    from mipnerf360_amd import _lib
    if not os.path.exists(_lib.DIAG_LIB_PATH):
        pytest.skip("libm360_diag.so not built (make -C mipnerf360_amd/csrc diag)")
    with _lib.use_library(_lib.DIAG_LIB_PATH) as lib:
        yield lib


def _packs(width, layers, dev, seed):
    from mipnerf360_amd import ops
    g = torch.Generator().manual_seed(seed)
    packs = []
    for _ in range(layers):
        w = (torch.randn(width, width, generator=g) * (2.0 / width) ** 0.5).to(dev)
        b = (torch.randn(width, generator=g) * 0.1).to(dev)
        packs.append(ops.pack_linear_bf16(w, b, width, width))
    return packs, g


def _layer_by_layer(x, packs):
    from mipnerf360_amd import _lib, ops
    flags = _lib.ACT_RELU | _lib.ROWS_PAIRED_IN | _lib.ROWS_PAIRED_OUT
    for wp, bp in packs:
        x = ops.linear_bf16(x, wp, bp, flags)
    return x


def _skip_unless_chain(M=32768, width=1024, layers=6):
    from mipnerf360_amd import ops
    if not ops.mlp_chain_bf16_supported(M, width, layers):
        pytest.skip("the layer chain needs a 256-CU device")


@pytest.mark.parametrize("M,layers", [(32768, 6), (65536, 3), (32768, 1)])
def test_safe_chain_clean_launch(dev, M, layers):
    """No fault: the chain's own rows stand (the gated launches find error == 0 and return), x_in is untouched, the counters say one clean
    launch; bit for bit the layer-by-layer result."""
    from mipnerf360_amd import ops
    _skip_unless_chain(M, 1024, layers)
    packs, g = _packs(1024, layers, dev, 100 + layers)
    x = ops.pair_rows(torch.randn(M, 1024, generator=g).to(dev).bfloat16())
    keep = x.clone()
    a, b = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    got, st = ops.mlp_chain_bf16_safe(x, a, b, packs)
    assert torch.equal(x, keep)
    assert torch.equal(got, _layer_by_layer(keep, packs))
    assert st == dict(launches=1, recoveries=0, timeouts=0, xcc_mismatch=0, last_error=0), st


@pytest.mark.parametrize("fault", [1, 2])
def test_safe_chain_repairs_an_injected_fault(dev, diag, fault):
    """fault 1: one workgroup reports a foreign XCD (placement check); fault 2: every wave treats its first wait as run out and stops
    waiting - it then reads rows its neighbours have not written yet.  Either way the launch flags itself, the gated re-run redoes all
    rows from x_in, and the result is the layer-by-layer bits; the sticky counters record what happened."""
    from mipnerf360_amd import ops
    _skip_unless_chain()
    packs, g = _packs(1024, 6, dev, 7 + fault)
    for rep in range(3):
        x = ops.pair_rows(torch.randn(65536, 1024, generator=g).to(dev).bfloat16())
        want = _layer_by_layer(x, packs)
        ops.set_chain_debug(0, fault)
        a, b = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
        got, st = ops.mlp_chain_bf16_safe(x, a, b, packs)
        ops.set_chain_debug(0, 0)
        assert st["launches"] == 1 and st["recoveries"] == 1 and st["last_error"] == (2 if fault == 1 else 1), st
        assert (st["xcc_mismatch"] >= 1 and st["timeouts"] == 0) if fault == 1 else (st["timeouts"] > 0 and st["xcc_mismatch"] == 0), st
        assert torch.equal(got, want), (fault, rep, int((got != want).sum()))


def test_unsafe_entry_point_reports_the_fault(dev, diag):
    """m360_mlp_chain_bf16 (no third buffer, no re-run) must at least say so: ops.mlp_chain_bf16 raises."""
    from mipnerf360_amd import ops
    _skip_unless_chain()
    packs, g = _packs(1024, 6, dev, 3)
    x = ops.pair_rows(torch.randn(32768, 1024, generator=g).to(dev).bfloat16())
    ops.set_chain_debug(0, 2)
    with pytest.raises(RuntimeError, match="reported an error"):
        ops.mlp_chain_bf16(x.clone(), torch.empty_like(x), packs)
    ops.set_chain_debug(0, 0)
    assert torch.equal(ops.mlp_chain_bf16(x.clone(), torch.empty_like(x), packs), _layer_by_layer(x, packs))


def test_product_library_has_no_test_hooks(dev):
    """the fault-injection hooks live in libm360_diag.so only.  The product library refuses a call whose m360_hyper_t
    carries them - through the op and through the whole forward - and renders normally again once they are cleared."""
    from mipnerf360_amd import _lib, ops
    _skip_unless_chain()
    assert getattr(_lib.lib(), "m360_path", "").endswith("libm360.so")
    packs, g = _packs(1024, 2, dev, 5)
    x = ops.pair_rows(torch.randn(32768, 1024, generator=g).to(dev).bfloat16())
    for hook in ((0, 2), (5000, 0), (0, -1)):
        ops.set_chain_debug(*hook)
        with pytest.raises(RuntimeError, match="diagnostics build"):
            ops.mlp_chain_bf16_safe(x, torch.empty_like(x), torch.empty_like(x), packs)
    model, rays = _bf16_model(dev), _rays(dev, 300)
    ops.set_chain_debug(0, 1)
    with pytest.raises(RuntimeError, match="diagnostics build"), torch.no_grad():
        model(rays)
    ops.set_chain_debug(0, 0)
    with torch.no_grad():
        assert all(bool(torch.isfinite(o).all()) for o in model(rays))
    got, st = ops.mlp_chain_bf16_safe(x, torch.empty_like(x), torch.empty_like(x), packs)
    assert torch.equal(got, _layer_by_layer(x, packs)) and st["last_error"] == 0


def _blocker(dev, stream, launches=40):
    """Kernels on `stream` that hold a quarter of the CUs (64 workgroups of the fp32 256 x 256-tile kernel, each with ~130 KB of LDS) for
    a few milliseconds: the chain's 256 workgroups (144 KB of LDS each: one per CU) cannot all be resident while they run."""
    from mipnerf360_amd import _lib, ops
    g = torch.Generator().manual_seed(1)
    xb = torch.randn(16384, 4096, generator=g).to(dev)
    wp, bp = ops.pack_linear((torch.randn(256, 4096, generator=g) * 0.02).to(dev), torch.zeros(256).to(dev), 256, 4096)
    yb = torch.empty(16384, 256, device=dev)
    torch.cuda.synchronize()

    def go():
        with torch.cuda.stream(stream):
            for _ in range(launches):
                ops.linear(xb, wp, bp, _lib.ACT_SIGMOID, out=yb)
    return go, (xb, wp, bp, yb)


def test_safe_chain_next_to_a_kernel_that_holds_cus(dev, diag):
    """The advisor's case: a kernel on ANOTHER stream of the same process holds CUs while the chain runs - what an RCCL
    all-gather waiting for a late peer does under the next frame's compute.  With the wait bound shrunk to 50 us the chain's resident
    workgroups give up on the ones that cannot start; whatever happens, the rows must be the layer-by-layer bits."""
    from mipnerf360_amd import ops
    _skip_unless_chain()
    packs, g = _packs(1024, 6, dev, 21)
    x = ops.pair_rows(torch.randn(65536, 1024, generator=g).to(dev).bfloat16())
    want = _layer_by_layer(x, packs)
    side = torch.cuda.Stream(device=dev)
    block, keep = _blocker(dev, side)
    seen = []
    for wait_ticks in (5000, 0):  # 50 us: gives up while the blocker runs; default (0.1 s): waits the blocker out
        ops.set_chain_debug(wait_ticks, 0)
        a, b = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
        torch.cuda.synchronize()
        block()
        got, st = ops.mlp_chain_bf16_safe(x, a, b, packs)
        torch.cuda.synchronize()
        seen.append(st)
        assert torch.equal(got, want), (wait_ticks, st, int((got != want).sum()))
        assert st["recoveries"] == (1 if st["last_error"] else 0), st
    print("chain next to a CU-holding kernel:", seen)
    assert seen[1]["timeouts"] == 0 or seen[1]["recoveries"] == 1  # with the default bound a 10 ms blocker is waited out or repaired


def _bf16_model(dev, N=128, seed=8):
    from mipnerf360_amd.model import mipNeRF360
    sd = {k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(256, 1024, seed=seed).items()}
    model = mipNeRF360(num_samples=N, hidden_proposal=256, hidden_nerf=1024, mlp_dtype="bf16", device=dev, randomized=False).eval()
    model.load_state_dict(sd)
    return model


def _rays(dev, B, seed=5):
    from mipnerf360_amd.intern.ray import Rays
    r = synthetic.make_rays("garden", B, seed=seed)
    return Rays(*[torch.from_numpy(r[k]).float().to(dev) for k in synthetic.RAY_FIELDS])


@pytest.mark.parametrize("fault", [1, 2])
def test_forward_repairs_a_faulty_chain_launch(dev, diag, fault):
    """The whole bf16 forward (m360_forward) with a fault injected into its chain launch: same bits as with six launches, and
    model.chain_status() counts the repair.  300 rays x 128 samples: 32768 rows in the chain + 5632 layer by layer."""
    from mipnerf360_amd import _lib, ops
    _skip_unless_chain()
    model, rays = _bf16_model(dev), _rays(dev, 300)
    was = ops.set_hidden_chain(False)
    try:
        with torch.no_grad():
            want = [o.clone() for o in model(rays)]
    finally:
        ops.set_hidden_chain(was)
    before = model.chain_status()
    ops.set_chain_debug(0, fault)
    with torch.no_grad():
        got = [o.clone() for o in model(rays)]
    ops.set_chain_debug(0, 0)
    after = model.chain_status()
    assert after["launches"] == before["launches"] + 1 and after["recoveries"] == before["recoveries"] + 1, (before, after)
    assert model.chain_error() is True
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    with torch.no_grad():
        again = model(rays)
    assert model.chain_error() is False and model.chain_status()["recoveries"] == after["recoveries"]
    for a, b in zip(again, want):
        assert torch.equal(a, b)


def test_chain_status_when_no_chain_runs(dev):
    """the status must be defined when the configuration runs no chain - plain rows, the chain switched
    off, a batch below 32768 rows, fp32: `launches` does not move and last_error is the last chain launch's, never uninitialised memory."""
    from mipnerf360_amd import _lib, ops
    _skip_unless_chain()
    model, rays = _bf16_model(dev), _rays(dev, 512)
    with torch.no_grad():
        model(rays)
    base = model.chain_status()
    assert base["last_error"] == 0 and base["launches"] >= 1

    def same_after(fn):
        with torch.no_grad():
            fn()
        st = model.chain_status()
        assert st == base, (st, base)

    was = ops.set_paired_rows(False)
    try:
        same_after(lambda: model(rays))
    finally:
        ops.set_paired_rows(was)
    was = ops.set_hidden_chain(False)
    try:
        same_after(lambda: model(rays))
    finally:
        ops.set_hidden_chain(was)
    same_after(lambda: model(_rays(dev, 200)))  # 25600 rows: below one chain block


def test_two_streams_of_one_process(dev):
    """a bf16 forward on stream A while a second bf16 model renders on stream B (each stream has its own scratch
    buffer, model._workspace) - two chain launches may meet on the GPU, each needing every CU.  Bits unchanged on both streams; whether
    a launch had to be repaired is reported, not hidden."""
    _skip_unless_chain()
    m1, m2 = _bf16_model(dev, seed=8), _bf16_model(dev, seed=9)
    r1, r2 = _rays(dev, 1024, seed=5), _rays(dev, 1024, seed=6)
    with torch.no_grad():
        want1 = [o.clone() for o in m1(r1)]
        want2 = [o.clone() for o in m2(r2)]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    got1, got2 = [], []
    with torch.no_grad():
        for _ in range(6):
            with torch.cuda.stream(sa):
                got1.append([o.clone() for o in m1(r1)])
            with torch.cuda.stream(sb):
                got2.append([o.clone() for o in m2(r2)])
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        st_a = m1.chain_status()
    with torch.cuda.stream(sb):
        st_b = m2.chain_status()
    print("two streams:", st_a, st_b)
    for outs in got1:
        for a, b in zip(outs, want1):
            assert torch.equal(a, b)
    for outs in got2:
        for a, b in zip(outs, want2):
            assert torch.equal(a, b)
    assert st_a["launches"] == 6 and st_b["launches"] == 6
    assert st_a["recoveries"] <= st_a["launches"] and st_b["recoveries"] <= st_b["launches"]


def test_cooperative_launch_same_bits(dev):
    """M360_TUNE_CHAIN_COOPERATIVE (ops.set_chain_cooperative): the chain through hipLaunchCooperativeKernel - same bits, clean status."""
    from mipnerf360_amd import ops
    _skip_unless_chain()
    packs, g = _packs(1024, 6, dev, 33)
    x = ops.pair_rows(torch.randn(65536, 1024, generator=g).to(dev).bfloat16())
    want = _layer_by_layer(x, packs)
    ops.set_chain_cooperative(True)
    a, b = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    got, st = ops.mlp_chain_bf16_safe(x, a, b, packs)
    ops.set_chain_cooperative(False)
    assert torch.equal(got, want) and st["last_error"] == 0 and st["launches"] == 1, st


def _run_bench(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


def test_bf16_frames_over_two_ranks_with_the_overlapped_gather(dev):
    """`bench.py --config c4 --mlp-dtype bf16 --gpus 2 --backend gloo` - two rank processes render bf16 frames
    (every chunk's NeRF MLP through the chain) while the pixel gather of the previous frame runs on a side stream.  The assembled frame
    must be the SAME BITS overlapped, serial, and rendered by one rank alone; the line carries the chain's counters."""
    size = "300x205"  # 61500 rays: 16 chunks of 4096 (the last one partial)
    res2, two = _run_bench("--gpus", "2", "--backend", "gloo", "--config", "c4", "--mlp-dtype", "bf16", "--steps", "3", "--warmup", "1", "--frame-size", size)
    assert res2.returncode == 0 and two is not None, (res2.stdout[-1500:], res2.stderr[-3000:])
    res1, one = _run_bench("--config", "c4", "--mlp-dtype", "bf16", "--steps", "1", "--warmup", "1", "--frame-size", size)
    assert res1.returncode == 0 and one is not None, (res1.stdout[-1500:], res1.stderr[-3000:])
    assert two["dtype"] == "bf16" and two["frame"]["overlapped"]["overlap"] and not two["frame"]["serial"]["overlap"]
    sums = {two["frame"]["overlapped"]["frame_bits_sum"], two["frame"]["serial"]["frame_bits_sum"], one["frame"]["overlapped"]["frame_bits_sum"]}
    assert len(sums) == 1, sums
    for line in (one, two):
        ch = line["chain"]
        assert ch["launches"] > 0 and ch["recoveries"] <= ch["launches"] and ch["chain_error"] == (ch["recoveries"] > 0), ch
    assert one["chain"]["recoveries"] == 0, one["chain"]  # alone on the GPU nothing needs repairing


@pytest.mark.parametrize("fault", [0, 2])
def test_bf16_forward_in_a_captured_hip_graph(dev, diag, fault):
    """the chain used to probe the device from inside the first bf16 forward (hipMalloc, a kernel on the null stream, a
    blocking copy) - impossible under stream capture.  Nothing is probed any more: the whole bf16 forward - chain launch, its memset,
    the six gated launches - is captured into ONE HIP graph and replayed; replays equal the eager bits, the status block counts every
    replay, and a fault baked into the captured launch (every wave gives up at its first wait) is repaired on every replay."""
    from mipnerf360_amd import ops
    _skip_unless_chain()
    model, rays = _bf16_model(dev), _rays(dev, 512)
    with torch.no_grad():
        want = [o.clone() for o in model(rays)]
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.no_grad(), torch.cuda.stream(side):
        for _ in range(2):
            model(rays)  # this stream's scratch buffer and the packing, outside the capture
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        before = model.chain_status()
    ops.set_chain_debug(0, fault)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g, stream=side):
        res = model(rays)
    ops.set_chain_debug(0, 0)
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        for a, b in zip(res, want):
            assert torch.equal(a, b)
    with torch.cuda.stream(side):
        after = model.chain_status()
    assert after["launches"] - before["launches"] == 3 and after["recoveries"] - before["recoveries"] == (3 if fault else 0), (before, after)


# ------------------------------------------------------------------------------- round 6: the chain for the bf16x3 hidden layers
# (m360_mlp_chain_bf16x3_safe: built, bit-identical, self-checking like the bf16 one - and NOT used by the forward: 16.98 against 17.01 ms per
# step, profiles/r06/bf16x3_chain_NO_GAIN.txt.  The op and its tests stay: the kernel's CHAIN form is one template for both modes.)
def _packs_x3(width, layers, dev, seed):
    from mipnerf360_amd import ops
    g = torch.Generator().manual_seed(seed)
    packs = []
    for _ in range(layers):
        w = (torch.randn(width, width, generator=g) * (2.0 / width) ** 0.5).to(dev)
        b = (torch.randn(width, generator=g) * 0.1).to(dev)
        packs.append(ops.pack_linear_bf16x3(w, b, width, width))
    return packs, g


def _hi_lo(v):
    hi = v.bfloat16()
    return torch.cat([hi, (v - hi.float()).bfloat16()], dim=1).contiguous()


def _layer_by_layer_x3(x, packs):
    from mipnerf360_amd import _lib, ops
    flags = _lib.ACT_RELU | _lib.ROWS_PAIRED_IN | _lib.ROWS_PAIRED_OUT
    for wp, bp in packs:
        x = ops.linear_bf16x3(x, wp, bp, flags)
    return x


@pytest.mark.parametrize("M,layers", [(32768, 6), (65536, 3), (32768, 1)])
def test_x3_chain_clean_launch(dev, M, layers):
    """m360_mlp_chain_bf16x3_safe: the bf16x3 mode's hidden layers ([hi | lo] rows, three products per 64-deep block) in ONE launch -
    bit for bit what `layers` calls of m360_linear_bf16x3 on paired rows give, x_in untouched, one clean launch in the counters."""
    from mipnerf360_amd import ops
    _skip_unless_chain(M, 1024, layers)
    packs, g = _packs_x3(1024, layers, dev, 200 + layers)
    x = ops.pair_rows(_hi_lo(torch.randn(M, 1024, generator=g).to(dev)))
    keep = x.clone()
    for rep in range(2):
        a, b = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
        got, st = ops.mlp_chain_bf16x3_safe(x, a, b, packs)
        assert torch.equal(x, keep)
        want = _layer_by_layer_x3(keep, packs)
        assert torch.equal(got, want), (rep, int((got != want).sum()))
        assert st == dict(launches=1, recoveries=0, timeouts=0, xcc_mismatch=0, last_error=0), st


@pytest.mark.parametrize("fault", [1, 2])
def test_x3_chain_repairs_an_injected_fault(dev, diag, fault):
    from mipnerf360_amd import ops
    _skip_unless_chain()
    packs, g = _packs_x3(1024, 6, dev, 17 + fault)
    x = ops.pair_rows(_hi_lo(torch.randn(65536, 1024, generator=g).to(dev)))
    want = _layer_by_layer_x3(x, packs)
    ops.set_chain_debug(0, fault)
    a, b = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    got, st = ops.mlp_chain_bf16x3_safe(x, a, b, packs)
    ops.set_chain_debug(0, 0)
    assert st["launches"] == 1 and st["recoveries"] == 1 and st["last_error"] == (2 if fault == 1 else 1), st
    assert torch.equal(got, want), (fault, int((got != want).sum()))
