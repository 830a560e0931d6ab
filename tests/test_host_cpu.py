"""CPU-only tests (no GPU): the C-ABI library builds/loads/exports every declared symbol, the host
mirror keeps the reference's API surface and checkpoint layout, the product path fails loudly
without a device, and the sharding logic is correct (incl. world_size-2 gloo runs)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from mipnerf360_amd import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from mipnerf360_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_symbols_are_exported_and_bound(lib):
    from mipnerf360_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "m360.h")).read()
    declared = set(re.findall(r"\b(m360_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/m360.h but not exported by libm360.so"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.m360_version() == 200
    assert lib.m360_contract_workspace_bytes() >= 8192


def test_library_has_no_mutable_globals(lib):
    """"no global mutable state, re-entrant per stream".  Checked on the built object: every writable data symbol of
    libm360.so (nm: .data / .bss) is a HIP kernel handle / runtime registration or one of three named exceptions - the thread-local
    error string, the per-device CU-count memo (an immutable hardware property, atomics) and the host shadow of a __device__ array of
    zeros nothing ever writes.  A switch, a stream, an event or a cache added as a `static` shows up here."""
    import subprocess
    from mipnerf360_amd import _lib
    out = subprocess.run(["nm", "-C", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    allowed = ("m360::g_err", "cu_count()::cached", "m360::g_zero_bias")
    runtime = ("__hip", "_GLOBAL_", "completed.", "__dso_handle", "__TMC_END__", "_DYNAMIC", "__bss_start", "_edata", "_end", "__frame_dummy",
               "__do_init", "__do_fini", "__init", "__fini", "__FRAME_END__", "__data_start", "data_start")
    left = []
    for line in out.splitlines():
        parts = line.split(None, 2)
        if len(parts) < 3 or parts[1] not in "dDbB":
            continue
        name = parts[2]
        if "_kernel" in name or name.startswith(runtime) or name in allowed:
            continue
        left.append(line)
    assert not left, left
    src = "".join(open(os.path.join(ROOT, "mipnerf360_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "mipnerf360_amd", "csrc")) if f.endswith((".hip", ".h")))
    hdr = open(os.path.join(ROOT, "include", "m360.h")).read()
    assert not re.findall(r"m360_set_[a-z]", src) and not re.findall(r"m360_set_[a-z]", hdr)  # no process-wide switch is declared or defined


def test_paired_rows_host_side(lib):
    """include/m360.h "paired rows": ops.pair_rows is its own inverse, moves 64-byte quarters inside 2-row x 64-column blocks of the
    full 256-row tiles only, and m360_linear_bf16_rows_pairable (host logic, no device) names the ring kernel's shapes."""
    from mipnerf360_amd import _lib, ops
    x = torch.arange(300 * 128, dtype=torch.float32).reshape(300, 128).bfloat16()
    y = ops.pair_rows(x)
    assert torch.equal(ops.pair_rows(y), x) and torch.equal(y[256:], x[256:])
    assert torch.equal(y[0, :32], x[0, :32]) and torch.equal(y[0, 32:64], x[1, :32]) and torch.equal(y[1, :32], x[0, 32:64]) and torch.equal(y[1, 32:64], x[1, 32:64])
    assert torch.equal(y[7, 64:96], x[6, 96:128]) and torch.equal(y[6, 96:128], x[7, 64:96])
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.pair_rows(torch.zeros(256, 96))
    q = lib.m360_linear_bf16_rows_pairable
    assert q(_lib.PAIRABLE_LINEAR, 1024, 1024) and q(_lib.PAIRABLE_LINEAR, 256, 256) and q(_lib.PAIRABLE_LINEAR, 1024, 64) and q(_lib.PAIRABLE_LINEAR, 256, 384)
    assert not q(_lib.PAIRABLE_LINEAR, 256, 192) and not q(_lib.PAIRABLE_LINEAR, 128, 256) and not q(_lib.PAIRABLE_LINEAR, 1000, 256) and not q(_lib.PAIRABLE_LINEAR, 256, 128)
    assert q(_lib.PAIRABLE_X3, 1024, 1024) and q(_lib.PAIRABLE_X3, 256, 192) and q(_lib.PAIRABLE_X3_BF16OUT, 1024, 64) and not q(_lib.PAIRABLE_X3, 96, 64)
    assert q(_lib.PAIRABLE_SPLIT, 1024, 384) and not q(_lib.PAIRABLE_SPLIT, 1024, 64)
    assert q(_lib.PAIRABLE_HEADS, 1024, 1024) and q(_lib.PAIRABLE_HEADS_X3, 256, 256) and not q(_lib.PAIRABLE_HEADS, 2048, 1024) and not q(_lib.PAIRABLE_HEADS, 256, 128)
    assert not q(99, 1024, 1024)
    # the layout switch is a per-call bit of m360_hyper_t.tuning (the host mirror keeps it per THREAD): no process-wide m360_set_* is left
    assert ops.set_paired_rows(False) is True and ops.tuning_bits() == _lib.TUNE_PLAIN_ROWS and ops.set_paired_rows(True) is False and ops.tuning_bits() == 0


def test_header_cites_reference_lines():
    hdr = open(os.path.join(ROOT, "include", "m360.h")).read()
    assert len(re.findall(r"(intern/\w+\.py|model\.py):\d+", hdr)) >= 25


def test_argument_validation_without_gpu(lib):
    """Invalid arguments are rejected before any device work (so this runs on CPU)."""
    from mipnerf360_amd import _lib
    assert lib.m360_sample_t(None, None, None, 4, 8, None, None) == -1
    assert "m360_sample_t" in _lib.last_error()
    assert lib.m360_linear(None, 4, 64, None, None, 32, 64, 0, None, 32, None) == -1
    assert lib.m360_resample_t(None, None, None, 1, 0, 0.01, None, None) == -1
    assert lib.m360_forward(None, None, None, 1, None, None, 0, None) == -1
    # round 5's entries: the NaN scan, the bf16 gradient GEMMs, the switches (which only return the previous setting)
    assert lib.m360_params_nan_flag(None, None, 3, None, None) == -1 and "m360_params_nan_flag" in _lib.last_error()
    assert lib.m360_linear_wgrad_bf16(None, 1024, None, 1024, 4096, 1024, 1024, None, None, None, 0, 0, None) == -1
    assert lib.m360_linear_dgrad_bf16(None, 4096, 1024, None, 1024, 1024, None, None, 1024, None) == -1
    assert lib.m360_linear_wgrad_bf16_workspace_bytes(524288, 1024, 1024) >= 16 * 1024 * 1024 * 4
    assert lib.m360_side_create(None) == -1 and "m360_side_create" in _lib.last_error()
    lib.m360_side_destroy(None)  # a no-op, like free(NULL)
    # round 6: m360_pack_many checks its whole list before it launches anything (no nan_flag: not even a memset)
    import ctypes as C
    assert lib.m360_pack_many(None, 2, None, None) == -1 and "m360_pack_many" in _lib.last_error()
    assert lib.m360_pack_many(None, 0, None, None) == 0  # an empty list packs nothing
    items = (_lib.PackItem * 2)()
    items[0] = _lib.PackItem(0x1000, 0, 0x2000, 0, 8, 8, 64, 64, _lib.PACK_BF16, 0)
    items[1] = _lib.PackItem(0x1000, 0, 0x2000, 0, 8, 8, 64, 96, _lib.PACK_BF16, 0)   # k_pad not a multiple of 64
    assert lib.m360_pack_many(C.cast(items, C.c_void_p), 2, None, None) == -1 and "item 1" in _lib.last_error()
    items[1] = _lib.PackItem(0x1000, 0x1000, 0x2000, 0, 8, 8, 64, 64, _lib.PACK_BF16_T, 0)  # a transposed packing has no bias
    assert lib.m360_pack_many(C.cast(items, C.c_void_p), 2, None, None) == -1 and "no bias" in _lib.last_error()
    items[1] = _lib.PackItem(0x1000, 0, 0x2000, 0, 8, 8, 64, 64, 9, 0)
    assert lib.m360_pack_many(C.cast(items, C.c_void_p), 2, None, None) == -1 and "format 9" in _lib.last_error()
    with pytest.raises(RuntimeError, match="m360_ipe"):
        _lib.check(lib.m360_ipe(None, None, 5, None, None), "m360_ipe")
    m = _lib.ModelStruct()
    m.in_ch, m.in_pad, m.hp_pad, m.hn_pad = 58, 64, 256, 1024
    need = lib.m360_forward_workspace_bytes(4096, 128, m)
    S = 4096 * 128
    assert need >= 2 * S * 1024 * 4 + S * 64 * 4 and need < 2 * S * 1024 * 4 + S * 64 * 4 + S * 8 * 4 * 4 + (64 << 20)  # + the fused last layer's partial head sums


def test_state_dict_layout_matches_reference():
    """30 tensors, names/shapes of SURVEY.md §8b (the checkpoint wire format)."""
    from mipnerf360_amd.model import mipNeRF360
    m = mipNeRF360(device=torch.device("cpu"))
    sd = m.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, tuple(s)) for k, s in synthetic.state_dict_spec()]
    assert len(sd) == 30 and sum(v.numel() for v in sd.values()) == 7_624_453
    assert len(list(m.buffers())) == 0
    w = sd["prop_net.model.0.weight"]
    assert float(w.abs().max()) <= (6.0 / 58) ** 0.5 + 1e-6  # kaiming_uniform_, model.py:8-12
    assert m.prop_net.input_size == 58 and m.nerf_net.input_size == 58
    assert isinstance(m.nerf_net.density_activation, torch.nn.Softplus)


def test_api_surface_matches_reference_signatures():
    import inspect
    from mipnerf360_amd import model
    from mipnerf360_amd.intern import encoding, parameterization, ray
    sig = lambda f: list(inspect.signature(f).parameters)  # noqa: E731
    assert sig(model.mipNeRF360.__init__)[1:12] == ["randomized", "num_samples", "hidden_proposal", "hidden_nerf",
                                                    "density_bias", "rgb_padding", "resample_padding", "white_bkgd",
                                                    "viewdir_min_deg", "viewdir_max_deg", "device"]
    assert sig(model.mipNeRF360.__init__)[12:] == ["num_samples_fine", "mlp_dtype"]  # keyword extensions, last
    assert sig(model.mipNeRF360.render_image)[1:] == ["rays", "height", "width", "chunks"]
    assert inspect.signature(model.mipNeRF360.render_image).parameters["chunks"].default == 4096
    assert sig(model.nerf_net.forward)[1:] == ["rays", "t_vals", "coarse_weights"]
    assert sig(ray.sample_along_rays) == ["origins", "directions", "radii", "num_samples", "near", "far", "randomized"]
    assert sig(ray.resample_along_rays) == ["origins", "directions", "radii", "t_vals", "weights", "randomized",
                                            "resample_padding"]
    assert sig(ray.volumetric_rendering) == ["rgb", "density", "t_vals", "dirs", "white_bkgd"]
    assert sig(ray.sorted_piecewise_constant_pdf) == ["bins", "weights", "num_samples", "randomized"]
    assert inspect.signature(ray.sorted_piecewise_constant_pdf).parameters["randomized"].default is True
    assert ray.Rays._fields == ("origins", "directions", "viewdirs", "radii", "near", "far")
    assert sig(parameterization.para_rays) == ["t_vals", "origins", "directions", "radii", "diag"]
    assert sig(parameterization.conical_frustum_to_gaussian) == ["d", "t0", "t1", "base_radius", "diag", "stable"]
    assert sig(parameterization.gaussian_to_xyz) == ["d", "t_mean", "t_var", "r_var", "diag"]
    for name in ("g", "t_to_s", "s_to_t", "contract", "gaussian_contract"):
        assert callable(getattr(parameterization, name))
    assert encoding.PositionalEncoding().P.shape == (21, 3)
    assert torch.equal(encoding.ViewdirectionEncoding(0, 4).scales, torch.tensor([1.0, 2.0, 4.0, 8.0]))
    t = ray.Rays(*[torch.ones(2, 1) * i for i in range(6)])
    assert ray.namedtuple_map(lambda x: x * 2, t).far[0, 0] == 10


def test_train_eval_mirror_reference_quirk():
    from mipnerf360_amd.model import mipNeRF360
    m = mipNeRF360(randomized=True, num_samples=4, hidden_proposal=32, hidden_nerf=32, device=torch.device("cpu"))
    # model.py:276-283: eval() sets randomized=False, but nn.Module.eval() calls self.train(False), which
    # restores init_randomized; the sub-nets keep the flag they were built with
    assert m.eval() is m and m.training is False
    assert m.randomized is True and m.prop_net.randomized is True and m.nerf_net.randomized is True
    assert m.train() is m and m.randomized is True and m.training is True


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: CPU tensors are rejected with a clear error, never silently computed."""
    from mipnerf360_amd.intern import ray
    from mipnerf360_amd.intern.ray import Rays
    from mipnerf360_amd.model import mipNeRF360
    m = mipNeRF360(num_samples=4, hidden_proposal=32, hidden_nerf=32, device=torch.device("cpu"))
    r = synthetic.make_rays("lego", 4)
    rays = Rays(*[torch.from_numpy(r[k]) for k in synthetic.RAY_FIELDS])
    with pytest.raises(RuntimeError, match="HIP"):
        m(rays)
    with pytest.raises(RuntimeError, match="HIP"):
        ray.volumetric_rendering(torch.zeros(1, 4, 3), torch.zeros(1, 4, 1), torch.zeros(1, 5), torch.zeros(1, 3), False)


def test_missing_library_is_an_error(tmp_path, monkeypatch):
    from mipnerf360_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libm360.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mipnerf360_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/", ""), f"{f} mentions the oracle"
                assert "/root/reference" not in src


def test_synthetic_generators_are_deterministic():
    a, b = synthetic.make_rays("garden", 64, seed=1), synthetic.make_rays("garden", 64, seed=1)
    assert all(np.array_equal(a[k], b[k]) and a[k].dtype == np.float32 for k in synthetic.RAY_FIELDS)
    assert np.allclose(np.linalg.norm(a["viewdirs"], axis=1), 1, atol=1e-6) and (a["near"] == 0).all()
    lego = synthetic.make_rays("lego", 64, seed=1)
    n = np.linalg.norm(lego["directions"], axis=1)
    assert (n >= 1 - 1e-6).all() and (n < 1.2).all() and (lego["near"] == 2).all() and (lego["far"] == 6).all()
    s1, s2 = synthetic.make_state_dict(32, 64, seed=3), synthetic.make_state_dict(32, 64, seed=3)
    assert all(np.array_equal(s1[k], s2[k]) for k in s1)


def test_dropin_aliases():
    import mipnerf360_amd
    saved = {k: sys.modules.get(k) for k in ("model", "intern", "intern.ray", "intern.parameterization",
                                             "intern.encoding", "intern.utils")}
    try:
        mipnerf360_amd.install_dropin()
        import model as ref_named_model
        from intern.ray import Rays, namedtuple_map  # noqa: F401
        from intern.parameterization import t_to_s  # noqa: F401
        from intern.utils import to8b  # noqa: F401
        assert ref_named_model.mipNeRF360 is mipnerf360_amd.model.mipNeRF360
        # the drop-in runs the reference's own scripts: its models reproduce the in-place g() drift by default (G14) ...
        assert mipnerf360_amd.model.MUTATE_LIKE_REFERENCE is True
        mipnerf360_amd.install_dropin(mutate_like_reference=False)
        assert mipnerf360_amd.model.MUTATE_LIKE_REFERENCE is False  # ... direct use of mipnerf360_amd.model never mutates
    finally:
        mipnerf360_amd.model.MUTATE_LIKE_REFERENCE = False
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


# ------------------------------------------------------------------------------- sharding
@pytest.mark.parametrize("n,chunks,world", [(1016814, 4096, 8), (1016814, 128, 4), (768, 128, 2), (100, 4096, 8),
                                            (0, 16, 2), (4097, 4096, 2), (33, 1, 3)])
def test_chunk_partition_keeps_reference_chunks(n, chunks, world):
    from mipnerf360_amd.distributed import chunk_partition
    spans = chunk_partition(n, chunks, world)
    assert len(spans) == world and spans[0][0] == 0 and spans[-1][1] == n
    for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
        assert e0 == b1 and b0 <= e0
    for b, e in spans:
        assert b % chunks == 0 or b == n            # every rank starts on a chunk boundary of the reference loop
        assert e % chunks == 0 or e == n
    sizes = [(e - b + chunks - 1) // chunks for b, e in spans]
    assert max(sizes) - min(s for s in sizes) <= max(sizes)  # contiguous blocks, last ranks may be short/empty


GLOO_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from mipnerf360_amd.distributed import chunk_partition, gather_pixels, render_rays_sharded
from mipnerf360_amd.intern.ray import Rays
from mipnerf360_amd import synthetic
from oracle import ref_path as O   # checker only (tests/)

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, chunks = int(sys.argv[2]), int(sys.argv[3])

# 1) gather_pixels with ragged spans
spans = chunk_partition(n, chunks, world)
b, e = spans[rank]
full = torch.arange(n * 5, dtype=torch.float32).reshape(n, 5)
got = gather_pixels(full[b:e].clone(), spans)
assert torch.equal(got, full), "gather mismatch"

# 2) sharded render == single-process render with the same chunk partition (fake renderer = CPU oracle;
#    the product renderer needs a GPU, the sharding/gather logic under test does not)
sd = O.to_torch_state_dict(synthetic.make_state_dict(32, 32, seed=1))
hp = O.Hyper(num_samples=8)
class OracleModel:
    def render_rays(self, rays, chunks):
        outs = [O.forward(O.Rays(*[f[i:i + chunks] for f in rays]), sd, hp) for i in range(0, rays[0].shape[0], chunks)]
        if not outs:
            return torch.zeros(0, 3), torch.zeros(0), torch.zeros(0)
        return tuple(torch.cat([o[j] for o in outs], 0) for j in range(3))
r = synthetic.make_rays("garden", n, seed=2)
rays = Rays(*[torch.from_numpy(r[k]) for k in synthetic.RAY_FIELDS])
rgb, d, a = render_rays_sharded(OracleModel(), rays, chunks)
rgb1, d1, a1 = OracleModel().render_rays(rays, chunks)
assert torch.equal(rgb, rgb1) and torch.equal(d, d1) and torch.equal(a, a1), "sharded != single"
# and a different chunking really changes the result (so the partition matters)
rgb2, _, _ = OracleModel().render_rays(rays, chunks * 2)
assert not torch.equal(rgb1, rgb2)
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
"""


@pytest.mark.parametrize("n,chunks", [(100, 16), (64, 16), (17, 16)])
def test_sharded_render_world2_gloo(tmp_path, n, chunks):
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29511 + n % 50), str(script), ROOT, str(n), str(chunks)]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count("OK") == 2


GLOO_SHARDED_BATCH_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from mipnerf360_amd.distributed import forward_sharded
from mipnerf360_amd import synthetic
from oracle import ref_path as O   # checker only (tests/): stands in for the HIP stages, which need a GPU

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
per = int(sys.argv[2])
sd = O.to_torch_state_dict(synthetic.make_state_dict(32, 32, seed=1))
hp = O.Hyper(num_samples=8)

class OracleStages:
    def sharded_sample(self, rays):
        return O.sample_t(rays.near, rays.far, hp.num_samples).expand(rays.origins.shape[0], -1).contiguous()
    def sharded_sumsq(self, rays, t):
        return O.mean_sumsq(t, rays.directions, rays.radii)
    def sharded_prop(self, rays, t, norm):
        return O.prop_forward_from_t(rays, sd, hp, t, norm)
    def sharded_nerf(self, rays, t_new, norm):
        return O.nerf_forward_from_t(rays, sd, hp, t_new, norm)

r = synthetic.make_rays("garden", per * world, seed=2)
r["origins"] = r["origins"] * 3.0
whole = O.rays_from_numpy(r)
mine = O.Rays(*[f[rank * per:(rank + 1) * per] for f in whole])
with torch.no_grad():
    got = forward_sharded(OracleStages(), mine)
    ref = O.forward(whole, sd, hp)
    alone = O.forward(mine, sd, hp)
for g, w in zip(got, ref):
    assert torch.allclose(g, w[rank * per:(rank + 1) * per], atol=2e-6, rtol=2e-6), "sharded batch != whole batch"
assert float((alone[0] - ref[0][rank * per:(rank + 1) * per]).abs().max()) > 1e-4, "the global norm must matter in this test"
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
"""


def test_sharded_batch_global_norm_world2_gloo(tmp_path):
    """one batch over two ranks, contraction norm rebuilt from an all-reduce of per-rank sums of squares."""
    script = tmp_path / "worker.py"
    script.write_text(GLOO_SHARDED_BATCH_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", str(script), ROOT, "24"]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count("OK") == 2


GLOO_C5_SPLIT_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from mipnerf360_amd.distributed import forward_sharded
from mipnerf360_amd import synthetic
from oracle import ref_path as O   # checker only (tests/): stands in for the HIP stages, which need a GPU

torch.set_num_threads(1)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
per, N = 1024, 256                      # BASELINE configs[4]: 8192 rays x 256 samples over 8 ranks, bf16 MLP
sd = O.to_torch_state_dict(synthetic.make_state_dict(32, 32, seed=1))
hp = O.Hyper(num_samples=N, mlp_bf16=1)

class OracleStages:
    def sharded_sample(self, rays):
        return O.sample_t(rays.near, rays.far, hp.num_samples).expand(rays.origins.shape[0], -1).contiguous()
    def sharded_sumsq(self, rays, t):
        return O.mean_sumsq(t, rays.directions, rays.radii)
    def sharded_prop(self, rays, t, norm):
        return O.prop_forward_from_t(rays, sd, hp, t, norm)
    def sharded_nerf(self, rays, t_new, norm):
        return O.nerf_forward_from_t(rays, sd, hp, t_new, norm)

r = synthetic.make_rays("garden", per * world, seed=2)
r["origins"] = r["origins"] * 3.0
whole = O.rays_from_numpy(r)
mine = O.Rays(*[f[rank * per:(rank + 1) * per] for f in whole])
with torch.no_grad():
    got = forward_sharded(OracleStages(), mine)
    # the whole 8192-ray batch on ONE rank (what a single device renders), handed to the others
    ref = [torch.empty(per * world, 3), torch.empty(per * world), torch.empty(per * world)]
    if rank == 0:
        torch.set_num_threads(8)
        ref = [t.contiguous() for t in O.forward(whole, sd, hp)]
        torch.set_num_threads(1)
    for t in ref:
        dist.broadcast(t, src=0)
    alone = O.forward(mine, sd, hp) if rank == world - 1 else None
    # (and in fp32 arithmetic, where what the norm is worth can be told from rounding noise)
    hp32 = O.Hyper(num_samples=N)
    alone32 = O.forward(mine, sd, hp32) if rank == world - 1 else None
sl = slice(rank * per, (rank + 1) * per)
# one 8-byte all-reduce per stage rebuilds the 8192-ray norm up to the summation order of 8 partial sums; the bf16 emulation rounds
# activations to 8 bits, so an ulp of the norm may flip a rounding somewhere: 5e-4 on colours in [0, 1], far below what the norm is worth
for g, w, tol in zip(got, ref, (5e-4, 5e-4, 5e-4)):
    err = float((g - w[sl]).abs().max() / max(1.0, float(w[sl].abs().max())))
    assert err <= tol, ("sharded batch != whole batch", rank, err)
if alone is not None:
    # the shard rendered with its OWN 1024-ray norm is another picture: in fp32 the difference is far above rounding ...
    with torch.no_grad():
        whole32 = O.forward(whole, sd, hp32) if rank == world - 1 else None
    gap32 = float((alone32[0] - whole32[0][sl]).abs().max())
    assert gap32 > 1e-4, ("the global norm must matter in this test", gap32)
    print("norm gap fp32", gap32, "bf16", float((alone[0] - ref[0][sl]).abs().max()), flush=True)
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
"""


def test_sharded_batch_c5_split_world8_gloo(tmp_path):
    """BASELINE configs[4] as it shards - 8192 rays x 256 samples as 8 x 1024 over a world of 8 (gloo, the oracle's
    bf16-emulating stages standing in for the HIP ones): every rank's rows equal those rows of the single-device 8192-ray forward."""
    script = tmp_path / "worker.py"
    script.write_text(GLOO_C5_SPLIT_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29583", str(script), ROOT]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count("OK") == 8


GLOO_FRAME_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from mipnerf360_amd.distributed import (PixelGather, check_replicas, chunk_partition, partition_efficiency,
                                        render_local_block, render_rays_sharded)
from mipnerf360_amd.intern.ray import Rays
from mipnerf360_amd import synthetic
from oracle import ref_path as O   # checker only (tests/): stands in for the HIP renderer, which needs a GPU

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
h, w, chunks = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = h * w
sd = O.to_torch_state_dict(synthetic.make_state_dict(32, 32, seed=1))
hp = O.Hyper(num_samples=8)
pose = np.concatenate([np.eye(3), np.array([[0.05], [-0.02], [0.1]])], 1).astype(np.float32)
full = O.generate_rays(pose[None], h, w, 0.9 * w, 0.0, 1.0, ndc=True)
full = Rays(*[torch.from_numpy(full[k]) for k in synthetic.RAY_FIELDS])

class OracleRenderer(torch.nn.Module):       # has parameters: check_replicas fingerprints them
    def __init__(self, scale=1.0):
        super().__init__()
        self.p = torch.nn.Parameter(torch.full((3,), float(scale)))
        self.num_samples = 8
    def render_rays(self, rays, chunks):
        outs = [O.forward(O.Rays(*[f[i:i + chunks] for f in rays]), sd, hp) for i in range(0, rays[0].shape[0], chunks)]
        if not outs:
            return torch.zeros(0, 3), torch.zeros(0), torch.zeros(0)
        return tuple(torch.cat([o[j] for o in outs], 0) for j in range(3))

m = OracleRenderer()
single = m.render_rays(full, chunks)
# the bench's c4 frame loop in miniature: two frames through the two slots of one PixelGather, every rank renders only
# the rays of its own span ("generated on the device per rank": here sliced from the oracle's frame)
pg = PixelGather(n, chunks, "cpu", slots=2)
assert pg.spans == chunk_partition(n, chunks, world)
outs = [(torch.empty(n, 3), torch.empty(n), torch.empty(n)) for _ in range(2)]
for frame in range(3):
    s = frame % 2
    b, e = pg.span
    local = Rays(*[f[b:e] for f in full])
    render_local_block(m, None, chunks, pg, slot=s, local_rays=local)
    pg.gather(s)
    got = pg.assemble(s, out=outs[s])
    assert all(torch.equal(g, x) for g, x in zip(got, single)), "sharded frame != single-process frame"
fresh = pg.assemble(0)
assert fresh[0].data_ptr() != outs[0][0].data_ptr() and torch.equal(fresh[0], single[0])
# the public entry point, and the efficiency bound the bench reports
got = render_rays_sharded(m, full, chunks)
assert all(torch.equal(g, x) for g, x in zip(got, single))
n_chunks = (n + chunks - 1) // chunks
assert partition_efficiency(n, chunks, world) <= 1.0
assert abs(partition_efficiency(n, chunks, world) - n / (world * max(e - b for b, e in pg.spans))) < 1e-12
# replicas with different weights are refused
bad = OracleRenderer(scale=1.0 + rank)
try:
    check_replicas(bad)
    raise SystemExit("check_replicas accepted different weights")
except RuntimeError as ex:
    assert "different weights" in str(ex)
check_replicas(m)
# ONE rank changes its weights in place after an agreeing check (ADVICE r3): every rank still enters the same collective
# (no per-rank cache decides that) and every rank gets the error - not a hang, not a mismatched collective
if rank == world - 1:
    with torch.no_grad():
        m.p.add_(0.5)
try:
    check_replicas(m)
    raise SystemExit("check_replicas missed an in-place update on one rank")
except RuntimeError as ex:
    assert "different weights" in str(ex)
if rank == world - 1:
    with torch.no_grad():
        m.p.sub_(0.5)
check_replicas(m)
# the same values in another order inside one tensor (equal plain sums of the bit patterns) are different weights
perm = OracleRenderer()
with torch.no_grad():
    perm.p.copy_(torch.tensor([1.0, 2.0, 3.0] if rank % 2 == 0 else [3.0, 2.0, 1.0]))
try:
    check_replicas(perm)
    raise SystemExit("check_replicas accepted permuted weights")
except RuntimeError as ex:
    assert "different weights" in str(ex)
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
"""


@pytest.mark.parametrize("h,w,chunks", [(11, 13, 16), (9, 7, 64), (8, 8, 16)])
def test_sharded_frame_pipeline_world2_gloo(tmp_path, h, w, chunks):
    """bench.py --config c4 / render_view_sharded in miniature (BASELINE configs[3]): ragged last chunk, a frame smaller than
    one chunk (rank 1 idle), an exact split."""
    script = tmp_path / "worker.py"
    script.write_text(GLOO_FRAME_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29601 + h), str(script), ROOT, str(h), str(w), str(chunks)]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count("OK") == 2


def test_sharded_frame_pipeline_world8_gloo(tmp_path):
    """The same at the node size the driver scales to: 8 ranks, 143 rays in 9 chunks of 16 -> 2 chunks per rank, ranks 0-3
    full, rank 4 a full + the ragged chunk, ranks 5-7 WITHOUT rays (empty spans must gather, assemble and fingerprint too)."""
    script = tmp_path / "worker.py"
    script.write_text(GLOO_FRAME_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29688", str(script), ROOT, "11", "13", "16"]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count("OK") == 8


def test_bench_dry_launch_prints_the_child_command():
    """bench.py --gpus N without a launcher starts the ranks itself: N fresh processes under torch.distributed.run."""
    import json
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "7", "--warmup", "2",
                          "--dry-launch"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads(res.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert d["ranks"] == 8 and cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 1024
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "7", "--warmup", "2"]      # the ranks get the same arguments, minus --dry-launch
    # the parent must not import torch (it must never touch the GPU before the ranks exist)
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--dry-launch']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit:\n    pass\n"
            "assert 'torch' not in sys.modules, 'launcher imported torch'\n" % os.path.join(ROOT, "bench.py"))
    res = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert res.returncode == 0, res.stderr[-2000:]


def test_bench_launcher_starts_ranks_and_reports_their_failure():
    """Here (no GPU) both ranks stop with 'needs a HIP device': the launcher must really start them (not refuse with its own
    SystemExit), relay their message and exit non-zero."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert res.returncode == 0 and '"n_gpus": 2' in res.stdout
        return
    if torch.cuda.is_available():  # a 1-GPU box: the parent's pre-flight refuses before any rank exists (see the test below)
        assert res.returncode == 2 and "only 1 GPU(s) are visible" in res.stderr
        return
    assert res.returncode != 0
    assert "launch with torch.distributed.run" not in res.stderr
    assert "2-rank child run failed" in res.stderr
    if not torch.cuda.is_available():
        assert "needs a HIP device" in res.stderr


def test_bench_preflight_counts_gpus_without_touching_them(tmp_path, monkeypatch):
    """`bench.py --gpus N` beyond the visible GPUs is refused by the PARENT with one line and rc = 2 (round 3
    failed only after N processes had initialised, with RCCL's 'Duplicate GPU').  The count comes from the KFD topology in sysfs
    (simd_count > 0 = a GPU node) narrowed by *_VISIBLE_DEVICES - no HIP call, no torch import.  Here: a fake topology."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for i, simd in enumerate((0, 0, 256, 256, 256)):       # two CPU nodes, three GPU nodes
        d = tmp_path / "nodes" / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {16 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count(str(tmp_path / "nodes")) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count(str(tmp_path / "nodes")) == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_gpu_count(str(tmp_path / "nodes")) == 1
    assert bench.visible_gpu_count(str(tmp_path / "missing")) is None     # no driver: unknown, the ranks report it themselves
    # and the refusal itself, through launch(): 4 ranks on (fake) 1 visible GPU
    monkeypatch.setattr(bench, "visible_gpu_count", lambda sysfs=None: 1)
    args = bench.parse_args(["--gpus", "4"])
    assert bench.launch(args, ["--gpus", "4"]) == 2
    args = bench.parse_args(["--gpus", "4", "--dry-launch"])
    assert bench.launch(args, ["--gpus", "4", "--dry-launch"]) == 0       # printing the command needs no GPU


def test_dropin_falls_through_to_reference_helpers(tmp_path):
    """install_dropin(reference_root=...): modules without a mirror (intern.scheduler) and names missing from a mirror
    (camera paths of intern.pose, intern.utils.normalize) resolve to the reference's own files; mirrored names win.
    Uses a stand-in tree (the real reference is not available where the GPU tests run)."""
    ref = tmp_path / "ref"
    (ref / "intern").mkdir(parents=True)
    (ref / "intern" / "__init__.py").write_text("")
    (ref / "intern" / "scheduler.py").write_text("def lr_decay(*a, **k):\n    return 'reference lr_decay'\n")
    (ref / "intern" / "pose.py").write_text("def generate_spiral_cam_to_world(*a):\n    return 'reference spiral'\n"
                                            "def visualize_depth(*a):\n    return 'reference visualize_depth'\n")
    (ref / "intern" / "utils.py").write_text("def normalize(x):\n    return 'reference normalize'\n")
    code = f"""
import sys
sys.path.insert(0, {ROOT!r})
import mipnerf360_amd
mipnerf360_amd.install_dropin(reference_root={str(ref)!r})
from intern.scheduler import lr_decay
from intern.pose import generate_spiral_cam_to_world, visualize_depth
from intern.utils import normalize, to8b
from intern.ray import Rays, convert_to_ndc, namedtuple_map
from intern.loss import Loss_prop, Loss_nerf, Loss_dist, mse_to_psnr
from model import mipNeRF360
assert lr_decay() == 'reference lr_decay' and generate_spiral_cam_to_world() == 'reference spiral'
assert normalize(1) == 'reference normalize'
assert visualize_depth.__module__.startswith('mipnerf360_amd') and to8b.__module__.startswith('mipnerf360_amd')
assert mipNeRF360.__module__ == 'mipnerf360_amd.model'
try:
    from intern.pose import no_such_name
except ImportError:
    print('OK')
"""
    res = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout[-3000:]


def test_hot_path_names_never_fall_through_to_the_reference(tmp_path):
    """every hot-path name of the reference's intern/* and model.py must resolve to the
    mirror (HIP path) even when a reference root is installed whose files define the same names - and a name that is NOT
    on the allowlist of out-of-scope host helpers must raise instead of silently running the reference's CPU code.
    No bytecode may be left in the reference tree."""
    ref = tmp_path / "ref"
    (ref / "intern").mkdir(parents=True)
    (ref / "intern" / "__init__.py").write_text("")
    hot = {
        "pose": ["depth_to_normals", "sinebow", "visualize_normals", "visualize_depth"],
        "utils": ["to8b"],
    }
    for mod, names in hot.items():
        body = "".join(f"def {n}(*a, **k):\n    return 'REFERENCE CPU CODE'\n" for n in names)
        body += "def secret_cpu_helper(*a):\n    return 'REFERENCE CPU CODE'\n"
        body += "def normalize(x):\n    return 'reference normalize'\n" if mod == "utils" else \
                "def poses_avg(p):\n    return 'reference poses_avg'\n"
        (ref / "intern" / f"{mod}.py").write_text(body)
    mirrored = {
        "intern.ray": ["Rays", "namedtuple_map", "sample_along_rays", "resample_along_rays", "volumetric_rendering",
                       "sorted_piecewise_constant_pdf", "convert_to_ndc"],
        "intern.parameterization": ["g", "t_to_s", "s_to_t", "contract", "gaussian_to_xyz", "gaussian_contract",
                                    "conical_frustum_to_gaussian", "para_rays"],
        "intern.encoding": ["PositionalEncoding", "ViewdirectionEncoding"],
        "intern.utils": ["to8b"],
        "intern.pose": ["depth_to_normals", "sinebow", "visualize_normals", "visualize_depth"],
        "intern.loss": ["Loss_prop", "Loss_nerf", "Loss_dist", "mse_to_psnr"],
        "intern.distillation": ["bounds", "loss_prop"],
        "intern.regularization": ["loss_dist"],
        "model": ["mipNeRF360", "prop_net", "nerf_net"],
    }
    code = f"""
import importlib, os, sys
sys.path.insert(0, {ROOT!r})
import mipnerf360_amd
mipnerf360_amd.install_dropin(reference_root={str(ref)!r})
mirrored = {mirrored!r}
for mod, names in mirrored.items():
    m = importlib.import_module(mod)
    for n in names:
        obj = getattr(m, n)
        owner = getattr(obj, '__module__', '')
        assert owner.startswith('mipnerf360_amd'), (mod, n, owner)
import intern.pose, intern.utils
assert intern.utils.normalize(1) == 'reference normalize' and intern.pose.poses_avg(1) == 'reference poses_avg'
for m in (intern.pose, intern.utils):
    try:
        m.secret_cpu_helper
    except AttributeError:
        pass
    else:
        raise SystemExit('a name outside the allowlist fell through to the reference')
pyc = [f for r, d, fs in os.walk({str(ref)!r}) for f in fs if f.endswith('.pyc')]
assert not pyc, pyc
print('OK')
"""
    res = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout[-3000:]


def test_header_is_plain_c99(tmp_path):
    """include/m360.h is the boundary a C caller binds: it must compile as C99 on its own (no torch / HIP / C++ types)."""
    import shutil
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "c.c"
    src.write_text('#include "m360.h"\nint main(void) { m360_hyper_t h = {128, 0, 4, 0, -1.0f, 0.001f, 0.01f, 0, 0, 0};\n'
                   '  m360_rays_t r = {0}; m360_model_t m = {0}; m360_outputs_t o = {0};\n'
                   '  return (int)(h.num_samples_fine + h.norm_group_rays + (r.origins != 0) + m.in_ch + (o.rgb != 0)) + M360_OK; }\n')
    res = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                          str(tmp_path / "c.o")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout


def test_generated_kernel_schedules_are_current(tmp_path):
    """The K-step / slab / stage bodies of the generated kernels are committed next to their sources; they must be exactly what
    the generators in tools/ emit (a hand edit of the .inc, or a generator change without regenerating, fails here)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for tool, outs in (("gen_hd_kstep.py", {"OUT": "mipnerf360_amd/csrc/m360_linear_hd_gen.inc"}),
                       ("gen_w32_slab.py", {"OUT": "mipnerf360_amd/csrc/diag/m360_linear_bf16_w32_gen.inc"}),
                       ("gen_w16_slab.py", {"OUT": "mipnerf360_amd/csrc/m360_linear_bf16_w16_gen.inc",
                                            "OUT_X3": "mipnerf360_amd/csrc/m360_linear_bf16_w16x3_gen.inc"})):
        spec = importlib.util.spec_from_file_location(tool[:-3], os.path.join(root, "tools", tool))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        for attr, rel in outs.items():
            setattr(mod, attr, str(tmp_path / os.path.basename(rel)))
        argv = sys.argv
        sys.argv = [tool]
        try:
            mod.main()
            if hasattr(mod, "main_x3"):
                mod.main_x3()
        finally:
            sys.argv = argv
        for attr, rel in outs.items():
            assert open(getattr(mod, attr)).read() == open(os.path.join(root, rel)).read(), f"{rel} is stale: run python tools/{tool}"


def test_ring_kernel_store_instructions_match_its_counted_waits(tmp_path):
    """The one-wave bf16 ring kernel waits with counted vmcnt across its epilogue: the generated stage bodies assume W16_STORES
    vector-memory stores per lane and tile (32; 64 as [hi | lo]; 8 partial-sum stores with fused heads).  The head stores are plain
    C++ the compiler could merge or split - then a counted wait would let a needed LDS-DMA piece stay in flight.  Compile the device
    code and count: every instantiation must hold exactly the number its waits were generated for."""
    import re
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "mipnerf360_amd", "csrc")
    out = str(tmp_path / "m360_linear.s")
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                          "-S", "--cuda-device-only", os.path.join(csrc, "m360_linear.hip"), "-o", out],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    text = open(out).read()
    seen = 0
    # _ZN4m3603w1622linear_bf16_w16_kernelILi<ACT>ELi<ABL>ELb<STAMP>ELb<X3>ELb<ONE_BLOCK>ELi<HEADS>ELb<SPLIT>ELb<LDSEPI>ELb<PAIR>ELb<CHAIN>ELb<KEEP_Y>EEE...
    # (ABL: 0, or 128 = the same kernel with temporal stores: the row blocks' and the layer chain's instantiations)
    split_without_x3 = paired = chains = 0
    keeps = 0
    for m in re.finditer(r"^(_ZN4m3603w1622linear_bf16_w16_kernelILi(\d)ELi(?:0|128)ELb0ELb([01])ELb([01])ELi(\d)ELb([01])ELb0ELb([01])ELb([01])ELb([01])EEE\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):  # (not up to the first s_endpgm: the gate's early return has one)
        x3, heads, split, pair, chain, keep, body = m.group(3) == "1", int(m.group(5)), m.group(6) == "1", m.group(7) == "1", m.group(8) == "1", m.group(9) == "1", m.group(10)
        chains += int(chain)
        keeps += int(keep)
        # every instantiation: the gate at the kernel's entry (one atomic add: the recovery counter of a gated re-run launch).  The layer
        # chain (m360_mlp_chain_bf16) on top of that: counter adds per tile, its self-checks (XCC_ID read, a compare-and-swap on its slot
        # word, the error bit), activation pieces that bypass the CU's L1
        assert (body.count("global_atomic_add") > 1) == chain and body.count("global_atomic_add") >= 1, f"{m.group(1)}: the gate / the chain's hand-over instructions"
        assert not chain or (len(re.findall(r"buffer_load_dword[^\n]* lds[^\n]* sc1|buffer_load_dword[^\n]* sc1[^\n]* lds", body)) == 64 and "HW_REG_XCC_ID" in body and "global_atomic_cmpswap" in body and "global_atomic_or" in body and "s_memrealtime" in body), f"{m.group(1)}: the chain's self-checks"
        want = 8 if heads else (64 if split else 32)
        if keep:
            want = 32 + 8  # KEEP_Y (the tape-keeping last layer of the bf16 training path): the layer's own rows AND the partial head sums
        if chain:
            want *= 2  # one of two store policies per tile (temporal inside the chain, non-temporal for its last layer): a wave-uniform branch
        split_without_x3 += int(split and not x3)
        paired += int(pair)
        # paired rows (include/m360.h): the lanes' own pieces are whole lines - no exchange between lanes in that epilogue
        assert (body.count("v_cndmask_b32_dpp") == 0) == (pair or (heads > 0 and not keep)), f"{m.group(1)}: lane exchange in a paired-rows / fused-heads epilogue, or none in a plain one"
        got = len(re.findall(r"\bglobal_store_dwordx4" if chain else r"\bglobal_store_", body))  # (the chain also stores its error word where a wait gives up)
        assert got == want, f"{m.group(1)}: {got} store instructions, the counted waits assume {want}"
        assert "scratch_" not in body, f"{m.group(1)} spills"
        seen += 1
    assert seen >= 24 and paired >= 10 and chains == 2 and keeps == 2, f"(chains: the bf16 one and, since round 6, the bf16x3 one) only {seen} ring-kernel instantiations found ({paired} with paired rows out, {chains} layer chain, {keeps} keeping y beside the heads)"
    assert split_without_x3 == 4, "the x6 first layer of the bf16x3 mode (plain K loop, [hi | lo] output; none / ReLU / ReLU with paired rows out, with non-temporal or temporal stores) is missing"


def test_design_md_numbers_are_generated_from_profiles():
    """DESIGN.md carries its measured numbers (headline, per-launch table, named workloads) in a block that
    tools/summarize_profiles.py --markdown regenerates from profiles/<tag>/: the block in the file must be exactly what the
    tool prints today - numbers in the docs are never typed in."""
    import re
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"<!-- profiles:begin (\w+) -->\n(.*?)<!-- profiles:end -->", design, re.S)
    assert m, "DESIGN.md lost its <!-- profiles:begin <tag> --> ... <!-- profiles:end --> block"
    tag, block = m.group(1), m.group(2)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profiles.py"), "--markdown", tag],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    assert block == res.stdout, "DESIGN.md's profile block is stale: re-run tools/summarize_profiles.py --sync-docs " + tag
    assert "rays/s" in block and "last_step_kernel_trace.csv" in block
    # README.md and INTEGRATION.md quote their few numbers from the SAME bench.json, through the same tool (one number per quantity)
    for name in ("README.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, name)).read()
        m2 = re.search(r"<!-- state:begin (\w+) -->\n(.*?)<!-- state:end -->", text, re.S)
        assert m2, f"{name} lost its <!-- state:begin <tag> --> block"
        assert m2.group(1) == tag, f"{name} quotes profiles/{m2.group(1)}, DESIGN.md profiles/{tag}"
        res2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profiles.py"), "--readme", tag],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        assert res2.returncode == 0 and m2.group(2) == res2.stdout, f"{name}'s state block is stale: re-run tools/summarize_profiles.py --sync-docs " + tag
        outside = text[:m2.start()] + text[m2.end():]
        assert not re.search(r"\d[\d.]* ?(k rays/s|ms per (step|iteration)|ms/step)", outside), f"{name} types a measured number outside its generated block"
