"""The sync-free pass: "_dn" entry points (row counts resident on the device, capacity-sized
buffers) and the captured FramePlan against the per-layer host-dispatched path. Bar: BITWISE equality -- tile
mapping, tail splits and summation order are the same functions of the row count on the device as on the host."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _bits(t):
    return t.contiguous().view(torch.int16 if t.element_size() == 2 else torch.int32)


def _sites(rng, batch, shape, n):
    cells = batch * shape[0] * shape[1] * shape[2]
    lin = np.sort(rng.choice(cells, size=min(n, cells), replace=False))
    x = lin % shape[2]
    y = (lin // shape[2]) % shape[1]
    z = (lin // (shape[2] * shape[1])) % shape[0]
    b = lin // (shape[2] * shape[1] * shape[0])
    return np.stack([b, z, y, x], 1).astype(np.int32)


@pytest.mark.parametrize("cin,cout,dtype", [(5, 16, "f32"), (16, 16, "f32"), (16, 32, "f32"), (32, 64, "f32"),
                                             (64, 64, "f32"), (64, 128, "f32"), (128, 128, "f32"), (128, 32, "f32"),
                                             (32, 64, "bf16"), (64, 64, "bf16"), (128, 128, "bf16")])
@pytest.mark.parametrize("n", [1, 300, 66900])       # 66900 rows = 262 tiles of 256: a split tail on 256 CUs
def test_dn_ops_match_host_count_ops_bitwise(cin, cout, dtype, n, dev):
    """rulebooks, conv forward (every kernel family) and dense() with the count on the device and a larger capacity."""
    from vista_amd import ops, ops_dn
    rng = np.random.default_rng(cin * 1000 + cout + n)
    shape = [11, 96, 96]
    coords = torch.from_numpy(_sites(rng, 2, shape, n)).to(dev)
    n = coords.shape[0]
    cap = n + 777
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    fdt = torch.float32 if cin == 5 else tdt
    feat = torch.from_numpy(rng.normal(size=(n, cin)).astype(np.float32)).to(dev).to(fdt)
    w = torch.from_numpy((rng.normal(size=(cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev).to(fdt)
    bias = torch.from_numpy(rng.normal(size=(cout,)).astype(np.float32)).to(dev)
    # host-count reference
    rb = ops.rulebook_subm(coords, 2, shape, 3)
    ref = ops.conv_fwd(feat, w, rb, bias=bias, relu=True)
    rbc = ops.rulebook_conv(coords, 2, shape, 3, 2, 1)
    refc = ops.conv_fwd(feat, w, rbc, bias=bias)
    refd = ops.to_dense(refc, rbc.out_indices, 2, rbc.out_shape)
    # device-count forms on capacity-sized buffers filled with junk past the count
    counts = torch.tensor([0, n, 0], dtype=torch.int64, device=dev)
    coords_c = torch.full((cap, 4), 12345, dtype=torch.int32, device=dev)
    coords_c[:n] = coords
    feat_c = torch.full((cap, cin), float("nan"), dtype=fdt, device=dev)
    feat_c[:n] = feat
    rb_d = ops_dn.rulebook_subm_dn(coords_c, counts[1:2], 2, shape, 3)
    assert torch.equal(rb_d.nbr[:, :n], rb.nbr)
    out = ops_dn.conv_fwd_dn(feat_c, w, rb_d, bias=bias, relu=True)
    assert torch.equal(_bits(out[:n]), _bits(ref))
    cap_out = rbc.n_out + 555
    rbc_d = ops_dn.rulebook_conv_dn(coords_c, counts[1:2], 2, shape, 3, 2, 1, cap_out, counts[2:3], counts[0:1])
    assert int(counts[2]) == rbc.n_out and int(counts[0]) == 0
    assert torch.equal(rbc_d.out_indices[:rbc.n_out], rbc.out_indices)
    assert torch.equal(rbc_d.nbr[:, :rbc.n_out], rbc.nbr)
    outc = ops_dn.conv_fwd_dn(feat_c, w, rbc_d, bias=bias)
    assert torch.equal(_bits(outc[:rbc.n_out]), _bits(refc))
    # submanifold rulebook over the strided conv's outputs through its rank table
    rb2 = ops.rulebook_subm(rbc.out_indices, 2, rbc.out_shape, 3, rank=rbc.out_rank)
    rb2_d = ops_dn.rulebook_subm_dn(rbc_d.out_indices, counts[2:3], 2, rbc_d.out_shape, 3, rank=rbc_d.out_rank)
    assert torch.equal(rb2_d.nbr[:, :rbc.n_out], rb2.nbr)
    dense = ops_dn.to_dense_dn(outc, rbc_d.out_indices, counts[2:3], 2, rbc_d.out_shape)
    assert torch.equal(_bits(dense), _bits(refd))


def test_dn_capacity_overflow_drops_rows_and_reports_the_count(dev):
    """A count above the capacity never writes out of bounds: the first `cap` rows are produced, the count is not
    clamped (that is how the caller notices)."""
    from vista_amd import ops, ops_dn
    rng = np.random.default_rng(5)
    shape = [9, 64, 64]
    coords = torch.from_numpy(_sites(rng, 1, shape, 5000)).to(dev)
    n = coords.shape[0]
    rbc = ops.rulebook_conv(coords, 1, shape, 3, 2, 1)
    cap_out = rbc.n_out - 1000
    counts = torch.tensor([0, n, 0], dtype=torch.int64, device=dev)
    rbc_d = ops_dn.rulebook_conv_dn(coords, counts[1:2], 1, shape, 3, 2, 1, cap_out, counts[2:3], counts[0:1])
    assert int(counts[2]) == rbc.n_out > cap_out
    assert torch.equal(rbc_d.out_indices, rbc.out_indices[:cap_out])
    # entries naming dropped rows are absent, the others intact
    ref = rbc.nbr[:, :cap_out]
    assert torch.equal(rbc_d.nbr, ref)


def test_rulebook_rejects_out_of_grid_coordinates(dev):
    from vista_amd import ops
    from vista_amd._lib import VxError
    rng = np.random.default_rng(6)
    shape = [9, 32, 32]
    c = _sites(rng, 2, shape, 500)
    bad = c.copy()
    bad[17, 0] = 2                       # batch index == batch_size
    with pytest.raises(VxError, match="outside batch"):
        ops.rulebook_conv(torch.from_numpy(bad).to(dev), 2, shape, 3, 2, 1)
    bad = c.copy()
    bad[3, 2] = 32                       # y == H
    with pytest.raises(VxError, match="outside batch"):
        ops.rulebook_conv(torch.from_numpy(bad).to(dev), 2, shape, 3, 2, 1)
    # submanifold tables skip such rows instead of reading the rank table out of bounds
    good = ops.rulebook_conv(torch.from_numpy(c).to(dev), 2, shape, 3, 2, 1)
    oc = good.out_indices.clone()
    oc[5, 0] = 7
    rb = ops.rulebook_subm(oc, 2, good.out_shape, 3, rank=good.out_rank)
    assert int((rb.nbr[:13, 5] >= 0).sum()) == 0


def _engine(kind, dev, dtype=torch.float32, dispatch="queued"):
    from vista_amd import models
    from vista_amd.engine import FusedBackbone
    from vista_amd.synth import backbone_state_dict
    net = models.load_numpy_state_dict(getattr(models, kind)(num_input_features=5), backbone_state_dict(kind))
    return FusedBackbone(net.to(dev).eval(), dtype=dtype, dispatch=dispatch)


def _frames(dev, seeds, sweeps):
    from vista_amd.synth import lidar_cloud
    return [torch.from_numpy(lidar_cloud(seed=s, sweeps=sweeps)).to(dev) for s in seeds]


@pytest.mark.parametrize("dispatch", ["queued", "graph"])
@pytest.mark.parametrize("kind,geometry,dtype", [("SpMiddleResNetFHD", "FHD", torch.float32),
                                                 ("SpMiddleResNetFHD", "FHD", torch.bfloat16),
                                                 ("MultiViewBackbone", "VISTA", torch.float32)])
def test_sync_free_plan_equals_eager_path_bitwise(kind, geometry, dtype, dispatch, dev):
    """Full-size frames: the sync-free plan (queued launches / one captured HIP graph) against per-layer launches,
    then the same plan on DIFFERENT frames (other counts, same capacities: for the graph this is a replay of nodes
    recorded on other data) and a batch-1 plan next to the batch-2 one."""
    from vista_amd.synth import GEOMETRIES, grid_size
    from vista_amd.voxel import VoxelGenerator
    rg, vs, T, mv = GEOMETRIES[geometry]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    eng = _engine(kind, dev, dtype, dispatch)
    eager = eng.eager_twin()
    assert eager.eager and not eng.eager
    builds = []
    real = type(eng)._build_plan
    eng._build_plan = lambda *a, **k: (builds.append(1), real(eng, *a, **k))[1]
    for seeds in ([0, 1], [7, 5], [3]):
        fr = _frames(dev, seeds, 10)
        got = eng.forward_frames(gen, fr, grid)
        ref = eager.forward_frames(gen, fr, grid)
        got, ref = (got, ref) if isinstance(got, tuple) else ((got,), (ref,))
        for g, r in zip(got, ref):
            assert g.shape == r.shape and g.dtype == r.dtype
            assert torch.equal(_bits(g), _bits(r))
    assert len(eng._plans) == 2 and len(builds) == 2   # batch 2 (built once, re-used for the second pair) and batch 1
    if dtype == torch.float32 and kind == "SpMiddleResNetFHD":
        # another fp32 matrix arithmetic is another plan (a captured graph keeps the kernels it recorded)
        from vista_amd import ops
        old = ops.set_f32_products(0)
        try:
            fr = _frames(dev, [3], 10)
            got = eng.forward_frames(gen, fr, grid)
            ref = eager.forward_frames(gen, fr, grid)
            assert torch.equal(_bits(got), _bits(ref)) and len(eng._plans) == 3
        finally:
            ops.set_f32_products(old)


def test_plan_grows_when_a_count_exceeds_its_capacity(dev, monkeypatch):
    from vista_amd.engine import FusedBackbone
    from vista_amd.synth import GEOMETRIES, grid_size
    from vista_amd.voxel import VoxelGenerator
    rg, vs, T, mv = GEOMETRIES["SMALL"]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    eng = _engine("SpMiddleResNetFHD", dev)
    real = FusedBackbone._probe
    builds = []

    def stingy(self, gen, frames, grid):
        rows, need = real(self, gen, frames, grid)
        builds.append(1)
        if len(builds) == 1:             # first plan: capacities far too small, the first pass must overflow
            return max(1, rows // 3), [max(1, n // 3) for n in need]
        return rows, need
    monkeypatch.setattr(FusedBackbone, "_probe", stingy)
    monkeypatch.setattr(FusedBackbone, "_cap", staticmethod(lambda n, h: max(64, int(n))))
    fr = _frames(dev, [11, 12], 3)
    got = eng.forward_frames(gen, fr, grid)
    ref = eng.eager_twin().forward_frames(gen, fr, grid)
    assert torch.equal(_bits(got), _bits(ref))
    assert len(builds) == 2              # one overflow, one re-plan from an exact geometry pass


# ---- full-size cases through the plan, and plan lifetime (round 3) ---------------------------------------------------
@pytest.mark.parametrize("dispatch", ["queued", "graph"])
def test_vista_frame_multiview_backbone_vs_oracle_planned(dispatch, dev):
    """BASELINE configs[2], sparse part, full size, through the sync-free plan (queued and captured)."""
    from _fullsize_cases import vista_multiview_case
    vista_multiview_case(dev, dispatch=dispatch)


@pytest.mark.parametrize("dispatch", ["queued", "graph"])
def test_stress_frame_end_to_end_vs_oracle_planned(dispatch, dev):
    """BASELINE configs[3], full size, fp32 then bf16 engines one after the other (two STRESS-sized captures in a row:
    the case that aborted in round 2's driver run, DESIGN.md "Plan lifetime")."""
    from _fullsize_cases import stress_case
    stress_case(dev, dispatch=dispatch)


def test_plans_of_two_engines_built_abandoned_and_rebuilt(dev):
    """Captured plans of two engines in one process: build, abandon (with and without a reference cycle that only the
    cyclic collector can break), rebuild, and capture again while garbage holding a graph is pending. Results stay
    bitwise those of the eager path throughout and every abandoned plan ends up closed."""
    import gc
    import weakref
    from vista_amd.synth import GEOMETRIES, grid_size
    from vista_amd.voxel import VoxelGenerator
    rg, vs, T, mv = GEOMETRIES["SMALL"]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    fr = _frames(dev, [21, 22], 3)
    ref32 = _engine("SpMiddleResNetFHD", dev, torch.float32, "graph").eager_twin().forward_frames(gen, fr, grid)
    ref16 = _engine("SpMiddleResNetFHD", dev, torch.bfloat16, "graph").eager_twin().forward_frames(gen, fr, grid)
    plans = []
    for round_ in range(3):
        a = _engine("SpMiddleResNetFHD", dev, torch.float32, "graph")
        out = a.forward_frames(gen, fr, grid)
        assert torch.equal(_bits(out), _bits(ref32))
        plans += [weakref.ref(p) for p in a._plans.values()]
        if round_ == 1:
            a.self_reference = a           # a cycle: only the collector can free this engine
        del a, out                         # abandoned; round 1's stays pending as garbage with its graph alive
        b = _engine("SpMiddleResNetFHD", dev, torch.bfloat16, "graph")
        out = b.forward_frames(gen, fr, grid)          # captures while the garbage above may still be pending
        assert torch.equal(_bits(out), _bits(ref16))
        # the same engine re-plans (larger batch -> new key; then an explicit close and a rebuild of the first key)
        out3 = b.forward_frames(gen, fr + _frames(dev, [23], 3), grid)
        assert out3.shape[0] == 3
        b.close()
        assert not b._plans
        out = b.forward_frames(gen, fr, grid)
        assert torch.equal(_bits(out), _bits(ref16))
        plans += [weakref.ref(p) for p in b._plans.values()]
        del b, out, out3
    gc.collect()
    assert all(p() is None or p().closed for p in plans)


def test_plan_error_word_is_not_sticky(dev):
    """A frame set with an out-of-grid coordinate raises; the next valid call on the SAME plan works."""
    from vista_amd._lib import VxError
    from vista_amd.synth import GEOMETRIES, grid_size
    from vista_amd.voxel import VoxelGenerator
    rg, vs, T, mv = GEOMETRIES["SMALL"]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    eng = _engine("SpMiddleResNetFHD", dev, torch.float32, "graph")
    fr = _frames(dev, [31], 3)
    ref = eng.eager_twin().forward_frames(gen, fr, grid)
    assert torch.equal(_bits(eng.forward_frames(gen, fr, grid)), _bits(ref))
    plan = next(iter(eng._plans.values()))
    plan.counts[0:1].fill_(1)              # what vx_rulebook_conv_dn leaves behind after a bad coordinate
    plan.h_counts[0] = 1
    with pytest.raises(VxError, match="outside batch"):
        plan.overflow()
    assert torch.equal(_bits(eng.forward_frames(gen, fr, grid)), _bits(ref))     # cleared inside the step
    assert next(iter(eng._plans.values())) is plan


@pytest.mark.parametrize("feat_streams", ["1", "2"])
@pytest.mark.parametrize("kind,geometry", [("SpMiddleResNetFHD", "FHD"), ("MultiViewBackbone", "VISTA")])
def test_two_plans_in_flight_equal_forward_frames_bitwise(kind, geometry, feat_streams, dev, monkeypatch):
    """FramePipeline2 (geometry graph of step k + 1 beside the feature graph of step k, two alternating slots): every
    output, in submission order, is bit for bit forward_frames' -- full-size frames, repeated and changing frame sets,
    and a change of the batch size in mid-stream (slots re-planned). An output is read AFTER the submit that returned
    it, on the caller's stream: it must not be the slot that submit has just queued work on."""
    from vista_amd.engine import FramePipeline2
    from vista_amd.synth import GEOMETRIES, grid_size
    from vista_amd.voxel import VoxelGenerator
    monkeypatch.setenv("VX_PIPE_FEAT_STREAMS", feat_streams)     # feature graphs of the two slots on one / two streams
    rg, vs, T, mv = GEOMETRIES[geometry]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    eng = _engine(kind, dev, dispatch="graph")
    sets = [_frames(dev, seeds, 10) for seeds in ([0, 1], [7, 5], [3, 9], [4])]

    def tup(o):
        return tuple(t.clone() for t in o) if isinstance(o, tuple) else (o.clone(),)
    ref = [tup(eng.forward_frames(gen, fr, grid)) for fr in sets]
    pipe = FramePipeline2(eng, gen, grid)
    order = [0, 1, 2, 0, 0, 3, 3, 1, 2]
    outs = []
    for j in order:
        o = pipe.submit(sets[j])
        if o is not None:
            outs.append(tup(o))
    outs += [tup(o) for o in pipe.drain()]
    assert len(outs) == len(order) and pipe.drain() == []
    for j, o in zip(order, outs):
        assert len(o) == len(ref[j])
        for a, b in zip(o, ref[j]):
            assert torch.equal(_bits(a), _bits(b))
    assert pipe.rebuilt == 6          # 2 slots x (batch 2, batch 1, batch 2 again)
    pipe.close()
    assert all(p is None for p in pipe.slots)


def test_two_plans_in_flight_overflow_is_rerun_exactly(dev, monkeypatch):
    """Slots whose capacities are far too small: the overflowing steps come back through forward_frames' exact re-plan
    (same bits as the eager path), and the slots are rebuilt at least that large before their next use."""
    from vista_amd.engine import FramePipeline2, FusedBackbone
    from vista_amd.synth import GEOMETRIES, grid_size
    from vista_amd.voxel import VoxelGenerator
    rg, vs, T, mv = GEOMETRIES["SMALL"]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    eng = _engine("SpMiddleResNetFHD", dev, dispatch="graph")
    real = FusedBackbone._probe
    builds = []

    def stingy(self, gen, frames, grid):
        rows, need = real(self, gen, frames, grid)
        builds.append(1)
        if len(builds) <= 2:             # the two slots' first plans
            return max(1, rows // 3), [max(1, n // 3) for n in need]
        return rows, need
    monkeypatch.setattr(FusedBackbone, "_probe", stingy)
    monkeypatch.setattr(FusedBackbone, "_cap", staticmethod(lambda n, h: max(64, int(n))))
    sets = [_frames(dev, [11, 12], 3), _frames(dev, [13, 14], 3)]
    pipe = FramePipeline2(eng, gen, grid)
    order = [0, 1, 0, 1, 1]
    outs = []
    for j in order:
        o = pipe.submit(sets[j])
        if o is not None:
            outs.append(o.clone())
    outs += [o.clone() for o in pipe.drain()]
    eager = eng.eager_twin()
    assert len(outs) == len(order)
    for j, o in zip(order, outs):
        assert torch.equal(_bits(o), _bits(eager.forward_frames(gen, sets[j], grid)))
    assert pipe.floor is not None and all(p.cap_rows >= pipe.floor.cap_rows for p in pipe.slots)
    pipe.close()


def test_plan_soak_random_frame_sets(dev):
    """Forty random frame sets (1-3 frames of 1-4 sweeps, empty-ish to large) through ONE graph engine and one queued
    engine: every result bitwise the eager path's, plans re-used / re-planned / rebuilt as sizes move, nothing leaks
    (at most one plan per batch size and arithmetic stays open)."""
    from vista_amd.synth import GEOMETRIES, grid_size, lidar_cloud
    from vista_amd.voxel import VoxelGenerator
    rg, vs, T, mv = GEOMETRIES["SMALL"]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    g_eng = _engine("SpMiddleResNetFHD", dev, torch.float32, "graph")
    q_eng = _engine("SpMiddleResNetFHD", dev, torch.bfloat16, "queued")
    eager32, eager16 = g_eng.eager_twin(), q_eng.eager_twin()
    rng = np.random.default_rng(2024)
    clouds = {}
    for it in range(40):
        nfr = int(rng.integers(1, 4))
        fr = []
        for _ in range(nfr):
            seed, sweeps = int(rng.integers(0, 6)), int(rng.integers(1, 5))
            if (seed, sweeps) not in clouds:
                clouds[seed, sweeps] = torch.from_numpy(lidar_cloud(seed=seed, sweeps=sweeps)).to(dev)
            c = clouds[seed, sweeps]
            keep = int(rng.integers(max(1, c.shape[0] // 8), c.shape[0] + 1))      # a random prefix: sizes move a lot
            fr.append(c[:keep].contiguous())
        for eng, ref in ((g_eng, eager32), (q_eng, eager16)):
            got, want = eng.forward_frames(gen, fr, grid), ref.forward_frames(gen, fr, grid)
            assert torch.equal(_bits(got), _bits(want)), it
    assert len(g_eng._plans) <= 3 and len(q_eng._plans) <= 3
    assert all(not p.closed for p in g_eng._plans.values())
    g_eng.close()
    q_eng.close()
