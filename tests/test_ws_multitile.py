"""GPU parity tests (``-m gpu``) of the weights-stationary Conv2D kernels
(``sup3r_amd/csrc/kernels_conv2d_ws.hip``) IN THE REGIME ``bench.py`` RUNS THEM
IN: every (half-)workgroup walks several tiles, so the steady state — the
register prefetch of the next halo under the tap loop, the one ``vmcnt(0)`` per
M phase, the zero-flag hand-over of the frame form, the second skip operand
riding the prefetch registers — is compared with the ORACLE, not with another
build of the same kernel (tests/test_mfma_gen.py compares at <= 220 tiles: one
tile per half-workgroup, prologue -> taps -> epilogue).

How many tiles a launch has (``launch_conv2d_ws``):

* ``conv2d_ws_pp_kernel`` (64 -> 64 k, no exogenous channel): single-image
  tiles of 16 x 16 positions, ``gp = min(num_cu / n_ct, (T1 + 1) / 2)``
  workgroups of two half-workgroups; a half-workgroup has >= 2 tiles when
  ``T1 > 2 * 256 = 512`` and >= 3 when ``T1 >= 1536`` (n_ct = 1, 256 CUs);
* ``conv2d_ws_kernel`` (the exogenous-channel form, the few-feature output
  conv): tiles of 2 images x 16 x 16, ``min(num_cu / n_ct, T)`` workgroups —
  several tiles each when ``T > 256``.

The shapes are those of the reference's spatial step
(``/root/reference/sup3r/pipeline/forward_pass.py:274-337``: a 2-D model sees
the chunk's time steps as its batch axis;
``examples/sup3rwind/run_configs/wind/config_fwp_spatial.json``: 75 x 75 x (38 +
2 x 5) chunks) — the shape of ``bench.py``'s ``fwd2d`` leg.

The nets are per-image independent, so the device runs the whole batch and the
oracle only a few of its images (first two + last); the batch is made of
DISTINCT images (a tile that read another image's halo would not pass).
Bounds as everywhere (tests/test_parity_r02.py): per op teacher-forced (<= one
bf16 spacing on <= 1 % of an op's elements) + 3e-2 end to end; gradients 2e-2
on the device's activations and masks.
"""
import numpy as np
import pytest

from sup3r_amd import spec as S
from tests.helpers import emulate_plan, rel_linf, rel_max, teacher_forced_check
from tests.test_ref_surface import _exo_for, load_surface

pytestmark = pytest.mark.gpu

NUM_CU = 256
# an inference plan with a buffer per tensor (and the last op writing ITS buffer,
# not the caller's output): s3_plan_tensor_read then sees every op's output
KEEP = {'KEEP_ACTIVATIONS': 1, 'NO_DIRECT_OUTPUT': 1}


def _selection(ph, field='fwd'):
    return [ph.op_info(i)[field] for i, op in enumerate(ph.plan.ops)
            if op['kind'] == S.OP_CONV]


def _tiles_pp(n, h, w):
    return n * ((h + 15) // 16) * ((w + 15) // 16)


def _tiles_lockstep(n, h, w):
    return ((n + 1) // 2) * ((h + 15) // 16) * ((w + 15) // 16)


def _pp_runs(t1, n_ct=1):
    """(tiles per half-workgroup, smallest .. largest) of a ping-pong launch"""
    gp = max(1, min(NUM_CU // n_ct, (t1 + 1) // 2))
    runs = [(r + 1) * t1 // gp - r * t1 // gp for r in range(gp)]
    halves = [(L - g + 1) // 2 for L in runs for g in (0, 1)]
    return min(halves), max(halves), len(set(runs)) > 1


def _per_op(spec, ref, pht, x, exo, samples, what):
    from oracle.network import Network as OracleNet
    from tests.test_parity_r02 import _assert_per_op
    emu = OracleNet(spec)
    emu.init_weights(x[:1], None if exo is None else
                     {k: v[:1] for k, v in exo.items()}, seed=0)
    emu.set_weights(ref.weights)
    n_ops, n_store, _ = emulate_plan(emu, pht)
    assert n_ops >= 30 and n_store >= 30, (n_ops, n_store)
    for sl in samples:
        stats = teacher_forced_check(emu, pht, x, exo, sample=sl)
        assert len(stats) == len(pht.plan.ops)
        _assert_per_op(stats, f'{what}, images {sl.start}:{sl.stop}')


# (a) + (d): the bench shape (48, 75, 75): 1200 lo-res tiles = runs of 4 and 5 per
# workgroup (the two half-workgroups of a 5-run have 3 and 2 tiles), 4800 hi-res
# tiles; and 66 x 75 x 70: 1650 tiles (>= 3 per half-workgroup), ragged 6-column
# last tile
@pytest.mark.parametrize('shape', [(48, 75, 75, 2), (66, 75, 70, 2)])
def test_gen_2x_2f_at_the_bench_shape_vs_oracle(shape):
    from tests.test_parity_r02 import _hip, _oracle
    rel = 'spatial/gen_2x_2f.json'
    spec = load_surface(rel)
    n = shape[0]
    t1 = _tiles_pp(*shape[:3])
    lo, hi, ragged = _pp_runs(t1)
    assert t1 > 2 * NUM_CU and lo >= 2 and ragged and hi > lo, (t1, lo, hi)
    if n > 48:
        assert t1 >= 6 * NUM_CU and lo >= 3, (t1, lo)
    rng = np.random.default_rng(601 + n)
    x = rng.standard_normal(shape).astype(np.float32)
    ref = _oracle(spec, x[:1], None, seed=61)
    y_head, y_last = ref.forward(x[:2]), ref.forward(x[-1:])
    net = _hip(spec, ref.weights, 'bf16')
    dev = net.dev
    xd = dev.to_device(x)
    # the inference plan: what bench.py's fwd2d leg times
    ph = net.plan(shape, training=False)
    sel = _selection(ph)
    assert sel.count('conv2d_ws') >= 34, sel
    y = ph.forward(xd).cpu().numpy()
    assert y.shape == (n, 2 * shape[1], 2 * shape[2], 2) and np.isfinite(y).all()
    e_head, e_last = rel_linf(y[:2], y_head), rel_linf(y[-1:], y_last)
    print(f'{rel} {shape}: {t1} tiles ({lo}..{hi} per half-workgroup), end to '
          f'end vs the fp32 oracle {e_head:.2e} / {e_last:.2e}')
    assert e_head < 3e-2 and e_last < 3e-2, (e_head, e_last)
    # per op, on the inference plan itself with a buffer per tensor (option
    # KEEP_ACTIVATIONS: same kernels, same bits)
    keep = net.plan(shape, training=False, options=KEEP)
    assert _selection(keep) == sel
    np.testing.assert_array_equal(keep.forward(xd).cpu().numpy(), y)
    _per_op(spec, ref, keep, x, None, [slice(0, 2), slice(n - 1, n)],
            f'{rel} bf16 inference plan {shape}')


def test_gen_2x_2f_bf16x3_at_the_bench_shape_meets_the_fp32_tolerance():
    """the mode that owns north_star's L-inf < 1e-3 on the production 2-D chunk:
    BF16X3 plans run the 64 -> 64 k convs on conv2d_ws_x3_pp_kernel (two K passes of
    [hi | lo] operands, ping-pong half-workgroups, 2 .. 3 tiles each at this shape)
    and the 64 -> 2 output conv on conv2d_out_kernel<X3>; the oracle on the first
    two and the last image"""
    from tests.test_parity_r02 import _hip, _oracle
    rel = 'spatial/gen_2x_2f.json'
    spec = load_surface(rel)
    shape = (48, 75, 75, 2)
    rng = np.random.default_rng(611)
    x = rng.standard_normal(shape).astype(np.float32)
    ref = _oracle(spec, x[:1], None, seed=64)
    y_head, y_last = ref.forward(x[:2]), ref.forward(x[-1:])
    net = _hip(spec, ref.weights, 'bf16x3')
    ph = net.plan(shape, training=False)
    sel = _selection(ph)
    assert sel.count('conv2d_ws') >= 35, sel
    y = ph.forward(net.dev.to_device(x)).cpu().numpy()
    e_head = float(np.abs(y[:2] - y_head).max())
    e_last = float(np.abs(y[-1:] - y_last).max())
    print(f'{rel} bf16x3 {shape}: L-inf vs the fp32 oracle {e_head:.2e} / '
          f'{e_last:.2e} (scale {np.abs(y_head).max():.2f})')
    assert e_head < 1e-3 and e_last < 1e-3, (e_head, e_last)
    # the logical-axes tile kernel (option NO_CONV2D_WS: what these plans ran on
    # before round 6) agrees at the same level
    alt = net.plan(shape, training=False, options={'NO_CONV2D_WS': 1})
    assert _selection(alt).count('conv2d_ws') == 0
    ya = alt.forward(net.dev.to_device(x)).cpu().numpy()
    assert float(np.abs(ya - y).max()) < 1e-3
    # lock-step form of the X3 kernel (option NO_WS_PP): same MFMAs per position in
    # the same pass order -> the same bits
    lock = net.plan(shape, training=False, options={'NO_WS_PP': 1})
    np.testing.assert_array_equal(lock.forward(net.dev.to_device(x)).cpu().numpy(), y)


# (b) the second step of the reference's wind chain (sup3rcc/gen_wind_5x_1x_6f +
# topography): 16 hi-res 64 -> 64 convs with skip operands, the conv with TWO skip
# operands (res2), the exogenous-channel form and the 64 -> 6 output conv, all at
# (24, 150, 150) = 2400 single-image / 1200 two-image tiles
def test_wind_chain_step_exo_res_res2_multi_tile_vs_oracle():
    from tests.test_parity_r02 import _hip, _oracle
    rel = 'sup3rcc/gen_wind_5x_1x_6f.json'
    spec = load_surface(rel)
    shape = (24, 30, 30, 7)
    assert _tiles_pp(24, 150, 150) > 8 * NUM_CU
    assert _tiles_lockstep(24, 150, 150) > 4 * NUM_CU
    rng = np.random.default_rng(77)
    x = rng.standard_normal(shape).astype(np.float32)
    plan = S.build_plan(S.parse_layers(spec), shape)
    exo = _exo_for(plan, 'topography', rng, np.float32)
    assert exo['topography'].shape[:3] == (24, 150, 150)

    def cut(sl):
        return {k: v[sl] for k, v in exo.items()}
    ref = _oracle(spec, x[:1], cut(slice(0, 1)), seed=62)
    y_head = ref.forward(x[:2], cut(slice(0, 2)))
    y_last = ref.forward(x[-1:], cut(slice(23, 24)))
    net = _hip(spec, ref.weights, 'bf16')
    dev = net.dev
    xd, ed = dev.to_device(x), {k: dev.to_device(v) for k, v in exo.items()}
    ph = net.plan(shape, training=False)
    kinds = [op['kind'] for op in ph.plan.ops]
    conv65 = next(i for i, op in enumerate(ph.plan.ops)
                  if op['kind'] == S.OP_CONV and op['cin'] == 65)
    assert ph.op_info(conv65)['fwd'] == 'conv2d_ws'              # EXO form
    assert ph.op_info(kinds.index(S.OP_CONCAT))['in_rep'] == 1   # fused away
    iadd = [i for i, k in enumerate(kinds) if k == S.OP_ADD]
    assert [ph.op_info(i)['in_rep'] for i in iadd][-1] == 1      # res2 form
    sel = _selection(ph)
    assert sel.count('conv2d_ws') >= 36 and sel[-1] == 'conv2d_ws', sel
    y = ph.forward(xd, ed).cpu().numpy()
    assert y.shape == (24, 150, 150, 6) and np.isfinite(y).all()
    e_head, e_last = rel_linf(y[:2], y_head), rel_linf(y[-1:], y_last)
    print(f'{rel} {shape}: end to end vs the fp32 oracle {e_head:.2e} / '
          f'{e_last:.2e}')
    assert e_head < 3e-2 and e_last < 3e-2, (e_head, e_last)
    # the separate-pass forms of the same plan (second skip as an add pass, concat
    # + two-pass 65-channel conv) agree like two valid bf16 evaluations do
    alt = net.plan(shape, training=False,
                   options={'NO_WS_RES2': 1}).forward(xd, ed).cpu().numpy()
    np.testing.assert_array_equal(alt, y)
    # per op, ON THE INFERENCE PLAN (option KEEP_ACTIVATIONS: a buffer per tensor, same
    # kernels, same fusions — asserted): the fused concat and the conv whose store
    # carries the second skip have no output of their own; the oracle steps over
    # them and the next op's output covers both.  The oracle rounds where the
    # device does: the first sum of the res2 conv is rounded to bf16 before the
    # second add (the plan with the add as a pass of its own says where).
    keep = net.plan(shape, training=False, options=KEEP)
    assert _selection(keep) == sel
    assert [keep.op_info(i)['in_rep'] for i in iadd] == \
        [ph.op_info(i)['in_rep'] for i in iadd]
    np.testing.assert_array_equal(keep.forward(xd, ed).cpu().numpy(), y)
    from oracle.network import Network as OracleNet
    from tests.test_parity_r02 import _assert_per_op
    emu = OracleNet(spec)
    emu.init_weights(x[:1], cut(slice(0, 1)), seed=0)
    emu.set_weights(ref.weights)
    emulate_plan(emu, net.plan(shape, training=False,
                               options=dict(KEEP, NO_WS_RES2=1)))
    n_ops, n_store, _ = emulate_plan(emu, keep)
    assert n_ops >= 36 and n_store >= 34, (n_ops, n_store)
    for sl in (slice(0, 1), slice(23, 24)):
        stats = teacher_forced_check(emu, keep, x, exo, sample=sl,
                                     allow_missing=True)
        skipped = [d['op'] for d in stats if d.get('skipped')]
        assert len(skipped) == 2, skipped       # the concat, the res2 conv
        _assert_per_op([d for d in stats if not d.get('skipped')],
                       f'{rel} bf16 inference plan {shape}, image {sl.start}')


# (c) + (d) the training frame form: the data gradient of a 2-D reflect-'same'
# 64 -> 64 conv as a conv over the zero-padded frame, 24 x (75 + 2) x (75 + 2):
# 600 tiles on 256 workgroups — runs of 2 and 3 (the half-workgroups of a 3-run
# have 2 and 1 tiles, the zero-flag hand-over between a border tile and an
# interior one happens inside a run)
def test_gen_2x_2f_training_frame_form_multi_tile_vs_oracle():
    from tests.test_parity_r02 import _assert_per_op, _hip, _oracle
    rel = 'spatial/gen_2x_2f.json'
    spec = load_surface(rel)
    n, n_base = 24, 3
    shape = (n, 75, 75, 2)
    t1 = _tiles_pp(n, 77, 77)
    lo, hi, ragged = _pp_runs(t1)
    assert t1 > 2 * NUM_CU and hi >= 2 and ragged, (t1, lo, hi)
    rng = np.random.default_rng(91)
    base = rng.standard_normal((n_base,) + shape[1:]).astype(np.float32)
    # three distinct images, interleaved (0 1 2 0 1 2 ..): the oracle runs three
    # images, a tile that took a neighbour image's cells would not pass
    idx = np.arange(n) % n_base
    x = base[idx]
    ref = _oracle(spec, base[:1], None, seed=63)
    y_ref = ref.forward(base)
    net = _hip(spec, ref.weights, 'bf16')
    dev = net.dev
    ph = net.plan(shape, training=True)
    dg = _selection(ph, 'dgrad')
    print('dgrad kernels:', {k: dg.count(k) for k in sorted(set(dg))})
    info = [ph.op_info(i) for i, op in enumerate(ph.plan.ops)
            if op['kind'] == S.OP_CONV and op['cin'] == 64 and op['cout'] == 64]
    assert sum(bool(f.get('dgrad_frame16')) for f in info) >= 30, \
        [f.get('dgrad_frame16') for f in info]
    y = ph.forward(dev.to_device(x)).cpu().numpy()
    assert rel_linf(y[:n_base], y_ref) < 3e-2
    for k in range(n_base, n):
        np.testing.assert_array_equal(y[k], y[k % n_base])
    emulate_plan(ref, ph, masks=False)
    stats = teacher_forced_check(ref, ph, x, None, sample=slice(0, n_base))
    _assert_per_op(stats, f'{rel} bf16 training {shape}')
    emulate_plan(ref, ph, masks=True, rounding=False, sample=slice(0, n_base))
    dy_base = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = ref.backward(dy_base)
    dx = ph.backward(dev.to_device(dy_base[idx]), need_dx=True).cpu().numpy()
    dx = dx.reshape((n,) + dx_ref.shape[1:])
    for k in range(n_base, n):
        np.testing.assert_array_equal(dx[k], dx[k % n_base])
    errs = {'dx': rel_max(dx[:n_base], dx_ref)}
    rep = n // n_base
    gmax = max(float(np.abs(g).max()) for g in ref.grads) * rep
    for i, (g, g_ref) in enumerate(zip(net.grads, ref.grads)):
        g_ref = g_ref * rep
        errs[i] = float(np.abs(g - g_ref).max()
                        / max(np.abs(g_ref).max(), 1e-3 * gmax))
    worst = max(errs.values())
    print(f'{rel} training {shape}: worst gradient error {worst:.2e} '
          f'(dx {errs["dx"]:.2e})')
    assert worst < 2e-2, errs
