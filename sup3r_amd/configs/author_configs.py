"""Authors the sup3r-spec JSON files shipped in this directory.

These are written in the reference's own ``hidden_layers`` spec language
(sup3r/configs/**) so they parse with phygnn unchanged; they are generated
here, not copied:

* ``gen_5x_12x_2f.json`` — BASELINE.json config C2.  The reference ships no
  5x/12x generator; this follows the topology of
  ``spatiotemporal/gen_2x_12x_14f.json`` (temporal head 2*2*3, 16 residual
  blocks x 64 ch) with the expansion conv at 8*5^2 = 200 filters,
  ``spatial_mult`` 5 and 2 output features.
* ``disc_st.json`` / ``disc_s.json`` — the 8-conv + dense patch discriminator
  topology (valid padding, production) and ``*_same`` variants (the padding the
  reference's own tests use so that tiny samples survive 4 stride-2 convs).
* ``test_*.json`` — small-channel nets of the same archetypes for fast parity.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def pcc(nd, filters, act=True, pad=3, crop=2, conv=None, **kw):
    """REFLECT pad -> conv(k=3, valid) -> crop [-> LeakyReLU 0.2]."""
    cls = conv or f'Conv{nd}D'
    if isinstance(pad, int):
        pad = [pad] * nd
    pads = [[0, 0]] + [[p, p] for p in pad] + [[0, 0]]
    conv_spec = {'class': cls, 'filters': filters, 'kernel_size': 3,
                 'strides': 1}
    conv_spec.update(kw)
    out = [{'class': 'FlexiblePadding', 'paddings': pads, 'mode': 'REFLECT'},
           conv_spec, {'class': f'Cropping{nd}D', 'cropping': crop}]
    if act:
        out.append({'alpha': 0.2, 'class': 'LeakyReLU'})
    return out


def st_gen(s, tmults, nf_out, body=16, ch=64, exo=None):
    hl = []
    if len(tmults) > 1:
        assert len(set(tmults[:-1])) == 1
        hl.append({'n': len(tmults) - 1, 'repeat': pcc(3, ch) + [
            {'class': 'SpatioTemporalExpansion', 'temporal_mult': tmults[0],
             'temporal_method': 'nearest'}]})
    hl += pcc(3, ch) + [{'class': 'SpatioTemporalExpansion',
                         'temporal_mult': tmults[-1],
                         'temporal_method': 'nearest'}]
    hl.append({'class': 'SkipConnection', 'name': 'a'})
    hl.append({'n': body, 'repeat': (
        [{'class': 'SkipConnection', 'name': 'b'}] + pcc(3, ch)
        + pcc(3, ch, act=False) + [{'class': 'SkipConnection', 'name': 'b'}])})
    hl += pcc(3, ch, act=False) + [{'class': 'SkipConnection', 'name': 'a'}]
    hl += pcc(3, 8 * s * s, act=False) + [
        {'class': 'SpatioTemporalExpansion', 'spatial_mult': s},
        {'alpha': 0.2, 'class': 'LeakyReLU'}]
    if exo:
        hl.append({'class': exo[0], 'name': exo[1]})
    hl += pcc(3, nf_out, act=False)
    return {'hidden_layers': hl}


def s_gen(s, nf_out, body=16, ch=64):
    """Spatial generator of the Conv2DTranspose archetype (pad 3 / crop 4)."""
    def blk(f, act):
        return pcc(2, f, act=False, crop=4, conv='Conv2DTranspose',
                   activation=act)
    hl = blk(ch, 'relu') + [{'class': 'SkipConnection', 'name': 'a'}]
    hl.append({'n': body, 'repeat': (
        [{'class': 'SkipConnection', 'name': 'b'}] + blk(ch, 'relu')
        + blk(ch, None) + [{'class': 'SkipConnection', 'name': 'b'}])})
    hl += blk(ch, None) + [{'class': 'SkipConnection', 'name': 'a'}]
    hl += blk(ch * s * s, None) + [
        {'class': 'SpatialExpansion', 'spatial_mult': s},
        {'class': 'Activation', 'activation': 'relu'}]
    hl += blk(nf_out, None)
    return {'hidden_layers': hl}


def disc(nd, padding, dense=(2048, 1024), widths=(32, 64, 128, 256)):
    hl = []
    for w in widths:
        for stride in (1, 2):
            hl += [{'class': f'Conv{nd}D', 'filters': w, 'kernel_size': 3,
                    'padding': padding, 'strides': stride},
                   {'alpha': 0.2, 'class': 'LeakyReLU'}]
    hl.append({'class': 'Flatten'})
    for u in dense:
        hl += [{'class': 'Dense', 'units': u},
               {'alpha': 0.2, 'class': 'LeakyReLU'}]
    hl.append({'class': 'Dense', 'units': 1})
    return {'hidden_layers': hl}


def toy_wind():
    """The layer list of the reference's sup3rcc/gen_wind_3x_4x_2f.json (a
    placeholder config: every hidden conv has ONE filter, 1 447 parameters,
    Sup3rConcat topography) — BASELINE config C4 names it."""
    hl = [{'n': 2, 'repeat': pcc(3, 1) + [
        {'class': 'SpatioTemporalExpansion', 'temporal_mult': 2,
         'temporal_method': 'nearest'}]}]
    hl.append({'class': 'SkipConnection', 'name': 'a'})
    hl.append({'n': 1, 'repeat': (
        [{'class': 'SkipConnection', 'name': 'b'}] + pcc(3, 1)
        + pcc(3, 1, act=False) + [{'class': 'SkipConnection', 'name': 'b'}])})
    hl += pcc(3, 1, act=False) + [{'class': 'SkipConnection', 'name': 'a'}]
    hl += pcc(3, 36, act=False) + [
        {'class': 'SpatioTemporalExpansion', 'spatial_mult': 3},
        {'alpha': 0.2, 'class': 'LeakyReLU'},
        {'class': 'Sup3rConcat', 'name': 'topography'}]
    hl += pcc(3, 2, act=False)
    return {'hidden_layers': hl}


# ---------------------------------------------------------------------------
# the reference's shipped spec surface (sup3r/configs/**: 16 generators + 2
# discriminators), authored from their topology parameters.  They are DATA of
# the drop-in boundary (north_star: "sup3r/configs JSON-spec surface"); the
# CPU test tests/test_ref_surface.py asserts that each authored spec equals
# the reference's file wherever /root/reference exists, and every one of them
# runs through the HIP path against the oracle (-m gpu, same file).
# ---------------------------------------------------------------------------
def _t_expand(m, meth='nearest', **kw):
    d = {'class': 'SpatioTemporalExpansion', 'temporal_mult': m,
         'temporal_method': meth}
    d.update(kw)
    return d


def _res_blocks(nd, n, name, ch=64, pad=3, crop=2, skip=True):
    body = pcc(nd, ch, pad=pad, crop=crop) + \
        pcc(nd, ch, act=False, pad=pad, crop=crop)
    if skip:
        s = {'class': 'SkipConnection', 'name': name}
        body = [s] + body + [dict(s)]
    return {'n': n, 'repeat': body}


def ref_st_gen(s, rep_n, extra_m, exp_filters, nf_out):
    """spatiotemporal/gen_*: ``rep_n`` x (conv + nearest x2) [+ conv +
    nearest x ``extra_m``], 16 residual blocks inside skip 'a', expansion conv
    + depth-to-space ``s``, output conv."""
    hl = [{'n': rep_n, 'repeat': pcc(3, 64) + [_t_expand(2)]}]
    if extra_m:
        hl += pcc(3, 64) + [_t_expand(extra_m)]
    hl.append({'class': 'SkipConnection', 'name': 'a'})
    hl.append(_res_blocks(3, 16, 'b'))
    hl += pcc(3, 64, act=False) + [{'class': 'SkipConnection', 'name': 'a'}]
    hl += pcc(3, exp_filters, act=False) + [
        {'class': 'SpatioTemporalExpansion', 'spatial_mult': s},
        {'alpha': 0.2, 'class': 'LeakyReLU'}]
    hl += pcc(3, nf_out, act=False)
    return {'hidden_layers': hl}


def ref_s_gen(mults, nf_out):
    """spatial/gen_*: the Conv2DTranspose archetype (pad 3 / crop 4) with one
    expansion stage (64 m^2 filters + SpatialExpansion m + relu) per mult."""
    def blk(f, act):
        return pcc(2, f, act=False, crop=4, conv='Conv2DTranspose',
                   activation=act)
    hl = blk(64, 'relu') + [{'class': 'SkipConnection', 'name': 'a'}]
    hl.append({'n': 16, 'repeat': (
        [{'class': 'SkipConnection', 'name': 'b'}] + blk(64, 'relu')
        + blk(64, None) + [{'class': 'SkipConnection', 'name': 'b'}])})
    hl += blk(64, None) + [{'class': 'SkipConnection', 'name': 'a'}]
    for m in mults:
        hl += blk(64 * m * m, None) + [
            {'class': 'SpatialExpansion', 'spatial_mult': m},
            {'class': 'Activation', 'activation': 'relu'}]
    hl += blk(nf_out, None)
    return {'hidden_layers': hl}


def ref_cc_temporal(t_mult, t_roll, exp_filters, nf_out, pad=3, crop=2):
    """sup3rcc/gen_solar_1x_8x_1f, gen_trh_1x_24x_2f: 1x spatial, temporal
    enhancement by depth_to_time (``exp_filters`` -> t_mult time steps of
    exp_filters / t_mult channels each, rolled by ``t_roll``)."""
    hl = pcc(3, 64, pad=pad, crop=crop)
    hl.append(_res_blocks(3, 16, 'small_skip', pad=pad, crop=crop))
    hl += pcc(3, 64, pad=pad, crop=crop)
    hl += pcc(3, exp_filters, act=False, pad=pad, crop=crop) + [
        _t_expand(t_mult, 'depth_to_time', t_roll=t_roll),
        {'alpha': 0.2, 'class': 'LeakyReLU'}]
    hl += pcc(3, nf_out, act=False)
    return {'hidden_layers': hl}


def ref_cc_solar_5x():
    hl = pcc(2, 64) + [{'class': 'SkipConnection', 'name': 'big_skip'}]
    hl.append(_res_blocks(2, 16, 'small_skip'))
    hl += pcc(2, 64, act=False) + [
        {'class': 'SkipConnection', 'name': 'big_skip'}]
    hl += pcc(2, 1600, act=False) + [
        {'class': 'SpatialExpansion', 'spatial_mult': 5},
        {'alpha': 0.2, 'class': 'LeakyReLU'}]
    hl += pcc(2, 1, act=False)
    return {'hidden_layers': hl}


def ref_cc_wind_5x():
    hl = pcc(2, 64) + [{'class': 'SkipConnection', 'name': 'big_skip_1'}]
    hl.append(_res_blocks(2, 8, 'small_skip_1'))
    hl += pcc(2, 64, act=False) + [
        {'class': 'SkipConnection', 'name': 'big_skip_1'}]
    hl += pcc(2, 1600, act=False) + [
        {'class': 'SpatialExpansion', 'spatial_mult': 5},
        {'alpha': 0.2, 'class': 'LeakyReLU'},
        {'class': 'Sup3rConcat', 'name': 'topography'}]
    hl += pcc(2, 64) + [{'class': 'SkipConnection', 'name': 'big_skip_2'}]
    hl.append(_res_blocks(2, 8, 'small_skip_2'))
    hl.append({'class': 'SkipConnection', 'name': 'big_skip_2'})
    hl += pcc(2, 6, act=False)
    return {'hidden_layers': hl}


def ref_cc_wind_24x():
    hl = pcc(3, 64, pad=2, crop=1) + [_t_expand(3)]
    hl.append({'n': 3, 'repeat': pcc(3, 64) + [_t_expand(2)]})
    hl.append(_res_blocks(3, 16, None, skip=False))
    hl += pcc(3, 64, act=False)
    hl += pcc(3, 6, act=False)
    return {'hidden_layers': hl}


def reference_surface():
    """{relative path under sup3r/configs: spec} of everything the reference
    ships there."""
    return {
        'spatial/disc.json': disc(2, 'valid', dense=(1024,)),
        'spatial/gen_2x_1f.json': ref_s_gen([2], 1),
        'spatial/gen_2x_2f.json': ref_s_gen([2], 2),
        'spatial/gen_10x_2f.json': ref_s_gen([2, 5], 2),
        'spatiotemporal/disc.json': disc(3, 'valid'),
        'spatiotemporal/gen_2x_2x_2f.json': ref_st_gen(2, 1, 0, 72, 2),
        'spatiotemporal/gen_2x_12x_14f.json': ref_st_gen(2, 2, 3, 72, 14),
        'spatiotemporal/gen_3x_4x_1f.json': ref_st_gen(3, 2, 0, 72, 1),
        'spatiotemporal/gen_3x_4x_2f.json': ref_st_gen(3, 2, 0, 72, 2),
        'spatiotemporal/gen_3x_4x_10f.json': ref_st_gen(3, 2, 0, 72, 10),
        'spatiotemporal/gen_3x_4x_14f.json': ref_st_gen(3, 2, 0, 576, 14),
        'spatiotemporal/gen_4x_24x_3f.json': ref_st_gen(4, 3, 3, 128, 3),
        'sup3rcc/gen_solar_1x_8x_1f.json': ref_cc_temporal(
            8, 4, 512, 1, pad=[3, 3, 2], crop=[2, 2, 1]),
        'sup3rcc/gen_trh_1x_24x_2f.json': ref_cc_temporal(24, 12, 768, 2),
        'sup3rcc/gen_solar_5x_1x_1f.json': ref_cc_solar_5x(),
        'sup3rcc/gen_wind_5x_1x_6f.json': ref_cc_wind_5x(),
        'sup3rcc/gen_wind_1x_24x_6f.json': ref_cc_wind_24x(),
        'sup3rcc/gen_wind_3x_4x_2f.json': toy_wind(),
    }


def main():
    files = {
        'gen_wind_3x_4x_2f_toy.json': toy_wind(),
        'gen_5x_12x_2f.json': st_gen(5, [2, 2, 3], 2),
        # C4/C5 body: the 3x/4x 2-feature topology (16 blocks x 64 ch)
        'gen_3x_4x_2f.json': st_gen(3, [2, 2], 2),
        # C1: spatial 2x generator of the Conv2DTranspose archetype
        'gen_2x_2f.json': s_gen(2, 2),
        'disc_st.json': disc(3, 'valid'),
        'disc_s.json': disc(2, 'valid', dense=(1024,)),
        'disc_st_same.json': disc(3, 'same'),
        'disc_s_same.json': disc(2, 'same', dense=(1024,)),
        'test_gen_st_2x_4x_2f.json': st_gen(2, [2, 2], 2, body=2, ch=16),
        'test_gen_st_3x_4x_2f_topo.json': st_gen(
            3, [2, 2], 2, body=1, ch=8, exo=('Sup3rConcat', 'topography')),
        'test_gen_st_64ch.json': st_gen(2, [2], 2, body=1, ch=64),
        'test_gen_s_2x_2f.json': s_gen(2, 2, body=2, ch=16),
        'test_disc_st_same.json': disc(3, 'same', dense=(64, 32),
                                       widths=(8, 8, 16, 16)),
        'test_disc_s_same.json': disc(2, 'same', dense=(32,),
                                      widths=(8, 8, 16, 16)),
        'test_disc_st_valid.json': disc(3, 'valid', dense=(32,),
                                        widths=(8, 16)),
    }
    for rel, spec in reference_surface().items():
        files[os.path.join('sup3r', rel)] = spec
    for name, spec in files.items():
        os.makedirs(os.path.dirname(os.path.join(HERE, name)), exist_ok=True)
        with open(os.path.join(HERE, name), 'w') as f:
            json.dump(spec, f, indent=1)


if __name__ == '__main__':
    main()
